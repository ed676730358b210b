mkdir -p gpurun_out/r04F
AB_STEPS=30 python tools/ab_run.py torus:6 default > gpurun_out/r04F/ab_tube2.txt 2>&1
AB_STEPS=30 python tools/ab_run.py torus:6 default >> gpurun_out/r04F/ab_tube2.txt 2>&1
cat gpurun_out/r04F/ab_tube2.txt
