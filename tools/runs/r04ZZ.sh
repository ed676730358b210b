mkdir -p gpurun_out/r04ZZ
AB_STEPS=20 python tools/ab_run.py torus:6 default > gpurun_out/r04ZZ/ab_torus_far_nolimit.txt 2>&1
cat gpurun_out/r04ZZ/ab_torus_far_nolimit.txt
