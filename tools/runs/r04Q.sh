mkdir -p gpurun_out/r04Q
AB_STEPS=30 python tools/ab_run.py default quadric torus:6 > gpurun_out/r04Q/ab_qboxpairs.txt 2>&1
AB_STEPS=30 python tools/ab_run.py default quadric torus:6 >> gpurun_out/r04Q/ab_qboxpairs.txt 2>&1
cat gpurun_out/r04Q/ab_qboxpairs.txt
