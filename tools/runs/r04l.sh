mkdir -p gpurun_out/r04l
AB_STEPS=30 python tools/ab_run.py default quadric torus:6 > gpurun_out/r04l/ab_spre.txt 2>&1; cat gpurun_out/r04l/ab_spre.txt
