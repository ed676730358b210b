mkdir -p gpurun_out/r04t
AB_STEPS=30 python tools/ab_run.py quadric torus:6 > gpurun_out/r04t/ab_wpe_heavy.txt 2>&1; cat gpurun_out/r04t/ab_wpe_heavy.txt
