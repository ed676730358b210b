mkdir -p gpurun_out/r04b
AB_STEPS=20 python tools/ab_run.py default quadric torus:6 > gpurun_out/r04b/ab_rebase.txt 2>&1
bash tools/ab_traffic.sh > gpurun_out/r04b/ab_traffic.txt 2>&1
cat gpurun_out/r04b/ab_rebase.txt gpurun_out/r04b/ab_traffic.txt
