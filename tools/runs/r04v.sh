mkdir -p gpurun_out/r04v
AB_STEPS=30 python tools/ab_run.py default quadric torus:6 > gpurun_out/r04v/ab_cubent.txt 2>&1
AB_STEPS=30 python tools/ab_run.py default quadric torus:6 >> gpurun_out/r04v/ab_cubent.txt 2>&1
cat gpurun_out/r04v/ab_cubent.txt
bash tools/ab_traffic.sh > gpurun_out/r04v/ab_cubent_traffic.txt 2>&1; cat gpurun_out/r04v/ab_cubent_traffic.txt
