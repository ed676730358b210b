mkdir -p gpurun_out/r04X
AB_STEPS=30 python tools/ab_run.py quadric default torus:6 > gpurun_out/r04X/ab_qboxrec.txt 2>&1
AB_STEPS=30 python tools/ab_run.py quadric default torus:6 >> gpurun_out/r04X/ab_qboxrec.txt 2>&1
cat gpurun_out/r04X/ab_qboxrec.txt
