mkdir -p gpurun_out/r04s
python tools/cull_audit.py --rays 1e11 --margin-rays 4e10 --scenes 12 --out gpurun_out/r04s/cull_audit.json > gpurun_out/r04s/cull_audit.txt 2>&1
grep -E "^==|VIOL|tube|largest" gpurun_out/r04s/cull_audit.txt
AB_STEPS=30 python tools/ab_run.py torus:6 > gpurun_out/r04s/ab.txt 2>&1; cat gpurun_out/r04s/ab.txt
