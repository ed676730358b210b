mkdir -p gpurun_out/r04g
python tools/cull_audit.py --rays 1e11 --margin-rays 5e10 --scenes 12 --out gpurun_out/r04g/cull_audit.json > gpurun_out/r04g/cull_audit.txt 2>&1
grep -E "^==|VIOL|largest|early" gpurun_out/r04g/cull_audit.txt
AB_STEPS=20 python tools/ab_run.py default quadric torus:6 > gpurun_out/r04g/ab.txt 2>&1; cat gpurun_out/r04g/ab.txt
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r04g/pytest_gpu.txt 2>&1; tail -5 gpurun_out/r04g/pytest_gpu.txt
