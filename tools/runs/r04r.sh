mkdir -p gpurun_out/r04r
python tools/cull_audit.py --families torus,torus_margin --rays 1e11 --margin-rays 2e10 --scenes 12 --out gpurun_out/r04r/cull_audit_torus.json > gpurun_out/r04r/cull_audit_torus.txt 2>&1
grep -E "^==|VIOL|tube|largest" gpurun_out/r04r/cull_audit_torus.txt
AB_STEPS=30 python tools/ab_run.py default quadric torus:6 > gpurun_out/r04r/ab.txt 2>&1; cat gpurun_out/r04r/ab.txt
timeout 1500 python -m pytest tests -x -q -m gpu --durations=8 > gpurun_out/r04r/pytest_gpu.txt 2>&1; tail -14 gpurun_out/r04r/pytest_gpu.txt
