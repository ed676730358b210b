mkdir -p gpurun_out/r04d
python tools/cull_audit.py --rays 1e11 --margin-rays 5e10 --scenes 12 --out gpurun_out/r04d/cull_audit.json > gpurun_out/r04d/cull_audit.txt 2>&1
grep -E "^==|VIOL|largest" gpurun_out/r04d/cull_audit.txt
