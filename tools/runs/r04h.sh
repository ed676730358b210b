mkdir -p gpurun_out/r04h
python tools/cull_audit.py --rays 1e11 --margin-rays 5e10 --scenes 12 --out gpurun_out/r04h/cull_audit.json > gpurun_out/r04h/cull_audit.txt 2>&1
grep -E "^==|VIOL|largest|early" gpurun_out/r04h/cull_audit.txt
