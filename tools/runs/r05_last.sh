# Last session of round 5, after the profiles were re-stamped for hash 64976f36: the cube-mip GPU tests incl. the fuzz, the bench lines again (now with
# roofline.traffic: the PMC figures' hash matches), the torus family of the cull audit at the size of the round's earlier record.
O=gpurun_out/${1:-r05x}; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_cube_mips.py -m gpu -q --durations=5 2>&1 | tail -12 > $O/pytest_cube_mips.txt; cat $O/pytest_cube_mips.txt
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python bench.py --scene quadric --no-cpu-baseline > $O/bench_n1_quadric.json 2>/dev/null
python bench.py --scene torus --depth 6 --no-cpu-baseline > $O/bench_n1_torus.json 2>/dev/null
python bench.py --width 1920 --height 1080 --no-cpu-baseline > $O/bench_config1_1080p.json 2>/dev/null
for f in $O/bench_*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['kernel_ms'], d['roofline'])
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
timeout 1500 python tools/cull_audit.py --rays ${AUDIT_RAYS:-3e11} --families torus --scenes 24 --out $O/audit_torus.json > $O/audit_torus.txt 2>&1; tail -22 $O/audit_torus.txt
