mkdir -p gpurun_out/r04L
timeout 900 python -m pytest tests/test_gpu_smaa.py tests/test_gpu_multi.py -x -q -m gpu > gpurun_out/r04L/pytest.txt 2>&1; tail -2 gpurun_out/r04L/pytest.txt
for rep in 1 2; do python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('kernel_ms', d['kernel_ms'], 'ms_per_step', d['ms_per_step'], d['value'], 'smaa', d['smaa']['ms_per_resolve'], d['smaa']['ms_per_resolve_inside_draw'], d['smaa']['roofline']['frac'])"; done | tee gpurun_out/r04L/bench.txt
