mkdir -p gpurun_out/r04G
timeout 1200 python -m pytest tests/test_gpu_multi.py -x -q > gpurun_out/r04G/pytest_multi.txt 2>&1; tail -5 gpurun_out/r04G/pytest_multi.txt
python bench.py --gpus 1 --transport loopback --no-cpu-baseline --also-bands > gpurun_out/r04G/bench_loopback_multi.json 2>/dev/null
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04G/bench_loopback_multi.json') if l.startswith('{')][-1]); c=d['config']
print(d['value'], d['ms_per_step'], c['trace_ms_max_rank'], c['gather_ms'], c['gather_bytes_per_frame'], c['gather_bytes_per_pixel'], c['gather_GB_s_into_rank0'], c['also_measured'], d['parity'])
PY
