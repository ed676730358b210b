mkdir -p gpurun_out/r04p
timeout 900 python -m pytest tests/test_gpu_smaa.py tests/test_gpu_multi.py -x -q > gpurun_out/r04p/pytest.txt 2>&1; tail -4 gpurun_out/r04p/pytest.txt
python bench.py --no-cpu-baseline > gpurun_out/r04p/bench.json 2> gpurun_out/r04p/bench.err; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04p/bench.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['kernel_ms'], d['smaa'])
PY
AB_STEPS=30 python tools/ab_run.py default > gpurun_out/r04p/ab.txt 2>&1; cat gpurun_out/r04p/ab.txt
