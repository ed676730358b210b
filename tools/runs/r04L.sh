mkdir -p gpurun_out/r04L
timeout 900 python -m pytest tests/test_gpu_multi.py -x -q -m gpu > gpurun_out/r04L/pytest.txt 2>&1; tail -12 gpurun_out/r04L/pytest.txt
