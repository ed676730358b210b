mkdir -p gpurun_out/r04n
AB_STEPS=30 python tools/ab_run.py default quadric torus:6 > gpurun_out/r04n/ab_others.txt 2>&1; cat gpurun_out/r04n/ab_others.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_fuzz_scenes.py tests/test_gpu_widened.py -x -q -m gpu > gpurun_out/r04n/pytest.txt 2>&1; tail -8 gpurun_out/r04n/pytest.txt
