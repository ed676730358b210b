mkdir -p gpurun_out/r04w
timeout 900 python -m pytest tests/test_gpu_smaa.py -x -q > gpurun_out/r04w/pytest.txt 2>&1; tail -3 gpurun_out/r04w/pytest.txt
bash tools/ab_smaa.sh > gpurun_out/r04w/ab_smaa_roles.txt 2>&1; cat gpurun_out/r04w/ab_smaa_roles.txt
