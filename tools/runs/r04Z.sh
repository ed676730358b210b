mkdir -p gpurun_out/r04Z
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3 | tee gpurun_out/r04Z/smoke.txt
timeout 600 python -m pytest tests -x -q -m gpu > gpurun_out/r04Z/pytest_gpu.txt 2>&1; grep -E "passed|failed|error" gpurun_out/r04Z/pytest_gpu.txt | tail -3
