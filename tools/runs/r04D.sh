mkdir -p gpurun_out/r04D
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > gpurun_out/r04D/gpu_pytest.log 2>&1; tail -5 gpurun_out/r04D/gpu_pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r04D/smoke.log 2>&1; tail -2 gpurun_out/r04D/smoke.log
