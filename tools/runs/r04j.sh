mkdir -p gpurun_out/r04j
timeout 900 python -m pytest tests/test_gpu_multi.py tests/test_gpu_parity.py::test_error_behaviour -x -q > gpurun_out/r04j/pytest_multi.txt 2>&1; tail -15 gpurun_out/r04j/pytest_multi.txt
python bench.py --gpus 1 --transport loopback --no-cpu-baseline > gpurun_out/r04j/bench_loopback.json 2> gpurun_out/r04j/bench_loopback.err; tail -c 2500 gpurun_out/r04j/bench_loopback.json; tail -3 gpurun_out/r04j/bench_loopback.err
RTX_HIP_LIB=$PWD/raytracing_opengl_amd/variants/librtx_hip_prof.so PCS_TIMEOUT=300 bash tools/pc_sample.sh default stochastic 1048576 --steps 200 --warmup 5 2>&1 | tail -5
ls -la gpurun_out/pcs_default | head; head -60 gpurun_out/pcs_default/agg.txt
