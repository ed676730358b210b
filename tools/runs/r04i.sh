mkdir -p gpurun_out/r04i
python bench.py > gpurun_out/r04i/bench_n1.json 2> gpurun_out/r04i/bench_n1.err; tail -c 1500 gpurun_out/r04i/bench_n1.json; tail -3 gpurun_out/r04i/bench_n1.err
timeout 1500 python -m pytest tests -x -q -m gpu --durations=45 > gpurun_out/r04i/pytest_gpu.txt 2>&1; tail -60 gpurun_out/r04i/pytest_gpu.txt
