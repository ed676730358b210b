mkdir -p gpurun_out/r04E
python bench.py > gpurun_out/r04E/bench_n1.json 2> gpurun_out/r04E/bench_n1.err; tail -c 600 gpurun_out/r04E/bench_n1.json
python bench.py --scene quadric --no-cpu-baseline > gpurun_out/r04E/bench_n1_quadric.json 2>/dev/null
python bench.py --scene torus --depth 6 --no-cpu-baseline > gpurun_out/r04E/bench_n1_torus.json 2>/dev/null
