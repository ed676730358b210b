mkdir -p gpurun_out/r04a
python bench.py > gpurun_out/r04a/bench_n1.json 2> gpurun_out/r04a/bench_n1.err
tools/micro/exec_rate > gpurun_out/r04a/exec_rate.txt 2>&1
AB_STEPS=20 python tools/ab_run.py default quadric torus:6 > gpurun_out/r04a/ab_cmul.txt 2>&1
RTX_HIP_LIB=$PWD/raytracing_opengl_amd/variants/librtx_hip_dk.so python tools/dk_stats.py default torus:6 > gpurun_out/r04a/dk_stats.txt 2>&1
tail -3 gpurun_out/r04a/bench_n1.json; cat gpurun_out/r04a/ab_cmul.txt
