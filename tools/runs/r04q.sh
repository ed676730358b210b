mkdir -p gpurun_out/r04q
AB_STEPS=30 python tools/ab_run.py default quadric torus:6 > gpurun_out/r04q/ab_bern.txt 2>&1; cat gpurun_out/r04q/ab_bern.txt
RTX_HIP_LIB=$PWD/raytracing_opengl_amd/variants/librtx_hip_dkbern.so python tools/dk_stats.py default torus:6 > gpurun_out/r04q/dk_stats_bern.txt 2>&1; cat gpurun_out/r04q/dk_stats_bern.txt
