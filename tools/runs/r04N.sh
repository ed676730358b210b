mkdir -p gpurun_out/r04N
RTX_HIP_LIB=$PWD/raytracing_opengl_amd/variants/librtx_hip_smaa_early.so timeout 600 python -m pytest tests/test_gpu_smaa.py -x -q -m gpu 2>&1 | tail -2
bash tools/ab_smaa.sh 2>&1 | grep -E "==|traced|pattern/ULTRA" | tee gpurun_out/r04N/ab_smaa_early_atomic.txt
