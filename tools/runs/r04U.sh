mkdir -p gpurun_out/r04U
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/r04U/smoke.txt
RTX_HIP_LIB=$PWD/raytracing_opengl_amd/variants/librtx_hip_scan.so python tools/scan_stats.py quadric 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04U/scan_stats_quadric_clipbox.txt
