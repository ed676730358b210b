mkdir -p gpurun_out/r04x
for P in ULTRA HIGH MEDIUM; do RTX_HIP_LIB=$PWD/raytracing_opengl_amd/variants/librtx_hip_smaa_phases.so python tools/smaa_role_times.py $P 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r04x/role_times.txt; cat gpurun_out/r04x/role_times.txt
