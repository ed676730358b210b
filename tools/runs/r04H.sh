mkdir -p gpurun_out/r04H
for S in "quadric 4" "default 4" "torus 6"; do RTX_HIP_LIB=$PWD/raytracing_opengl_amd/variants/librtx_hip_prof.so python tools/phase_profile.py $S 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r04H/phase_profile.txt; cat gpurun_out/r04H/phase_profile.txt
