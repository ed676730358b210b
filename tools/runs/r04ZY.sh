mkdir -p gpurun_out/r04ZY
CULL_AUDIT_LIB=$PWD/raytracing_opengl_amd/variants/libcull_audit_far12.audit timeout 290 python tools/cull_audit.py --rays 1.2e11 --families torus --scenes 24 --out gpurun_out/r04ZY/audit_torus_far12 2>&1 | grep -v amdgpu.ids > gpurun_out/r04ZY/audit_torus_far12.txt; head -20 gpurun_out/r04ZY/audit_torus_far12.txt | cut -c1-150
