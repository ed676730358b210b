mkdir -p gpurun_out/r04Y
AB_STEPS=30 python tools/ab_run.py torus:6 default quadric > gpurun_out/r04Y/ab_torus_limit_widen.txt 2>&1
cat gpurun_out/r04Y/ab_torus_limit_widen.txt
CULL_AUDIT_LIB=$PWD/raytracing_opengl_amd/variants/libcull_audit_tlw125.audit timeout 700 python tools/cull_audit.py --rays 2.5e11 --families torus --scenes 24 --out gpurun_out/r04Y/audit_torus_2e11_widen125 2>&1 | grep -v amdgpu.ids > gpurun_out/r04Y/audit_torus_2e11_widen125.txt; head -20 gpurun_out/r04Y/audit_torus_2e11_widen125.txt | cut -c1-150
