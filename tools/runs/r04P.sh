mkdir -p gpurun_out/r04P
python tools/cull_audit.py --rays 1e11 --families quadric --out gpurun_out/r04P/audit_quadric_final 2>&1 | grep -v amdgpu.ids | tail -12
