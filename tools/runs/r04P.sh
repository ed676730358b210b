mkdir -p gpurun_out/r04P
python tools/cull_audit.py --rays 1e11 --families quadric --scenes 12 --out gpurun_out/r04P/audit_quadric_final 2>&1 | grep -v amdgpu.ids | tail -14
AB_STEPS=30 python tools/ab_run.py quadric 2>&1 | tail -1
