mkdir -p gpurun_out/r04V
timeout 900 python tools/cull_audit.py --rays 1e12 --families quadric --scenes 24 --out gpurun_out/r04V/audit_quadric_1e12 2>&1 | grep -v amdgpu.ids > gpurun_out/r04V/audit_quadric_1e12.txt; cat gpurun_out/r04V/audit_quadric_1e12.txt | head -16
