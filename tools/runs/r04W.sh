mkdir -p gpurun_out/r04W
timeout 300 python tools/cull_audit.py --rays 1e12 --families ring --scenes 24 --out gpurun_out/r04W/audit_ring_1e12 2>&1 | grep -v amdgpu.ids > gpurun_out/r04W/audit_ring_1e12.txt; grep "==" gpurun_out/r04W/audit_ring_1e12.txt
timeout 600 python tools/cull_audit.py --rays 3e11 --families tables --scenes 24 --out gpurun_out/r04W/audit_tables_3e11 2>&1 | grep -v amdgpu.ids > gpurun_out/r04W/audit_tables_3e11.txt; grep "==" gpurun_out/r04W/audit_tables_3e11.txt
timeout 700 python tools/cull_audit.py --rays 2.5e11 --families torus --scenes 24 --out gpurun_out/r04W/audit_torus_2e11 2>&1 | grep -v amdgpu.ids > gpurun_out/r04W/audit_torus_2e11.txt; grep "==" gpurun_out/r04W/audit_torus_2e11.txt
