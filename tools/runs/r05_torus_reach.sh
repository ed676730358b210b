# Round 5, session 3: the torus culls as they ship -- never the ray's own limit, only the reference's t < 100 (rt_device.h RT_TORUS_REACH).
O=gpurun_out/r05c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_culls.py -x -q -m gpu -s > $O/pytest_gpu_culls.txt 2>&1; tail -3 $O/pytest_gpu_culls.txt
AB_STEPS=20 timeout 900 python tools/ab_run.py torus:6 default quadric > $O/ab_reach.txt 2>&1; cat $O/ab_reach.txt
timeout 900 python tools/cull_audit.py --rays 6e10 --families torus_far --scenes 24 --out $O/audit_torus_far.json 2>&1 | grep -v amdgpu.ids > $O/audit_torus_far.txt; head -16 $O/audit_torus_far.txt | cut -c1-200
timeout 900 python tools/cull_audit.py --rays 2.7e11 --families torus --scenes 24 --out $O/audit_torus_2e11.json 2>&1 | grep -v amdgpu.ids > $O/audit_torus_2e11.txt; grep "==" $O/audit_torus_2e11.txt
timeout 900 python tools/cull_audit.py --rays 1e11 --families tables --scenes 24 --out $O/audit_tables_1e11.json 2>&1 | grep -v amdgpu.ids > $O/audit_tables_1e11.txt; grep "==" $O/audit_tables_1e11.txt
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" > $O/pytest_gpu_all.txt; tail -3 $O/pytest_gpu_all.txt
