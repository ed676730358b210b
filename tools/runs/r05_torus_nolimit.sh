# Round 5, session 2: the torus culls without any length limit (rt_device.h torus_cull).
#  1. the -m gpu cull tests; 2. bench frames against round 4's behaviour (variant nearinf = limit everywhere) -- time and frame hash;
#  3. the torus audit at round 4's size and seeds (2.7e11 rays over 101 scenes: the run that met the four offenders) and the candidate tables;
#  4. the whole -m gpu suite
O=gpurun_out/r05b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_culls.py -x -q -m gpu -s > $O/pytest_gpu_culls.txt 2>&1; tail -3 $O/pytest_gpu_culls.txt
AB_STEPS=20 timeout 900 python tools/ab_run.py torus:6 default quadric > $O/ab_nolimit.txt 2>&1; cat $O/ab_nolimit.txt
timeout 900 python tools/cull_audit.py --rays 2.7e11 --families torus --scenes 24 --out $O/audit_torus_2e11.json 2>&1 | grep -v amdgpu.ids > $O/audit_torus_2e11.txt; grep "==" $O/audit_torus_2e11.txt
timeout 900 python tools/cull_audit.py --rays 1e11 --families tables --scenes 24 --out $O/audit_tables_1e11.json 2>&1 | grep -v amdgpu.ids > $O/audit_tables_1e11.txt; grep "==" $O/audit_tables_1e11.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu_all.txt 2>&1; tail -3 $O/pytest_gpu_all.txt
