# Round 5, session 1: the near-origin length-limit rule of the torus culls (rt_device.h torus_limit).
#  1. the -m gpu cull tests (recorded far rays + short audit)
#  2. the lead family at scale: how early does the solver report, by distance of the origin -> where RT_TORUS_NEAR may sit
#  3. what the rule costs on the bench frames, by RT_TORUS_NEAR (variants built by tools/ab_build.sh: 0 = no length limit at all, 4, 8, 12; product = 6)
O=gpurun_out/r05a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_culls.py -x -q -m gpu -s > $O/pytest_gpu_culls.txt 2>&1; tail -5 $O/pytest_gpu_culls.txt
timeout 900 python tools/cull_audit.py --rays 4e10 --families torus_lead --scenes 12 --out $O/lead_4e10.json 2>&1 | grep -v amdgpu.ids > $O/lead_4e10.txt; grep -A12 "==" $O/lead_4e10.txt | cut -c1-330
AB_STEPS=20 timeout 900 python tools/ab_run.py torus:6 default > $O/ab_torus_near.txt 2>&1; cat $O/ab_torus_near.txt
