# torus_far at scale: the premise behind RT_TORUS_REACH (a solve from beyond the reach reports no root below 100)
timeout 1200 python tools/cull_audit.py --rays 1e12 --families torus_far --scenes 24 --out gpurun_out/r05d/audit_torus_far_1e12.json 2>&1 | grep -v amdgpu.ids > gpurun_out/r05d_far.txt; mkdir -p gpurun_out/r05d; mv gpurun_out/r05d_far.txt gpurun_out/r05d/audit_torus_far_1e12.txt; head -16 gpurun_out/r05d/audit_torus_far_1e12.txt | cut -c1-200
# Round 5, session 4: camera-ray tile masks (rt_device.h tile_mask): audit, cost / gain against the same build without them (variant notile)
O=gpurun_out/r05d; mkdir -p $O
timeout 600 python tools/cull_audit.py --rays 2e10 --families tile --scenes 24 --out $O/audit_tile.json 2>&1 | grep -v amdgpu.ids > $O/audit_tile.txt; head -14 $O/audit_tile.txt | cut -c1-200
AB_STEPS=20 timeout 900 python tools/ab_run.py default torus:6 quadric > $O/ab_tile.txt 2>&1; cat $O/ab_tile.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_culls.py -x -q -m gpu 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" > $O/pytest_gpu_parity.txt; tail -3 $O/pytest_gpu_parity.txt
# hot rows' waves at s_setprio 3 (RTX_HOT_PRIO, rt_kernel.hip): small frames and one rank's share of a frame
for prio in 0 1 0 1; do for wh in "1920 1080" "1280 720" "640 480"; do set -- $wh; echo -n "prio=$prio $1x$2 "; RTX_HOT_PRIO=$prio python bench.py --width $1 --height $2 --steps 40 --warmup 10 --no-cpu-baseline --no-smaa 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'kernel_ms', d.get('kernel_ms'), 'parity', d.get('parity',{}).get('max_abs_diff'))"; done; done > $O/hot_prio_small_frames.txt 2>&1; cat $O/hot_prio_small_frames.txt
for prio in 0 1; do echo "RTX_HOT_PRIO=$prio"; RTX_HOT_PRIO=$prio timeout 600 python tools/time_bands.py 2>/dev/null | grep -v amdgpu; done > $O/hot_prio_bands.txt 2>&1; cat $O/hot_prio_bands.txt
# scene tables staged in LDS (north_star) against scalar loads (shipped), on this round's kernel: time, and the instruction counters of both
{ echo "LDS-staged scene tables vs scalar (SMEM) loads on the round-5 kernel, 3840x2160 depth 4, bench.py --lds {0,1} --steps 40 (kernel ms by HIP events)";
for sc in default quadric; do for lds in 0 1 0 1; do echo -n "scene=$sc lds=$lds "; python bench.py --scene $sc --lds $lds --steps 40 --warmup 10 --no-cpu-baseline --no-smaa 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kernel_ms', d.get('kernel_ms'), 'ms_per_step', d['ms_per_step'], 'scene_in_lds', d['config'].get('scene_in_lds'))"; done; done; } > $O/lds_vs_smem.txt 2>&1
R=$(pwd); export TMPDIR=/tmp; ( cd /tmp; for sc in default quadric; do for lds in 0 1; do rm -rf /tmp/ldspmc; rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d /tmp/ldspmc -- python $R/bench.py --scene $sc --lds $lds --no-cpu-baseline --no-smaa --steps 5 --warmup 2 > /tmp/ldspmc.log 2>&1
python3 - $sc $lds <<'PY'
import csv,glob,collections,sys
acc=collections.defaultdict(list)
for f in glob.glob("/tmp/ldspmc/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "rt_trace_kernel<true, false" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("scene=%s lds=%s"%(sys.argv[1],sys.argv[2]), "  ".join("%s %.2fM"%(k.replace("SQ_",""), sum(v)/len(v)/1e6) for k,v in sorted(acc.items())))
PY
done; done ) >> $O/lds_vs_smem.txt 2>&1; cat $O/lds_vs_smem.txt
