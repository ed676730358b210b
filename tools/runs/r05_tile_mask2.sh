# Round 5, session 5: the timing half of session 4 again, on a build whose s_setprio does not clobber the scalar loads
O=gpurun_out/r05e; mkdir -p $O
AB_STEPS=20 timeout 900 python tools/ab_run.py default torus:6 quadric > $O/ab_tile.txt 2>&1; cat $O/ab_tile.txt
for prio in 0 1 0 1; do for wh in "1920 1080" "1280 720" "640 480" "3840 2160"; do set -- $wh; echo -n "prio=$prio $1x$2 "; RTX_HOT_PRIO=$prio python bench.py --width $1 --height $2 --steps 40 --warmup 10 --no-cpu-baseline --no-smaa 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'kernel_ms', d.get('kernel_ms'))"; done; done > $O/hot_prio_small_frames.txt 2>&1; cat $O/hot_prio_small_frames.txt
for prio in 0 1; do echo "RTX_HOT_PRIO=$prio"; RTX_HOT_PRIO=$prio timeout 600 python tools/time_bands.py 2>/dev/null | grep -v amdgpu; done > $O/hot_prio_bands.txt 2>&1; cat $O/hot_prio_bands.txt | cut -c1-300
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
