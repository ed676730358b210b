# ON THE GPU BOX: the round's final measurements (kernel-trace + PMC passes of the three bench scenes, the bench lines of every configuration,
# the SMAA split). usage: bash tools/runs/r05_final.sh [tag]
T=${1:-r05z}; O=gpurun_out/$T; mkdir -p $O
bash tools/profile_configs.sh r05 > $O/profile_configs.log 2>&1; tail -5 $O/profile_configs.log
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/bench_trace -- python $GRAFT_REPO_ROOT/bench.py > $GRAFT_REPO_ROOT/$O/bench_n1_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/bench_n1_under_rocprof.err; cd $GRAFT_REPO_ROOT
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python bench.py --scene quadric --no-cpu-baseline > $O/bench_n1_quadric.json 2>/dev/null
python bench.py --scene torus --depth 6 --no-cpu-baseline > $O/bench_n1_torus.json 2>/dev/null
python bench.py --width 1920 --height 1080 --no-cpu-baseline > $O/bench_config1_1080p.json 2>/dev/null
python bench.py --width 7680 --height 4320 --no-cpu-baseline > $O/bench_config4_8k_1gpu.json 2>/dev/null
python bench.py --gpus 1 --transport loopback --no-cpu-baseline --also-bands > $O/bench_loopback_multi.json 2>/dev/null
for f in $O/bench_*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['kernel_ms'], d['roofline']['frac'], (d.get('smaa') or {}).get('ms_per_resolve'))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
find $O/bench_trace -name "*kernel_stats.csv" | head -2
REPS=20 bash tools/ab_smaa.sh > $O/smaa_split.txt 2>&1; cat $O/smaa_split.txt
