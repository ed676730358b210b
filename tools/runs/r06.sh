#!/bin/bash
# Round 6: the GPU-box steps of the round, one parameterised script (tools/runs/r05_*.sh were one file per step).
#   gpurun -- 'bash tools/runs/r06.sh <step> [args]'      output under gpurun_out/r06_<step>/
# steps:
#   behind   the "behind" rule of the torus culls: audits (behind, behind_far, torus, tables) + what it costs (A/B against variants built by
#            tools/ab_build.sh: r5rule = -DRT_TORUS_BEHIND_RULE=0, notube = -DRT_TORUS_BACK_TUBE=0, backinf = no backward reach)
#   audit    tools/cull_audit.py <families> <rays> <scenes> [timeout] [seed0]    e.g. r06.sh audit torus_behind 1e12 24
#   ab       tools/ab_run.py <scenes...> over every library under raytracing_opengl_amd/variants/   (AB_SIZE, AB_STEPS from the environment)
#   dk       tools/dk_stats.py over counting builds <variant names>
#   sized    torus / torus_margin / torus_lead audits over the sized-torus scenes only, violations by scene
#   gpu      the -m gpu suite
#   final    the round's final measurements (profiles of the three bench scenes, every configuration's bench line, SMAA split)
#   bench    bench.py at N = 1 (the driver's line)
step=$1; shift
O=gpurun_out/r06_$step; mkdir -p $O
export PYTHONUNBUFFERED=1
case $step in
behind)
  timeout 1500 python tools/cull_audit.py --rays ${1:-3e10} --families torus_behind,torus_behind_far,torus,tables --scenes 8 --out $O/audit.json 2>&1 | grep -v amdgpu.ids | cut -c1-420 > $O/audit.txt
  grep "==\|VIOLATION\|phantom\|hits reported" $O/audit.txt | cut -c1-200
  AB_STEPS=20 timeout 900 python tools/ab_run.py default quadric torus:6 > $O/ab_4k.txt 2>&1; cat $O/ab_4k.txt
  AB_SIZE=1920x1080 AB_STEPS=40 timeout 600 python tools/ab_run.py default > $O/ab_1080p.txt 2>&1; cat $O/ab_1080p.txt
  ;;
audit)
  timeout ${4:-3000} python tools/cull_audit.py --families $1 --rays $2 --scenes ${3:-12} --seed0 ${5:-1000} --out $O/$1_$2_s${5:-1000}.json 2>&1 | grep -v amdgpu.ids | cut -c1-420 > $O/$1_$2_s${5:-1000}.txt
  grep "==\|VIOLATION\|phantom\|hits reported" $O/$1_$2_s${5:-1000}.txt | cut -c1-200
  ;;
ab)
  timeout 1500 python tools/ab_run.py "$@" > $O/ab_${AB_SIZE:-4k}.txt 2>&1; cat $O/ab_${AB_SIZE:-4k}.txt
  ;;
dk)   # solver statistics of counting builds (tools/ab_build.sh NAME_dk "-DRT_DK_STATS ..."): RTX_HIP_LIB per variant
  for v in "$@"; do echo "== $v"; RTX_HIP_LIB=raytracing_opengl_amd/variants/librtx_hip_$v.so timeout 600 python tools/dk_stats.py default torus:6; done > $O/dk.txt 2>&1; cat $O/dk.txt
  ;;
sized)  # the torus premises on tori of every size: per-scene violations (tests/random_scenes.py sized_torus_scene)
  timeout 1500 python tools/cull_audit.py --rays ${1:-2e10} --margin-rays ${1:-2e10} --families torus,torus_margin,torus_lead --scenes ${2:-16} --only sized_torus_scene --out $O/sized.json 2>&1 | grep -v amdgpu.ids | cut -c1-420 > $O/sized.txt
  grep "==\|VIOLATION\|violations by\|largest" $O/sized.txt | cut -c1-900
  ;;
gpu)
  timeout 1500 python -m pytest tests -m gpu -x -q "$@" > $O/pytest_gpu.txt 2>&1; tail -15 $O/pytest_gpu.txt
  ;;
final)  # the round's final measurements: kernel-trace + PMC passes of the three bench scenes (-> profiles/valu*.json, traffic*.json through
        # tools/prof_to_json.py), the bench line of every configuration, the driver's line under rocprofv3 (kernel-trace), the SMAA split
  bash tools/profile_configs.sh r06 > $O/profile_configs.log 2>&1; tail -5 $O/profile_configs.log
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/bench_trace -- python $GRAFT_REPO_ROOT/bench.py > $GRAFT_REPO_ROOT/$O/bench_n1_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/bench_n1_under_rocprof.err )
  python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
  python bench.py --scene quadric --no-cpu-baseline > $O/bench_n1_quadric.json 2>/dev/null
  python bench.py --scene torus --depth 6 --no-cpu-baseline > $O/bench_n1_torus.json 2>/dev/null
  python bench.py --width 1920 --height 1080 --no-cpu-baseline > $O/bench_config1_1080p.json 2>/dev/null
  python bench.py --width 7680 --height 4320 --no-cpu-baseline > $O/bench_config4_8k_1gpu.json 2>/dev/null
  python bench.py --gpus 1 --transport loopback --no-cpu-baseline --also-bands > $O/bench_loopback_multi.json 2>/dev/null
  for f in $O/bench_*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['kernel_ms'], d['roofline']['frac'], (d.get('smaa') or {}).get('ms_per_resolve'), (d.get('cpu_baseline') or {}).get('value'))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
  done
  find $O/bench_trace -name "*kernel_stats.csv" | head -2
  REPS=20 bash tools/ab_smaa.sh > $O/smaa_split.txt 2>&1; cat $O/smaa_split.txt
  ;;
bench)
  timeout 900 python bench.py "$@" > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json
  ;;
*) echo "unknown step $step"; exit 2;;
esac
