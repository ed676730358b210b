#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 PC sampling of the bench workload, aggregated per instruction on the box (the raw sample
# files are too large to travel).  Usage: tools/pc_sample.sh <tag> <method: host_trap|stochastic> <interval> [bench args...]
# Output: gpurun_out/pcs_<tag>/{agg.txt, head.csv, log}.  PC sampling is never combined with --pmc or any trace domain.
set -u
TAG=${1:-pcs}; METHOD=${2:-host_trap}; INTERVAL=${3:-1}
shift 3 || true
EXTRA="$*"
REPO=$(pwd)
OUT=$REPO/gpurun_out/pcs_$TAG
RAW=/tmp/pcs_raw_$TAG
mkdir -p "$OUT" "$RAW"
export TMPDIR=/tmp
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
UNIT=time
[ "$METHOD" = "stochastic" ] && UNIT=cycles
BENCH=${PROF_CMD:-"python $REPO/bench.py --no-cpu-baseline --no-smaa $EXTRA"}
cd /tmp
timeout ${PCS_TIMEOUT:-400} rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $METHOD --pc-sampling-unit $UNIT \
  --pc-sampling-interval $INTERVAL --output-format csv -d "$RAW" -- $BENCH > "$OUT/log" 2>&1
echo "rocprofv3 exit $?" >> "$OUT/log"
cd "$REPO"
find "$RAW" -type f | head -20 >> "$OUT/log"
python tools/pc_sample_agg.py "$RAW" "$OUT" >> "$OUT/log" 2>&1
tail -5 "$OUT/log"
