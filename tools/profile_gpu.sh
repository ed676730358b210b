#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats + PMC passes of the bench workload.
# Outputs under gpurun_out/prof_<tag>/ ; copy the summaries worth keeping into profiles/.
# PMC passes are separate runs with --pmc only (never combined with trace domains other than kernel-trace).
set -u
TAG=${1:-r01}
shift || true
EXTRA="$*"
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH=${PROF_CMD:-"python $REPO/bench.py --no-cpu-baseline $EXTRA"}
cd /tmp
echo "== kernel trace + stats"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1
pass() {  # name, counters...
  local name=$1; shift
  echo "== pmc $name: $*"
  rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -- $BENCH > "$OUT/pmc_$name.log" 2>&1
}
if [ "${PROF_PASSES:-all}" = "all" ]; then
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS
pass sq2 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_WR
pass cache SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_IFETCH
pass level SQ_IFETCH_LEVEL SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD
fi
pass valu SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
# instruction classes, for the mix-weighted issue ceiling (tools/prof_to_json.py -> bench.py roofline.valu)
pass mix SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64
pass fetch FETCH_SIZE GRBM_GUI_ACTIVE
pass write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
cd "$REPO"
python tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
