#!/bin/bash
# Runs ON THE GPU BOX: PMC passes of ONE SMAA configuration (default traced:ULTRA), per kernel means -> stdout
export TMPDIR=/tmp; R=$(pwd); CFG=${1:-traced:ULTRA}
pass() { local name=$1; shift; rm -rf /tmp/smaapmc_$name; cd /tmp; ONLY=$CFG REPS=12 rocprofv3 --pmc "$@" --output-format csv -d /tmp/smaapmc_$name -- python $R/tools/bench_smaa.py > /dev/null 2>&1; cd $R
  python3 - /tmp/smaapmc_$name <<'PY'
import csv,glob,collections,sys
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "smaa" in k: acc[k.split("::")[1].split("(")[0].split("<")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    print("%-22s"%k, "  ".join("%s %.0f"%(c.replace("SQ_",""), sorted(x)[len(x)//2]) for c,x in sorted(v.items())))
PY
}
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS
pass b SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS
pass c TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
