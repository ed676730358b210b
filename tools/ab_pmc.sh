#!/bin/bash
# GPU box: dynamic VALU counters of the 4K default-scene kernel for the product library and every A/B variant.
R=$(pwd); export TMPDIR=/tmp; cd /tmp
for lib in $R/raytracing_opengl_amd/librtx_hip.so $R/raytracing_opengl_amd/variants/*.so; do
  tag=$(basename $lib .so); rm -rf /tmp/abpmc_$tag
  RTX_HIP_LIB=$lib rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU --output-format csv -d /tmp/abpmc_$tag -- python $R/bench.py --no-cpu-baseline --steps 5 --warmup 2 ${AB_BENCH_ARGS:-} > /tmp/abpmc_$tag.log 2>&1
  python3 - $tag <<'PY'
import csv,glob,collections,sys
acc=collections.defaultdict(list)
for f in glob.glob("/tmp/abpmc_%s/**/*counter_collection.csv"%sys.argv[1],recursive=True):
    for r in csv.DictReader(open(f)):
        if "rt_trace_kernel<true, false" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("%-22s"%sys.argv[1], "  ".join("%s %.1fM"%(k.replace("SQ_",""), sum(v)/len(v)/1e6) for k,v in sorted(acc.items())))
PY
done
