#!/bin/bash
R=$(pwd); export TMPDIR=/tmp; cd /tmp
for lib in $R/raytracing_opengl_amd/librtx_hip.so $R/raytracing_opengl_amd/variants/*.so; do
  tag=$(basename $lib .so)
  for ctr in WRITE_SIZE FETCH_SIZE; do
  rm -rf /tmp/abt_$tag
  RTX_HIP_LIB=$lib rocprofv3 --pmc $ctr --output-format csv -d /tmp/abt_$tag -- python $R/bench.py --no-cpu-baseline --steps 5 --warmup 2 > /tmp/abt_$tag.log 2>&1
  python3 - $tag <<'PY'
import csv,glob,collections,sys
acc=collections.defaultdict(list)
for f in glob.glob("/tmp/abt_%s/**/*counter_collection.csv"%sys.argv[1],recursive=True):
    for r in csv.DictReader(open(f)):
        if "rt_trace_kernel<true, false" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("%-22s"%sys.argv[1], "  ".join("%s %.1f MB"%(k, sum(v)/len(v)*1024/1e6) for k,v in sorted(acc.items())))
PY
  done
done
