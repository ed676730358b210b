#!/bin/bash
# Runs ON THE GPU BOX: per-kernel SMAA times for the shipped library and the timing-ablation builds in raytracing_opengl_amd/variants/
export TMPDIR=/tmp; R=$(pwd); mkdir -p gpurun_out/abl
for V in base abl1 abl16 abl18 abl48; do
  LIB=$R/raytracing_opengl_amd/librtx_hip.so; [ $V != base ] && LIB=$R/raytracing_opengl_amd/variants/librtx_$V.so
  [ -f $LIB ] || continue
  cd /tmp; RTX_HIP_LIB=$LIB REPS=12 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/abl/$V -- python $R/tools/bench_smaa.py > /dev/null 2>&1
  cd $R; echo "== $V"; python tools/smaa_trace_split.py $(find gpurun_out/abl/$V -name "*kernel_trace.csv") 12 | grep "LOW"
done
