#!/bin/bash
# Runs ON THE GPU BOX: per-kernel SMAA times (rocprofv3 --kernel-trace of tools/bench_smaa.py, medians) for the product library and every
# raytracing_opengl_amd/variants/librtx_hip_smaa_*.so; byte-exactness is the GPU suite's business (tests/test_gpu_smaa.py), not this tool's.
export TMPDIR=/tmp; R=$(pwd); mkdir -p gpurun_out/absmaa
for LIB in $R/raytracing_opengl_amd/librtx_hip.so $R/raytracing_opengl_amd/variants/librtx_hip_smaa_*.so; do
  [ -f $LIB ] || continue
  V=$(basename $LIB .so); rm -rf /tmp/absmaa_$V
  cd /tmp; RTX_HIP_LIB=$LIB REPS=${REPS:-20} rocprofv3 --kernel-trace --output-format csv -d /tmp/absmaa_$V -- python $R/tools/bench_smaa.py > $R/gpurun_out/absmaa/$V.json 2>/dev/null
  cd $R; echo "== $V"; python tools/smaa_trace_split.py $(find /tmp/absmaa_$V -name "*kernel_trace.csv") ${REPS:-20}
done
