# ON THE GPU BOX: per-kernel SMAA times of the product library and every variants/librtx_hip_smaa_*.so at strip heights 8 and 16 (RTX_SMAA_STRIP_H)
export TMPDIR=/tmp; R=$(pwd); O=$R/gpurun_out/${TAG:-r05_smaa_strips}; mkdir -p $O
for LIB in $R/raytracing_opengl_amd/librtx_hip.so $R/raytracing_opengl_amd/variants/librtx_hip_smaa_*.so; do
  [ -f $LIB ] || continue
  for H in ${HEIGHTS:-8 16}; do
    V=$(basename $LIB .so)_h$H; rm -rf /tmp/absmaa_$V
    cd /tmp; RTX_SMAA_STRIP_H=$H RTX_HIP_LIB=$LIB REPS=${REPS:-20} ONLY=${ONLY:-} rocprofv3 --kernel-trace --output-format csv -d /tmp/absmaa_$V -- python $R/tools/bench_smaa.py > $O/$V.json 2>/dev/null
    cd $R; echo "== $V"; python tools/smaa_trace_split.py $(find /tmp/absmaa_$V -name "*kernel_trace.csv") ${REPS:-20} | grep -E "${SHOW:-LOW|ULTRA}"
  done
done
