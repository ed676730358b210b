export TMPDIR=/tmp; R=$(pwd); mkdir -p gpurun_out/r02e
python -m pytest tests/test_gpu_smaa.py -x -q 2>&1 | tail -3
for H in 2 4 8; do
  cd /tmp; RTX_SMAA_STRIP_H=$H REPS=12 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r02e/h$H -- python $R/tools/bench_smaa.py > $R/gpurun_out/r02e/bench_h$H.jsonl 2>/dev/null
  cd $R; echo "== STRIP_H $H"; python tools/smaa_trace_split.py $(find gpurun_out/r02e/h$H -name "*kernel_trace.csv") 12
done
