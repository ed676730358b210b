mkdir -p gpurun_out/r04k
export TMPDIR=/tmp; export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
(cd /tmp && rocprofv3 -L 2>&1 | grep -i -A12 "pc.sampl" | head -40) > gpurun_out/r04k/pcs_avail.txt 2>&1; cat gpurun_out/r04k/pcs_avail.txt
for cfg in "host_trap 1" "host_trap 10" "stochastic 65536"; do
  set -- $cfg
  RTX_HIP_LIB=$PWD/raytracing_opengl_amd/variants/librtx_hip_prof.so PCS_TIMEOUT=300 bash tools/pc_sample.sh default_$1_$2 $1 $2 --steps 300 --warmup 5 2>&1 | tail -3
  ls gpurun_out/pcs_default_$1_$2; head -c 600 gpurun_out/pcs_default_$1_$2/log
  if [ -s gpurun_out/pcs_default_$1_$2/agg.txt ] && [ $(wc -l < gpurun_out/pcs_default_$1_$2/agg.txt) -gt 5 ]; then break; fi
done
