mkdir -p gpurun_out/r04T
for G in crowd_scene pencil_scene nasty_scene; do
  FUZZ_GEN=$G timeout 400 python tools/fuzz_gpu.py 120000 400 2>&1 | grep -v amdgpu.ids | tail -2 > gpurun_out/r04T/fuzz_gpu_$G.txt; cat gpurun_out/r04T/fuzz_gpu_$G.txt
done
