mkdir -p gpurun_out/r04M
for G in random_scene nasty_scene scaled_quat_scene crowd_scene pencil_scene; do
  FUZZ_GEN=$G timeout 700 python tools/fuzz_gpu.py 90000 30 3840 2160 2>&1 | grep -v amdgpu.ids | tail -2 > gpurun_out/r04M/fuzz_gpu_4k_$G.txt; cat gpurun_out/r04M/fuzz_gpu_4k_$G.txt
done
