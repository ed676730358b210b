mkdir -p gpurun_out/r04z
python tools/fuzz_smaa_gpu.py 70000 600 > gpurun_out/r04z/fuzz_smaa.txt 2>&1; tail -2 gpurun_out/r04z/fuzz_smaa.txt
for G in random_scene nasty_scene scaled_quat_scene crowd_scene pencil_scene; do
  FUZZ_GEN=$G timeout 900 python tools/fuzz_gpu.py 50000 1200 > gpurun_out/r04z/fuzz_gpu_$G.txt 2>&1; tail -2 gpurun_out/r04z/fuzz_gpu_$G.txt
done
