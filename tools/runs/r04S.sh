mkdir -p gpurun_out/r04S
python bench.py > gpurun_out/r04S/bench_n1.json 2> gpurun_out/r04S/bench_n1.err; tail -c 400 gpurun_out/r04S/bench_n1.json
python bench.py --scene quadric --no-cpu-baseline > gpurun_out/r04S/bench_n1_quadric.json 2>/dev/null
python bench.py --scene torus --depth 6 --no-cpu-baseline > gpurun_out/r04S/bench_n1_torus.json 2>/dev/null
timeout 900 python tools/cull_audit.py --rays 1e11 --scenes 12 --out gpurun_out/r04S/audit_final_sources 2>&1 | grep -v amdgpu.ids > gpurun_out/r04S/audit_final_sources.txt; grep "==" gpurun_out/r04S/audit_final_sources.txt
for G in crowd_scene pencil_scene; do
  FUZZ_GEN=$G timeout 420 python tools/fuzz_gpu.py 90000 5 3840 2160 2>&1 | grep -v amdgpu.ids | tail -2 > gpurun_out/r04S/fuzz_gpu_4k_$G.txt; cat gpurun_out/r04S/fuzz_gpu_4k_$G.txt
done
