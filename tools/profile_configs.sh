#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel-trace + the PMC passes that feed profiles/valu*.json / traffic*.json for the three bench scenes
# (BASELINE configs: default 4K d4 = the metric, configs[2] quadric 4K d4, configs[3] torus 4K d6). Output: gpurun_out/prof_r03_<scene>/.
set -u
for spec in "default 4" "quadric 4" "torus 6"; do
  set -- $spec
  PROF_PASSES=min bash tools/profile_gpu.sh r03_$1 --scene $1 --depth $2 --steps 20 --no-smaa > gpurun_out/prof_r03_$1.log 2>&1
  tail -30 gpurun_out/prof_r03_$1.log
done
