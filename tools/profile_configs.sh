#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel-trace + the PMC passes that feed profiles/valu*.json / traffic*.json for the three bench scenes
# (BASELINE configs: default 4K d4 = the metric, configs[2] quadric 4K d4, configs[3] torus 4K d6). Output: gpurun_out/prof_<round>_<scene>/.
# usage: tools/profile_configs.sh [round tag, default r04]
set -u
R=${1:-r04}
for spec in "default 4" "quadric 4" "torus 6"; do
  set -- $spec
  PROF_PASSES=min bash tools/profile_gpu.sh ${R}_$1 --scene $1 --depth $2 --steps 20 --no-smaa > gpurun_out/prof_${R}_$1.log 2>&1
  tail -30 gpurun_out/prof_${R}_$1.log
done
