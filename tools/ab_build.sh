#!/bin/bash
# Build A/B variants of librtx_hip.so into raytracing_opengl_amd/variants/ (git-ignored, shipped by gpurun).
#   tools/ab_build.sh name1 "-DFLAG_A" name2 "-DFLAG_B -DRT_WAVES_PER_EU=3" ...
# The variant's flags are appended to the product flags (a later -D/-f wins).
set -e
cd "$(dirname "$0")/.."
mkdir -p raytracing_opengl_amd/variants
cfg() { sed -n "s/^$1[ \t]*?=[ \t]*//p" raytracing_opengl_amd/kernel_build.cfg; }   # the product's own configuration
BASE="-DRT_WAVES_PER_EU=$(cfg WAVES_PER_EU) -DRT_WPE_HEAVY=$(cfg WPE_HEAVY) --offload-arch=gfx950 $(cfg KERNEL_FLAGS) -Iinclude -Iraytracing_opengl_amd/csrc"
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc $BASE $flags -shared -o raytracing_opengl_amd/variants/librtx_hip_$name.so \
      raytracing_opengl_amd/csrc/rt_kernel.hip raytracing_opengl_amd/csrc/smaa_kernel.hip raytracing_opengl_amd/csrc/bands_kernel.hip -x hip raytracing_opengl_amd/csrc/rtx_capi.cpp -ldl 2>&1 | grep -v "warning\|^$" | head -5; echo "built $name" ) &
done
wait
