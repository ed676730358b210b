mkdir -p gpurun_out/r04I
for rep in 1 2 3; do for LIB in raytracing_opengl_amd/librtx_hip.so raytracing_opengl_amd/variants/librtx_hip_u8plain.so; do
RTX_HIP_LIB=$PWD/$LIB python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$LIB'.split('/')[-1], 'kernel_ms', d['kernel_ms'], 'ms_per_step', d['ms_per_step'], 'smaa', d['smaa']['ms_per_resolve'], d['smaa']['ms_per_resolve_inside_draw'])"
done; done | tee gpurun_out/r04I/u8_plain_store.txt
