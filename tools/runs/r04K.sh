mkdir -p gpurun_out/r04K
for rep in 1 2; do for M in ext marker; do
if [ $M = marker ]; then export RTX_MARKER_EVENTS=1; else unset RTX_MARKER_EVENTS; fi
python bench.py --no-cpu-baseline --steps 200 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$M', 'kernel_ms', d['kernel_ms'], 'ms_per_step', d['ms_per_step'], d['value'], d['smaa']['ms_per_resolve'], d['smaa']['ms_per_resolve_inside_draw'], d['animated']['mean_ms'])"; done; done | tee gpurun_out/r04K/ext_launch_events.txt
unset RTX_MARKER_EVENTS
