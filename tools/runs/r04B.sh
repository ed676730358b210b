mkdir -p gpurun_out/r04B
for WH in "2560 1440" "3200 1800" "3840 2160"; do for HW in 0 100000; do set -- $WH
  for rep in 1 2; do RTX_HOT_WG=$HW python bench.py --width $1 --height $2 --no-cpu-baseline --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('hot_wg=$HW', '$1x$2', 'kernel_ms', d['kernel_ms'], 'ms_per_step', d['ms_per_step'], 'animated', d.get('animated',{}).get('kernel_ms_per_frame'))"
  done; done; done > gpurun_out/r04B/hot_rows_large_frames.txt 2>&1
cat gpurun_out/r04B/hot_rows_large_frames.txt
for HW in 0 100000; do RTX_HOT_WG=$HW python bench.py --width 1920 --height 1080 --no-cpu-baseline --steps 200 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('hot_wg=$HW 1080p animated', d.get('animated',{}).get('kernel_ms_per_frame'), d['parity'] if 'parity' in d else '')"; done | tee gpurun_out/r04B/hot_rows_1080p_animated.txt
