mkdir -p gpurun_out/r04C
for WH in "1280 720" "1920 1080" "2560 1440" "3840 2160"; do for MODE in 0 1 2 3; do set -- $WH
  RTX_HOT_WG=100000 RTX_HOT_MODE=$MODE python bench.py --width $1 --height $2 --no-cpu-baseline --steps 150 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('mode=$MODE', '$1x$2', 'kernel_ms', d['kernel_ms'], 'animated', d.get('animated',{}).get('kernel_ms_per_frame'), d.get('parity',{}).get('max_abs_diff'))"
  done; done > gpurun_out/r04C/hot_rows_interleaved.txt 2>&1
cat gpurun_out/r04C/hot_rows_interleaved.txt
RTX_HOT_WG=0 python bench.py --no-cpu-baseline --steps 150 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('4K plain order', d['kernel_ms'])"
