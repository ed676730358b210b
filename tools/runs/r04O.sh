mkdir -p gpurun_out/r04O
for WH in "1280 720" "1920 1080" "2560 1440" "3840 2160"; do for PAD in 0 8192 16384 28672; do set -- $WH
  RTX_XP_LDS_PAD=$PAD python bench.py --width $1 --height $2 --no-cpu-baseline --no-smaa --steps 150 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('pad=$PAD', '$1x$2', 'kernel_ms', d['kernel_ms'], 'animated', d.get('animated',{}).get('mean_ms'))"
  done; done | tee gpurun_out/r04O/lds_pad_occupancy.txt
