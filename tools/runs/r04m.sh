mkdir -p gpurun_out/r04m
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "contiguous_row_ranges or row_bands" 2>&1 | tail -3
python tools/time_bands.py 3840 2160 4 > gpurun_out/r04m/time_bands_4k.txt 2>&1; cat gpurun_out/r04m/time_bands_4k.txt
python tools/time_bands.py 7680 4320 4 > gpurun_out/r04m/time_bands_8k.txt 2>&1; cat gpurun_out/r04m/time_bands_8k.txt
