mkdir -p gpurun_out/r04e
python tools/cull_audit.py --families torus_margin --margin-rays 3e10 --scenes 12 --out gpurun_out/r04e/margin.json > gpurun_out/r04e/margin.txt 2>&1
cat gpurun_out/r04e/margin.txt | grep -v first
