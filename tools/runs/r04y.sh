mkdir -p gpurun_out/r04y
bash tools/ab_smaa.sh > gpurun_out/r04y/ab_smaa_planetex.txt 2>&1; cat gpurun_out/r04y/ab_smaa_planetex.txt
