mkdir -p gpurun_out/r04o
bash tools/ab_pmc.sh > gpurun_out/r04o/ab_pmc_default.txt 2>&1; cat gpurun_out/r04o/ab_pmc_default.txt
