# Last session of round 5: random-scene sweeps on the final sources -- with a mip-mapped sky box (FUZZ_CUBE_MIPS) over three generators, and the two
# long-table generators (many tori: the START part of the tube test is in their product variant) as they are.
O=gpurun_out/${1:-r05y}; mkdir -p $O
for g in random_scene nasty_scene scaled_quat_scene; do FUZZ_CUBE_MIPS=1 FUZZ_GEN=$g timeout 900 python tools/fuzz_gpu.py 52000 ${N_CUBE:-1200} > $O/fuzz_cube_mips_$g.txt 2>&1; tail -3 $O/fuzz_cube_mips_$g.txt; done
for g in pencil_scene crowd_scene; do FUZZ_GEN=$g timeout 1200 python tools/fuzz_gpu.py 53000 ${N_LONG:-150} > $O/fuzz_$g.txt 2>&1; tail -3 $O/fuzz_$g.txt; done
