# A/B of whatever is under raytracing_opengl_amd/variants/ against the product library: AB_SCENES (default: the three bench scenes), 4K
O=gpurun_out/${AB_TAG:-r05ab}; mkdir -p $O
AB_STEPS=${AB_STEPS:-20} timeout 1200 python tools/ab_run.py ${AB_SCENES:-default torus:6 quadric} > $O/ab.txt 2>&1; cat $O/ab.txt
