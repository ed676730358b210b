# The START part of the torus tube cull (rays that start on the inner half of a torus): A/B against the build without it (variants/nostart) and
# with it in the default kernel variant too (variants/startlight), at 4K and at 1920x1080; then the torus family of the cull audit.
O=gpurun_out/${AB_TAG:-r05u_start}; mkdir -p $O
AB_STEPS=20 timeout 900 python tools/ab_run.py default torus:6 > $O/ab_4k.txt 2>&1; cat $O/ab_4k.txt
AB_SIZE=1920x1080 AB_STEPS=40 timeout 600 python tools/ab_run.py default torus:6 > $O/ab_1080.txt 2>&1; cat $O/ab_1080.txt
timeout 1500 python tools/cull_audit.py --rays ${AUDIT_RAYS:-1e11} --families torus --out $O/audit_torus.json > $O/audit_torus.txt 2>&1; tail -25 $O/audit_torus.txt
