mkdir -p gpurun_out/r04c
python tools/cull_audit.py --rays 2e8 --scenes 3 --out gpurun_out/r04c/audit_trial.json > gpurun_out/r04c/audit_trial.txt 2>&1
tail -80 gpurun_out/r04c/audit_trial.txt
