mkdir -p gpurun_out/r04R
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r04R/pytest_gpu.txt 2>&1; tail -3 gpurun_out/r04R/pytest_gpu.txt
sed -e 's/r04u/r04R/g' tools/runs/r04u.sh > /tmp/r04R_body.sh; bash /tmp/r04R_body.sh
