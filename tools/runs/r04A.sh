mkdir -p gpurun_out/r04A
timeout 600 python -m pytest tests/test_gpu_smaa.py -x -q > gpurun_out/r04A/pytest_smaa.txt 2>&1; tail -2 gpurun_out/r04A/pytest_smaa.txt
mkdir -p /tmp/v; mv raytracing_opengl_amd/variants/librtx_hip_qsign.so /tmp/v/
bash tools/ab_smaa.sh 2>&1 | grep -E "==|traced" > gpurun_out/r04A/ab_smaa_packed_entries.txt; cat gpurun_out/r04A/ab_smaa_packed_entries.txt
mv /tmp/v/librtx_hip_qsign.so raytracing_opengl_amd/variants/
AB_STEPS=30 python tools/ab_run.py default quadric torus:6 > gpurun_out/r04A/ab_qsign.txt 2>&1
AB_STEPS=30 python tools/ab_run.py default quadric torus:6 >> gpurun_out/r04A/ab_qsign.txt 2>&1
cat gpurun_out/r04A/ab_qsign.txt
python tools/cull_audit.py --rays 2e10 --families quadric --out gpurun_out/r04A/audit_quadric_sign_exit 2>&1 | grep -v amdgpu.ids | tail -14
