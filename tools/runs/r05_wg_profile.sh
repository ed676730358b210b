# Round 5, session 6: what a 4K launch looks like over time (tools/wg_times.py on a device-wide clock), with and without the camera tile masks
O=gpurun_out/r05f; mkdir -p $O
for v in wgtimes wgtimes_notile; do echo "== $v"; RTX_HIP_LIB=raytracing_opengl_amd/variants/librtx_hip_$v.so WG_WORLDS=1,8 python tools/wg_times.py 2>/dev/null | grep -v amdgpu; done > $O/wg_profile.txt 2>&1; cat $O/wg_profile.txt | cut -c1-400
