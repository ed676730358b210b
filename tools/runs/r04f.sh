mkdir -p gpurun_out/r04f
python - <<'PY'
import sys, json
sys.path.insert(0,'tools'); sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import cull_audit, ctypes, struct
lib=ctypes.CDLL('tools/audit/libcull_audit.so')
lib.cull_audit_run.restype = ctypes.c_int
lib.cull_audit_run.argtypes = [ctypes.POINTER(cull_audit.Defines), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64), ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_float), ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
lib.cull_audit_error.restype = ctypes.c_char_p
for name, sc in cull_audit.scene_list(12):
    if sc.defines[4] == 0: continue
    counters=(ctypes.c_uint64*48)(); rows=[]
    cull_audit.run(lib, sc, 'torus_margin', 1e8, 7, counters, rows, max_bad=4)
    c=list(counters)
    print(name, 'tori', sc.defines[4], 'hits', c[1], 'pt>=0.1', c[18], 't>=4', sum(c[23:28]))
    for r in rows:
        if r[0]==99.0:
            i=int(r[1]); rec=sc.blocks['toruses_buf'][112*i:112*i+112]
            print('    ', [round(x,5) for x in r[:11]], 'torus quat,pos,R,r', struct.unpack('<4f4f2f', rec[64:104]))
PY
