#!/usr/bin/env python3
"""GPU audit of the tracer's culls (tools/audit/cull_audit.hip): random and near-boundary rays per primitive record, the
product's cull and the literal intersector in the same lane, "culled and hit" counted -- at 1e10 rays per family.

    python tools/cull_audit.py --rays 1e10 --out profiles/r04_cull_audit.json         (GPU box; about 25 GPU-minutes, most of it Durand-Kerner)
    python tools/cull_audit.py --rays 2e7 --families quadric,ring                     (a quick pass)

Scenes: the three bench scenes and seeded scenes of every generator of tests/random_scenes.py (random / nasty / scaled-quaternion /
crowd / pencil / sized-torus), so the records include rotated, non-unit-quaternion, degenerate, open-clip-box and far-away primitives. The rays of a
family are split evenly over the scenes that have primitives of that family."""
import argparse
import ctypes
import json
import os
import struct
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

BLOCKS = ("scene_buf", "spheres_buf", "planes_buf", "surfaces_buf", "boxes_buf", "toruses_buf", "rings_buf", "lights_point_buf", "lights_direct_buf")
FAMILIES = {"torus": 0, "torus_margin": 1, "quadric": 2, "ring": 3, "tables": 4, "torus_lead": 5, "torus_far": 6, "torus_behind": 7, "torus_behind_far": 8}
N_COUNTERS = 128
LEAD_BINS = ("<2", "2..4", "4..6", "6..8", "8..10", "10..12", "12..16", "16..24", "24..48", ">=48")      # |o|: origin to torus centre
FAR_BINS = ("< 120", "120..150", "150..200", "200..400", "400..1000", ">= 1000")
NEEDS = {"torus": 4, "torus_margin": 4, "torus_lead": 4, "torus_far": 4, "torus_behind": 4, "torus_behind_far": 4, "quadric": 2, "ring": 5}      # index into defines of the count that must be > 0
LABELS = {
    "torus": {0: "rays", 1: "culled by any test", 2: "sphere cull", 3: "group sphere", 4: "convex-hull cull", 5: "puck / hole cull", 6: "the ray up to the reference's reach (t < 100) stays >= 6 mm clear of the real tube (exact)",
              7: "solver runs", 8: "solver hits among them", 16: "rays with a non-unit direction", 17: "tube (Bernstein) cull, behind the puck test",
              18: "VIOLATIONS tube (Bernstein) cull",
              10: "VIOLATIONS sphere cull", 11: "VIOLATIONS group sphere", 12: "VIOLATIONS convex-hull cull", 13: "VIOLATIONS puck / hole cull",
              14: "VIOLATIONS clearance premise: >= 6 mm clear of the real tube, yet a hit is reported", 15: "VIOLATIONS a non-unit direction was culled"},
    "torus_margin": {0: "rays (every one solved)", 1: "hits reported", 2: "hits whose ray touches the real tube", 3: "phantom hits, clearance < 1e-6", 4: "1e-6 .. 1e-5",
                     5: "1e-5 .. 1e-4", 6: "1e-4 .. 1e-3", 7: "1e-3 .. 1e-2", 8: "1e-2 .. 1e-1", 9: "1e-1 .. 1", 10: "1 .. 10", 11: ">= 10",
                     12: "VIOLATIONS phantom hits whose ray clears the tube by more than the culls' inflation (the tube's own: rt_pack.h rinf - r)",
                     13: "hit point within 1e-5 of the surface", 14: "1e-5 .. 1e-4", 15: "1e-4 .. 1e-3", 16: "1e-3 .. 1e-2", 17: "1e-2 .. 1e-1", 18: ">= 1e-1",
                     22: "hits with t < 4", 23: "4 .. 8", 24: "8 .. 16", 25: "16 .. 32", 26: "32 .. 64", 27: ">= 64",
                     28: "reported more than 1e-3 t + 0.01 before the ray enters the inflated tube: t < 4", 29: "... 4 .. 8", 30: "... 8 .. 16", 31: "... 16 .. 32",
                     32: "... 32 .. 64", 33: "... >= 64", 40: "reported more than 1e-3 t + 0.01 + 0.025 (t - 8) early, t > 8", 41: "reported more than 1 early", 42: "more than 5 early"},
    "torus_lead": dict([(0, "rays (every one solved; origins 1.5 .. 64 from the centre, aimed at the tube's surface, half of them grazing)"), (1, "hits reported")] +
                       [(90 + b, f"VIOLATIONS |o| {LEAD_BINS[b]}: a hit although the ray never enters the inflated tube") for b in range(10)]),
    "torus_far": dict([(0, "rays that enter the inflated torus beyond the reference's own reach (t < 100; RT_TORUS_REACH), every one culled by torus_cull and SOLVED"),
                       (10, "VIOLATIONS a root below the ray's limit is reported")] +
                      [(20 + b, f"rays from {n} units out") for b, n in enumerate(("< 120", "120..150", "150..200", "200..400", "400..1000", ">= 1000"))] +
                      [(30 + b, f"hits reported from {n} units out") for b, n in enumerate(("< 120", "120..150", "150..200", "200..400", "400..1000", ">= 1000"))]),
    "torus_behind": dict([(0, "rays that point AWAY from a torus their backward extension goes through (origins 1.5 .. 100 from the centre), every one solved"),
                          (1, "culled by the product's composition"), (2, "hits reported")] +
                         [(20 + b, f"rays from |o| {LEAD_BINS[b]}") for b in range(10)] +
                         [(30 + b, f"phantom hits from |o| {LEAD_BINS[b]}: a hit is reported although the half-line clears the real tube by more than 1 mm") for b in range(10)] +
                         [(40 + b, f"VIOLATIONS |o| {LEAD_BINS[b]}: a phantom hit the product culls (does not reproduce)") for b in range(10)]),
    "torus_behind_far": dict([(0, "rays that point AWAY from a torus their backward extension goes through, origins 104 .. 3000+ units out, every one solved"),
                              (1, "culled by the product's composition")] +
                             [(20 + b, f"rays from {n} units out") for b, n in enumerate(FAR_BINS)] +
                             [(30 + b, f"hits reported (phantoms by construction) from {n} units out") for b, n in enumerate(FAR_BINS)] +
                             [(40 + b, f"VIOLATIONS from {n} units out: a phantom hit the product culls") for b, n in enumerate(FAR_BINS)]),
    "quadric": {0: "rays", 1: "culled by surface_cull", 2: "culled by the group test", 3: "literal hits", 4: "literal hits on the degenerate branch", 5: "left early by the product intersector (no real root)", 6: "culled by the clip-box test behind surface_cull",
                10: "VIOLATIONS surface_cull", 11: "VIOLATIONS group test", 12: "VIOLATIONS product intersector != literal rt.frag:513-572", 13: "VIOLATIONS clip-box test"},
    "ring": {0: "rays", 1: "culled", 2: "literal hits", 10: "VIOLATIONS"},
    "tables": {0: "rays", 1: "camera-pencil rays", 2: "light-pencil rays", 3: "slab-table rays", 4: "rays whose mask has every bit set", 5: "set bits", 6: "quadric checks (clear bit)",
               7: "torus checks (clear bit)", 8: "torus checks from a far origin (the 'behind' rule: the line's backward half as well)",
               10: "VIOLATIONS quadric: bit clear, literal intersector hits", 11: "VIOLATIONS torus: bit clear, the ray up to the reference's reach (t < 100) comes within 5 mm of the real tube",
               12: "VIOLATIONS torus: bit clear, far origin, the line's part BEHIND the origin (up to the backward reach) comes within 5 mm of the real tube"},
}


class Defines(ctypes.Structure):
    _fields_ = [(f"i{k}", ctypes.c_int32) for k in range(9)] + [("ambient", ctypes.c_float * 3), ("shadow", ctypes.c_float * 3)]


def load():
    if os.environ.get("CULL_AUDIT_LIB"):      # a variant built by hand (other -D flags), e.g. to audit a candidate change before it ships
        lib = ctypes.CDLL(os.environ["CULL_AUDIT_LIB"])
    else:
        subprocess.run(["make", "-C", os.path.join(ROOT, "tools", "audit")], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        lib = ctypes.CDLL(os.path.join(ROOT, "tools", "audit", "libcull_audit.so"))
    lib.cull_audit_run.restype = ctypes.c_int
    lib.cull_audit_run.argtypes = [ctypes.POINTER(Defines), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64), ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64,
                                   ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_float), ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
    lib.cull_audit_error.restype = ctypes.c_char_p
    lib.cull_audit_probe.restype = ctypes.c_int
    lib.cull_audit_probe.argtypes = [ctypes.POINTER(Defines), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64), ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    return lib


def scene_list(n_random):
    from raytracing_opengl_amd import scenes
    import random_scenes as rs
    out = [("bench:default", scenes.build_scene("default", 640, 480, 4)), ("bench:quadric", scenes.build_scene("quadric", 640, 480, 4)),
           ("bench:torus", scenes.build_scene("torus", 640, 480, 6)), ("bench:default t=7.5", scenes.build_scene("default", 640, 480, 4, time=7.5, delta=0.016))]
    for gen in ("random_scene", "nasty_scene", "scaled_quat_scene", "crowd_scene", "pencil_scene", "sized_torus_scene"):
        for seed in range(n_random):
            out.append((f"{gen}:{seed}", getattr(rs, gen)(seed, 96, 64)))
    return out


def _scene_args(sc):
    d = Defines()
    for k in range(9):
        setattr(d, f"i{k}", int(sc.defines[k]))
    d.ambient = (ctypes.c_float * 3)(*sc.defines[9:12])
    d.shadow = (ctypes.c_float * 3)(*sc.defines[12:15])
    keep = [ctypes.create_string_buffer(sc.blocks.get(n, b""), max(len(sc.blocks.get(n, b"")), 1)) for n in BLOCKS]
    ptrs = (ctypes.c_void_p * 9)(*[ctypes.cast(b, ctypes.c_void_p) for b in keep])
    sizes = (ctypes.c_uint64 * 9)(*[len(sc.blocks.get(n, b"")) for n in BLOCKS])
    return d, keep, ptrs, sizes


def probe(lib, sc, rays):
    """Single rays through the product's own scans on the GPU (cull_audit.hip probe_kernel). rays: (n, 8) float32 -- ro, rd, limit, torus
    index; returns (n, 12) float32 (layout: tests/harness.py probe, the host build of the same)."""
    import numpy as np
    d, keep, ptrs, sizes = _scene_args(sc)
    rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
    out = np.zeros((rays.shape[0], 12), dtype=np.float32)
    if lib.cull_audit_probe(ctypes.byref(d), ptrs, sizes, rays.ctypes.data, rays.shape[0], out.ctypes.data) != 0:
        raise RuntimeError(lib.cull_audit_error().decode())
    return out


def run(lib, sc, family, rays, seed, counters, bad_rows, max_bad=16):
    d, keep, ptrs, sizes = _scene_args(sc)
    bad = (ctypes.c_float * (12 * max_bad))()
    secs = ctypes.c_double(0)
    n = lib.cull_audit_run(ctypes.byref(d), ptrs, sizes, FAMILIES[family], int(rays), seed, counters, bad, max_bad, ctypes.byref(secs))
    if n < 0:
        raise RuntimeError(lib.cull_audit_error().decode())
    for k in range(n):
        bad_rows.append([float(v) for v in bad[12 * k:12 * k + 12]])
    return secs.value


def _spread(rows, n):
    """up to n rows, taken round-robin over the scenes they come from (so that one scene's violations do not hide the others')"""
    by = {}
    for r in rows:
        by.setdefault(r[0], []).append(r)
    out = []
    while len(out) < n and any(by.values()):
        for k in list(by):
            if by[k]:
                out.append(by[k].pop(0))
    return out[:n]


def run_family(lib, scs, fam, want, seed0=1000):
    """`want` rays of family `fam`, split evenly over the scenes of `scs` (name, scene) that have primitives of that family. Returns the
    report entry; violation rows carry the scene's name in front (a recorded ray can be replayed: tests/golden/torus_far_rays.json)."""
    use = [(n, s) for n, s in scs if (fam == "tables" and (s.defines[2] >= 16 or s.defines[4] >= 16)) or (fam != "tables" and s.defines[NEEDS[fam]] > 0)]
    counters = (ctypes.c_uint64 * N_COUNTERS)()
    bad_rows, t0, gpu_s = [], time.time(), 0.0
    per = max(1, int(want / max(1, len(use))))
    vk = [k for k in range(N_COUNTERS) if LABELS[fam].get(k, "").startswith("VIOLATIONS")]
    per_scene = {}
    for k, (name, sc) in enumerate(use):
        rows = []
        before = [counters[j] for j in vk]
        gpu_s += run(lib, sc, fam, per, seed0 + k, counters, rows)
        bad_rows += [[name] + r for r in rows]
        got = sum(counters[j] - b for j, b in zip(vk, before))
        if got:
            per_scene[name] = got
    c = list(counters)
    viol = sum(c[k] for k in vk)
    entry = {"scenes": len(use), "gpu_seconds": round(gpu_s, 2), "wall_seconds": round(time.time() - t0, 2), "violations": viol, "violations_by_scene": per_scene,
             "counters": {LABELS[fam][k]: c[k] for k in sorted(LABELS[fam])}, "first_violations": bad_rows[:32] if len(per_scene) < 2 else _spread(bad_rows, 64), "raw": c}
    f32 = lambda v: struct.unpack("<f", struct.pack("<I", v & 0xffffffff))[0]
    if fam == "torus_margin":
        entry["largest_clearance_of_a_phantom_hit"] = f32(c[20])
        entry["largest_distance_of_a_hit_point_from_the_surface"] = f32(c[21])
        entry["largest_imaginary_part_taken_for_real"] = f32(c[43])
        entry["largest_lead_by_class_of_t"] = {name: f32(c[34 + k]) for k, name in enumerate(("<4", "4..8", "8..16", "16..32", "32..64", ">=64"))}
    if fam == "torus_lead":
        entry["by_origin_distance"] = [
            {"|o|": LEAD_BINS[b], "hits": c[10 + b], "early>1e-3t+0.01": c[20 + b], "early>widened": c[30 + b], "early>0.1": c[40 + b], "early>1": c[50 + b],
             "largest_lead": f32(c[60 + b]), "phantoms_clear>1mm": c[70 + b], "largest_phantom_clearance": f32(c[80 + b]), "never_enters": c[90 + b]} for b in range(10)]
    return entry


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=float, default=1e8, help="rays per family")
    ap.add_argument("--families", default="torus,torus_margin,quadric,ring,tables")
    ap.add_argument("--scenes", type=int, default=6, help="seeds per random generator")
    ap.add_argument("--margin-rays", type=float, default=None, help="rays of the torus_margin family (every ray is solved: default rays / 20)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--only", default=None, help="keep the scenes whose name starts with this (e.g. sized_torus_scene)")
    ap.add_argument("--seed0", type=int, default=1000, help="first ray-stream seed (scene k of a family draws from seed0 + k): another value = other rays")
    args = ap.parse_args()
    lib = load()
    scs = scene_list(args.scenes)
    if args.only:
        scs = [(n, s) for n, s in scs if n.startswith(args.only)]
    report = {"rays_per_family": args.rays, "seed0": args.seed0, "families": {}, "scenes": [n for n, _ in scs]}
    for fam in args.families.split(","):
        want = args.rays if fam != "torus_margin" else (args.margin_rays or args.rays / 20)
        entry = run_family(lib, scs, fam, want, seed0=args.seed0)
        c, gpu_s, viol, bad_rows = entry.pop("raw"), entry["gpu_seconds"], entry["violations"], entry["first_violations"]
        report["families"][fam] = entry
        print(f"== {fam}: {c[0]:.3e} rays over {entry['scenes']} scenes in {gpu_s:.1f} s of kernels: {viol} violations", flush=True)
        for k in sorted(LABELS[fam]):
            print(f"     {LABELS[fam][k]:70s} {c[k]}", flush=True)
        if fam == "torus_margin":
            print(f"     largest |Im| of a root pair the solver took for real (sqrt(clearance (2 r + clearance))): {entry['largest_imaginary_part_taken_for_real']:.6g}")
            print(f"     largest clearance of a phantom hit: {entry['largest_clearance_of_a_phantom_hit']:.6g}; largest distance of a reported hit point from the surface: "
                  f"{entry['largest_distance_of_a_hit_point_from_the_surface']:.6g}")
            print("     largest lead (true entry into the inflated tube minus the reported t) by class of t:", entry["largest_lead_by_class_of_t"])
        if fam == "torus_lead":
            for row in entry["by_origin_distance"]:
                print("     ", row, flush=True)
        if entry.get("violations_by_scene"):
            print("     violations by scene:", entry["violations_by_scene"])
        for row in bad_rows[:12]:
            print("     first violations (scene, kind, prim, ro, rd, tmin, t, extra):", row[:12])
    if args.out:
        with open(args.out, "w") as f:
            json.dump(report, f, indent=1)
    return 1 if any(e["violations"] for e in report["families"].values()) else 0


if __name__ == "__main__":
    sys.exit(main())
