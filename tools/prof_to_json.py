#!/usr/bin/env python3
"""tools/profile_gpu.sh output directory -> profiles/valu[_scene].json + profiles/traffic[_scene].json, stamped with the kernel-source
hash (raytracing_opengl_amd/build_info.py) so that bench.py can tell whether they still describe the kernel it runs.
usage: prof_to_json.py gpurun_out/prof_<tag> <scene> <width> <height> <depth> <summary file kept under profiles/>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_opengl_amd import build_info  # noqa: E402

out, scene, W, H, depth, kept = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]


def newest(pattern):
    """gpurun MERGES a call's output into gpurun_out/, so a tag that was profiled before still holds the earlier runs' files: per pass
    directory only the most recent file counts."""
    by_dir = {}
    for f in glob.glob(os.path.join(out, pattern), recursive=True):
        d = f[len(out):].lstrip(os.sep).split(os.sep)[0]
        if d not in by_dir or os.path.getmtime(f) > os.path.getmtime(by_dir[d]):
            by_dir[d] = f
    return sorted(by_dir.values())


KERNEL = "rt_trace_kernel"
# the timed variant only: the single launch of the ray-counting variant and nothing else shares the name stem
names = defaultdict(int)
for f in newest(os.path.join("trace", "**", "*kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        if KERNEL in r.get("Kernel_Name", ""):
            names[r["Kernel_Name"]] += 1
KERNEL = max(names, key=names.get) if names else KERNEL
vals, n = defaultdict(float), defaultdict(int)
for f in newest(os.path.join("pmc_*", "**", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if KERNEL == r.get("Kernel_Name", "") or (not names and KERNEL in r.get("Kernel_Name", "")):
            vals[r["Counter_Name"]] += float(r["Counter_Value"])
            n[r["Counter_Name"]] += 1
mean = {k: vals[k] / n[k] for k in vals}
dur = []
for f in newest(os.path.join("trace", "**", "*kernel_trace.csv")):
    dur += [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if KERNEL == r.get("Kernel_Name", "")]
dur = sorted(dur)[: max(1, len(dur) - 2)]   # the first launches after a module load are outliers (clock ramp): drop the two slowest
suffix = "" if scene == "default" else "_" + scene
stamp = dict(kernel=KERNEL, kernel_hash=build_info.kernel_source_hash(), scene=scene, width=W, height=H, depth=depth, source=kept,
             kernel_us_rocprof=round(sum(dur) / len(dur) / 1e3, 1) if dur else None)
valu = dict(stamp, valu_insts_per_launch=int(mean["SQ_INSTS_VALU"]), salu_insts_per_launch=int(mean.get("SQ_INSTS_SALU", 0)),
            active_inst_valu_quad_cycles=int(mean.get("SQ_ACTIVE_INST_VALU", 0)), thread_cycles_valu=int(mean.get("SQ_THREAD_CYCLES_VALU", 0)),
            busy_cu_cycles=int(mean.get("SQ_BUSY_CU_CYCLES", 0)), grbm_gui_active=int(mean.get("GRBM_GUI_ACTIVE", 0)))
# instruction classes (pass "mix"): what the mix-weighted issue ceiling is computed from. "other" = everything the class counters do not
# name: v_mov, v_cmp, v_cndmask, v_min / v_max / v_med3, v_readlane / v_writelane, DPP moves, bit operations counted nowhere else.
classes = {k: int(mean.get("SQ_INSTS_VALU_" + c, 0)) for k, c in (("add_f32", "ADD_F32"), ("mul_f32", "MUL_F32"), ("fma_f32", "FMA_F32"),
                                                                   ("trans_f32", "TRANS_F32"), ("int32", "INT32"), ("int64", "INT64"), ("cvt", "CVT"))}
if any(classes.values()):
    classes["other"] = max(0, valu["valu_insts_per_launch"] - sum(classes.values()))
    valu["classes"] = classes
if valu["active_inst_valu_quad_cycles"]:
    valu["lane_utilisation"] = round(valu["thread_cycles_valu"] / (64.0 * valu["active_inst_valu_quad_cycles"]), 4)
    valu["cycles_per_valu_inst"] = round(4.0 * valu["active_inst_valu_quad_cycles"] / valu["valu_insts_per_launch"], 3)
if valu["grbm_gui_active"]:
    # GRBM_GUI_ACTIVE: busy cycles summed over the 8 XCDs -> the shader clock the launches ran at. (Rounds 1-3 also derived a `valu_pipe_busy`
    # from SQ_ACTIVE_INST_VALU / GRBM_GUI_ACTIVE; the two counters' units are nominal and the ratio came out above 1 on a saturated pipe, so it
    # is no longer written: cycles_per_valu_inst and the mix-weighted issue ceiling in bench.py carry the information.)
    valu["shader_clock_GHz"] = round(valu["grbm_gui_active"] / 8.0 / (stamp["kernel_us_rocprof"] * 1e3), 3) if stamp["kernel_us_rocprof"] else None
json.dump(valu, open(os.path.join(ROOT, "profiles", f"valu{suffix}.json"), "w"), indent=1)
fetch_kb, write_kb = mean.get("FETCH_SIZE", 0.0), mean.get("WRITE_SIZE", 0.0)
traffic = dict(stamp, hbm_bytes_per_launch=int((fetch_kb + write_kb) * 1024), fetch_size_kb=round(fetch_kb, 1), write_size_kb=round(write_kb, 1),
               algorithmic_bytes_per_launch=W * H * 16,
               note="FETCH_SIZE raw (the guide's x2 correction applies to wide coalesced streams; the kernel's reads are 4-byte texture taps and scratch re-loads); WRITE_SIZE 1:1 for 16 B/lane stores (calibrated in round 1 on the spill-free build)")
json.dump(traffic, open(os.path.join(ROOT, "profiles", f"traffic{suffix}.json"), "w"), indent=1)
print(json.dumps(valu), json.dumps(traffic), sep="\n")
