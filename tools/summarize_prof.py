#!/usr/bin/env python3
"""Condense a tools/profile_gpu.sh output directory into a text summary (kernel stats + PMC sums
of the trace kernel, per launch)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
KERNEL = "rt_trace_kernel"


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


print(f"# profile summary: {os.path.basename(out)}")
for f in find("trace/**/*kernel_stats.csv"):
    print(f"\n## kernel stats ({os.path.relpath(f, out)})")
    for row in csv.DictReader(open(f)):
        name = row.get("Name", "")
        print(f"{name[:90]:90s} calls={row.get('Calls')} total_ns={row.get('TotalDurationNs')} avg_ns={row.get('AverageNs')} "
              f"min_ns={row.get('MinNs')} max_ns={row.get('MaxNs')} pct={row.get('Percentage')}")
for f in find("trace/**/*kernel_trace.csv"):
    allrows = [r for r in csv.DictReader(open(f)) if KERNEL in r.get("Kernel_Name", "")]
    by_name = defaultdict(list)
    for r in allrows:
        by_name[r["Kernel_Name"]].append(r)
    # one line per instantiation (the timed variant, and the single launch of the ray-counting variant). VGPR_Count / Scratch_Size are the
    # dispatch packet's fields as rocprofv3 prints them (the ISA's figures -- NumVgprs, ScratchSize per lane -- are in tools/isa_stats.sh:
    # the light variant is 80 VGPRs / 44 B, the heavy one 72 / 96 B; rocprofv3 shows 40 / 72 for the light one: VGPRs in units of two, the
    # private segment rounded up by the runtime)
    for name, rows in sorted(by_name.items(), key=lambda kv: -len(kv[1])):
        print(f"\n## {name}")
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
        r0 = rows[0]
        print(f"## trace kernel launches: n={len(d)} mean={sum(d)/len(d)/1e3:.1f} us min={min(d)/1e3:.1f} max={max(d)/1e3:.1f} "
              f"grid={r0.get('Grid_Size_X')}x{r0.get('Grid_Size_Y')} wg={r0.get('Workgroup_Size_X')} vgpr={r0.get('VGPR_Count')} "
              f"accum_vgpr={r0.get('Accum_VGPR_Count')} sgpr={r0.get('SGPR_Count')} lds={r0.get('LDS_Block_Size')} scratch={r0.get('Scratch_Size')}")
for d in find("pmc_*/"):
    sums, n = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if KERNEL in r.get("Kernel_Name", ""):
                sums[r["Counter_Name"]] += float(r["Counter_Value"])
                n[r["Counter_Name"]] += 1
    if sums:
        print(f"\n## {os.path.basename(os.path.normpath(d))} (mean per launch of {KERNEL})")
        for k in sorted(sums):
            print(f"{k:28s} {sums[k]/max(1,n[k]):18.1f}   (launches {n[k]})")
