"""Per-kernel medians of the SMAA resolve from a rocprofv3 --kernel-trace CSV of tools/bench_smaa.py (8 configurations x REPS resolves).
A resolve starts with its smaa_edges_kernel launch; whatever smaa_* kernels follow belong to it."""
import collections
import csv
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
resolves = []
for r in rows:
    if "smaa" not in r["Kernel_Name"] or "expand" in r["Kernel_Name"]:
        continue
    name = r["Kernel_Name"].split("::")[1].split("(")[0].split("<")[0]
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if name == "smaa_edges_kernel":
        resolves.append({})
    if resolves:
        resolves[-1][name] = resolves[-1].get(name, 0) + d
labels = ["traced/LOW", "traced/MEDIUM", "traced/HIGH", "traced/ULTRA", "pattern/LOW", "pattern/MEDIUM", "pattern/HIGH", "pattern/ULTRA"]
for cfg in range(len(resolves) // reps):
    chunk = resolves[cfg * reps:(cfg + 1) * reps][2:]
    names = sorted({n for c in chunk for n in c})
    us = {n[5:-7]: round(statistics.median(c.get(n, 0) for c in chunk) / 1000, 1) for n in names}
    print(f"{labels[cfg] if cfg < len(labels) else cfg:16}", us, "sum", round(sum(us.values()), 1), "us")
