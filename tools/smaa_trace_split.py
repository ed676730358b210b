"""Per-kernel medians of the SMAA resolve from a rocprofv3 --kernel-trace CSV of tools/bench_smaa.py (8 configurations x REPS resolves)."""
import collections
import csv
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
smaa = [(r["Kernel_Name"].split("::")[1].split("(")[0].split("<")[0], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows if "smaa" in r["Kernel_Name"]]
per = collections.defaultdict(list)
for i in range(0, len(smaa), 4):
    for name, d in smaa[i:i + 4]:
        per[(i // 4 // reps, name)].append(d)
names = ("smaa_clear_kernel", "smaa_edges_kernel", "smaa_weights_kernel", "smaa_blend_kernel")
labels = ["traced/LOW", "traced/MEDIUM", "traced/HIGH", "traced/ULTRA", "pattern/LOW", "pattern/MEDIUM", "pattern/HIGH", "pattern/ULTRA"]
for cfg in range(len(smaa) // 4 // reps):
    us = {n[5:-7]: round(statistics.median(per[(cfg, n)]) / 1000, 1) for n in names}
    print(f"{labels[cfg]:16s}", us, "sum", round(sum(us.values()), 1), "us")
