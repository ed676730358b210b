#!/usr/bin/env python3
"""Join an `ablate.py` log with the rocprofv3 --pmc counter CSV of the same run: dynamic instruction
counts per scene variant. Every variant starts with ONE launch of the ray-counting kernel
(<true, true, false>), which is used as the separator."""
import csv
import glob
import sys
from collections import OrderedDict


def main(out_dir, log):
    names = [l[:34].strip() for l in open(log) if " us " in l and "rays closest" in l]
    rows = []
    for f in glob.glob(out_dir + "/**/*counter_collection.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    # dispatch id -> {counter: value}
    disp = OrderedDict()
    for r in rows:
        if "rt_trace_kernel" not in r["Kernel_Name"]:
            continue
        d = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"]})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    cases, cur = [], None
    for d in disp.values():
        if "<true, true, false" in d["name"]:
            cur = []
            cases.append(cur)
        elif cur is not None:
            cur.append(d)
    ctrs = sorted({k for c in cases for d in c for k in d if k != "name"})
    print(f"{'variant':34s} " + " ".join(f"{c:>22s}" for c in ctrs))
    for i, c in enumerate(cases):
        nm = names[i] if i < len(names) else f"case {i}"
        vals = [sum(d.get(k, 0.0) for d in c) / max(len(c), 1) for k in ctrs]
        print(f"{nm:34s} " + " ".join(f"{v:22.0f}" for v in vals))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
