"""Aggregate rocprofv3 PC-sampling CSVs on the GPU box (tools/pc_sample.sh): samples per instruction, per kernel (dispatch), per stall
reason -- whatever columns the file has.  Writes agg.txt (tables) and head.csv (the first rows, to see the format)."""
import csv
import collections
import glob
import os
import sys

raw, out = sys.argv[1], sys.argv[2]
files = sorted(glob.glob(os.path.join(raw, "**", "*pc_sampling*.csv"), recursive=True))
kfiles = sorted(glob.glob(os.path.join(raw, "**", "*kernel_trace*.csv"), recursive=True))
lines = []
for f in files:
    n = 0
    by_inst = collections.Counter()
    by_col = collections.defaultdict(collections.Counter)
    lanes = collections.Counter()
    with open(f, newline="") as fh:
        rd = csv.reader(fh)
        header = next(rd)
        hi = {h: i for i, h in enumerate(header)}
        with open(os.path.join(out, "head_" + os.path.basename(f)), "w") as hh:
            hh.write(",".join(header) + "\n")
            for row in rd:
                n += 1
                if n <= 300:
                    hh.write(",".join(row) + "\n")
                inst = row[hi["Instruction"]] if "Instruction" in hi else ""
                cmt = row[hi["Instruction_Comment"]] if "Instruction_Comment" in hi else ""
                did = row[hi["Dispatch_Id"]] if "Dispatch_Id" in hi else ""
                key = (inst, cmt)
                by_inst[key] += 1
                if "Exec_Mask" in hi:
                    try:
                        lanes[key] += bin(int(row[hi["Exec_Mask"]])).count("1")
                    except ValueError:
                        pass
                for h in header:
                    if h in ("Sample_Timestamp", "Exec_Mask", "Instruction", "Instruction_Comment", "Correlation_Id"):
                        continue
                    by_col[h][row[hi[h]]] += 1
    lines.append("== %s: %d samples, %d distinct instructions" % (os.path.basename(f), n, len(by_inst)))
    for h, c in by_col.items():
        if len(c) <= 64:
            lines.append("-- by %s: %s" % (h, ", ".join("%s=%d" % kv for kv in c.most_common())))
        else:
            lines.append("-- by %s: %d distinct; top: %s" % (h, len(c), ", ".join("%s=%d" % kv for kv in c.most_common(12))))
    mn = collections.Counter()
    for (inst, cmt), c in by_inst.items():
        mn[inst.split(" ")[0]] += c
    lines.append("-- by mnemonic (samples, %):")
    for m, c in mn.most_common(80):
        lines.append("   %-28s %9d %6.2f" % (m, c, 100.0 * c / max(n, 1)))
    lines.append("-- by instruction (samples, %, mean live lanes):")
    for (inst, cmt), c in by_inst.most_common(2500):
        lines.append("   %9d %6.3f %5.1f  %s  ; %s" % (c, 100.0 * c / max(n, 1), lanes[(inst, cmt)] / c if c else 0, inst, cmt))
open(os.path.join(out, "agg.txt"), "w").write("\n".join(lines) + "\n")
print("aggregated", len(files), "files")
