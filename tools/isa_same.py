#!/usr/bin/env python3
"""Are the kernels of two `hipcc -S --cuda-device-only` listings the same machine code? Usage: isa_same.py old.s new.s [old-suffix new-suffix]
Kernels are matched by mangled name (optionally after replacing a name suffix: a template parameter added at the end); block labels are
normalised; prints one line per kernel of the old listing and the register / scratch figures of kernels only the new listing has.
Round 5 used it to show that adding the SKYLOD instantiations (mip-mapped sky box) left all nine existing tracer kernels untouched."""
import re
import sys


def kernels(path):
    txt = open(path).read()
    out = {}
    for m in re.finditer(r'^(_Z\w+):\s*; @\1\n(.*?)^\s*\.end_amdhsa_kernel', txt, re.S | re.M):
        lines = [l.split(';')[0].strip() for l in m.group(2).split('\n')]
        out[m.group(1)] = [re.sub(r'\.LBB\d+_', '.LBB_', l).replace(m.group(1), 'K') for l in lines if l and not l.startswith('.')]
    return out, txt


def main():
    (a, _), (b, txt) = kernels(sys.argv[1]), kernels(sys.argv[2])
    old_sfx, new_sfx = (sys.argv[3], sys.argv[4]) if len(sys.argv) > 4 else ("", "")
    seen = set()
    for k, la in a.items():
        k2 = k.replace(old_sfx, new_sfx) if old_sfx and old_sfx in k else k
        seen.add(k2)
        if k2 not in b:
            print(f"{k}: not in the new listing")
            continue
        d = sum(1 for x, y in zip(la, b[k2]) if x != y) + abs(len(la) - len(b[k2]))
        print(f"{k2}: {len(la)} instructions, " + ("IDENTICAL" if d == 0 else f"{d} lines differ ({len(b[k2])} instructions now)"))
    for m in re.finditer(r'\.amdhsa_kernel (_Z\w+)\n(.*?)\.end_amdhsa_kernel', txt, re.S):
        if m.group(1) in seen or m.group(1) not in b:
            continue
        v = re.search(r'\.amdhsa_next_free_vgpr (\d+)', m.group(2)).group(1)
        sc = re.search(r'\.amdhsa_private_segment_fixed_size (\d+)', m.group(2)).group(1)
        print(f"{m.group(1)}: new, {len(b[m.group(1)])} instructions, {v} VGPRs, {sc} B scratch")


if __name__ == "__main__":
    main()
