#!/usr/bin/env python3
"""Static listings of the trace kernel's ISA:
    python tools/isa_listing.py walk    > profiles/r06_quadric_walk_isa.txt       the quadric candidate walks of the many-primitive variant: every scalar load
                                                                                  with the distance (instructions) to the s_waitcnt that waits for it and to its first use
    python tools/isa_listing.py scratch > profiles/r06_scratch_account.txt        every scratch (spill) access of both variants with the loop it sits in
    python tools/isa_listing.py mix     > profiles/r06_valu_other_static.txt      static VALU mix per variant: what the "other" class is made of
Compiles rt_kernel.hip with the product's flags (kernel_build.cfg) to assembly first."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cfg(name):
    for l in open(os.path.join(ROOT, "raytracing_opengl_amd", "kernel_build.cfg")):
        m = re.match(rf"{name}\s*\?=\s*(.*)", l)
        if m:
            return m.group(1).strip()
    raise KeyError(name)


def assembly():
    out = "/tmp/rt_kernel_listing.s"
    cmd = (["/opt/rocm/bin/hipcc", f"-DRT_WAVES_PER_EU={cfg('WAVES_PER_EU')}", f"-DRT_WPE_HEAVY={cfg('WPE_HEAVY')}", "--offload-arch=gfx950"] + cfg("KERNEL_FLAGS").split() +
           ["-Iinclude", "-Iraytracing_opengl_amd/csrc", "-S", "--cuda-device-only", "-o", out, "raytracing_opengl_amd/csrc/rt_kernel.hip"])
    subprocess.run(cmd, check=True, cwd=ROOT, stderr=subprocess.DEVNULL)
    return open(out).read()


def kernel_body(txt, heavy):
    m = [x for x in re.finditer(r"rt_trace_kernelILb1ELb0ELb0ELi\d+ELb([01])ELb0EEEv14RtLaunchParams:", txt) if x.group(1) == ("1" if heavy else "0")][0]
    j = txt.find(".end_amdhsa_kernel", m.start())
    return txt[m.start():j].split("\n")


def is_instr(l):
    s = l.strip()
    return l.startswith("\t") and s and not s.startswith((".", ";"))


def sregs(tok):
    """scalar registers an operand names: s12, s[16:23] -> set of numbers"""
    out = set()
    for m in re.finditer(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b", tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def walk(txt):
    body = kernel_body(txt, True)
    heads = [k for k, l in enumerate(body) if "s_ff1_i32_b32" in l]
    print("# The quadric / torus candidate walks of rt_trace_kernel<true,false,false,6,HEAVY=true,false> (calc_inter and in_shadow): a wave-uniform walk over the set bits of the")
    print("# wave's OR of candidate words. For every scalar load inside a walk: how many instructions later the wave WAITS for it (s_waitcnt lgkmcnt) and how many later the")
    print("# loaded registers are first READ. A distance of 1-2 = the full scalar-cache round trip is exposed, once per candidate.\n")
    for h in heads:
        # the loop: from the header label above the s_ff1 to the backward branch that targets it
        k0 = h
        while k0 > 0 and "Loop Header" not in body[k0]:
            k0 -= 1
        while k0 > 0 and not body[k0].startswith(".LBB"):      # the header's label is the last label line above its comment block
            k0 -= 1
        label = body[k0].split(":")[0]
        # the loop's blocks are annotated "in Loop: Header=BB7_248": it ends with the last block that names this header (the latch may precede it)
        tag = "Header=" + label.lstrip(".L") + " "
        inloop = [k for k in range(max(0, k0 - 400), min(len(body), h + 4000)) if tag in body[k]]
        k0 = min([k0] + inloop)
        k1 = max(inloop) if inloop else h
        while k1 + 1 < len(body) and not body[k1 + 1].startswith(".LBB") and "; %bb." not in body[k1 + 1]:
            k1 += 1
        ins = [(k, body[k].strip()) for k in range(k0, k1 + 1) if is_instr(body[k])]
        print(f"== walk at line {h} ({label}): {len(ins)} instructions in the loop body, {sum(1 for _k, s in ins if s.startswith('s_load'))} scalar loads, "
              f"{sum(1 for _k, s in ins if s.startswith('s_waitcnt'))} waits, {sum(1 for _k, s in ins if s.startswith('v_'))} VALU, {sum(1 for _k, s in ins if s.startswith('s_cbranch'))} branches")
        for n, (k, s) in enumerate(ins):
            if not s.startswith("s_load"):
                continue
            dst = sregs(s.split(",")[0])
            wait = use = None
            for m2, (k2, s2) in enumerate(ins[n + 1:], start=1):
                if wait is None and s2.startswith("s_waitcnt") and "lgkmcnt" in s2:
                    wait = m2
                if use is None and not s2.startswith("s_load") and sregs(s2.split(None, 1)[1] if " " in s2 else "") & dst:
                    use = m2
                if wait is not None and use is not None:
                    break
            print(f"   +{n:4d}  {s:70s} wait after {wait} instr, first use after {use}")
        print()


def scratch(txt):
    for heavy in (False, True):
        body = kernel_body(txt, heavy)
        print(f"== rt_trace_kernel<true,false,false,6,HEAVY={'true' if heavy else 'false'},false>: scratch accesses (compiler spills) with the innermost loop they sit in")
        loop = "(straight-line)"
        per = collections.Counter()
        for l in body:
            m = re.search(r"=>\s+This (Inner )?Loop Header: Depth=(\d+)", l)
            if m:
                loop = f"loop depth {m.group(2)}"
            if "scratch_" in l and is_instr(l):
                s = l.strip()
                per[(s.split()[0], loop)] += 1
                print(f"   {loop:16s} {s}")
        print("   -- totals:", dict(per))
        tail = "\n".join(body[-1:])
        print()


def mix(txt):
    for heavy in (False, True):
        body = kernel_body(txt, heavy)
        ins = [l.strip() for l in body if is_instr(l)]
        c = collections.Counter(s.split()[0].split("_e32")[0].split("_e64")[0] for s in ins if s.startswith("v_"))
        named = ("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_fmamk_f32", "v_fmaak_f32", "v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_log_f32", "v_exp_f32",
                 "v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32")
        other = {k: v for k, v in c.items() if k not in named and not k.startswith(("v_add_u32", "v_add_co", "v_sub_u32", "v_mul_lo", "v_mul_hi", "v_lshl", "v_lshr", "v_ashr", "v_and_b32", "v_or_b32", "v_xor", "v_cvt", "v_mad_u", "v_add3", "v_lshl_add", "v_bfe", "v_bfi"))}
        print(f"== HEAVY={'true' if heavy else 'false'}: {len(ins)} instructions, {sum(c.values())} VALU (static counts; the PMC classes count EXECUTED wave-instructions)")
        print("   the classes the SQ_INSTS_VALU_* counters do not name ('other'), most frequent first:")
        for k, v in sorted(other.items(), key=lambda kv: -kv[1])[:24]:
            print(f"      {k:28s} {v}")
        print()


if __name__ == "__main__":
    t = assembly()
    {"walk": walk, "scratch": scratch, "mix": mix}[sys.argv[1] if len(sys.argv) > 1 else "walk"](t)
