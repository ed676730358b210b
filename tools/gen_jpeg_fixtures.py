#!/usr/bin/env python3
"""Golden vectors for include/rtx/jpeg_decode.h: small JPEG files + the texels the REFERENCE's decoder gives for them.

The reference loads its textures with stb_image v2.25 (GLWrapper.cpp:293,325). `make -C oracle refstb` compiles that
decoder from the reference checkout where it lies (oracle/_ref/libstbref.so, build container only); this script
  1. writes small JPEG files: Pillow/libjpeg encodings (baseline + progressive, 4:4:4 / 4:2:2 / 4:2:0 / 4:1:1, grey, CMYK,
     RGB-coded, restart intervals, optimised tables) and hand-encoded baseline files for what libjpeg does not emit
     (arbitrary sampling factors such as 1x2, 4x1, 1x4, mixed 2x1/1x2 chroma, 16-bit quantisation tables,
     one scan per component, fill bytes before markers, a DNL segment);
  2. decodes each with libstbref.so (stbi_load(path, &w, &h, &c, 0));
  3. stores the files under tests/golden/jpeg/ and the expected texels in tests/golden/jpeg/expected.npz.
tests/test_jpeg_decode.py replays them anywhere (no reference checkout, no Pillow needed).
"""
import ctypes
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "jpeg")
ZIGZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
          35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]


def stb_decode(lib, path):
    w, h, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    p = lib.stbi_load(path.encode(), ctypes.byref(w), ctypes.byref(h), ctypes.byref(c), 0)
    if not p:
        return None
    n = w.value * h.value * c.value
    a = np.ctypeslib.as_array((ctypes.c_uint8 * n).from_address(p)).copy().reshape(h.value, w.value, c.value)
    lib.stbi_image_free(p)
    return a


def load_stb():
    path = os.path.join(ROOT, "oracle", "_ref", "libstbref.so")
    if not os.path.exists(path):
        return None
    lib = ctypes.CDLL(path)
    lib.stbi_load.restype = ctypes.c_void_p
    lib.stbi_load.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.c_int]
    lib.stbi_image_free.argtypes = [ctypes.c_void_p]
    return lib


# ---- a minimal baseline encoder working on random quantised coefficients -------------------------------------------
class BitWriter:
    def __init__(self):
        self.out, self.acc, self.n = bytearray(), 0, 0

    def put(self, value, nbits):
        for k in range(nbits - 1, -1, -1):
            self.acc = (self.acc << 1) | ((value >> k) & 1)
            self.n += 1
            if self.n == 8:
                self.out.append(self.acc)
                if self.acc == 0xFF:
                    self.out.append(0)
                self.acc, self.n = 0, 0

    def flush(self):
        while self.n:
            self.put(1, 1)


def category(v):
    a, n = abs(v), 0
    while a:
        a >>= 1
        n += 1
    return n


def amplitude_bits(v, n):
    return v if v >= 0 else v + (1 << n) - 1


# DC: categories 0..11 as 4-bit codes; AC: (run, size) for size 1..10 plus EOB and ZRL as 8-bit codes, canonical order
DC_SYMS = list(range(12))
AC_SYMS = [0x00, 0xF0] + [(r << 4) | s for r in range(16) for s in range(1, 11)]


def dht_segment():
    seg = bytearray()
    for cls, length, syms in ((0, 4, DC_SYMS), (1, 8, AC_SYMS)):
        counts = [0] * 16
        counts[length - 1] = len(syms)
        seg += bytes([cls << 4]) + bytes(counts) + bytes(syms)
    return b"\xff\xc4" + struct.pack(">H", 2 + len(seg)) + bytes(seg)


def encode_block(bw, coef, pred):
    dc = int(coef[0])
    diff = dc - pred
    n = category(diff)
    bw.put(DC_SYMS.index(n), 4)
    bw.put(amplitude_bits(diff, n), n)
    run = 0
    last = max([k for k in range(1, 64) if coef[ZIGZAG[k]] != 0], default=0)
    for k in range(1, last + 1):
        v = int(coef[ZIGZAG[k]])
        if v == 0:
            run += 1
            continue
        while run > 15:
            bw.put(AC_SYMS.index(0xF0), 8)
            run -= 16
        n = category(v)
        bw.put(AC_SYMS.index((run << 4) | n), 8)
        bw.put(amplitude_bits(v, n), n)
        run = 0
    if last < 63:
        bw.put(AC_SYMS.index(0x00), 8)
    return dc


def handmade(rng, w, h, factors, wide_q=False, separate_scans=False, restart=0, fill_bytes=False, dnl=False, ids=None, adobe=None, jfif=True):
    """factors: [(h, v)] per component. Random sparse coefficients, one quantisation table per component."""
    nc = len(factors)
    hmax, vmax = max(f[0] for f in factors), max(f[1] for f in factors)
    mx, my = -(-w // (8 * hmax)), -(-h // (8 * vmax))
    ids = ids or list(range(1, nc + 1))
    out = bytearray(b"\xff\xd8")
    if jfif:
        out += b"\xff\xe0" + struct.pack(">H", 16) + b"JFIF\0\x01\x01\0\0\x01\0\x01\0\0"
    if adobe is not None:
        out += b"\xff\xee" + struct.pack(">H", 14) + b"Adobe\0\x64\0\0\0\0" + bytes([adobe])
    qt = []
    for t in range(min(nc, 4)):
        q = rng.integers(1, 40, 64) if not wide_q else rng.integers(200, 700, 64)
        qt.append(q)
        body = bytes([(1 << 4 | t) if wide_q else t]) + (b"".join(struct.pack(">H", int(q[ZIGZAG[i]])) for i in range(64)) if wide_q
                                                          else bytes(int(q[ZIGZAG[i]]) for i in range(64)))
        out += b"\xff\xdb" + struct.pack(">H", 2 + len(body)) + body
    out += b"\xff\xc0" + struct.pack(">HBHHB", 8 + 3 * nc, 8, h, w, nc)
    for i, (fh, fv) in enumerate(factors):
        out += bytes([ids[i], (fh << 4) | fv, i])
    out += dht_segment()
    if restart:
        out += b"\xff\xdd" + struct.pack(">HH", 4, restart)

    def blocks_for(i):
        # random coefficients, magnitudes chosen so that sample values saturate now and then
        fh, fv = factors[i]
        lim = 2 if wide_q else 14
        blk = {}
        for by in range(my * fv):
            for bx in range(mx * fh):
                c = np.zeros(64, np.int64)
                c[0] = rng.integers(-lim * 2, lim * 2 + 1)
                for _ in range(int(rng.integers(0, 9))):
                    c[int(rng.integers(1, 64))] = rng.integers(-lim, lim + 1)
                blk[(bx, by)] = c
        return blk
    coefs = [blocks_for(i) for i in range(nc)]

    def scan(comps):
        seg = b"\xff\xda" + struct.pack(">HB", 6 + 2 * len(comps), len(comps)) + b"".join(bytes([ids[i], 0x00]) for i in comps) + b"\0\x3f\0"
        bw = BitWriter()
        pred = {i: 0 for i in comps}
        units = []
        if len(comps) == 1:
            i = comps[0]
            fh, fv = factors[i]
            cx, cy = -(-w * fh // hmax), -(-h * fv // vmax)
            for by in range((cy + 7) // 8):
                for bx in range((cx + 7) // 8):
                    units.append([(i, bx, by)])
        else:
            for j in range(my):
                for ii in range(mx):
                    units.append([(i, ii * factors[i][0] + bx, j * factors[i][1] + by) for i in comps for by in range(factors[i][1]) for bx in range(factors[i][0])])
        data = bytearray()
        for n, unit in enumerate(units):
            for (i, bx, by) in unit:
                pred[i] = encode_block(bw, coefs[i][(bx, by)], pred[i])
            if restart and (n + 1) % restart == 0 and n + 1 < len(units):
                bw.flush()
                data += bw.out + (b"\xff\xff" if fill_bytes else b"") + bytes([0xFF, 0xD0 + ((n + 1) // restart - 1) % 8])
                bw = BitWriter()
                pred = {i: 0 for i in comps}
        bw.flush()
        return seg + bytes(data) + bytes(bw.out)

    if separate_scans:
        for i in range(nc):
            out += scan([i])
    else:
        out += scan(list(range(nc)))
    if dnl:
        out += b"\xff\xdc" + struct.pack(">HH", 4, h)
    out += (b"\xff\xff" if fill_bytes else b"") + b"\xff\xd9"
    return bytes(out)


def handmade_cases(rng):
    c = {}
    c["hm_grey_9x7"] = handmade(rng, 9, 7, [(1, 1)])
    c["hm_v2_13x21"] = handmade(rng, 13, 21, [(1, 2), (1, 1), (1, 1)])              # 4:4:0
    c["hm_v2_1x1"] = handmade(rng, 1, 1, [(1, 2), (1, 1), (1, 1)])
    c["hm_h2_1x9"] = handmade(rng, 1, 9, [(2, 1), (1, 1), (1, 1)])                  # w_lores == 1
    c["hm_h2v2_1x1"] = handmade(rng, 1, 1, [(2, 2), (1, 1), (1, 1)])
    c["hm_h2v2_2x3"] = handmade(rng, 2, 3, [(2, 2), (1, 1), (1, 1)])
    c["hm_h4_37x9"] = handmade(rng, 37, 9, [(4, 1), (1, 1), (1, 1)])                # replication 4x
    c["hm_v4_9x37"] = handmade(rng, 9, 37, [(1, 4), (1, 1), (1, 1)])
    c["hm_h4v2_30x20"] = handmade(rng, 30, 20, [(4, 2), (2, 1), (1, 2)])            # mixed: cb 2x2 up, cr 4x1 replication
    c["hm_mixed_27x19"] = handmade(rng, 27, 19, [(2, 2), (2, 1), (1, 2)])           # cb: v only, cr: h only
    c["hm_h3_20x10"] = handmade(rng, 20, 10, [(3, 1), (1, 1), (1, 1)])              # factor 3
    c["hm_luma_sub_21x13"] = handmade(rng, 21, 13, [(1, 1), (2, 2), (2, 2)])        # chroma finer than luma
    c["hm_wideq_17x17"] = handmade(rng, 17, 17, [(2, 2), (1, 1), (1, 1)], wide_q=True)
    c["hm_scans_25x18"] = handmade(rng, 25, 18, [(2, 2), (1, 1), (1, 1)], separate_scans=True)
    c["hm_scans_rst_25x18"] = handmade(rng, 25, 18, [(2, 1), (1, 1), (1, 1)], separate_scans=True, restart=2, fill_bytes=True)
    c["hm_rst_fill_40x24"] = handmade(rng, 40, 24, [(2, 2), (1, 1), (1, 1)], restart=1, fill_bytes=True, dnl=True)
    c["hm_rgb_ids_11x6"] = handmade(rng, 11, 6, [(1, 1), (1, 1), (1, 1)], ids=[ord("R"), ord("G"), ord("B")])
    c["hm_adobe0_nojfif_11x6"] = handmade(rng, 11, 6, [(1, 1), (1, 1), (1, 1)], adobe=0, jfif=False)   # RGB by Adobe transform 0
    c["hm_adobe0_jfif_11x6"] = handmade(rng, 11, 6, [(1, 1), (1, 1), (1, 1)], adobe=0, jfif=True)      # JFIF wins: YCbCr
    c["hm_cmyk_12x10"] = handmade(rng, 12, 10, [(1, 1)] * 4, adobe=0)
    c["hm_ycck_12x10"] = handmade(rng, 12, 10, [(2, 2), (1, 1), (1, 1), (2, 2)], adobe=2)
    c["hm_4comp_plain_12x10"] = handmade(rng, 12, 10, [(1, 1)] * 4)
    return c


def pillow_cases(rng):
    from PIL import Image

    def img(w, h, ch):
        yy, xx = np.mgrid[0:h, 0:w]
        base = (np.sin(xx / 5.0) * 70 + np.cos(yy / 3.0) * 60 + 128)[..., None] + rng.integers(-40, 40, (h, w, ch)) + np.array([0, 30, -30, 10])[:ch]
        base[: h // 3, : w // 2] = rng.integers(0, 256, (h // 3, w // 2, ch))
        return np.clip(base, 0, 255).astype(np.uint8)
    import io
    c = {}
    spec = [("L", 1, (19, 13), None), ("RGB", 3, (33, 17), "4:4:4"), ("RGB", 3, (33, 17), "4:2:2"), ("RGB", 3, (33, 17), "4:2:0"),
            ("RGB", 3, (35, 9), "4:1:1"), ("CMYK", 4, (18, 12), None), ("RGB", 3, (1, 1), "4:2:0"), ("RGB", 3, (48, 32), "4:2:0")]
    for mode, ch, (w, h), sub in spec:
        arr = img(w, h, ch)
        im = Image.fromarray(arr[..., 0] if ch == 1 else arr, mode)
        for prog in (False, True):
            for q, rst in ((35, 0), (90, 2)):
                kw = dict(quality=q, progressive=prog, optimize=(q == 90))
                if sub:
                    kw["subsampling"] = sub
                if rst:
                    kw["restart_marker_blocks"] = rst
                b = io.BytesIO()
                im.save(b, "JPEG", **kw)
                c[f"pil_{mode}_{(sub or 'n').replace(':', '')}_{w}x{h}_{'prog' if prog else 'base'}_q{q}_r{rst}"] = b.getvalue()
    arr = img(21, 14, 3)
    b = io.BytesIO()
    Image.fromarray(arr, "RGB").save(b, "JPEG", keep_rgb=True, quality=85)
    c["pil_keep_rgb_21x14"] = b.getvalue()
    return c


def main():
    lib = load_stb()
    if lib is None:
        sys.exit("oracle/_ref/libstbref.so missing: run `make -C oracle refstb` in a container that has /root/reference")
    rng = np.random.default_rng(20260928)
    cases = {}
    cases.update(handmade_cases(rng))
    cases.update(pillow_cases(rng))
    os.makedirs(OUT, exist_ok=True)
    for f in os.listdir(OUT):
        if f.endswith(".jpg") or f == "expected.npz":
            os.remove(os.path.join(OUT, f))
    expected = {}
    for name, data in sorted(cases.items()):
        path = os.path.join(OUT, name + ".jpg")
        with open(path, "wb") as f:
            f.write(data)
        a = stb_decode(lib, path)
        if a is None:
            sys.exit(f"the reference decoder rejects {name}")
        expected[name] = a
    np.savez_compressed(os.path.join(OUT, "expected.npz"), **expected)
    total = sum(len(d) for d in cases.values())
    print(f"{len(cases)} files, {total} bytes of JPEG, expected.npz {os.path.getsize(os.path.join(OUT, 'expected.npz'))} bytes")


if __name__ == "__main__":
    main()
