"""include/rtx/png_decode.h (the shim's built-in PNG reader) against Pillow: every colour
type, bit depth and interlace mode, tRNS, odd sizes, all five scanline filters (Pillow picks them adaptively on natural
content) and stored / fixed / dynamic DEFLATE blocks. Output convention = stb_image's stbi_load(..., req_comp = 0), which
is what the reference feeds to GL (GLWrapper.cpp:293,325). The reference's own two PNG textures are checked when the
reference checkout is present (build container)."""
import ctypes
import os
import struct
import zlib

import numpy as np
import pytest

from raytracing_opengl_amd import scenes

PIL = pytest.importorskip("PIL.Image")


def _decode(path):
    lib = scenes._host_lib()
    lib.rtxh_decode_image.restype = ctypes.c_size_t
    lib.rtxh_decode_image.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                      ctypes.c_void_p, ctypes.c_size_t]
    w, h, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    need = lib.rtxh_decode_image(str(path).encode(), ctypes.byref(w), ctypes.byref(h), ctypes.byref(c), None, 0)
    if need == 0:
        return None
    out = np.empty(need, np.uint8)
    lib.rtxh_decode_image(str(path).encode(), ctypes.byref(w), ctypes.byref(h), ctypes.byref(c), out.ctypes.data, need)
    return out.reshape(h.value, w.value, c.value)


def _image(rng, w, h, ch):
    yy, xx = np.mgrid[0:h, 0:w]
    base = (np.sin(xx / 7.0) * 60 + np.cos(yy / 5.0) * 50 + 128)[..., None] + rng.integers(-20, 20, (h, w, ch))
    base[: h // 3] = base[: h // 3].round(-1)          # flat-ish area: other filters win there
    return np.clip(base, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("mode,ch", [("L", 1), ("LA", 2), ("RGB", 3), ("RGBA", 4)])
@pytest.mark.parametrize("size", [(1, 1), (5, 3), (64, 48), (257, 31)])
@pytest.mark.parametrize("interlace", [False, True])
def test_eight_bit_colour_types(built, tmp_path, mode, ch, size, interlace):
    rng = np.random.default_rng(hash((mode, size, interlace)) & 0xffff)
    arr = _image(rng, size[0], size[1], ch)
    img = PIL.fromarray(arr[..., 0] if ch == 1 else arr, mode)
    p = tmp_path / "t.png"
    if interlace:
        _write_png(p, arr, {1: 0, 2: 4, 3: 2, 4: 6}[ch], 8, interlace=True)
    else:
        img.save(p, compress_level=int(rng.integers(0, 10)))
    got = _decode(p)
    assert got is not None and got.shape == (size[1], size[0], ch)
    assert np.array_equal(got, arr)


def _adam7_rows(arr, bits_per_sample=8):
    """scanlines (filter type 0) of the 7 Adam7 passes, for _write_png"""
    h, w = arr.shape[:2]
    x0, y0, dx, dy = (0, 4, 0, 2, 0, 1, 0), (0, 0, 4, 0, 2, 0, 1), (8, 8, 4, 4, 2, 2, 1), (8, 8, 8, 4, 4, 2, 2)
    out = bytearray()
    for p in range(7):
        sub = arr[y0[p]::dy[p], x0[p]::dx[p]]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        for row in sub:
            out += b"\x00" + _pack_row(row, bits_per_sample)
    return bytes(out)


def _pack_row(row, bits):
    if bits == 8:
        return row.astype(np.uint8).tobytes()
    if bits == 16:
        return row.astype(">u2").tobytes()
    flat = row.reshape(-1).astype(np.uint8)
    per = 8 // bits
    pad = (-len(flat)) % per
    flat = np.concatenate([flat, np.zeros(pad, np.uint8)])
    acc = np.zeros(len(flat) // per, np.uint8)
    for k in range(per):
        acc |= flat[k::per] << (8 - bits * (k + 1))
    return acc.tobytes()


def _write_png(path, arr, ctype, depth, interlace=False, plte=None, trns=None, level=6):
    """minimal PNG writer (filter type 0 only) for the cases Pillow cannot produce"""
    h, w = arr.shape[:2]
    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xffffffff)
    if interlace:
        raw = _adam7_rows(arr, depth)
    else:
        raw = b"".join(b"\x00" + _pack_row(row, depth) for row in arr)
    data = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1 if interlace else 0))
    if plte is not None:
        data += chunk(b"PLTE", bytes(plte))
    if trns is not None:
        data += chunk(b"tRNS", bytes(trns))
    comp = zlib.compress(raw, level)
    half = len(comp) // 2
    data += chunk(b"IDAT", comp[:half]) + chunk(b"IDAT", comp[half:]) + chunk(b"IEND", b"")   # split IDAT on purpose
    open(path, "wb").write(data)


@pytest.mark.parametrize("depth", [1, 2, 4])
@pytest.mark.parametrize("interlace", [False, True])
def test_low_bit_depth_grey_and_palette(built, tmp_path, depth, interlace):
    rng = np.random.default_rng(depth * 2 + interlace)
    w, h = 37, 21
    idx = rng.integers(0, 1 << depth, (h, w)).astype(np.uint8)
    p = tmp_path / "g.png"
    _write_png(p, idx[..., None], 0, depth, interlace)
    got = _decode(p)
    assert np.array_equal(got[..., 0], idx * {1: 255, 2: 85, 4: 17}[depth]) and got.shape[2] == 1
    pal = rng.integers(0, 256, (1 << depth, 3)).astype(np.uint8)
    _write_png(p, idx[..., None], 3, depth, interlace, plte=pal.tobytes())
    got = _decode(p)
    assert got.shape == (h, w, 3) and np.array_equal(got, pal[idx])
    alpha = rng.integers(0, 256, (1 << depth) - 1).astype(np.uint8) if depth > 1 else np.array([7], np.uint8)
    _write_png(p, idx[..., None], 3, depth, interlace, plte=pal.tobytes(), trns=alpha.tobytes())
    got = _decode(p)
    full_alpha = np.concatenate([alpha, np.full((1 << depth) - len(alpha), 255, np.uint8)])
    assert got.shape == (h, w, 4) and np.array_equal(got[..., :3], pal[idx]) and np.array_equal(got[..., 3], full_alpha[idx])


def test_sixteen_bit_and_colour_key(built, tmp_path):
    rng = np.random.default_rng(9)
    w, h = 33, 17
    a16 = rng.integers(0, 65536, (h, w, 3)).astype(np.uint16)
    p = tmp_path / "s.png"
    _write_png(p, a16, 2, 16)
    got = _decode(p)
    assert got.shape == (h, w, 3) and np.array_equal(got, (a16 >> 8).astype(np.uint8))
    a8 = rng.integers(0, 4, (h, w, 3)).astype(np.uint8) * 80
    key = a8[3, 5]
    _write_png(p, a8, 2, 8, trns=struct.pack(">3H", *[int(v) for v in key]))
    got = _decode(p)
    assert got.shape == (h, w, 4) and np.array_equal(got[..., :3], a8)
    assert np.array_equal(got[..., 3] == 0, (a8 == key).all(axis=2))
    for level in (0, 1, 9):   # stored, fast (mostly fixed Huffman), best (dynamic Huffman) DEFLATE blocks
        _write_png(p, a8, 2, 8, level=level)
        assert np.array_equal(_decode(p), a8)


def test_rejects_garbage(built, tmp_path):
    p = tmp_path / "bad.png"
    p.write_bytes(b"\x89PNG\r\n\x1a\n" + b"\x00" * 40)
    assert _decode(p) is None
    p.write_bytes(b"not an image")
    assert _decode(p) is None


def test_reference_png_assets(built):
    root = "/root/reference/assets/textures"
    names = [n for n in ("container.png", "8k_saturn_ring_alpha.png") if os.path.exists(os.path.join(root, n))]
    if not names:
        pytest.skip("reference checkout not present")
    for n in names:
        want = np.asarray(PIL.open(os.path.join(root, n)))
        got = _decode(os.path.join(root, n))
        assert got is not None and got.shape == want.shape and np.array_equal(got, want), n


@pytest.mark.parametrize("ch", [3, 4])
@pytest.mark.parametrize("size", [(1, 1), (7, 3), (300, 200)])   # 300x200x4 > 65535 bytes: several stored blocks
@pytest.mark.parametrize("bottom_up", [False, True])
def test_png_writer_round_trips(built, tmp_path, ch, size, bottom_up):
    """include/rtx/png_write.h: Pillow and the shim's own reader get the pixels back; bottom_up flips
    rtx_read_pixels' row order (row 0 = bottom) into the file's (row 0 = top)."""
    lib = scenes._host_lib()
    lib.rtxh_write_png.restype = ctypes.c_int
    lib.rtxh_write_png.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    rng = np.random.default_rng(size[0] * 7 + ch)
    arr = rng.integers(0, 256, (size[1], size[0], ch), dtype=np.uint8)
    p = tmp_path / "w.png"
    assert lib.rtxh_write_png(str(p).encode(), arr.ctypes.data, size[0], size[1], ch, int(bottom_up)) == 1
    want = arr[::-1] if bottom_up else arr
    assert np.array_equal(np.asarray(PIL.open(p)), want)
    assert np.array_equal(_decode(p), want)
    assert lib.rtxh_write_png(str(p).encode(), arr.ctypes.data, size[0], size[1], 2, 0) == 0   # unsupported channel count
