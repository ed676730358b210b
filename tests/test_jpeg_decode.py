"""include/rtx/jpeg_decode.h (the shim's built-in JPEG reader) against the REFERENCE's decoder.

JPEG leaves the decoder's arithmetic open, so the pin is the reference's own stb_image v2.25 (GLWrapper.cpp:293,325
stbi_load(path, &w, &h, &c, 0)): tests/golden/jpeg/ holds small files and the texels stb_image returns for them
(tools/gen_jpeg_fixtures.py, run in the build container against oracle/_ref/libstbref.so = the reference's stb_image.cpp
compiled where it lies). The fixtures replay anywhere; where the reference checkout and Pillow exist, a wider generated
sweep and the reference's nine JPEG assets are compared live as well. Bar: identical bytes."""
import ctypes
import io
import os

import numpy as np
import pytest

from raytracing_opengl_amd import scenes

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(HERE, "golden", "jpeg")
STBREF = os.path.join(ROOT, "oracle", "_ref", "libstbref.so")
REF_TEXTURES = "/root/reference/assets/textures"


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


def _expected():
    return np.load(os.path.join(GOLD, "expected.npz"))


def _names():
    return sorted(f[:-4] for f in os.listdir(GOLD) if f.endswith(".jpg"))


def test_fixture_set_is_complete():
    exp = _expected()
    assert sorted(exp.files) == _names() and len(exp.files) >= 50
    kinds = {n.split("_")[0] for n in exp.files}
    assert kinds == {"hm", "pil"}
    assert any("prog" in n for n in exp.files) and any("CMYK" in n for n in exp.files)


@pytest.mark.parametrize("name", _names())
def test_golden_file_decodes_to_the_reference_texels(built, name):
    want = _expected()[name]
    got = _decode(os.path.join(GOLD, name + ".jpg"))
    assert got is not None, "decoder rejected the file"
    assert got.shape == want.shape
    assert np.array_equal(got, want), f"{int((got != want).sum())} differing bytes"


def test_channel_convention(built):
    exp = _expected()
    assert exp["hm_grey_9x7"].shape[2] == 1           # one component -> 1 channel
    assert exp["hm_cmyk_12x10"].shape[2] == 3         # four components fold into RGB
    assert exp["pil_RGB_420_33x17_base_q35_r0"].shape == (17, 33, 3)


@pytest.mark.parametrize("damage", ["truncated_header", "truncated_scan", "no_eoi", "arithmetic", "twelve_bit", "empty", "not_jpeg"])
def test_bad_files_are_rejected_not_crashed(built, tmp_path, damage):
    data = open(os.path.join(GOLD, "pil_RGB_420_33x17_base_q35_r0.jpg"), "rb").read()
    sof = data.index(b"\xff\xc0")
    if damage == "truncated_header":
        data = data[: sof + 6]
    elif damage == "truncated_scan":
        data = data[: len(data) * 3 // 4]      # stb_image: no marker after the entropy data -> failure
    elif damage == "no_eoi":
        data = data[:-2]
    elif damage == "arithmetic":
        data = data[:sof] + b"\xff\xc9" + data[sof + 2:]
    elif damage == "twelve_bit":
        data = data[: sof + 4] + b"\x0c" + data[sof + 5:]
    elif damage == "empty":
        data = b""
    else:
        data = b"\xff\xd8" + b"hello world" * 10
    p = tmp_path / "bad.jpg"
    p.write_bytes(data)
    got = _decode(p)
    if damage in ("truncated_scan", "no_eoi") and got is not None:
        pytest.fail("a file without an end-of-image marker must be refused (stb_image does)")
    assert got is None


@pytest.mark.parametrize("factors", [(0x32, 0x21, 0x11), (0x23, 0x12, 0x11), (0x31, 0x21, 0x11), (0x41, 0x31, 0x11), (0x14, 0x13, 0x11)])
def test_sampling_factors_that_do_not_divide_the_maximum_are_refused(built, tmp_path, factors):
    """SOF with H or V factors that do not divide hmax / vmax, e.g. H = (3,2,1): accepting them made the up-sampler read past the end of the narrower plane rows (advisor finding, round 1:
    heap-buffer-overflow under ASan with this very file patched to (3,2,1))."""
    data = bytearray(open(os.path.join(GOLD, "pil_RGB_420_48x32_base_q35_r0.jpg"), "rb").read())
    sof = data.index(b"\xff\xc0")
    assert data[sof + 9] == 3                      # three components: id, HV, Tq each from sof + 10
    for i, hv in enumerate(factors):
        data[sof + 11 + 3 * i] = hv
    p = tmp_path / "badhv.jpg"
    p.write_bytes(bytes(data))
    # A deliberate divergence from the reference's stb_image v2.25, which only checks 1 <= H,V <= 4 (stb_image.h:3193-3194),
    # accepts these files and up-samples `width` samples out of plane rows that are narrower than that -- it reads across
    # row ends (past the buffer on the last rows) and returns garbage. Later stb_image releases refuse such files; so do we.
    assert _decode(p) is None


def _stb():
    if not os.path.exists(STBREF):
        pytest.skip("oracle/_ref/libstbref.so not built (needs the reference checkout)")
    lib = ctypes.CDLL(STBREF)
    lib.stbi_load.restype = ctypes.c_void_p
    lib.stbi_load.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.c_int]
    lib.stbi_image_free.argtypes = [ctypes.c_void_p]
    return lib


def _stb_decode(lib, path):
    w, h, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    p = lib.stbi_load(str(path).encode(), ctypes.byref(w), ctypes.byref(h), ctypes.byref(c), 0)
    if not p:
        return None
    n = w.value * h.value * c.value
    a = np.ctypeslib.as_array((ctypes.c_uint8 * n).from_address(p)).copy().reshape(h.value, w.value, c.value)
    lib.stbi_image_free(p)
    return a


def test_bad_files_agree_with_the_reference_decoder(built, tmp_path):
    """accept/reject decisions on damaged files are the reference decoder's"""
    lib = _stb()
    data = open(os.path.join(GOLD, "pil_RGB_420_48x32_base_q90_r2.jpg"), "rb").read()
    rng = np.random.default_rng(11)
    agree = 0
    for k in range(60):
        d = bytearray(data)
        if k % 3 == 0:
            d = d[: int(rng.integers(2, len(d)))]
        elif k % 3 == 1:
            pos = int(rng.integers(2, 200))       # header area
            d[pos] = int(rng.integers(0, 256))
        else:
            d = d[:-2]                            # no EOI
            d += bytes(int(rng.integers(0, 4)))
        p = tmp_path / f"d{k}.jpg"
        p.write_bytes(bytes(d))
        a, b = _decode(p), _stb_decode(lib, p)
        if (a is None) == (b is None):
            agree += 1
    assert agree >= 54        # header damage can hit fields only one of the two validates (e.g. unused table slots)


def test_generated_sweep_against_the_reference_decoder(built, tmp_path):
    lib = _stb()
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(77)

    def img(w, h, ch):
        yy, xx = np.mgrid[0:h, 0:w]
        base = (np.sin(xx / 4.0) * 70 + np.cos(yy / 6.0) * 60 + 128)[..., None] + rng.integers(-50, 50, (h, w, ch))
        base[h // 2:, : w // 3] = rng.integers(0, 256, (h - h // 2, w // 3, ch))
        return np.clip(base, 0, 255).astype(np.uint8)
    n = 0
    for (w, h) in ((1, 1), (7, 5), (16, 16), (17, 33), (65, 31), (130, 70)):
        for mode, ch in (("L", 1), ("RGB", 3), ("CMYK", 4)):
            arr = img(w, h, ch)
            im = PIL.fromarray(arr[..., 0] if ch == 1 else arr, mode)
            for sub in ((None,) if ch != 3 else ("4:4:4", "4:2:2", "4:2:0", "4:1:1")):
                for prog in (False, True):
                    for q, rst in ((25, 0), (75, 1), (95, 5)):
                        kw = dict(quality=q, progressive=prog, optimize=(q == 95))
                        if sub:
                            kw["subsampling"] = sub
                        if rst:
                            kw["restart_marker_blocks"] = rst
                        p = tmp_path / "s.jpg"
                        im.save(p, **kw)
                        a, b = _decode(p), _stb_decode(lib, p)
                        assert a is not None and b is not None and a.shape == b.shape
                        assert np.array_equal(a, b), (w, h, mode, sub, prog, q, rst)
                        n += 1
    assert n == 6 * (1 + 4 + 1) * 2 * 3


def test_reference_assets_decode_like_the_reference(built):
    """the nine JPEG files of the default scene (3 planet maps 4:4:4, 6 sky-box faces 4:2:0)"""
    lib = _stb()
    files = [os.path.join(REF_TEXTURES, f) for f in ("2k_mars.jpg", "8k_jupiter.jpg", "8k_saturn.jpg")]
    files += [os.path.join(REF_TEXTURES, "sb_nebula", f"GalaxyTex_{s}{a}.jpg") for s in ("Positive", "Negative") for a in "XYZ"]
    if not all(os.path.exists(f) for f in files):
        pytest.skip("reference assets not present")
    for f in files:
        a, b = _decode(f), _stb_decode(lib, f)
        assert a is not None and b is not None and a.shape == b.shape and a.shape[2] == 3
        assert np.array_equal(a, b), f
