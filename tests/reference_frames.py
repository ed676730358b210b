"""Shared by tools/gen_reference_frames.py and tests/test_reference_frames.py: the cases for which the REFERENCE's own
fragment shader was executed (Mesa llvmpipe, oracle/ref_gl) and its RGBA32F output committed as
tests/golden/ref_frame_<name>.npz.

Each fixture holds the nine std140 scene blocks, the texture-set scale (the procedural textures are regenerated from
raytracing_opengl_amd/textures.py and checked against a stored SHA-256) and the reference frame.

`limits` are acceptance bounds for |candidate - reference| per pixel (max over RGB): fraction of pixels allowed over
1e-4 and over 1e-2. They are the measured oracle-vs-reference figures with head-room; the reasons the two cannot agree
everywhere are listed in DESIGN.md section 2 (llvmpipe evaluates normalize() as v*rsqrt(dot), fuses differently, and its
mip-map level selection is an approximation the GL specification allows; Durand-Kerner stops at 1e-3)."""
import hashlib
import os
import struct

import numpy as np

from raytracing_opengl_amd import scenes, textures
from raytracing_opengl_amd.scenes import BLOCK_NAMES, SceneBlocks

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
W, H = 256, 144
TEX_SCALE = 8


def _strip_textures(sc, spheres=True, rings=True, boxes=True):
    """textureNum := 0 (rt.frag:749,763,772): untextured variant of a scene."""
    def patch(name, rec, off):
        b = bytearray(sc.blocks[name])
        for i in range(len(b) // rec):
            b[i * rec + off:i * rec + off + 4] = struct.pack("<i", 0)
        sc.blocks[name] = bytes(b)
    if spheres: patch("spheres_buf", 112, 96)
    if rings: patch("rings_buf", 112, 92)
    if boxes: patch("boxes_buf", 112, 108)
    return sc


def _trap(name):
    import trap_scenes
    return trap_scenes.ALL[name](W, H)


# name -> (scene builder, textured?, (max fraction > 1e-4, max fraction > 1e-2))
CASES = {
    "default_untextured": (lambda: _strip_textures(scenes.build_scene("default", W, H, 4)), False, (0.004, 0.0006)),
    "default": (lambda: scenes.build_scene("default", W, H, 4), True, (0.10, 0.012)),
    "quadric": (lambda: scenes.build_scene("quadric", W, H, 4), False, (0.006, 0.0006)),
    "torus": (lambda: scenes.build_scene("torus", W, H, 4), False, (0.09, 0.003)),
    "trap_inside_box": (lambda: _trap("inside_box"), False, (0.0005, 0.0)),
    "trap_degenerate_rings_untextured": (lambda: _strip_textures(_trap("degenerate_rings")), False, (0.0005, 0.0)),
    "trap_degenerate_rings": (lambda: _trap("degenerate_rings"), True, (0.12, 0.03)),
    "trap_glass_tir": (lambda: _trap("glass_tir"), False, (0.06, 0.012)),
    "trap_planes_glass": (lambda: _trap("planes_glass"), False, (0.003, 0.0005)),
    # ---- round 3: more of the reference's own runs -------------------------------------------------------------------------------------
    # BASELINE configs[0]: the animated default scene at t = 12.5 with the camera the reference's host code produces for yaw 25 / pitch 10
    # (the same recipe as tests/golden/default_t12.5_640x480_d1.rtxb, which is the output of the reference's own SceneManager), reflection
    # depth 1, at half the configuration's 640 x 480; with the textures off and as the reference runs it
    "config0_untextured": (lambda: _strip_textures(scenes.build_scene("default", 320, 240, 1, time=12.5, delta=0.75, yaw=25.0, pitch=10.0)), False, (0.004, 0.0006)),
    "config0": (lambda: scenes.build_scene("default", 320, 240, 1, time=12.5, delta=0.75, yaw=25.0, pitch=10.0), True, (0.10, 0.012)),
    # ---- round 5: the two textured plain runs again with BAND-LIMITED textures (textures.default_texture_set(smooth=True): no per-texel grain,
    # no steps): which mip level an implementation takes then hardly matters, and the `texture` class is held at 5e-3 instead of 0.1
    "default_smooth": (lambda: scenes.build_scene("default", W, H, 4), True, (0.10, 0.004)),
    "config0_smooth": (lambda: scenes.build_scene("default", 320, 240, 1, time=12.5, delta=0.75, yaw=25.0, pitch=10.0), True, (0.10, 0.004)),
    # the reference's own default run (main.cpp:7-8: 1280 x 720; SceneManager.cpp:233: reflect_depth 5; main.cpp:197-246: animated) at a
    # quarter of its size, two animation times
    "app_default_t3": (lambda: _strip_textures(scenes.build_scene("default", 320, 180, 5, time=3.0, delta=0.016)), False, (0.004, 0.0006)),
    "app_default_t7_5": (lambda: _strip_textures(scenes.build_scene("default", 320, 180, 5, time=7.5, delta=0.016)), False, (0.004, 0.0006)),
    # ---- round 4: the two configurations the reference itself is run at, at their FULL size and as the reference runs them (textured,
    # glGenerateMipmap): BASELINE configs[0] (640 x 480, depth 1, the animated scene at t = 12.5 under the yaw 25 / pitch 10 camera) and the
    # reference's own program (main.cpp:7-8: 1280 x 720; SceneManager.cpp:233: reflect_depth 5) at animation time t = 3
    "config0_full": (lambda: scenes.build_scene("default", 640, 480, 1, time=12.5, delta=0.75, yaw=25.0, pitch=10.0), True, (0.10, 0.012)),
    "app_default_full": (lambda: scenes.build_scene("default", 1280, 720, 5, time=3.0, delta=0.016), True, (0.10, 0.012)),
    # ---- round 5: GLWrapper::load_cubemap(faces, genMipmap = true) (GLWrapper.cpp:307-310: glGenerateMipmap(GL_TEXTURE_CUBE_MAP) and
    # GL_LINEAR_MIPMAP_LINEAR): the sky fetch rt.frag:893 is mip-mapped. Objects untextured, so the sky is the ONLY mip-mapped fetch of
    # these frames; band-limited sky faces (textures.nebula_face(smooth=True)) hold the `texture` class at 5e-3
    "moved_cube_mips": (lambda: _strip_textures(scenes.build_scene("default", W, H, 4, time=1.25, delta=0.016, yaw=-38.0, pitch=-14.0, cam_pos=(2.5, 1.5, -6.0))), True, (0.30, 0.016)),
    "config0_cube_mips": (lambda: _strip_textures(scenes.build_scene("default", 320, 240, 1, time=12.5, delta=0.75, yaw=25.0, pitch=10.0)), True, (0.03, 0.002)),
    # a moved and rotated camera (SceneManager.cpp:43-50: quat(vec3(radians(-pitch), radians(yaw), 0)))
    "moved_camera": (lambda: _strip_textures(scenes.build_scene("default", W, H, 4, time=1.25, delta=0.016, yaw=-38.0, pitch=-14.0, cam_pos=(2.5, 1.5, -6.0))), False, (0.004, 0.0006)),
}
# random content: eight seeds of tests/random_scenes.py (what tools/fuzz_reference.py sweeps, textures off) as committed frames
FUZZ_SEEDS = (3, 11, 19, 28, 42, 57, 64, 90)
FUZZ_SIZE = (112, 64)


def _fuzz(seed):
    import random_scenes
    return _strip_textures(random_scenes.random_scene(seed, *FUZZ_SIZE))


for _s in FUZZ_SEEDS:
    CASES[f"fuzz_{_s}"] = ((lambda s=_s: _fuzz(s)), False, (0.02, 0.004))
SIZES = {"config0_untextured": (320, 240), "config0": (320, 240), "config0_cube_mips": (320, 240), "config0_smooth": (320, 240), "app_default_t3": (320, 180), "app_default_t7_5": (320, 180),
         "config0_full": (640, 480), "app_default_full": (1280, 720)}
SIZES.update({f"fuzz_{_s}": FUZZ_SIZE for _s in FUZZ_SEEDS})


def size(name):
    """(width, height) of a case's frame; the *_same_mips / *_level0 variants have their base case's size."""
    for suffix in ("_same_mips", "_level0"):
        if name.endswith(suffix):
            name = name[: -len(suffix)]
    return SIZES.get(name, (W, H))
# Diagnostic fixture: the default scene again, but the GL textures were given the ORACLE's mip levels (glTexImage2D per
# level) instead of glGenerateMipmap, so that only level selection and filtering are compared. Checked with the oracle in
# its llvmpipe-LOD mode (texture_lod = 2): what is left is llvmpipe's atan/asin approximation on the planets + silhouettes.
SAME_MIPS = ("default_same_mips", 0.025, 0.008)   # name, max fraction > 1e-4, > 1e-2
# Variants of the two textured cases (same scene blocks and textures, different GL texture state; oracle/ref_gl/ref_gl.py):
#   <case>_same_mips : GL was given the oracle's mip texels -> compared with the oracle in its llvmpipe LOD mode (texture_lod = 2)
#   <case>_level0    : GL was given level 0 only (GL_TEXTURE_MAX_LEVEL = 0) -> compared with the oracle at texture_lod = 0
TEXTURED = ("default", "trap_degenerate_rings")     # pinned three ways (variants below)
TEXTURED_PLAIN_ONLY = ("config0",)                  # textured, plain run only (with llvmpipe's generated mip levels stored)
SMOOTH = ("default_smooth", "config0_smooth")       # textured with the band-limited set, plain run only, llvmpipe's mip levels stored
CUBE_MIPS = ("moved_cube_mips", "config0_cube_mips")   # the sky box loaded with genMipmap = true (objects untextured, band-limited faces)
FULL_SIZE = ("config0_full", "app_default_full")    # textured, plain run only, at the configuration's own size: the pixel-by-pixel accounting of
                                                    # these runs on the GPU box (GPU_PLAN; its host has the cores for the oracle's probes), the
                                                    # CPU suite holds the oracle to their limits only
# fixtures that also hold what the FIRST calcInter of every pixel returned in the reference's shader (t, type, num): instrumented run
PRIMARY_HITS = ("torus", "default_untextured")
VARIANTS = {f"{c}_{v}": (c, v) for c in TEXTURED for v in ("same_mips", "level0")}


def texture_set(name: str = ""):
    """The texture set of a case: the band-limited one for the *_smooth cases, band-limited sky faces for the *_cube_mips cases."""
    return textures.default_texture_set(scale=TEX_SCALE, smooth=name.endswith("_smooth"), smooth_sky=cube_mipmap(name))


def cube_mipmap(name: str) -> bool:
    """Was the case's sky box loaded with load_cubemap(faces, genMipmap = true)?"""
    return name.endswith("_cube_mips")


def input_digest(sc, ts) -> str:
    h = hashlib.sha256()
    h.update(repr(tuple(int(v) for v in sc.defines[:9])).encode())
    h.update(np.asarray(sc.defines[9:15], dtype=np.float32).tobytes())
    for name in BLOCK_NAMES:
        h.update(sc.blocks.get(name, b""))
    for uniform, unit, img in ts["textures"]:
        h.update(uniform.encode()); h.update(bytes([unit])); h.update(np.ascontiguousarray(img).tobytes())
    for f in ts["cubemap"]:
        h.update(np.ascontiguousarray(f).tobytes())
    return h.hexdigest()


def path(name):
    return os.path.join(GOLDEN, f"ref_frame_{name}.npz")


_LOADED = {}


def load(name):
    """-> dict(scene, textures, cubemap, frame (H,W,3 float32, row 0 = bottom), limits, renderer)"""
    if name in _LOADED:
        return _LOADED[name]
    z = np.load(path(name))
    d = z["defines"]
    defines = tuple(int(v) for v in d[:9]) + tuple(float(np.float32(v)) for v in d[9:15])
    blocks = {n: z["block_" + n].tobytes() if ("block_" + n) in z.files else b"" for n in BLOCK_NAMES}
    sc = SceneBlocks(defines=defines, blocks=blocks)
    ts = texture_set(name)
    if input_digest(sc, ts) != str(z["digest"]):
        raise RuntimeError(f"inputs of reference frame '{name}' no longer reproduce (textures.py changed?)")
    # llvmpipe's own mip levels (plain textured runs only), stored as differences from the oracle's integer-mean levels
    gl_mips = None
    if any(k.startswith("glmip_") for k in z.files):
        from oracle import oracle
        O = oracle.OracleScene(sc, int(z["width"]), int(z["height"]), ts["textures"], ts["cubemap"])
        gl_mips = {}
        for uniform, _unit, _img in ts["textures"]:
            ours = O.mip_levels(uniform)
            gl_mips[uniform] = [(a.astype(np.int16) + z[f"glmip_{uniform}_{L}"].astype(np.int16)).astype(np.uint8) for L, a in enumerate(ours, start=1)]
    primary = None
    if "primary_t" in z.files:
        primary = dict(t=z["primary_t"], type=z["primary_type"].astype(np.int32), num=z["primary_num"].astype(np.int32))
    _LOADED[name] = dict(cube_mipmap=cube_mipmap(name), primary=primary, gl_mips=gl_mips, name=name, scene=sc, width=int(z["width"]), height=int(z["height"]), textures=ts["textures"], cubemap=ts["cubemap"],
                         frame=z["frame"], limits=CASES[name][2] if name in CASES else SAME_MIPS[1:], renderer=str(z["renderer"]))
    return _LOADED[name]


def textured(name) -> bool:
    return CASES[name][1] if name in CASES else True


def compare(candidate, reference_rgb):
    """-> (fraction of pixels over 1e-4, fraction over 1e-2, max abs difference); NaN anywhere counts as over both."""
    d = np.abs(candidate[..., :3].astype(np.float64) - reference_rgb.astype(np.float64)).max(axis=2)
    d = np.where(np.isnan(d), np.inf, d)
    return float((d > 1e-4).mean()), float((d > 1e-2).mean()), float(d.max())
