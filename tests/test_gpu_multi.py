"""Multi-device contexts behind the C ABI (rtx_create_multi; SURVEY section 8(e), BASELINE north_star "GLWrapper dispatch -> HIP launch +
RCCL tile gather"): the frame a group of N ranks assembles on its root must be BIT-IDENTICAL to the frame one device traces -- both colour
targets, ray counters summed over the ranks, across consecutive frames (the buffer sets alternate), scene updates in between and with the
SMAA resolve on top. A GPU box here has ONE device, so the band split, the transfer plumbing and the placement kernel are exercised with
RTX_GATHER_PEER_COPY and all ranks on device 0; the RCCL transport runs where at least two devices exist (skipped otherwise) and is checked
for its error behaviour everywhere."""
import numpy as np
import pytest

from raytracing_opengl_amd import scenes, smaa_tables, wrapper

pytestmark = pytest.mark.gpu


def _n_devices():
    import torch
    return torch.cuda.device_count()


def _frames(gl, sc_list):
    out = []
    for sc in sc_list:
        if sc is not None:
            gl.uploader.update(sc)
        gl.draw()
        out.append((gl.read_pixels(wrapper.RTX_RGBA32F), gl.read_pixels(wrapper.RTX_RGBA8), gl.stats()))
    return out


@pytest.mark.parametrize("ranks", [2, 3, 5])
@pytest.mark.parametrize("kind,w,h,depth", [("default", 640, 360, 4), ("torus", 333, 207, 6), ("quadric", 320, 100, 4)])
def test_peer_copy_group_reproduces_the_single_device_frame(small_textures, ranks, kind, w, h, depth):
    seq = [None, scenes.build_scene(kind, w, h, depth, time=3.0, delta=0.1, yaw=20.0), scenes.build_scene(kind, w, h, depth, time=6.5, delta=0.1, yaw=-15.0, pitch=4.0), None]
    sc0 = scenes.build_scene(kind, w, h, depth)
    single = wrapper.make_renderer(sc0, w, h, small_textures["textures"], small_textures["cubemap"])
    single.set_option(wrapper.RTX_OPT_COUNT_RAYS, 1)
    want = _frames(single, seq)
    single.stop()
    group = wrapper.make_renderer(sc0, w, h, small_textures["textures"], small_textures["cubemap"], devices=[0] * ranks, gather=wrapper.RTX_GATHER_PEER_COPY)
    group.set_option(wrapper.RTX_OPT_COUNT_RAYS, 1)
    got = _frames(group, seq)
    group.stop()
    for k, ((f32, u8, st), (g32, g8, gst)) in enumerate(zip(want, got)):
        assert np.array_equal(f32.view(np.uint32), g32.view(np.uint32)), (k, int((f32.view(np.uint32) != g32.view(np.uint32)).sum()))
        assert np.array_equal(u8, g8), k
        assert (st["rays_closest"], st["rays_shadow"]) == (gst["rays_closest"], gst["rays_shadow"]), k
        assert gst["last_gather_ms"] > 0.0


@pytest.mark.parametrize("gather,devices", [(wrapper.RTX_GATHER_PEER_COPY, [0, 0, 0]), (wrapper.RTX_GATHER_RCCL_LOOPBACK, [0])])
def test_float_bands_travel_without_their_alpha_or_with_it(small_textures, gather, devices):
    """RTX_OPT_GATHER_RGB (default 1): the RGBA32F bands go to the root at 12 bytes per pixel -- a pack kernel on every sending rank's transfer
    stream, the 1.0f written back by the placement on the root -- or, switched off between two frames, as whole pixels. Either way the
    assembled frame is the single-device frame bit for bit, alpha included, at an odd width (rows of 333 x 12 bytes are not 16-byte units)."""
    w, h, depth = 333, 207, 4
    seq = [None, scenes.build_scene("default", w, h, depth, time=2.0, delta=0.1, yaw=10.0), None]
    sc0 = scenes.build_scene("default", w, h, depth)
    single = wrapper.make_renderer(sc0, w, h, small_textures["textures"], small_textures["cubemap"])
    want = _frames(single, seq)
    single.stop()
    assert all(np.all(f32[..., 3] == 1.0) for f32, _, _ in want)          # what the option rests on (rt.frag:902)
    group = wrapper.make_renderer(sc0, w, h, small_textures["textures"], small_textures["cubemap"], devices=devices, gather=gather)
    assert group.get_option(wrapper.RTX_OPT_GATHER_RGB) == 1
    for value in (1, 0, 1):
        group.set_option(wrapper.RTX_OPT_GATHER_RGB, value)
        assert group.get_option(wrapper.RTX_OPT_GATHER_RGB) == value
        group.uploader.update(sc0)
        got = _frames(group, seq)
        for k, ((f32, u8, _), (g32, g8, _)) in enumerate(zip(want, got)):
            assert np.array_equal(f32.view(np.uint32), g32.view(np.uint32)), (value, k, int((f32.view(np.uint32) != g32.view(np.uint32)).sum()))
            assert np.array_equal(u8, g8), (value, k)
    group.stop()


@pytest.mark.parametrize("kind,w,h,depth,ranks", [("default", 640, 360, 4, 4), ("torus", 333, 207, 6, 3), ("quadric", 320, 100, 4, 2)])
def test_contiguous_bands_reproduce_the_single_device_frame(small_textures, kind, w, h, depth, ranks):
    """RTX_OPT_BAND_LAYOUT 1 / 2 (round 4): one contiguous range of rows per rank, the root's range traced straight into the colour targets,
    the others received straight into place (no landing buffers, no placement pass) -- equal ranges, a caller's weighted split
    (rtx_set_band_split), the library's own re-balancing over a run of frames with scene updates in between, back to interleaved bands:
    every frame bit-identical to the single device's, both targets, summed ray counters; odd frame heights incl."""
    seq = [None, scenes.build_scene(kind, w, h, depth, time=3.0, delta=0.1, yaw=20.0), None, scenes.build_scene(kind, w, h, depth, time=6.5, delta=0.1, yaw=-15.0, pitch=4.0), None, None]
    sc0 = scenes.build_scene(kind, w, h, depth)
    single = wrapper.make_renderer(sc0, w, h, small_textures["textures"], small_textures["cubemap"])
    single.set_option(wrapper.RTX_OPT_COUNT_RAYS, 1)
    want = _frames(single, seq)
    single.stop()

    def check(group, tag):
        got = _frames(group, seq)
        for k, ((f32, u8, st), (g32, g8, gst)) in enumerate(zip(want, got)):
            assert np.array_equal(f32.view(np.uint32), g32.view(np.uint32)), (tag, k, int((f32.view(np.uint32) != g32.view(np.uint32)).sum()))
            assert np.array_equal(u8, g8), (tag, k)
            assert (st["rays_closest"], st["rays_shadow"]) == (gst["rays_closest"], gst["rays_shadow"]), (tag, k)
        group.uploader.update(sc0)

    group = wrapper.make_renderer(sc0, w, h, small_textures["textures"], small_textures["cubemap"], devices=[0] * ranks, gather=wrapper.RTX_GATHER_PEER_COPY)
    group.set_option(wrapper.RTX_OPT_COUNT_RAYS, 1)
    group.set_option(wrapper.RTX_OPT_BAND_LAYOUT, 1)
    split = group.band_split()
    assert sum(split) == h and all(v % 8 == 0 for v in split[:-1]) and len(split) == ranks, split
    check(group, "equal ranges")
    # a weighted split: most of the frame to the last rank, one tile row to the first
    units = (h + 7) // 8
    rows = [8] * (ranks - 1)
    rows.append(h - sum(rows))
    assert rows[-1] > 0 and units >= ranks
    group.set_band_split(rows)
    assert group.band_split() == rows
    check(group, "weighted split")
    # a rank without any rows (a split may starve one): its launch is empty, its events still complete, nothing travels for it
    if ranks >= 3:
        starved = [8, 0] + [8] * (ranks - 3) + [h - 8 * (ranks - 2)]
        group.set_band_split(starved)
        assert group.band_split() == starved
        check(group, "a rank without rows")
        assert group.rank_draw_ms()[1] >= 0.0
    with pytest.raises(wrapper.RtxError, match="multiple of 8|cover"):
        group.set_band_split([5] + [h - 5] + [0] * (ranks - 2))
    with pytest.raises(wrapper.RtxError, match="cover"):
        group.set_band_split([8] * ranks)
    # the library's own re-balancing: more frames than it needs to move the boundaries, scene updates in between
    group.set_option(wrapper.RTX_OPT_BAND_LAYOUT, 2)
    before = group.band_split()
    for _ in range(3):
        check(group, "re-balanced")
    after = group.band_split()
    assert sum(after) == h and all(v % 8 == 0 and v >= 8 for v in after[:-1]), after
    ms = group.rank_draw_ms()
    assert len(ms) == ranks and all(v > 0 for v in ms), ms
    group.set_option(wrapper.RTX_OPT_BAND_LAYOUT, 0)
    check(group, "back to interleaved bands")
    group.stop()
    print(f"{kind} {w}x{h} over {ranks} ranks: split {before} -> {after} after re-balancing, rank kernel ms {['%.3f' % v for v in ms]}")


def test_contiguous_bands_over_the_rccl_transport_and_with_smaa(small_textures):
    """The same layout through RCCL (loopback: the root's own range travels too and is received in place) in both launch forms, with the
    SMAA resolve on the assembled frame."""
    w, h, depth = 480, 272, 4
    sc = scenes.build_scene("default", w, h, depth)
    tables = smaa_tables.area_table(), smaa_tables.search_table()
    single = wrapper.make_renderer(sc, w, h, small_textures["textures"], small_textures["cubemap"])
    single.enable_SMAA("ULTRA")
    single.set_smaa_tables(*tables)
    single.draw()
    want32, want8, want_screen = single.read_pixels(wrapper.RTX_RGBA32F), single.read_pixels(wrapper.RTX_RGBA8), single.read_pixels(wrapper.RTX_SCREEN_RGBA8)
    single.stop()
    uid = wrapper.rccl_unique_id()
    for kw in (dict(devices=[0], gather=wrapper.RTX_GATHER_RCCL_LOOPBACK), dict(gather=wrapper.RTX_GATHER_RCCL_LOOPBACK, rank=(0, 1, uid))):
        g = wrapper.make_renderer(sc, w, h, small_textures["textures"], small_textures["cubemap"], **kw)
        g.set_option(wrapper.RTX_OPT_BAND_LAYOUT, 1)
        g.enable_SMAA("ULTRA")
        g.set_smaa_tables(*tables)
        for _ in range(3):
            g.draw()
        assert np.array_equal(g.read_pixels(wrapper.RTX_RGBA32F).view(np.uint32), want32.view(np.uint32))
        assert np.array_equal(g.read_pixels(wrapper.RTX_RGBA8), want8)
        assert np.array_equal(g.read_pixels(wrapper.RTX_SCREEN_RGBA8), want_screen)
        if "rank" in kw:
            with pytest.raises(wrapper.RtxError, match="one process"):
                g.set_option(wrapper.RTX_OPT_BAND_LAYOUT, 2)
        g.stop()


def test_group_with_smaa_and_target_selection(small_textures):
    w, h = 480, 272
    sc = scenes.build_scene("default", w, h, 4)
    tables = smaa_tables.area_table(), smaa_tables.search_table()
    single = wrapper.make_renderer(sc, w, h, small_textures["textures"], small_textures["cubemap"])
    single.enable_SMAA("ULTRA")
    single.set_smaa_tables(*tables)
    single.draw()
    want8, want_screen = single.read_pixels(wrapper.RTX_RGBA8), single.read_pixels(wrapper.RTX_SCREEN_RGBA8)
    single.stop()
    group = wrapper.make_renderer(sc, w, h, small_textures["textures"], small_textures["cubemap"], devices=[0, 0, 0, 0], gather=wrapper.RTX_GATHER_PEER_COPY)
    group.set_option(wrapper.RTX_OPT_GATHER_TARGETS, 2)          # only what the reference's framebuffer holds travels
    group.enable_SMAA("ULTRA")
    group.set_smaa_tables(*tables)
    for _ in range(3):
        group.draw()
    assert np.array_equal(group.read_pixels(wrapper.RTX_RGBA8), want8)
    assert np.array_equal(group.read_pixels(wrapper.RTX_SCREEN_RGBA8), want_screen)
    with pytest.raises(wrapper.RtxError, match="not gathered"):
        group.read_pixels(wrapper.RTX_RGBA32F)
    group.stop()


def test_one_rank_group_is_a_plain_context_and_bad_arguments(small_textures):
    w, h = 160, 96
    sc = scenes.build_scene("default", w, h, 2)
    a = wrapper.make_renderer(sc, w, h, small_textures["textures"], small_textures["cubemap"])
    a.draw()
    want = a.read_pixels()
    a.stop()
    b = wrapper.make_renderer(sc, w, h, small_textures["textures"], small_textures["cubemap"], devices=[0])
    b.draw()
    assert np.array_equal(b.read_pixels().view(np.uint32), want.view(np.uint32))
    b.stop()
    gl = wrapper.GLWrapper(w, h, devices=[0, 0])                 # RCCL needs one device per rank
    assert not gl.init_window() and "listed twice" in gl.last_error
    gl = wrapper.GLWrapper(w, h, devices=[0, 99], gather=wrapper.RTX_GATHER_PEER_COPY)
    assert not gl.init_window() and "out of range" in gl.last_error


@pytest.mark.parametrize("ranks", [2, 4, 8])
def test_rccl_group_reproduces_the_single_device_frame(small_textures, ranks):
    """The RCCL transport proper: needs `ranks` devices (the driver's 8-GPU node; skipped on the single-GPU boxes)."""
    if _n_devices() < ranks:
        pytest.skip(f"{ranks} devices needed, {_n_devices()} present")
    w, h, depth = 1280, 720, 4
    seq = [None, scenes.build_scene("default", w, h, depth, time=2.0, delta=0.1), None]
    sc0 = scenes.build_scene("default", w, h, depth)
    single = wrapper.make_renderer(sc0, w, h, small_textures["textures"], small_textures["cubemap"])
    want = _frames(single, seq)
    single.stop()
    group = wrapper.make_renderer(sc0, w, h, small_textures["textures"], small_textures["cubemap"], devices=list(range(ranks)), gather=wrapper.RTX_GATHER_RCCL)
    got = _frames(group, seq)
    group.stop()
    for (f32, u8, _), (g32, g8, _g) in zip(want, got):
        assert np.array_equal(f32.view(np.uint32), g32.view(np.uint32)) and np.array_equal(u8, g8)


# ---- the RCCL transport on a single-GPU box: loopback (rank 0 sends its own bands to itself) -------------------------------------------
def test_rccl_loopback_runs_the_transport_on_one_device(small_textures):
    """RTX_GATHER_RCCL_LOOPBACK with one device: librccl is loaded, a communicator created, and every frame's bands go through a grouped
    ncclSend/ncclRecv pair and the placement kernel on the transfer stream -- the code of the N-device gather, executed."""
    w, h, depth = 640, 360, 4
    seq = [None, scenes.build_scene("default", w, h, depth, time=2.0, delta=0.1, yaw=10.0), None, None]
    sc0 = scenes.build_scene("default", w, h, depth)
    single = wrapper.make_renderer(sc0, w, h, small_textures["textures"], small_textures["cubemap"])
    single.set_option(wrapper.RTX_OPT_COUNT_RAYS, 1)
    want = _frames(single, seq)
    single.stop()
    group = wrapper.make_renderer(sc0, w, h, small_textures["textures"], small_textures["cubemap"], devices=[0], gather=wrapper.RTX_GATHER_RCCL_LOOPBACK)
    group.set_option(wrapper.RTX_OPT_COUNT_RAYS, 1)
    got = _frames(group, seq)
    for k, ((f32, u8, st), (g32, g8, gst)) in enumerate(zip(want, got)):
        assert np.array_equal(f32.view(np.uint32), g32.view(np.uint32)), k
        assert np.array_equal(u8, g8), k
        assert (st["rays_closest"], st["rays_shadow"]) == (gst["rays_closest"], gst["rays_shadow"]), k
        assert gst["last_gather_ms"] > 0.0
    # only one target travels when asked so, and the other is then refused
    group.set_option(wrapper.RTX_OPT_GATHER_TARGETS, 2)
    group.draw()
    assert np.array_equal(group.read_pixels(wrapper.RTX_RGBA8), want[-1][1])
    with pytest.raises(wrapper.RtxError, match="not gathered"):
        group.read_pixels(wrapper.RTX_RGBA32F)
    group.stop()


def test_rank_context_of_a_one_process_per_gpu_split(small_textures):
    """rtx_create_rank (what bench.py uses under torch.distributed.run): rank 0 of 1 with the loopback transport -- unique id,
    ncclCommInitRank, send/recv, placement, SMAA on the assembled frame -- against the plain context."""
    w, h, depth = 480, 272, 4
    sc = scenes.build_scene("default", w, h, depth)
    tables = smaa_tables.area_table(), smaa_tables.search_table()
    single = wrapper.make_renderer(sc, w, h, small_textures["textures"], small_textures["cubemap"])
    single.enable_SMAA("ULTRA")
    single.set_smaa_tables(*tables)
    single.draw()
    want32, want8, want_screen = single.read_pixels(wrapper.RTX_RGBA32F), single.read_pixels(wrapper.RTX_RGBA8), single.read_pixels(wrapper.RTX_SCREEN_RGBA8)
    single.stop()
    uid = wrapper.rccl_unique_id()
    assert len(uid) == 128 and any(uid)
    r0 = wrapper.make_renderer(sc, w, h, small_textures["textures"], small_textures["cubemap"], gather=wrapper.RTX_GATHER_RCCL_LOOPBACK, rank=(0, 1, uid))
    r0.enable_SMAA("ULTRA")
    r0.set_smaa_tables(*tables)
    for _ in range(3):
        r0.draw()
    assert np.array_equal(r0.read_pixels(wrapper.RTX_RGBA32F).view(np.uint32), want32.view(np.uint32))
    assert np.array_equal(r0.read_pixels(wrapper.RTX_RGBA8), want8)
    assert np.array_equal(r0.read_pixels(wrapper.RTX_SCREEN_RGBA8), want_screen)
    r0.stop()
    # a plain rank (1 of 1, no loopback) is a plain context; bad arguments are refused before any communicator is built
    p = wrapper.make_renderer(sc, w, h, small_textures["textures"], small_textures["cubemap"], rank=(0, 1, uid))
    p.draw()
    assert np.array_equal(p.read_pixels(wrapper.RTX_RGBA32F).view(np.uint32), want32.view(np.uint32))
    p.stop()
    gl = wrapper.GLWrapper(w, h, rank=(2, 2, uid))
    assert not gl.init_window() and "bad arguments" in gl.last_error
    gl = wrapper.GLWrapper(w, h, rank=(0, 1, uid), gather=wrapper.RTX_GATHER_PEER_COPY)
    assert not gl.init_window() and "RCCL" in gl.last_error


def test_first_contact_between_ranks_in_separate_processes():
    """ranks that disagree about the frame configuration are told so (RTX_ERR_INVALID naming the rank) and a rank that never
    calls is a bounded wait's timeout (RTX_ERR_DEVICE), not a hang -- on real devices, two processes (tools/first_contact_check.py). Needs two GPUs; the
    protocol itself runs on the CPU against a fake transport in tests/test_band_math.py."""
    import os, subprocess, sys
    if _n_devices() < 2:
        pytest.skip(f"2 devices needed, {_n_devices()} present")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29544",
           os.path.join(root, "tools", "first_contact_check.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "1. 2 ranks agree" in r.stdout and "3. rank 0: a silent rank is a timeout" in r.stdout, r.stdout[-2000:]


def _run_bench(extra, torchrun=False, timeout=600):
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    head = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", "29533"] if torchrun else [sys.executable]
    cmd = head + [os.path.join(root, "bench.py"), "--width", "640", "--height", "360", "--steps", "4", "--warmup", "2", "--texture-scale", "16", "--no-cpu-baseline", "--no-smaa"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode == 0:   # stdout is the ONE line of JSON and nothing else (RCCL's version banner used to precede it)
        assert [l for l in r.stdout.splitlines() if l.strip()] == lines[-1:], r.stdout[:2000]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bench_multi_gpu_forms_run_through_the_c_boundary():
    """bench.py's N > 1 code on one GPU: the single-process form (rtx_create_multi) and the torch.distributed.run form (rtx_create_rank),
    both with the loopback transport; and --gpus 2 on a box without a second GPU is one clear line, exit code 2."""
    r, line = _run_bench(["--gpus", "1", "--transport", "loopback", "--also-bands"])
    assert r.returncode == 0 and line, r.stderr[-2000:]
    assert line["n_gpus"] == 1 and line["parity"]["vs_one_device_tracing_the_whole_frame"] == "bit-identical"
    assert "rtx_create_multi" in line["config"]["parallelism"] and line["config"]["gather_ms"] > 0 and line["config"]["trace_ms_max_rank"] > 0
    # one line carries what a single run on an 8-GPU node has to answer: per-rank kernel times, the split, gather rate, and -- timed like the
    # value -- the other colour target and the other band layout
    cfg = line["config"]
    assert len(cfg["trace_ms_per_rank"]) == 1 and cfg["rows_per_rank"] == [360] and cfg["gather_GB_s_into_rank0"] > 0 and cfg["clock_ramp"]["ms"] >= 100
    also = cfg["also_measured"]
    assert also["rgba8"]["ms_per_step"] > 0 and also["bands_balanced"]["ms_per_step"] > 0 and sum(also["bands_balanced"]["rows_per_rank"]) == 360
    r, line = _run_bench(["--gpus", "1", "--transport", "loopback", "--bands", "balanced", "--target", "rgba8", "--also-bands"])
    assert r.returncode == 0 and line and "weighted by kernel time" in line["config"]["parallelism"], r.stderr[-2000:]
    assert line["parity"]["vs_one_device_tracing_the_whole_frame"] == "bit-identical" and "bands_interleaved" in line["config"]["also_measured"]
    r, line = _run_bench(["--gpus", "1", "--transport", "loopback"], torchrun=True)
    assert r.returncode == 0 and line, r.stderr[-2000:]
    assert "rtx_create_rank" in line["config"]["parallelism"] and line["parity"]["vs_one_device_tracing_the_whole_frame"] == "bit-identical"
    if _n_devices() < 2:
        r, line = _run_bench(["--gpus", "2"])
        assert r.returncode == 2 and line is None
        err = [l for l in r.stderr.splitlines() if l.strip() and "amdgpu.ids" not in l]
        assert len(err) == 1 and "--gpus 2 requested but this box has 1 GPU" in err[0], r.stderr
    r, line = _run_bench(["--gpus", "1"])
    assert r.returncode == 0 and line["n_gpus"] == 1 and line["config"]["parallelism"] == "single GPU"
