#!/usr/bin/env python3
"""bench.py -- headline benchmark of the rt.frag path on MI355X.

Metric (BASELINE.json): Mray/s (+ ms/frame) at 3840x2160, reflection depth 4, default scene.
A "step" = one full frame traced from blocks/textures already resident in HBM. A ray = one
closest-hit scan (calcInter) or one shadow scan (inShadow) as the REFERENCE would execute them
(SURVEY.md section 8(d)); the per-frame count is exact (kernel counter == oracle counter, see
tests/test_gpu_parity.py) and is measured once, untimed, with the counting kernel variant.

N > 1: the SAME frame is split into interleaved 8-row bands (band b -> rank b mod N), every rank traces its bands, and the frame is
gathered to rank 0 over RCCL each step -> "scaling": "strong". The draw goes through the C boundary north_star names ("GLWrapper dispatch
-> HIP launch + RCCL tile gather", reference GLWrapper.cpp:155-165 called from main.cpp:188) in both launch forms:
  python bench.py --gpus N                      one process drives N devices: rtx_create_multi(..., RTX_GATHER_RCCL), one rtx_draw per step
  python -m torch.distributed.run ... bench.py  one process per GPU: every rank holds an rtx_create_rank context (RCCL communicator from a
                                                unique id that torch.distributed only hands round), one rtx_draw per step on every rank
(--launcher torch makes the first form re-exec itself as the second; --transport torch keeps the round-1/2 Python path, bands.py +
dist.gather, for comparison.) The target is the same at every N (default: the RGBA32F parity buffer the metric is defined on; --target
rgba8 traces and gathers what the reference's framebuffer holds, GLWrapper.cpp:127,209-222, a quarter of the bytes), so a 1/2/4/8 series
is one workload. A box with fewer than N GPUs answers --gpus N with one line and exit code 2.

Extra objects on the JSON line:
  roofline     HBM-write roofline of the trace kernel: W*H*16 B of RGBA32F per launch / mean kernel
               time from HIP events on the launch stream, against 8 TB/s; .valu = the ceiling that actually binds
               (VALU instructions per launch from profiles/valu.json / live duration vs the chip's issue rate).
  cpu_baseline the oracle (scalar C restatement of the shader) on the CPUs this container may use (affinity mask capped by the cgroup
               quota: 16 of the 256 hardware threads a GPU box shows), whole frames of the same workload for >= 10 s, rank 0, N = 1 only;
               with the one-thread rate and the parallel efficiency beside it.
  kernel_ms_stats  min / median / max of the timed draws' own HIP-event durations.
  smaa         the SMAA post-process on the traced frame: time of one resolve and its HBM roofline (untimed addition).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

WIDTH, HEIGHT, DEPTH, SCENE = 3840, 2160, 4, "default"
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)


def emit(out):
    """The ONE JSON line, last on stdout: libraries that print through C stdio (RCCL's version banner at communicator creation) sit in a
    buffer of their own when stdout is a pipe and would otherwise come out after Python's line, at exit."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    print(json.dumps(out), file=_JSON_OUT or sys.stdout, flush=True)


_JSON_OUT = None


def keep_stdout_for_the_json_line():
    """From here on file descriptor 1 is stderr for everything but emit(): RCCL prints a five-line version banner through C stdio when a
    communicator is created (N > 1, and the loopback diagnostic), and whoever reads this process' stdout expects ONE line of JSON."""
    global _JSON_OUT
    if _JSON_OUT is None:
        sys.stdout.flush()
        _JSON_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


CLOCK_RAMP_MS = 150.0


def clock_ramp(draw, finish):
    """Untimed: keep the device busy with the bench's own draws for CLOCK_RAMP_MS before the warm-up, so that the shader clock has settled when
    the W warm-up steps start (round 3's timed region was 10 ms after 2.5 ms of warm-up: the PMC passes ran at 2.29 GHz, the longer kernels
    at 2.39-2.40). Returns the number of draws and the wall time it took."""
    t0 = time.perf_counter()
    n = 0
    while (time.perf_counter() - t0) * 1e3 < CLOCK_RAMP_MS:
        for _ in range(8):
            draw()
        finish()
        n += 8
    return n, (time.perf_counter() - t0) * 1e3


def bench_c_boundary(args, mode, n_ranks, rank, local_rank):
    """N > 1 through the C boundary: one rtx_draw per step on a context that splits the frame itself (rtx_create_multi in one process,
    rtx_create_rank in one process per GPU). Timed exactly like the single-GPU line: W warm-up draws, barrier + finish, K draws, finish +
    barrier, max over ranks."""
    from raytracing_opengl_amd import bands, scenes, textures, wrapper

    W, H = args.width, args.height
    per_process = mode == "ranks"
    multi_proc = per_process and n_ranks > 1

    from raytracing_opengl_amd import ranks   # rendezvous only: unique id, barrier, max / sum of a few numbers (gloo, CPU tensors)

    barrier = ranks.barrier

    def reduce_(values, op):
        return ranks.reduce_values(values, "max" if op == dist.ReduceOp.MAX else "sum")

    gather_kind = {"rccl": wrapper.RTX_GATHER_RCCL, "peer": wrapper.RTX_GATHER_PEER_COPY, "loopback": wrapper.RTX_GATHER_RCCL_LOOPBACK}[args.transport]
    sc = scenes.build_scene(args.scene, W, H, args.depth)
    ts = textures.default_texture_set(scale=args.texture_scale)
    if per_process:
        uid = ranks.exchange_unique_id(rank, wrapper.rccl_unique_id)
        gl = wrapper.make_renderer(sc, W, H, ts["textures"], ts["cubemap"], device=local_rank, texture_lod=args.lod,
                                   gather=gather_kind, rank=(rank, n_ranks, uid))
    else:
        try:
            gl = wrapper.make_renderer(sc, W, H, ts["textures"], ts["cubemap"], texture_lod=args.lod, devices=list(range(n_ranks)), gather=gather_kind)
        except wrapper.RtxError as e:
            if args.transport != "rccl" or "rccl" not in str(e).lower():
                raise
            # no usable RCCL on this box: the same bands over the same links with hipMemcpyPeerAsync issued by rank 0 -- said loudly, and in the line
            print(f"bench.py: RCCL transport unavailable ({e}); falling back to --transport peer", file=sys.stderr, flush=True)
            args.transport = "peer"
            gather_kind = wrapper.RTX_GATHER_PEER_COPY
            gl = wrapper.make_renderer(sc, W, H, ts["textures"], ts["cubemap"], texture_lod=args.lod, devices=list(range(n_ranks)), gather=gather_kind)
    target = args.target
    px_bytes = 16 if target == "rgba32f" else 4
    gl.set_option(wrapper.RTX_OPT_GATHER_TARGETS, 1 if target == "rgba32f" else 2)
    gl.set_option(wrapper.RTX_OPT_CULL, args.cull)
    gl.set_option(wrapper.RTX_OPT_SCENE_LDS, args.lds)
    gl.set_option(wrapper.RTX_OPT_XCD_REMAP, args.xcd)
    gl.set_option(wrapper.RTX_OPT_RAY_PENCILS, args.pencils)
    gl.set_option(wrapper.RTX_OPT_BAND_LAYOUT, {"interleaved": 0, "contiguous": 1, "balanced": 1}[args.bands])

    # exact reference-defined ray count of the frame (untimed, counting kernel variant; summed over the ranks)
    gl.set_option(wrapper.RTX_OPT_COUNT_RAYS, 1)
    gl.draw()
    gl.finish()
    st = gl.stats()
    rays_frame, rays_cast_frame = (int(v) for v in reduce_([st["rays_closest"] + st["rays_shadow"], st["rays_closest"] + st["rays_shadow_cast"]], dist.ReduceOp.SUM))
    gl.set_option(wrapper.RTX_OPT_COUNT_RAYS, 0)

    def rank_ms():
        """every rank's kernel time of its last launch, in rank order (all of them known to this process, or gathered over the rendezvous)"""
        ms = gl.rank_draw_ms()
        return ranks.gather_values(ms[rank]) if multi_proc else [float(v) for v in ms]

    def timed(steps):
        """`steps` draws bracketed like the contract says (barrier + finish on both sides); -> (seconds, slowest rank's mean kernel ms, gather ms)"""
        gl.finish()
        gl.stats()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            gl.draw()
        gl.finish()   # every device's launch and transfer streams, incl. the placement of the last frame on rank 0
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        n_ev = min(steps, 128)
        tr = gl.sum_recent_draw_ms(n_ev) / n_ev     # HIP events on the launch streams; the slowest rank of this process
        gms = gl.stats()["last_gather_ms"]
        dt, tr = reduce_([dt, tr], dist.ReduceOp.MAX)
        return dt, tr, gms

    def balance(rounds=6):
        """contiguous bands weighted by measured kernel time: a few frames with the split in use, every rank's time, the new split -- the same
        arithmetic on every rank (bands.weighted_split), so a per-process group agrees without any further exchange"""
        for _ in range(rounds):
            for _ in range(3):
                gl.draw()
            gl.finish()
            rows = gl.band_split()
            ms = rank_ms()
            if max(ms) <= 1.04 * min(ms):
                break
            gl.set_band_split(bands.weighted_split(H, rows, ms, damping=0.7))
        return gl.band_split(), rank_ms()

    ramp_draws, ramp_ms = clock_ramp(gl.draw, gl.finish)
    if args.bands == "balanced":
        balance()
    for _ in range(args.warmup):
        gl.draw()
    elapsed, trace_ms_max, gather_ms = timed(args.steps)
    ranks_trace_ms = rank_ms()
    layout_name = {"interleaved": "interleaved 8-row bands", "contiguous": "contiguous equal ranges", "balanced": "contiguous ranges weighted by kernel time"}[args.bands]
    gather_rgb = bool(gl.get_option(wrapper.RTX_OPT_GATHER_RGB))
    split_used = gl.band_split()
    st_end = gl.stats()

    # Untimed additions to the SAME line (one run on an 8-GPU node is all there may be): the other colour target and the other band layout,
    # each timed exactly like the value above, with K steps.
    extras = {}
    try:
        if n_ranks > 1 or args.transport == "loopback":
            other_target = "rgba8" if target == "rgba32f" else "rgba32f"
            gl.set_option(wrapper.RTX_OPT_GATHER_TARGETS, 2 if other_target == "rgba8" else 1)
            for _ in range(3):
                gl.draw()
            e2, t2, g2 = timed(args.steps)
            extras[other_target] = {"ms_per_step": round(e2 / args.steps * 1e3, 4), "trace_ms_max_rank": round(t2, 4), "gather_ms": round(g2, 4),
                                    "value_Mray_s": round(rays_frame * args.steps / e2 / 1e6, 2)}
            gl.set_option(wrapper.RTX_OPT_GATHER_TARGETS, 1 if target == "rgba32f" else 2)
            for other_layout in (("balanced", "interleaved") if args.also_bands else ()):
                if other_layout == args.bands:
                    continue
                gl.set_option(wrapper.RTX_OPT_BAND_LAYOUT, 1 if other_layout == "balanced" else 0)
                if other_layout == "balanced":
                    bal_split, bal_ms = balance()
                for _ in range(3):
                    gl.draw()
                e3, t3, g3 = timed(args.steps)
                extras["bands_" + other_layout] = {"ms_per_step": round(e3 / args.steps * 1e3, 4), "trace_ms_max_rank": round(t3, 4), "gather_ms": round(g3, 4),
                                                   "value_Mray_s": round(rays_frame * args.steps / e3 / 1e6, 2), "rows_per_rank": gl.band_split(),
                                                   "trace_ms_per_rank": [round(v, 4) for v in rank_ms()]}
                break
            # back to the configuration the value was measured with (the parity check below reads its frame)
            gl.set_option(wrapper.RTX_OPT_BAND_LAYOUT, {"interleaved": 0, "contiguous": 1, "balanced": 1}[args.bands])
            if args.bands != "interleaved":
                gl.set_band_split(split_used)
            gl.draw()
            gl.finish()
    except Exception as e:      # the additions must never cost the line itself (N > 1 on distinct devices has not run before the driver runs it)
        extras["error"] = f"{type(e).__name__}: {e}"
        try:
            gl.set_option(wrapper.RTX_OPT_GATHER_TARGETS, 1 if target == "rgba32f" else 2)
            gl.set_option(wrapper.RTX_OPT_BAND_LAYOUT, {"interleaved": 0, "contiguous": 1, "balanced": 1}[args.bands])
            gl.draw()
            gl.finish()
        except Exception:
            pass
    if rank != 0:
        gl.stop()
        return
    # the assembled frame against ONE device tracing all of it (untimed): bit-identical or the split is wrong
    fmt = wrapper.RTX_RGBA32F if target == "rgba32f" else wrapper.RTX_RGBA8
    got = gl.read_pixels(fmt)
    gl.stop()
    one = wrapper.make_renderer(sc, W, H, ts["textures"], ts["cubemap"], device=local_rank, texture_lod=args.lod)
    for o, v in ((wrapper.RTX_OPT_CULL, args.cull), (wrapper.RTX_OPT_SCENE_LDS, args.lds), (wrapper.RTX_OPT_XCD_REMAP, args.xcd), (wrapper.RTX_OPT_RAY_PENCILS, args.pencils)):
        one.set_option(o, v)
    one.draw()
    want = one.read_pixels(fmt)
    one.stop()
    same = bool(np.array_equal(got.view(np.uint32 if target == "rgba32f" else np.uint8), want.view(np.uint32 if target == "rgba32f" else np.uint8)))
    ms_per_step = elapsed / args.steps * 1e3
    rows0 = split_used[0]
    achieved = rows0 * W * px_bytes / (trace_ms_max * 1e-3) / 1e9
    # what travels per pixel: the float target of the interleaved layout goes without its alpha (RTX_OPT_GATHER_RGB, default on: 12 bytes)
    link_px_bytes = 12 if (target == "rgba32f" and args.bands == "interleaved" and gather_rgb) else px_bytes
    moved = sum(split_used[r] for r in range(0 if args.transport == "loopback" else 1, n_ranks)) * W * link_px_bytes
    links = max(1, n_ranks - 1)
    out = {
        "metric": f"Mray/s at {W}x{H} depth-{args.depth} {args.scene} scene (reference-defined rays: closest-hit + shadow scans)",
        "value": round(rays_frame * args.steps / elapsed / 1e6, 2),
        "unit": "Mray/s",
        "n_gpus": n_ranks,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{args.scene} scene (reference main.cpp:43-132, t=0), {W}x{H}, reflection depth {args.depth}, "
                               f"{target.upper()} target, seeded synthetic textures at reference sizes/{args.texture_scale}",
                   "rays_per_frame": rays_frame, "rays_executed_per_frame": rays_cast_frame,
                   "parallelism": f"{n_ranks} GPU{'s' if n_ranks > 1 else ''}, {layout_name}, one rtx_draw per frame on "
                                  + ("one rtx_create_rank context per process" if per_process else "one rtx_create_multi context")
                                  + f", gather of the {target.upper()} frame to rank 0",
                   "launcher": "torch.distributed.run, one process per GPU" if per_process else "one process, N devices",
                   "transport": {"rccl": "RCCL: grouped ncclSend/ncclRecv, every peer straight to rank 0 (librtx_hip.so)",
                                 "loopback": "RCCL incl. rank 0 -> rank 0 (diagnostic)",
                                 "peer": "hipMemcpyPeerAsync issued by rank 0 (librtx_hip.so)"}[args.transport],
                   "trace_ms_max_rank": round(trace_ms_max, 4), "trace_ms_per_rank": [round(v, 4) for v in ranks_trace_ms], "rows_per_rank": split_used,
                   "gather_ms": round(gather_ms, 4), "gather_bytes_per_frame": int(moved), "gather_bytes_per_pixel": link_px_bytes,
                   "gather_GB_s_into_rank0": round(moved / max(gather_ms, 1e-6) / 1e6, 1), "gather_GB_s_per_link": round(moved / links / max(gather_ms, 1e-6) / 1e6, 1),
                   "also_measured": extras,
                   "cull": args.cull, "scene_in_lds": args.lds, "texture_lod": args.lod, "xcd_remap": args.xcd,
                   "clock_ramp": {"draws": ramp_draws, "ms": round(ramp_ms, 1), "note": "untimed draws in front of the warm-up"}},
        "ms_per_frame": round(ms_per_step, 4),
        "kernel_ms": round(trace_ms_max, 4),
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None,
                     "note": f"rank 0's launch: its {rows0} rows of the {target.upper()} frame ({px_bytes} B/pixel) / the slowest rank's mean kernel time; "
                             "the path is bound by VALU issue (DESIGN.md); gather_ms = transfer + band placement of the last frame on rank 0's "
                             "transfer stream (overlaps the next frame's trace)"},
        "parity": {"vs_one_device_tracing_the_whole_frame": "bit-identical" if same else "DIFFERENT", "checked": target.upper()},
    }
    emit(out)
    if not same:
        raise SystemExit("bench.py: the gathered frame differs from the single-device frame")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)     # (round 6: 100 timed draws = 47 ms at 4K; kernel_ms_stats has their spread)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--width", type=int, default=WIDTH)
    ap.add_argument("--height", type=int, default=HEIGHT)
    ap.add_argument("--depth", type=int, default=DEPTH)
    ap.add_argument("--scene", default=SCENE)
    ap.add_argument("--texture-scale", type=int, default=1, help="divide the reference texture sizes (1 = reference sizes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lds", type=int, default=0)
    ap.add_argument("--cull", type=int, default=1)
    ap.add_argument("--pencils", type=int, default=1, help="1: ray-pencil candidate masks for scenes with long quadric / torus tables; 0: two-level scans only")
    ap.add_argument("--xcd", type=int, default=0, help="1: XCD-aware super-tile workgroup order; 0: row-major")
    ap.add_argument("--lod", type=int, default=1, help="1: mip chain + quad-derivative LOD (reference texture state); 0: level-0 bilinear")
    ap.add_argument("--target", choices=("rgba32f", "rgba8"), default="rgba32f",
                    help="colour target the bands are traced into and gathered as -- the SAME at every N, so that a 1/2/4/8 series is one "
                         "workload: rgba32f (default) = the 16 B/pixel parity buffer of the BASELINE metric; rgba8 = what the reference's "
                         "framebuffer holds (GLWrapper.cpp:127,209-222), a quarter of the gather traffic")
    ap.add_argument("--bands", choices=("interleaved", "contiguous", "balanced"), default="interleaved",
                    help="N > 1: how the frame is split (rtx.h RTX_OPT_BAND_LAYOUT): interleaved 8-row bands (default; landing buffers + a placement "
                         "pass on rank 0), contiguous = one equal range of rows per rank traced / received straight into place, balanced = contiguous "
                         "ranges weighted by the ranks' measured kernel times. The line's value is this layout; the other one and the other colour "
                         "target are measured too and reported under config.also_measured")
    ap.add_argument("--also-bands", action="store_true",
                    help="N > 1: also time the other band layout (config.also_measured.bands_*). Off by default: measured alone on one GPU the contiguous "
                         "ranges lose to interleaved bands at every N (profiles/r04_time_bands_*.txt), and the line's run on real devices should not "
                         "depend on a second layout's code")
    ap.add_argument("--no-smaa", action="store_true", help="skip the untimed SMAA post-process measurement (N = 1)")
    ap.add_argument("--launcher", choices=("auto", "torch"), default="auto",
                    help="N > 1 without torch.distributed.run around it: auto = one process drives the N devices through rtx_create_multi; "
                         "torch = re-exec this script under torch.distributed.run (one process per GPU, rtx_create_rank)")
    ap.add_argument("--transport", choices=("rccl", "peer", "torch", "loopback"), default="rccl",
                    help="how the bands reach rank 0: rccl = grouped ncclSend/ncclRecv inside the C library (default); peer = hipMemcpyPeerAsync by "
                         "the root (single-process form only); torch = the Python path (bands.py, dist.gather; torch.distributed.run form only); "
                         "loopback = rccl with rank 0's own bands sent to itself as well -- a diagnostic that runs the N > 1 code, RCCL included, "
                         "on a box with one GPU (--gpus 1 --transport loopback)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    under_torchrun = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if under_torchrun and world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but torch.distributed.run started {world} rank(s)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP tracer has no CPU fallback)")
    n_dev = torch.cuda.device_count()
    if args.gpus < 1 or (not under_torchrun and args.gpus > n_dev) or (under_torchrun and local_rank >= n_dev):
        print(f"bench.py: --gpus {args.gpus} requested but this box has {n_dev} GPU{'s' if n_dev != 1 else ''}", file=sys.stderr, flush=True)
        raise SystemExit(2)
    if not under_torchrun and args.gpus > 1 and (args.launcher == "torch" or args.transport == "torch"):
        # the one-process-per-GPU form, started from here (what the driver does itself for N > 1)
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    keep_stdout_for_the_json_line()
    # how the N ranks are driven: "single" = plain context (N = 1); "multi" = one process, rtx_create_multi; "ranks" = one process per GPU,
    # rtx_create_rank; "torch" = one process per GPU, rtx_draw_bands + dist.gather (the Python path)
    if args.gpus == 1 and args.transport != "loopback":
        mode = "single"
    elif not under_torchrun:
        mode = "multi"
    else:
        mode = "torch" if args.transport == "torch" else "ranks"
    if mode == "ranks" and args.transport == "peer":
        raise SystemExit("bench.py: --transport peer needs the single-process form (ranks in separate processes exchange their bands over RCCL)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if under_torchrun and world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if mode == "torch":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            # torch.distributed is only the rendezvous here (unique id, barrier, max / sum of a few numbers, all on CPU tensors over gloo):
            # the frame's bytes move inside the C library, on its own RCCL communicator
            dist.init_process_group("gloo", rank=rank, world_size=world)
    if mode in ("multi", "ranks"):
        try:
            bench_c_boundary(args, mode, world if mode == "ranks" else args.gpus, rank, local_rank)
        finally:
            if dist.is_initialized():
                dist.destroy_process_group()
        return

    from raytracing_opengl_amd import bands, scenes, textures, wrapper

    W, H = args.width, args.height
    sc = scenes.build_scene(args.scene, W, H, args.depth)
    ts = textures.default_texture_set(scale=args.texture_scale)
    gl = wrapper.make_renderer(sc, W, H, ts["textures"], ts["cubemap"], device=local_rank, texture_lod=args.lod)
    gl.set_option(wrapper.RTX_OPT_CULL, args.cull)
    gl.set_option(wrapper.RTX_OPT_SCENE_LDS, args.lds)
    gl.set_option(wrapper.RTX_OPT_XCD_REMAP, args.xcd)
    gl.set_option(wrapper.RTX_OPT_RAY_PENCILS, args.pencils)

    band_rows = ((H + 7) // 8) * 8 if world == 1 else bands.choose_band_rows(H, world)
    target = args.target
    tgt_dtype, tgt_format, px_bytes = (torch.float32, wrapper.RTX_RGBA32F, 16) if target == "rgba32f" else (torch.uint8, wrapper.RTX_RGBA8, 4)
    gather = bands.FrameGather(H, W, 4, band_rows, tgt_dtype, device, dst=0)
    bufs = [gather.new_local(tgt_dtype, device) for _ in range(2)]
    stream = torch.cuda.current_stream(device).cuda_stream

    # exact reference-defined ray count of this rank's bands (untimed, counting kernel variant)
    gl.set_option(wrapper.RTX_OPT_COUNT_RAYS, 1)
    gl.draw_bands(band_rows, rank, world, bufs[0].data_ptr(), tgt_format, stream)
    torch.cuda.synchronize(device)
    st = gl.stats()
    rays_local = st["rays_closest"] + st["rays_shadow"]
    rays_cast_local = st["rays_closest"] + st["rays_shadow_cast"]
    gl.set_option(wrapper.RTX_OPT_COUNT_RAYS, 0)
    rays_t = torch.tensor([rays_local, rays_cast_local], dtype=torch.int64, device=device)
    if world > 1:
        dist.all_reduce(rays_t)
    rays_frame, rays_cast_frame = int(rays_t[0]), int(rays_t[1])

    def step(k, pending):
        buf = bufs[k & 1]
        if pending[k & 1] is not None:          # the gather that last read this buffer must be done
            gather.frame(pending[k & 1])
            pending[k & 1] = None
        gl.draw_bands(band_rows, rank, world, buf.data_ptr(), tgt_format, stream)
        pending[k & 1] = gather.gather(buf, k & 1)  # async; overlaps the next step's trace (recv slot k&1 on the root)

    def drain(pending):
        out = None
        for j in (0, 1):
            if pending[j] is not None:
                out = gather.frame(pending[j])
                pending[j] = None
        return out

    pending = [None, None]
    ramp_k = [0]

    def ramp_draw():
        step(ramp_k[0], pending)
        ramp_k[0] += 1
    ramp_draws, ramp_ms = clock_ramp(ramp_draw, lambda: (drain(pending), torch.cuda.synchronize(device)))
    for k in range(args.warmup):
        step(k, pending)
    drain(pending)
    torch.cuda.synchronize(device)
    gl.stats()  # retire warm-up events
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k, pending)
    frame = drain(pending)
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0

    n_ev = min(args.steps, 128)
    each_ms = sorted(gl.recent_draw_ms(n_ev))       # the spread of the timed draws: a 9 ms window is not the whole measurement
    kernel_ms = gl.sum_recent_draw_ms(n_ev) / n_ev  # HIP events on the launch stream
    st_end = gl.stats()
    t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, kernel_ms_max = float(t[0]), float(t[1])

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        mrays = rays_frame * args.steps / elapsed / 1e6
        px_bytes_launch = gather.rows_local * W * px_bytes  # algorithmic HBM bytes of one launch on this rank
        achieved = px_bytes_launch / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": f"Mray/s at {W}x{H} depth-{args.depth} {args.scene} scene (reference-defined rays: closest-hit + shadow scans)",
            "value": round(mrays, 2),
            "unit": "Mray/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.scene} scene (reference main.cpp:43-132, t=0), {W}x{H}, reflection depth {args.depth}, "
                                   f"{target.upper()} target, seeded synthetic textures at reference sizes/{args.texture_scale}",
                       "rays_per_frame": rays_frame, "rays_executed_per_frame": rays_cast_frame,
                       "parallelism": "single GPU" if world == 1 else f"{world} GPUs, interleaved {band_rows}-row bands, RCCL gather of the {target.upper()} frame to rank 0",
                       "cull": args.cull, "scene_in_lds": args.lds, "texture_lod": args.lod, "xcd_remap": args.xcd,
                       # ray pencils: candidate masks built on the device when the scene changes (not per frame: the bench scene is static,
                       # like its packed tables); build_ms is the cost a scene update adds
                       "ray_pencils": {"count": st_end["pencils"], "build_ms": round(st_end["last_pencil_build_ms"], 4)},
                       "clock_ramp": {"draws": ramp_draws, "ms": round(ramp_ms, 1), "note": "untimed draws in front of the warm-up"}},
            "ms_per_frame": round(ms_per_step, 4),
            "kernel_ms": round(kernel_ms, 4),
            "kernel_ms_stats": {"n": n_ev, "min": round(each_ms[0], 4), "median": round(each_ms[len(each_ms) // 2], 4), "max": round(each_ms[-1], 4),
                                "note": "HIP-event durations of the timed draws one by one (rtx_recent_draw_ms); kernel_ms is their mean"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None,
                         "note": f"HBM-write roofline of the {target.upper()} frame ({px_bytes} B/pixel); the path is bound by VALU issue, see roofline.valu and DESIGN.md"},
        }
        # The kernel's real ceiling is VALU issue, which the bound/peak vocabulary above cannot name: report it beside. The instruction and
        # traffic counts come from rocprofv3 PMC passes (tools/profile_gpu.sh -> tools/prof_to_json.py); they are quoted only while the
        # kernel sources they were measured on are the ones running (kernel_hash), and the peak is the measured issue rate of
        # tools/micro/valu_rate.hip (profiles/valu_peak.json), not a nominal figure.
        from raytracing_opengl_amd import build_info
        khash = build_info.kernel_source_hash()
        suffix = "" if args.scene == "default" else "_" + args.scene

        def prof(name):
            f = os.path.join(ROOT, "profiles", f"{name}{suffix}.json")
            try:
                v = json.load(open(f))
            except Exception:
                return None
            if v.get("kernel_hash") != khash or (v.get("width"), v.get("height"), v.get("depth")) != (W, H, args.depth):
                return None          # measured on other sources or another workload: stale, not reported
            return v
        if world == 1 and (args.cull, args.lod, args.lds, args.xcd) == (1, 1, 0, 0):
            v = prof("valu")
            try:
                peak = json.load(open(os.path.join(ROOT, "profiles", "valu_peak.json")))
            except Exception:
                peak = None
            if v and peak:
                rate = v["valu_insts_per_launch"] / (kernel_ms * 1e-3)
                cyc = peak["cycles_per_wave_inst_per_simd"]
                clock = (v.get("shader_clock_GHz") or 2.3) * 1e9
                # frac_nominal: against the guide's figure -- one wave64 VALU instruction per 2 cycles per SIMD, 1024 SIMDs
                # (/opt/skills/guides/MI355X_MICROARCH.md) -- beside the mix-weighted `frac`, whose ceiling is this repository's own micro-benchmark
                # (profiles/valu_peak.json: v_fma_f32 3.95, v_add / v_mul 2.34 cycles). Both are quoted so that neither has to be taken on trust.
                nominal_peak = 1024.0 * clock / 2.0
                val = {"insts_per_launch": v["valu_insts_per_launch"], "achieved": round(rate / 1e9, 1), "unit": "G wave-instructions/s",
                       "cycles_per_valu_inst": v.get("cycles_per_valu_inst"), "lane_utilisation": v.get("lane_utilisation"), "kernel_hash": khash,
                       "peak_nominal": round(nominal_peak / 1e9, 1), "frac_nominal": round(rate / nominal_peak, 4),
                       "frac_nominal_x_lane_utilisation": (round(rate / nominal_peak * v["lane_utilisation"], 4) if v.get("lane_utilisation") else None),
                       "nominal_note": "1024 SIMDs x shader clock / 2 cycles per wave64 VALU instruction (MI355X_MICROARCH.md); x lane utilisation = the share of FP32 lanes doing work"}
                cls = v.get("classes")
                if cls:
                    # Mix-weighted issue ceiling: every instruction class at its MEASURED issue cost (cycles per wave-instruction per SIMD,
                    # tools/micro/valu_rate.hip); the classes the counters do not name ("other": v_mov / v_cmp / v_cndmask / v_min / v_max /
                    # lane moves) at the cheapest cost measured (v_mov: 2.38 -- compares and selects cost 4.1-4.4, so this is a lower bound
                    # on the cycles the mix needs). frac = those cycles / the cycles the kernel took = achieved / peak <= 1 by construction.
                    cost = {"add_f32": cyc["v_add_f32"], "mul_f32": cyc["v_mul_f32"], "fma_f32": cyc["v_fma_f32"], "trans_f32": cyc["v_rcp_f32"],
                            "int32": cyc["v_add_u32"], "int64": cyc["v_add_u32"], "cvt": cyc["v_cvt_f32_ubyte0"], "other": cyc["v_mov_b32"]}
                    need_cycles = sum(cls[k] * cost[k] for k in cost if k in cls) / 1024.0      # per SIMD
                    mix_peak = v["valu_insts_per_launch"] / (need_cycles / clock)
                    val.update({"peak": round(mix_peak / 1e9, 1), "frac": round(rate / mix_peak, 4),
                                "classes_per_launch": cls, "issue_cycles_per_class": {k: cost[k] for k in cost},
                                "mean_issue_cycles_of_the_mix": round(need_cycles * 1024.0 / v["valu_insts_per_launch"], 3),
                                "shader_clock_GHz": v.get("shader_clock_GHz")})
                    val["note"] = ("mix-weighted VALU issue ceiling: per-class instruction counts (rocprofv3 PMC, " + v.get("source", "") + ") x measured issue "
                                   "cycles per class (" + peak.get("source", "") + "), unnamed classes at the cheapest measured cost, so frac <= 1 by "
                                   "construction; duration live. 'other' = SQ_INSTS_VALU minus the classes the SQ_INSTS_VALU_* counters name (add / mul / fma / "
                                   "trans f32, int32, int64, cvt): v_mov, v_cmp, v_cndmask, v_min / v_max, lane moves (readlane / writelane / DPP)")
                else:
                    val.update({"peak": peak["simple_op_peak_G_per_s"], "frac": round(rate / 1e9 / peak["simple_op_peak_G_per_s"], 4),
                                "note": "no per-class counts in the profile: peak = the issue rate of an all-v_add stream (a loose ceiling)"})
                out["roofline"]["valu"] = val
            t = prof("traffic")
            if t:
                out["roofline"]["traffic"] = t["hbm_bytes_per_launch"]
                out["roofline"]["traffic_over_algorithmic"] = round(t["hbm_bytes_per_launch"] / float(W * H * px_bytes), 3)
        if world == 1 and args.scene == "default":
            # The reference animates every frame (main.cpp:197-246: update_scene + an update_buffer of every block); the line above times the
            # t = 0 frame. Untimed addition: the kernel time of the same workload at other animation times, blocks re-uploaded like the
            # reference does (SceneUploader.update), 10 launches each.
            anim = {}
            for tt in (0.0, 1.0, 3.0, 7.5, 12.5):
                gl.uploader.update(scenes.build_scene(args.scene, W, H, args.depth, time=tt, delta=tt))
                for _ in range(3):
                    gl.draw()
                gl.finish()
                gl.stats()
                for _ in range(10):
                    gl.draw()
                gl.finish()
                anim[f"t={tt:g}"] = round(gl.sum_recent_draw_ms(10) / 10, 4)
            gl.uploader.update(sc)
            out["animated"] = {"kernel_ms_per_frame": anim, "mean_ms": round(sum(anim.values()) / len(anim), 4),
                               "note": "default scene at animation times t (update_scene of main.cpp:197-246 with time = deltaTime = t, every block "
                                       "re-uploaded); value / ms_per_step above are the t = 0 frame"}
        if world == 1 and not args.no_smaa:
            # the post-process that follows the tracer in the reference's draw(). Untimed addition to the line: ULTRA (main.cpp:32)
            # on the frame just traced, HIP events around the four kernels of one resolve. Algorithmic bytes: W*H*4 read + W*H*4 written.
            gl.enable_SMAA(wrapper.ULTRA)   # the library's own area / search tables (== the reference's arrays, include/rtx/smaa_tables.h)
            gl.draw()
            ts_ms = []
            for _ in range(12):
                gl.smaa_resolve()
                ts_ms.append(gl.stats()["last_smaa_ms"])
            sm = gl.stats()
            smaa_ms = float(np.median(ts_ms[2:]))
            in_draw = []
            for _ in range(10):             # the same passes right behind the tracer, as draw() runs them (GLWrapper.cpp:155-204): the colour target
                gl.draw()                   # then comes from HBM, not from the 256 MB last-level cache a repeated resolve finds it in
                in_draw.append(gl.stats()["last_smaa_ms"])
            gl.enable_SMAA(wrapper.RTX_SMAA_OFF)
            out["smaa"] = {"preset": "ULTRA", "ms_per_resolve": round(smaa_ms, 4), "ms_per_resolve_inside_draw": round(float(np.median(in_draw[2:])), 4),
                           "edge_pixels": int(sm["smaa_edge_pixels"]),
                           "roofline": {"bound": "hbm", "achieved": round(W * H * 8 / smaa_ms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": round(W * H * 8 / smaa_ms / 1e6 / HBM_PEAK_GBS, 4)},
                           "note": "all kernels of one resolve of the traced frame (RGBA8 in, RGBA8 screen out); area and search tables generated "
                                   "by the library, byte-identical to the reference's AreaTex.h / SearchTex.h (tests/test_smaa_tables.py); "
                                   "byte-exact against the oracle in tests/test_gpu_smaa.py"}
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle
            o = oracle.OracleScene(sc, W, H, ts["textures"], ts["cubemap"], texture_lod=args.lod)
            cores, cpu_quota = oracle.effective_cpus()
            cpu_s, reps = 0.0, 0
            while cpu_s < 10.0 and reps < 20:   # bounded sample: whole frames until >= 10 s of wall time
                c0 = time.perf_counter()
                ref, cnt = o.render(0, H, threads=cores)
                cpu_s += time.perf_counter() - c0
                reps += 1
            cpu_s /= reps
            # one thread, on a thin slice of rows spread over the frame (SURVEY section 8(d) asks for both figures)
            # (pairs of rows that make whole 2x2 quads: a single row would run -- and discard -- its quads' other halves)
            rows = sorted({min(H - 2, ((H * (2 * j + 1)) // 32) & ~1) for j in range(16)})
            c0 = time.perf_counter()
            one_rays = 0
            for y in rows:
                _r, c1 = o.render(y, y + 2, threads=1)
                one_rays += c1["rays_closest"] + c1["rays_shadow"]
            one_s = time.perf_counter() - c0
            # The oracle applies the quad-derivative texture rule by running every 2x2 quad as four ucontext coroutines -- a checker's
            # structure, not a renderer's. The same frame with level-0 textures needs no quads and runs as plain loops: reported beside, as
            # the closer stand-in for "the shader math on the host cores" (a different texture state: not the value).
            o0 = oracle.OracleScene(sc, W, H, ts["textures"], ts["cubemap"], texture_lod=0)
            c0 = time.perf_counter()
            _r0, cnt0 = o0.render(0, H, threads=cores)
            plain_s = time.perf_counter() - c0
            try:
                cpu_model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
            except Exception:
                cpu_model = "unknown"
            out["cpu_baseline"] = {"value": round((cnt["rays_closest"] + cnt["rays_shadow"]) / cpu_s / 1e6, 3), "unit": "Mray/s",
                                   "cores": cores, "kind": "port", "cpu": cpu_model,
                                   "cores_note": (f"threads used = the CPUs this container may use: {os.cpu_count()} hardware threads visible, cgroup CPU quota "
                                                  f"{cpu_quota:g}" if cpu_quota else f"threads used = the affinity mask ({os.cpu_count()} hardware threads visible, no cgroup quota)"),
                                   "one_thread": {"value": round(one_rays / one_s / 1e6, 4), "unit": "Mray/s",
                                                  "sample": f"{len(rows)} pairs of rows spread over the frame, {one_s:.2f} s"},
                                   "parallel_efficiency": round((cnt["rays_closest"] + cnt["rays_shadow"]) / cpu_s / (one_rays / one_s) / cores, 3),
                                   "without_quad_coroutines": {"value": round((cnt0["rays_closest"] + cnt0["rays_shadow"]) / plain_s / 1e6, 3), "unit": "Mray/s",
                                                               "sample": f"one full frame with level-0 textures (plain loops, no 2x2-quad coroutines), {plain_s:.2f} s"},
                                   "note": "the oracle is the CHECKER (scalar, contraction-free, quads as coroutines): a baseline for orientation, not a tuned CPU renderer",
                                   "sample": f"{reps} full {W}x{H} depth-{args.depth} frames of the same workload (mean {cpu_s:.2f} s each), "
                                             f"oracle/rt_oracle.c, OpenMP over rows"}
            # the oracle frame is there anyway: report full-size parity next to the timing
            img = frame.cpu().numpy() if (frame is not None and target == "rgba32f") else None
            if img is not None:
                d = np.abs(img - ref)
                out["parity"] = {"max_abs_diff": float(np.nanmax(d)), "over_1e-4": int((d > 1e-4).sum()),
                                 "nan_mismatch": int((np.isnan(img) != np.isnan(ref)).sum()),
                                 "rays_match": bool(cnt["rays_closest"] + cnt["rays_shadow"] == rays_frame)}
        emit(out)
    gl.stop()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
