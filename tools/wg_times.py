#!/usr/bin/env python3
"""GPU box, diagnostic build (-DRT_WG_TIMES, RTX_HIP_LIB=.../librtx_hip_wgtimes.so): duration of every workgroup of one
launch -- how long is the long pole, and where is it? Runs one GPU's share of the 4K default frame for N = 1 and N = 8."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from raytracing_opengl_amd import _capi, bands, scenes, textures, wrapper  # noqa: E402


def main():
    lib = _capi.load()
    fn = lib.rtx_debug_wg_times
    w, h = 3840, 2160
    sc = scenes.build_scene("default", w, h, 4)
    ts = textures.default_texture_set()
    gl = wrapper.make_renderer(sc, w, h, ts["textures"], ts["cubemap"])
    stream = torch.cuda.current_stream().cuda_stream
    for world in ([int(v) for v in os.environ.get('WG_WORLDS', '1,8').split(',')]):
        rows_max = bands.max_local_rows(h, 8, world)
        buf = torch.empty((rows_max, w, 4), dtype=torch.uint8, device="cuda:0")
        gx, gy = (w + 31) // 32, (bands.local_rows(h, 8, 0, world) + 7) // 8
        n = gx * gy
        for _ in range(3):
            gl.draw_bands(8, 0, world, buf.data_ptr(), wrapper.RTX_RGBA8, stream)
        gl.finish()
        ms = gl.sum_recent_draw_ms(1)
        start = (ctypes.c_ulonglong * 65536)(); dur = (ctypes.c_ulonglong * 65536)()
        fn(start, dur, 65536)
        m = min(n, 65536)
        s = np.array(start[:m], dtype=np.float64); d = np.array(dur[:m], dtype=np.float64)
        # s_memrealtime: 100 MHz, one clock for the device -- durations AND start times (round 5: the launch's occupancy over time)
        us = d / 100.0
        t0 = (s - s.min()) / 100.0
        t1 = t0 + us
        span = t1.max()
        edges = np.linspace(0.0, span, 41)
        inflight = [(int(((t0 <= a) & (t1 > a)).sum())) for a in edges[:-1]]
        print(f"    span first start .. last end {span:.1f} us; workgroups in flight at 40 instants (1536 slots at 6 per CU): {inflight}")
        print(f"    last start at {t0.max():.1f} us; work-time after the last start: {np.clip(t1 - t0.max(), 0, None).sum() / 1536:.1f} us per slot")
        order = np.argsort(-us)
        print(f"N={world}: {n} workgroups, launch {ms*1000:.1f} us (instrumented); workgroup duration us: median {np.median(us):.1f} p90 {np.percentile(us,90):.1f} p99 {np.percentile(us,99):.1f} max {us.max():.1f}; sum/1024 slots {us.sum()/1024:.1f}")
        print("    longest:", [(int(i % gx), int(i // gx), round(float(us[i]), 1)) for i in order[:8]])
        print(f"    workgroups longer than half the launch: {(us > ms*500).sum()}, longer than a quarter: {(us > ms*250).sum()}")
    gl.stop()


if __name__ == "__main__":
    main()
