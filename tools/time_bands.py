#!/usr/bin/env python3
"""GPU box: kernel time of ONE rank's share of the default frame for N = 1, 2, 4, 8 ranks, every share measured ALONE on one GPU -- what each
GPU of a multi-GPU run traces, without the other ranks' launches beside it. Both band layouts of rtx.h RTX_OPT_BAND_LAYOUT:
  interleaved  8-row bands, band b -> rank b mod N (rtx_draw_bands)
  balanced     one contiguous range per rank (rtx_draw_rows), the ranges weighted by the measured times (bands.weighted_split, the arithmetic
               bench.py --bands balanced and the library's layout 2 use), iterated until the ranks agree within 4 %
and the prediction they give for the driver's 1 / 2 / 4 / 8 series: frame time >= max over ranks of the trace, + the gather of (N-1)/N of the
frame into rank 0 where it does not overlap.
usage: tools/time_bands.py [W H [depth]]     (default 3840 2160 4; RGBA32F target like the metric)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from raytracing_opengl_amd import bands, scenes, textures, wrapper  # noqa: E402


def main():
    w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
    depth = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    sc = scenes.build_scene("default", w, h, depth)
    if os.environ.get("NO_TORUS"):   # ablation: is the per-rank floor the torus solver's long waves?
        d = list(sc.defines); d[4] = 0
        sc.defines = tuple(d); sc.blocks["toruses_buf"] = b""
    ts = textures.default_texture_set()
    gl = wrapper.make_renderer(sc, w, h, ts["textures"], ts["cubemap"])
    stream = torch.cuda.current_stream().cuda_stream
    buf = torch.empty((h + 8, w, 4), dtype=torch.float32, device="cuda:0")

    def best_of(launch, reps=3, n=12):
        for _ in range(4):
            launch()
        gl.finish(); gl.stats()
        best = 1e9
        for _ in range(reps):
            for _ in range(n):
                launch()
            gl.finish()
            best = min(best, gl.sum_recent_draw_ms(n) / n)
        return best * 1000.0

    print(f"default scene {w}x{h} depth {depth}, RGBA32F, every share alone on one GPU (kernel us)")
    base = None
    for world in (1, 2, 4, 8):
        inter = [best_of(lambda r=rank: gl.draw_bands(8, r, world, buf.data_ptr(), wrapper.RTX_RGBA32F, stream)) for rank in range(world)]
        if world == 1:
            base = inter[0]
        rows = bands.weighted_split(h, [h // world] * world, [1.0] * world)      # equal ranges
        starts = lambda rr: [sum(rr[:k]) for k in range(len(rr))]
        equal = [best_of(lambda y=y0, n=n: gl.draw_rows(y, n, buf.data_ptr(), wrapper.RTX_RGBA32F, stream)) for y0, n in zip(starts(rows), rows)]
        bal, brows = equal, rows
        for _ in range(8):
            if world == 1 or max(bal) <= 1.04 * min(bal):
                break
            brows = bands.weighted_split(h, brows, bal, damping=0.7)
            bal = [best_of(lambda y=y0, n=n: gl.draw_rows(y, n, buf.data_ptr(), wrapper.RTX_RGBA32F, stream), reps=2, n=8) for y0, n in zip(starts(brows), brows)]
        print(f"N={world}: 1/N of the one-GPU frame {base / world:7.1f} | interleaved: slowest rank {max(inter):7.1f} (fastest {min(inter):7.1f}) | "
              f"contiguous equal: slowest {max(equal):7.1f} (fastest {min(equal):7.1f}) | balanced: slowest {max(bal):7.1f} (fastest {min(bal):7.1f}), rows {brows}", flush=True)
    gl.stop()


if __name__ == "__main__":
    main()
