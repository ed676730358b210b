#!/usr/bin/env python3
"""First contact between ranks in separate processes, on real devices (needs >= 2 GPUs):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 tools/first_contact_check.py

 1. same configuration on every rank          -> the frame assembled on rank 0 is the one-device frame, bit for bit
 2. RTX_OPT_GATHER_RGB differs on the last rank -> every rank's draw fails with RTX_ERR_INVALID naming that rank (no hang, no garbage)
 3. the last rank never draws                 -> the other ranks' draw fails with RTX_ERR_DEVICE within RTX_GATHER_TIMEOUT_MS (set to 3 s here) instead of hanging
Prints one line per check on rank 0 and exits non-zero on the first failure. tests/test_gpu_multi.py runs it where two devices exist."""
import os
import sys
import time

os.environ.setdefault("RTX_GATHER_TIMEOUT_MS", "3000")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch.distributed as dist  # noqa: E402

from raytracing_opengl_amd import ranks, scenes, textures, wrapper  # noqa: E402


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    dist.init_process_group("gloo")
    w, h = 640, 360
    sc = scenes.build_scene("default", w, h, 4)
    ts = textures.default_texture_set(scale=16)

    def make():
        uid = ranks.exchange_unique_id(rank, wrapper.rccl_unique_id)
        return wrapper.make_renderer(sc, w, h, ts["textures"], ts["cubemap"], device=local, gather=wrapper.RTX_GATHER_RCCL, rank=(rank, world, uid))

    # 1. agreement
    gl = make()
    for _ in range(3):
        gl.draw()
    gl.finish()
    if rank == 0:
        one = wrapper.make_renderer(sc, w, h, ts["textures"], ts["cubemap"], device=local)
        one.draw()
        same = np.array_equal(one.read_pixels(wrapper.RTX_RGBA32F).view(np.uint32), gl.read_pixels(wrapper.RTX_RGBA32F).view(np.uint32))
        one.stop()
        print(f"1. {world} ranks agree: frame bit-identical to one device: {same}", flush=True)
        assert same
    ranks.barrier()
    # 2. one rank sets an option the others do not
    if rank == world - 1:
        gl.set_option(wrapper.RTX_OPT_GATHER_RGB, 0)
    try:
        gl.draw()
        gl.finish()
        err = None
    except wrapper.RtxError as e:
        err = str(e)
    ok = err is not None and f"rank {world - 1}" in err and "differs" in err
    print(f"2. rank {rank}: mismatch reported: {ok} ({(err or 'no error')[:120]})", flush=True)
    assert ok
    gl.stop()
    ranks.barrier()
    # 3. a rank that never draws
    gl = make()
    gl.draw(); gl.finish()      # first contact under the common configuration
    gl.set_option(wrapper.RTX_OPT_GATHER_TARGETS, 1)     # a change every rank makes -> a new handshake at the next draw ...
    t0 = time.time()
    if rank != world - 1:                                # ... which the last rank never starts
        try:
            gl.draw()
            gl.finish()
            err = None
        except wrapper.RtxError as e:
            err = str(e)
        if rank == 0:
            ok = err is not None and "did not finish within" in err and time.time() - t0 < 30
            print(f"3. rank 0: a silent rank is a timeout after {time.time() - t0:.1f} s, not a hang: {ok} ({(err or 'no error')[:140]})", flush=True)
            assert ok
    ranks.barrier()
    os._exit(0)     # (contexts with a transfer in flight that will never complete: leave without their destructors)


if __name__ == "__main__":
    main()
