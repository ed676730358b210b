"""MI355X-native drop-in for the fragment-shader ray tracer of engilas/raytracing-opengl.

Layout
  csrc/            HIP kernels (rt_kernel.hip, rt_device.h), scene packer, C ABI (rtx_capi.cpp)
  csrc/host/       C++ scene recipes built on include/rtx/*.h (SceneManager / SurfaceFactory shims)
  wrapper.py       Python mirror of GLWrapper / SceneManager upload path over the C ABI
  scenes.py        scene blocks (default / quadric / torus) from librtx_host.so
  textures.py      seeded synthetic textures with the reference assets' formats and sizes
  bands.py         row-band partition over GPUs + RCCL gather (torch.distributed)

The tracer itself lives in librtx_hip.so; importing `wrapper` fails loudly if it is not built.
"""
__all__ = ["scenes", "textures", "wrapper", "bands"]
