"""MI355X-native eikonal ray march + adjoint (drop-in for the `drrt` / `core.tracer` path of
ArjunTeh/AdjointNonlinearRayTracing).

Sub-modules
  _lib    ctypes binding of libdrrt_hip.so (C ABI: include/drrt_hip.h); fails loudly when missing
  drrt    `TracerC` / `TracerS` objects mirroring the reference pybind module (src/drrt.cpp:21-59)
  tracer  torch.autograd.Function classes mirroring core/tracer.py:294-526
  dist    ray-sharded multi-GPU execution with one all-reduce of dL/dn (torch.distributed / RCCL)
"""
__version__ = "0.1.0"
