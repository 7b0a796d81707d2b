#!/usr/bin/env python3
"""Builds examples/pybind_drrt/_build/drrt_native.so: g++ on drrt_module.cpp against the torch headers,
linked to the in-tree libdrrt_hip.so (rpath).  No device code in this translation unit."""
import os
import subprocess
import sysconfig


def build(verbose=False):
    import torch
    from torch.utils import cpp_extension as ce
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(os.path.dirname(here))
    pkg = os.path.join(root, "adjointnonlinearraytracing_amd")
    out = os.path.join(here, "_build", "drrt_native.so")
    src = os.path.join(here, "drrt_module.cpp")
    hdr = os.path.join(root, "include", "drrt_hip.h")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(src), os.path.getmtime(hdr)):
        return out
    if not os.path.exists(os.path.join(pkg, "libdrrt_hip.so")):
        raise RuntimeError("build libdrrt_hip.so first (make -C adjointnonlinearraytracing_amd/csrc)")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           "-DTORCH_EXTENSION_NAME=drrt_native", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
           src, "-o", out, "-I" + sysconfig.get_paths()["include"], "-I/opt/rocm/include"]
    cmd += ["-I" + i for i in ce.include_paths()] + ["-L" + l for l in ce.library_paths()]
    cmd += ["-L" + pkg, "-Wl,-rpath,$ORIGIN/../../../adjointnonlinearraytracing_amd", "-l:libdrrt_hip.so", "-lc10", "-ltorch", "-ltorch_cpu", "-ltorch_python",
            "-lc10_hip"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("pybind example failed to build:\n" + r.stderr[-4000:])
    if verbose:
        print("built", out)
    return out


if __name__ == "__main__":
    build(verbose=True)
