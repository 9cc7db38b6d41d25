"""Build oracle/_ref/libfd_ref_iou.so from the reference's OWN source where it lies
(/root/reference/det3d/ops/iou3d_nms/src/iou3d_cpu.cpp) -- nothing is copied into the repo.

The file needs <torch/extension.h>, <cuda.h> and <cuda_runtime_api.h>.  All three exist in this
image (the CUDA headers ship inside the triton wheel's nvidia backend), so the build uses real
headers only; no stand-in header, library or generated file is written.  g++ ignores the CUDA
`__device__` attribute with a warning.  Output goes only to oracle/_ref/ (git-ignored, travels
to the GPU box with the snapshot).  Skipped silently when /root/reference is absent.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/det3d/ops/iou3d_nms/src/iou3d_cpu.cpp"
OUT = os.path.join(HERE, "_ref", "libfd_ref_iou.so")


def cuda_header_dir():
    try:
        import triton
    except Exception:
        return None
    d = os.path.join(os.path.dirname(triton.__file__), "backends", "nvidia", "include")
    return d if os.path.isfile(os.path.join(d, "cuda_runtime_api.h")) else None


def build(force=False):
    if not os.path.isfile(REF_SRC):
        return None
    if os.path.isfile(OUT) and not force and os.path.getmtime(OUT) > os.path.getmtime(__file__):
        return OUT
    inc = cuda_header_dir()
    if inc is None:
        print("[oracle/_ref] CUDA headers not present in image: reference IoU is unbuildable here")
        return None
    import sysconfig

    import torch
    from torch.utils import cpp_extension

    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-w",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch.compiled_with_cxx11_abi()),
           REF_SRC, os.path.join(HERE, "ref_iou_bind.cpp"),
           "-I" + os.path.dirname(REF_SRC), "-I" + sysconfig.get_paths()["include"], "-I" + inc]
    cmd += ["-I" + p for p in cpp_extension.include_paths()]
    cmd += ["-L" + tlib, "-Wl,-rpath," + tlib, "-ltorch", "-ltorch_cpu", "-lc10", "-o", OUT]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
