"""Builds futuredet_amd/libfuturedet_hip.so (gfx950 only) with hipcc.  In-tree so the .so travels with the repo."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libfuturedet_hip.so")
SOURCES = ["fd_error.hip", "fd_voxelize.hip", "fd_index.hip", "fd_spconv.hip", "fd_spconv_v2.hip", "fd_spconv_c32.hip", "fd_spconv_f32r.hip", "fd_spconv_bf16.hip", "fd_spconv_bf16win.hip", "fd_densify.hip", "fd_conv2d.hip", "fd_conv2d_f32.hip", "fd_conv2d_wino.hip", "fd_conv2d_wino_pc.hip", "fd_decode.hip", "fd_sweeps.hip", "fd_pillars.hip", "fd_forecast.hip"]
# geometry / voxel membership follow the reference's operation order: no fma contraction there.
# -fno-slp-vectorize: the SLP vectoriser turns scalar geometry into packed-fp32 instructions, some with an op_sel swizzle of src1
# (v_pk_mul_f32 / v_pk_add_f32 ... op_sel:[0,1]) -- a form that returns wrong values in lanes 48-63 on MI355X while bf16 dense-convolution
# waves of another stream share the compute unit (round 6: flipped NMS decisions in 0.5-3 % of the bf16 sweeps with several in flight;
# fd_decode.hip's comment on footprint_overlap, profiles/round6_determinism_soak.txt).  Same IEEE operations either way: bit-identical
# results, no slower.  A CPU test disassembles the built library and refuses the form in ANY kernel.
EXTRA = {"fd_decode.hip": ["-ffp-contract=off", "-fno-slp-vectorize"], "fd_sweeps.hip": ["-ffp-contract=off"], "fd_forecast.hip": ["-ffp-contract=off"],
         "fd_voxelize.hip": ["-ffp-contract=off"], "fd_pillars.hip": ["-fno-slp-vectorize"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isfile(c) or c == "hipcc"):
            return c
    return "hipcc"


def needs_build():
    if not os.path.isfile(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "futuredet_hip.h"), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_locked(force=False, verbose=False):
    """build() under an exclusive file lock: every rank of a node may call it (no collective involved -- a rank never waits
    in an RCCL barrier for another rank's compiler); the first one builds, the others find the library up to date."""
    import fcntl

    with open(os.path.join(HERE, ".build.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            return build(force=force, verbose=verbose)
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    objdir = os.path.join(HERE, "csrc", "_obj")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-c",
               os.path.join(CSRC, src), "-o", obj] + EXTRA.get(src, [])
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd)))
        objs.append(obj)
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + src)
    subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
