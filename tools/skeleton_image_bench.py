"""Stop-rule experiment for the fp32 sparse convolutions (round-4 review, item 3): what a launch costs when the compacted per-tap lists
and item lists come ready-made instead of being built per convolution.
   tools/probes/build_exp.sh fd_spconv_c32 img -DFD_SKELETON_IMAGE   (and / or fd_spconv_v2)
   FD_LIB_PATH=tools/probes/libfd_fd_spconv_c32_img.so python tools/skeleton_image_bench.py 1
A dump launch (mode 1) writes every workgroup's lists to a global image, load launches (mode 2) read them back; results must be identical."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from futuredet_amd import build_backbone, hip_ops, lib  # noqa: E402
from futuredet_amd.synth import seeded_state_dict, synthetic_cloud  # noqa: E402

lib.load()
raw = ctypes.CDLL(lib.LIB_PATH)
raw.fd_debug_set_skeleton_image.restype = ctypes.c_int
raw.fd_debug_set_skeleton_image.argtypes = [ctypes.c_void_p, ctypes.c_int]
raw.fd_debug_skeleton_image_ints.restype = ctypes.c_int
dev = torch.device("cuda")
pts = torch.from_numpy(synthetic_cloud(0, 300000)).to(dev)
out = hip_ops.voxelize(pts, [0.075, 0.075, 0.2], [-54, -54, -5.0, 54, 54, 3.0], 10, 160000, want_voxels=False, want_mean=True, mean_stride=16, coor_cols=4)
m = int(out["num_voxels"].cpu()[0])
bb = build_backbone(dict(type="SpMiddleResNetFHD", num_input_features=5, ds_factor=8)).to(dev).eval()
idx = bb.build_indexes(lambda i0: i0.mark(out["coors"][:m].contiguous()), 1, [1440, 1440, 40], dev)
per_wg = raw.fd_debug_skeleton_image_ints()


def timed(fn, iters=30):
    for _ in range(3):
        y = fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        y = fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters, y


for lvl in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1").split(",")]:
    C = [16, 32, 64, 128][lvl]
    ix = idx[lvl]
    x = torch.randn((ix.n, C), device=dev)
    wpk = hip_ops.pack_spconv_weight(torch.randn((27, C, C)) * (2.0 / (27 * C)) ** 0.5).to(dev)
    bias = torch.zeros(C, device=dev)
    nbr = ix.rulebook(ix, [3, 3, 3], [1, 1, 1], [1, 1, 1])
    run = lambda: hip_ops.spconv_apply(x, wpk, bias, nbr, ix.n, C, residual=x, relu=True)  # noqa: E731
    image = torch.zeros((4096 * per_wg,), dtype=torch.int32, device=dev)
    for rep in range(2):
        raw.fd_debug_set_skeleton_image(None, 0)
        us0, y0 = timed(run)
        raw.fd_debug_set_skeleton_image(image.data_ptr(), 1)
        y1 = run()
        torch.cuda.synchronize()
        raw.fd_debug_set_skeleton_image(image.data_ptr(), 2)
        us2, y2 = timed(run)
        raw.fd_debug_set_skeleton_image(None, 0)
        print("level %d C=%d rows %d: product %.1f us; lists ready-made (image load) %.1f us; results identical: %s / %s" % (
            lvl, C, ix.n, us0, us2, bool(torch.equal(y0, y1)), bool(torch.equal(y0, y2))), flush=True)
