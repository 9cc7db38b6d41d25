"""Phase timeline of the fp32 sparse conv (tuning build only):
   tools/probes/build_trace.sh && FD_LIB_PATH=tools/probes/libfd_trace.so python tools/spconv_trace.py [--levels 1,2,3]
Prints, per level, the mean/max shader-clock cycles of each phase of a workgroup's FIRST chunk, and the whole-kernel span."""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from futuredet_amd import build_backbone, hip_ops, lib  # noqa: E402
from futuredet_amd.synth import seeded_state_dict, synthetic_cloud  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--levels", default="0,1,2,3")
ap.add_argument("--mode", default="balanced")
ap.add_argument("--rpc", type=int, default=0)
ap.add_argument("--down", action="store_true", help="the strided convolution leaving each level (16 -> 32, 32 -> 64, 64 -> 128) instead of its SubM layers")
args = ap.parse_args()
L = lib.load()
L.fd_debug_set_trace.restype = ctypes.c_int
L.fd_debug_set_trace.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda")
pts = torch.from_numpy(synthetic_cloud(0, 300000)).to(dev)
out = hip_ops.voxelize(pts, [0.075, 0.075, 0.2], [-54, -54, -5.0, 54, 54, 3.0], 10, 160000, want_voxels=False, want_mean=True, mean_stride=16, coor_cols=4)
m = int(out["num_voxels"].cpu()[0])
bb = build_backbone(dict(type="SpMiddleResNetFHD", num_input_features=5, ds_factor=8))
bb.load_state_dict(seeded_state_dict(bb, 7), strict=False)
bb = bb.to(dev).eval()
idx = bb.build_indexes(lambda i0: i0.mark(out["coors"][:m].contiguous()), 1, [1440, 1440, 40], dev)
MODE = {"tiles": "tiles", "uniform": False, "balanced": True}
hip_ops.set_tuning("v2_ranges_per_cu", args.rpc)
names = ["stage+zero", "compaction", "items+first gathers", "main loop", "loop-end barrier", "epilogue"]
for lvl in [int(v) for v in args.levels.split(",")]:
    C = [16, 32, 64, 128][lvl]
    ix = idx[lvl]
    x = torch.randn((ix.n, C), device=dev)
    if args.down:
        if lvl > 2:
            continue
        ox, CO = idx[lvl + 1], 2 * C
        wpk = hip_ops.pack_spconv_weight(torch.randn((27, C, CO)) * 0.05).to(dev)
        nbr = ix.rulebook(ox, [3, 3, 3], [2, 2, 2], [1, 1, 1] if lvl < 2 else [0, 1, 1])
        run = lambda: hip_ops.spconv_apply(x, wpk, None, nbr, ox.n, CO, relu=True, balanced=MODE[args.mode])  # noqa: E731
    else:
        wpk = hip_ops.pack_spconv_weight(torch.randn((27, C, C)) * 0.05).to(dev)
        nbr = ix.rulebook(ix, [3, 3, 3], [1, 1, 1], [1, 1, 1])
        run = lambda: hip_ops.spconv_apply(x, wpk, None, nbr, ix.n, C, residual=x, relu=True, balanced=MODE[args.mode])  # noqa: E731
    trace = torch.zeros((4096 * 16,), dtype=torch.int64, device=dev)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    assert L.fd_debug_set_trace(trace.data_ptr()) == 0
    run()
    torch.cuda.synchronize()
    L.fd_debug_set_trace(None)
    t = trace.cpu().numpy().reshape(-1, 16)
    t = t[t[:, 0] > 0]
    d = np.diff(t[:, :7].astype(np.float64), axis=1)
    span = float(t[:, 7].max()) - float(t[:, 0].min())
    print("level %d C=%d: %d workgroups, kernel span %.0f cycles; first chunk: items/wave %.1f, chunks/wg %.2f, rows %.1f" %
          (lvl, C, len(t), span, t[:, 8].mean(), t[:, 9].mean(), t[:, 10].mean()))
    for i, nme in enumerate(names):
        print("   %-22s mean %8.0f  max %8.0f cycles  (%.1f %% of the first chunk)" % (nme, d[:, i].mean(), d[:, i].max(), 100 * d[:, i].mean() / d.sum(1).mean()))
    life = (t[:, 7] - t[:, 0]).astype(np.float64)
    items = t[:, 13].astype(np.float64)  # items of wave 0 summed over the chunks
    hw, xcc = t[:, 11], t[:, 12] & 15
    cu = ((xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15))
    ucu, inv, cnt = np.unique(cu, return_inverse=True, return_counts=True)
    cu_items = np.bincount(inv, weights=items)
    cu_end = np.array([t[inv == i, 7].max() for i in range(len(ucu))], np.float64) - t[:, 0].min()
    print("   items/wg: mean %.1f min %.0f max %.0f; lifetime/item mean %.0f; corr(life, items) %.2f" %
          (items.mean(), items.min(), items.max(), (life / np.maximum(items, 1)).mean(), np.corrcoef(life, items)[0, 1]))
    print("   CUs used %d; workgroups per CU: %s; items per CU mean %.0f max %.0f; CU end time mean %.0f max %.0f (cycles after first start)" %
          (len(ucu), dict(zip(*np.unique(cnt, return_counts=True))), cu_items.mean(), cu_items.max(), cu_end.mean(), cu_end.max()))
    print("   workgroup lifetime     mean %8.0f  max %8.0f; start spread %.0f, end spread %.0f" %
          ((t[:, 7] - t[:, 0]).mean(), (t[:, 7] - t[:, 0]).max(), t[:, 0].max() - t[:, 0].min(), t[:, 7].max() - t[:, 7].min()))
