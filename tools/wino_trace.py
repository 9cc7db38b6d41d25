"""Phase timeline of the Winograd conv (tuning build): tools/probes/build_trace.sh && FD_LIB_PATH=tools/probes/libfd_trace.so python tools/wino_trace.py"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from futuredet_amd import hip_ops, lib  # noqa: E402

L = lib.load()
L.fd_debug_set_wino_trace.restype = ctypes.c_int
L.fd_debug_set_wino_trace.argtypes = [ctypes.c_void_p]
names = ["store_raw+barrier", "transform", "barrier", "xi loop (MFMA)", "end barrier", "whole k-loop"]
for (cin, cout, hw0) in ((128, 128, 180), (256, 256, 90), (64, 384, 180)):
    hw = hw0
    x = torch.randn(1, hw, hw, cin, device="cuda")
    w = torch.randn(cout, cin, 3, 3) * 0.02
    wpk = hip_ops.pack_conv2d_weight_wino(w).cuda()
    b = torch.zeros(cout, device="cuda")
    out = torch.empty((1, hw, hw, cout), device="cuda")
    for tile in (6, 5, 3):
        for _ in range(3):
            hip_ops.conv2d_wino_nhwc_f32(x, wpk, b, cout, True, out=out, tile=tile)
        tr = torch.zeros((8192 * 8,), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        assert L.fd_debug_set_wino_trace(tr.data_ptr()) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        hip_ops.conv2d_wino_nhwc_f32(x, wpk, b, cout, True, out=out, tile=tile)
        e1.record()
        torch.cuda.synchronize()
        print("   launch %.1f us" % (e0.elapsed_time(e1) * 1e3))
        L.fd_debug_set_wino_trace(None)
        t = tr.cpu().numpy().reshape(-1, 8)
        ti = t[t[:, 7] > 0]
        t = ti.astype(np.float64)
        ns = cin // 16
        os.makedirs("gpurun_out", exist_ok=True)
        np.save("gpurun_out/wino_trace_%d_%d_%d_t%d.npy" % (cin, cout, hw, tile), t)
        hwid, xcc = (ti[:, 5] >> 32) & 0xffffffff, ti[:, 5] & 15
        cu = ((xcc << 16) | (((hwid >> 13) & 7) << 8) | (((hwid >> 12) & 1) << 4) | ((hwid >> 8) & 15))
        life = t[:, 7] - t[:, 6]
        spans, conc, first = [], [], []
        for c in np.unique(cu):
            m = cu == c
            s0 = t[m, 6].min()
            spans.append(t[m, 7].max() - s0)
            first.append(int(((t[m, 6] - s0) < 3000).sum()))
            conc.append(life[m].sum() / spans[-1])
        spans = np.array(spans)
        print("%d->%d @%d tile %d: %d workgroups on %d CUs, %d slices; per slice cycles: %s ; workgroup life %.0f; per-CU span mean %.0f max %.0f; "
              "workgroups resident per CU (life-weighted) %.2f, co-started %s" %
              (cin, cout, hw0, tile, len(t), len(spans), ns, ", ".join("%s %.0f" % (n, v) for n, v in zip(names[:5], t[:, :5].mean(0) / ns)),
               life.mean(), spans.mean(), spans.max(), float(np.mean(conc)), np.bincount(first).tolist()))
