"""Phase timeline of the producer/consumer Winograd conv (tuning build):
tools/probes/build_trace.sh && FD_LIB_PATH=tools/probes/libfd_trace.so python tools/wino_pc_trace.py"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from futuredet_amd import hip_ops, lib  # noqa: E402

L = lib.load()
L.fd_debug_set_wino_pc_trace.restype = ctypes.c_int
L.fd_debug_set_wino_pc_trace.argtypes = [ctypes.c_void_p]
for (cin, cout, hw) in ((128, 128, 180), (256, 256, 90), (64, 384, 180), (512, 64, 180)):
    x = torch.randn(1, hw, hw, cin, device="cuda")
    w = torch.randn(cout, cin, 3, 3) * 0.02
    wpk = hip_ops.pack_conv2d_weight_wino(w).cuda()
    b = torch.zeros(cout, device="cuda")
    out = torch.empty((1, hw, hw, cout), device="cuda")
    for _ in range(3):
        hip_ops.conv2d_wino_nhwc_f32(x, wpk, b, cout, True, out=out, tile=7)
    tr = torch.zeros((1024 * 8,), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    assert L.fd_debug_set_wino_pc_trace(tr.data_ptr()) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    hip_ops.conv2d_wino_nhwc_f32(x, wpk, b, cout, True, out=out, tile=7)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    L.fd_debug_set_wino_pc_trace(None)
    t = tr.cpu().numpy().reshape(-1, 8).astype(np.float64)
    t = t[t[:, 6] > 0]
    life = t[:, 6] - t[:, 5]
    span = t[:, 6].max() - t[:, 5].min()
    ns = cin // 16
    n_items = -(-((hw + 1) // 2) ** 2 // 32) * ((cout + 63) // 64)
    steps = n_items * ns / len(t)
    print("%d->%d @%d: %.1f us, %d workgroups, %.1f steps each; kernel span %.0f cycles (%.2f GHz if the span is the launch), workgroup life %.0f; "
          "per step: multiply %.0f (ideal 4096), consumer barrier wait %.0f, producer work %.0f, producer barrier wait %.0f; per workgroup: prologue %.0f, epilogues %.0f"
          % (cin, cout, hw, us, len(t), steps, span, span / us / 1e3, life.mean(), t[:, 0].mean() / steps, t[:, 1].mean() / steps, t[:, 3].mean() / steps,
             t[:, 4].mean() / steps, t[:, 7].mean(), t[:, 2].mean()), flush=True)
