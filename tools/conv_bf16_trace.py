"""Phase timeline of the bf16 dense conv (tuning build):
   tools/probes/build_exp.sh fd_conv2d trace -DFD_V2_TRACE && FD_LIB_PATH=tools/probes/libfd_fd_conv2d_trace.so python tools/conv_bf16_trace.py [batch]"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from futuredet_amd import hip_ops, lib  # noqa: E402
L = lib.load()
L.fd_debug_set_conv_trace.restype = ctypes.c_int
L.fd_debug_set_conv_trace.argtypes = [ctypes.c_void_p]
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for (cin, cout, hw) in ((128, 128, 180), (256, 128, 180), (256, 256, 90), (512, 64, 180)):
    x = torch.randn(NB, hw, hw, cin, device="cuda").bfloat16()
    w = torch.randn(cout, cin, 3, 3) * 0.02
    wpk = hip_ops.pack_conv2d_weight(w).cuda()
    b = torch.zeros(cout, device="cuda")
    out = torch.empty((NB, hw, hw, cout), device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        hip_ops.conv2d_nhwc_bf16(x, wpk, b, cout, 3, 1, True, out=out)
    tr = torch.zeros((4096 * 8,), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    assert L.fd_debug_set_conv_trace(tr.data_ptr()) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    hip_ops.conv2d_nhwc_bf16(x, wpk, b, cout, 3, 1, True, out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    L.fd_debug_set_conv_trace(None)
    t = tr.cpu().numpy().reshape(-1, 8).astype(np.float64)
    t = t[t[:, 7] > 0]
    # back to back: the layer's time without launch latency
    e0.record()
    for _ in range(20):
        hip_ops.conv2d_nhwc_bf16(x, wpk, b, cout, 3, 1, True, out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    ns = cin // 32
    # o[7] = s_memrealtime at the workgroup's start (100 MHz, device-wide): who starts late = who had to wait for a slot
    rt = (t[:, 7] - t[:, 7].min()) * 0.01  # us
    span = 0.0
    print("   workgroup start times (us after the first): median %.2f, p75 %.2f, p90 %.2f, p95 %.2f, max %.2f; started after 2 us: %d of %d" % (
        np.median(rt), np.percentile(rt, 75), np.percentile(rt, 90), np.percentile(rt, 95), rt.max(), int((rt > 2.0).sum()), len(rt)))
    print("B=%d " % NB + "%d->%d @%d: %.1f us back to back, %d workgroups, %d slices; span %.0f cycles (%.2f GHz); life mean %.0f max %.0f; prologue %.0f; per slice: steps %.0f (MFMA %d), hand-over %.0f, barrier %.0f; epilogue %.0f (of it transpose into LDS + barrier %.0f)" % (cin, cout, hw, us, len(t), ns, span, span / us / 1e3, t[:, 5].mean(), t[:, 5].max(), t[:, 0].mean(), t[:, 1].mean() / ns, 18 * 4 * 32,
                                                   t[:, 2].mean() / ns, t[:, 3].mean() / ns, t[:, 4].mean(), t[:, 6].mean()), flush=True)
