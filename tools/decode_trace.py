"""Phase stamps of dec_select_hist (fd_decode.hip) on the head maps of one bench sweep.
   tools/probes/build_exp.sh fd_decode trace -DFD_DEC_TRACE && FD_LIB_PATH=tools/probes/libfd_fd_decode_trace.so python tools/decode_trace.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from futuredet_amd import build_detector, lib  # noqa: E402
from futuredet_amd.configs import centerpoint_config  # noqa: E402
from futuredet_amd.synth import seeded_state_dict, synthetic_cloud, tame_box_dims  # noqa: E402

lib.load()
raw = ctypes.CDLL(lib.LIB_PATH)
raw.fd_debug_set_dectrace.restype = ctypes.c_int
raw.fd_debug_set_dectrace.argtypes = [ctypes.c_void_p]
variant = sys.argv[1] if len(sys.argv) > 1 else "forecast_n0"
cfg = centerpoint_config(variant)
net = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
net.load_state_dict(tame_box_dims(seeded_state_dict(net, 7)), strict=False)
net = net.cuda().eval()
pts = torch.from_numpy(synthetic_cloud(seed=0, target_points=300000)).cuda()
with torch.no_grad():
    for _ in range(2):
        net.forward_points([pts], cfg.voxel_generator, padded="packed")
    trace = torch.zeros((64 * 16,), dtype=torch.int64, device="cuda")
    assert raw.fd_debug_set_dectrace(trace.data_ptr()) == 0
    p, c = net.forward_points([pts], cfg.voxel_generator, padded="packed")
    torch.cuda.synchronize()
    raw.fd_debug_set_dectrace(None)
t = trace.cpu().numpy().reshape(-1, 16)
for g in range(len(t)):
    if t[g, 0] == 0:
        continue
    d = t[g]
    print("group %d: valid %d, in the threshold bin %d, needed from it %d; cycles: histogram sum + key loads %d, total + bin search %d, keys above the bin %d, "
          "keys inside the bin %d" % (g, d[8], d[9], d[10], d[1] - d[0], d[2] - d[1], d[3] - d[2], (d[4] - d[3]) if d[4] else -1))
    if d[11]:
        print("  nms_sweep_tail: staging %d, sweep %d cycles" % (d[12] - d[11], d[13] - d[12]))
print("detections", int(c.sum()))
