"""Determinism soak: four captured sweeps (detectors.StaticStep, packed outputs) in flight on four streams, each replayed ROUNDS times on
its own pair of clouds; every replay's packed detections + counts + level counts must equal the first replay's bit for bit.  A race
inside a graph or between passes sharing the device (workspaces, the conv plan's buffers) shows up as a differing replay.
    python tools/soak_determinism.py [fp32|bf16] [rounds] [passes in flight]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from futuredet_amd import build_detector, hip_ops  # noqa: E402
from futuredet_amd.configs import centerpoint_config  # noqa: E402
from futuredet_amd.detectors import StaticStep  # noqa: E402
from futuredet_amd.synth import seeded_state_dict, synthetic_cloud, tame_box_dims  # noqa: E402

dtype = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float32
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 200
variant = sys.argv[4] if len(sys.argv) > 4 else ("forecast_n3" if dtype == torch.bfloat16 else "forecast_n0")
IS_PP = variant == "pp"
if IS_PP:  # the PointPillars configs (reader + scatter instead of the sparse backbone)
    from futuredet_amd.configs import pointpillars_config
    cfg = pointpillars_config("car")
elif variant == "config5":  # BASELINE configs[4]: pedestrian forecast_n3 on the 0.05 m grid, 500k-point clouds (BEV 270 x 270: the decode's streaming selection)
    cfg = centerpoint_config("forecast_n3", "pedestrian", voxel_size=(0.05, 0.05, 0.2), max_voxel_num=(300000, 400000))
else:
    cfg = centerpoint_config(variant)
POINTS = 500000 if variant == "config5" else 300000
net = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
net.load_state_dict(tame_box_dims(seeded_state_dict(net, 7)), strict=False)
net = net.cuda().eval()
net.set_precision(dtype)
B, NS = 2, (int(sys.argv[3]) if len(sys.argv) > 3 else 4)
clouds = [[torch.from_numpy(synthetic_cloud(seed=10 * s + b, target_points=POINTS)).cuda() for b in range(B)] for s in range(NS)]
cap = max(c.shape[0] for cs in clouds for c in cs) + 1024
streams = [torch.cuda.Stream() for _ in range(NS)]
steps, taps = [], []


class LayerTaps:
    """backbone.profile_hook: keeps every sparse convolution's output tensor of the capture (static memory of the graph)"""

    def __init__(self):
        self.outs = []

    def __call__(self, tag, info, fn):
        y = fn()
        self.outs.append((tag, y))
        return y


with torch.no_grad():
    for s, st in enumerate(streams):
        with torch.cuda.stream(st):
            net.forward_points(clouds[s], cfg.voxel_generator, padded="packed")
            step = StaticStep(net, cfg.voxel_generator, cap, batch_size=B, ndim=5, packed=True, row_caps="datafree" if IS_PP else "auto")
            step.warm_up(clouds[s])
            hook, stage = LayerTaps(), {}
            if not IS_PP:
                net.backbone.profile_hook, net.__dict__["debug_taps"] = hook, stage
            step.capture()
            if not IS_PP:
                net.backbone.profile_hook, net.__dict__["debug_taps"] = None, None
            steps.append(step)
            if IS_PP:
                taps.append([])
                continue
            ws_dec = hip_ops.workspace._bufs.get(("decode", torch.cuda.current_device(), ("scope", id(step))))
            named = ([("decode workspace", ws_dec)] if ws_dec is not None else []) + [(k, stage[k]) for k in ("mean", "coors", "num_points", "num_voxels", "feats0")] + \
                    [("%02d %s" % (i, t), y) for i, (t, y) in enumerate(hook.outs)] + [(k, stage[k]) for k in ("bev", "neck", "head") if stage.get(k) is not None]
            taps.append(named)
    torch.cuda.synchronize()
    print("stages compared per replay:", ", ".join(n for n, _ in taps[0]))
    first, bad = [None] * NS, 0
    for r in range(rounds):
        snaps = []
        for s, st in enumerate(streams):
            with torch.cuda.stream(st):
                packed, counts = steps[s](clouds[s], check=False)
                lc = steps[s].level_counts if steps[s].level_counts is not None and torch.is_tensor(steps[s].level_counts) else torch.zeros(1, dtype=torch.int32, device="cuda")
                snaps.append((packed.clone(), counts.clone(), lc.clone()) + tuple(t.clone() for _, t in taps[s]))
        torch.cuda.synchronize()
        for s, snap in enumerate(snaps):
            assert IS_PP or not steps[s].overflowed(snap[2].cpu().tolist())
            if first[s] is None:
                first[s] = snap
                assert int(snap[1].sum()) > 0
            elif not all(torch.equal(a, b) for (a, b), nm in zip(zip(snap, first[s]), ["", "", ""] + [n for n, _ in taps[s]]) if nm != "decode workspace"):
                bad += 1
                dp = snap[0] != first[s][0]
                print("round %d stream %d differs: %d packed elements; per column %s; per (sample, step) %s" % (
                    r, s, int(dp.sum()), dp.sum((0, 1, 2)).tolist(), dp.any(-1).sum(-1).tolist()))
                if bad <= 2 and bool(dp.any()):
                    b_, s_ = [int(v) for v in dp.any(-1).any(-1).nonzero()[0]]
                    k_ = int(dp[b_, s_].any(-1).nonzero()[0])
                    print("    first differing row (sample %d, step %d, row %d):\n      now   %s\n      first %s" % (
                        b_, s_, k_, snap[0][b_, s_, k_].tolist(), first[s][0][b_, s_, k_].tolist()))
                for (name, _), a, b in zip(taps[s], snap[3:], first[s][3:]):  # the first stage whose tensor differs is where it starts
                    if not torch.equal(a, b):
                        d = (a != b)
                        rows = d.reshape(d.shape[0], -1).any(1).nonzero().flatten() if d.dim() > 1 else d.nonzero().flatten()
                        print("    %-28s %s: %d elements in %d rows differ, rows %s ... %s" % (
                            name, tuple(a.shape), int(d.sum()), len(rows), rows[:6].tolist(), rows[-3:].tolist()))
                        if name == "decode workspace":  # dec_layout (fd_decode.hip): regions in order, each rounded up to 256 bytes
                            G_, HW, pre, post = B, 180 * 180, 1000, 83
                            cb = (pre + 63) // 64
                            sizes = [("keys", 4 * G_ * HW), ("sel_boxes", 28 * G_ * pre), ("nms_boxes", 28 * G_ * pre), ("sel_scores", 4 * G_ * pre), ("sel_cell", 4 * G_ * pre),
                                     ("sel_count", 8 * G_), ("mask", 8 * G_ * pre * cb), ("keep", 4 * G_ * post), ("foot", 64 * G_ * pre), ("hist", 4096 * G_ * ((HW + 1023) // 1024)),
                                     ("words", 8 * G_ * pre)]
                            off = 0
                            for nm, sz in sizes:
                                seg = d[off:off + sz]
                                if bool(seg.any()):
                                    idx = seg.nonzero().flatten()
                                    print("        region %-10s %6d bytes differ, first at byte %d, last at %d of %d" % (nm, int(seg.sum()), int(idx[0]), int(idx[-1]), sz))
                                if nm == "mask" and bool(seg.any()):  # which (row, column) decisions flipped, and how close to the threshold they are
                                    wa = a[off:off + sz].view(torch.int64).view(G_, pre, cb)
                                    wb = b[off:off + sz].view(torch.int64).view(G_, pre, cb)
                                    nb_off = sum((z + 255) // 256 * 256 for _, z in sizes[:1])
                                    selb = a[nb_off:nb_off + 28 * G_ * pre].view(torch.float32).view(G_, pre, 7)  # x y z d0 d1 d2 yaw -> the NMS layout (box_torch_ops.py:256-257)
                                    nmsb = torch.stack([selb[..., 0], selb[..., 1], selb[..., 2], selb[..., 4], selb[..., 3], selb[..., 5], -selb[..., 6] - 1.5707963267948966], -1)
                                    for g_, r_, c_ in (wa != wb).nonzero().tolist():
                                        x = int(wa[g_, r_, c_]) ^ int(wb[g_, r_, c_])
                                        cols = [c_ * 64 + t for t in range(64) if (x >> t) & 1]
                                        for col in cols:
                                            pair = torch.stack([nmsb[g_, r_], nmsb[g_, col]])
                                            ious = [float(hip_ops.boxes_iou_bev(pair[:1].contiguous(), pair[1:].contiguous())[0, 0]) for _ in range(3)]
                                            print("          group %d row %d col %d: bit now %d first %d; IoU by fd_boxes_iou_bev x3 %s; row >= col? %s" % (
                                                g_, r_, col, (int(wa[g_, r_, c_]) >> (col % 64)) & 1, (int(wb[g_, r_, c_]) >> (col % 64)) & 1, ious, r_ >= col))
                                off += (sz + 255) // 256 * 256
    print("%s: %d rounds x %d passes in flight, %d detections per pass pair, %d differing replays" % (
        str(dtype).split(".")[-1], rounds, NS, int(first[0][1].sum()), bad))
    import ctypes
    L = hip_ops._lib.load()
    if hasattr(L, "fd_debug_mask_counters"):  # tuning build -DFD_MASK_DEBUG: nms_mask evaluates every near pair three times
        c4 = (ctypes.c_int * 8)()
        L.fd_debug_mask_counters(c4)
        print("    two more evaluations of each near pair, first quantity that differs between them: point count %d, point set %d, angles %d, area %d" % tuple(c4[4:8]))
        c4 = list(c4)[:4]
        print("nms_mask self-check: %d near pairs evaluated; second evaluation (inputs pinned in registers) differs %d times; third (inputs loaded "
              "again) differs from the first %d times; the re-loaded footprints differ from the pinned ones %d times" % tuple(c4))
    if hasattr(L, "fd_debug_mask_log"):
        lg = (ctypes.c_uint * 1024)()
        L.fd_debug_mask_log(lg)
        import struct
        f = lambda u: struct.unpack("f", struct.pack("I", u))[0]  # noqa: E731
        for i in range(64):
            o = lg[16 * i:16 * i + 16]
            if not any(o):
                break
            print("  log %2d: group %d row %d col %d | n %d vs %d | made %06x vs %06x (xor %06x) | side values %s | points %s | area %.6f vs %.6f | hw_id %08x thread %d block %d" % (
                i, o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[5] ^ o[6], "same" if o[7] == o[8] else "DIFFER", "same" if o[9] == o[10] else "differ", f(o[11]), f(o[12]), o[13], o[14], o[15]))
    sys.exit(1 if bad else 0)
