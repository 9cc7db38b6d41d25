"""Which part of a sweep disturbs nms_mask?  Stream A replays one whole captured sweep (the victim; with the -DFD_MASK_DEBUG build its
nms_mask evaluates every near pair twice and counts disagreements); streams B.. replay, for the whole duration of it, one disturber
component in a loop: whole sweeps, the sparse backbone, the neck + head plan, the neck plan's blocks / deblocks, or decodes.
    FD_LIB_PATH=tools/probes/libfd_maskdbg.so python tools/soak_pairs.py [rounds per disturber] [fp32|bf16] [disturber streams] [name filter, comma separated]
(every graph's inputs, weights and plans are kept alive for the whole run: a disturber whose inputs were freed faults, which is the harness, not the kernels)"""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
import futuredet_amd.detectors as D  # noqa: E402
from futuredet_amd import build_detector, hip_ops  # noqa: E402
from futuredet_amd.configs import centerpoint_config  # noqa: E402
from futuredet_amd.detectors import StaticStep  # noqa: E402
from futuredet_amd.synth import seeded_state_dict, synthetic_cloud, tame_box_dims  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dtype = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == "fp32") else torch.bfloat16
NB = int(sys.argv[3]) if len(sys.argv) > 3 else 3
cfg = centerpoint_config("forecast_n3")
net = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
net.load_state_dict(tame_box_dims(seeded_state_dict(net, 7)), strict=False)
net = net.cuda().eval()
net.set_precision(dtype)
B = 2
sa = torch.cuda.Stream()
sbs = [torch.cuda.Stream() for _ in range(NB)]
L = hip_ops._lib.load()


def counters():
    if not hasattr(L, "fd_debug_mask_counters"):
        return None
    c = (ctypes.c_int * 8)()
    L.fd_debug_mask_counters(c)
    return list(c)


def capture(fn, stream, scope):
    with torch.cuda.stream(stream):
        with hip_ops.workspace.scope(scope):
            out = fn()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream):
                out = fn()
    torch.cuda.synchronize()
    return g, out


def sweep_graph(stream, clouds):
    with torch.cuda.stream(stream):
        step = StaticStep(net, cfg.voxel_generator, max(c.shape[0] for c in clouds) + 1024, batch_size=B, ndim=5, packed=True, row_caps="auto")
        step.warm_up(clouds)
        step.capture()
        step._load(clouds)
    torch.cuda.synchronize()
    return step


with torch.no_grad():
    clouds = [torch.from_numpy(synthetic_cloud(seed=b, target_points=300000)).cuda() for b in range(B)]
    stage = {}
    net.__dict__["debug_taps"] = stage
    D._NO_GRAPH = True
    net.forward_points(clouds, cfg.voxel_generator, padded="packed")
    net.__dict__["debug_taps"] = None
    torch.cuda.synchronize()
    preds, bev, idx, feats0 = stage["preds"], stage["bev"], stage["idx"], stage["feats0"]
    victim = sweep_graph(sa, clouds)
    alive = [sweep_graph(sb, clouds) for sb in sbs]
    dist = {"nothing": (None, 0), "whole sweep": ([s.graph for s in alive], 1)}
    for name, reps, fn in (("sparse backbone", 2, lambda: net.backbone.run_fused(idx, feats0)[0]),
                           ("neck + head plan", 6, lambda: net.bbox_head(net.neck(bev))),
                           ("decode", 30, lambda: net.bbox_head.predict_packed(preds, net.test_cfg))):
        dist[name] = ([capture(fn, sb, ("soak", name, i))[0] for i, sb in enumerate(sbs)], reps)
    neck_out = net.neck(bev)
    dist["neck plan"] = ([capture(lambda: net.neck(bev), sb, ("soak", "neck", i))[0] for i, sb in enumerate(sbs)], 8)
    dist["head plan"] = ([capture(lambda: net.bbox_head(neck_out), sb, ("soak", "head", i))[0] for i, sb in enumerate(sbs)], 16)
    rp = net.neck._plan[1]
    xin = bev.to(dtype).permute(0, 2, 3, 1).contiguous()

    def block(i, x):
        for conv in rp.blocks[i]:
            x = conv(x)
        return x
    x1 = block(0, xin)
    x2 = block(1, x1)
    ups = torch.empty((B, x1.shape[1], x1.shape[2], rp.cout_total), dtype=dtype, device="cuda")
    keep_alive = [xin, x1, x2, ups]

    def deblock(j, x, co):
        kind, k, op, cout = rp.deblocks[j]
        if kind == "conv":
            op(x, out=ups, co_off=co)
        elif kind == "up":
            for sub, dy, dx in op:
                sub(x, out=ups, co_off=co, osy=k, osx=k, ooy=dy, oox=dx)
        return ups
    for nm, reps, fn in (("neck block 0 (3x3 @180)", 12, lambda: block(0, xin)), ("neck block 1 (s2 + 3x3 @90)", 16, lambda: block(1, x1)),
                         ("neck deblock 0 (%s)" % rp.deblocks[0][0], 60, lambda: deblock(0, x1, 0)),
                         ("neck deblock 1 (%s)" % rp.deblocks[1][0], 30, lambda: deblock(1, x2, rp.deblocks[0][3]))):
        dist[nm] = ([capture(fn, sb, ("soak", nm, i))[0] for i, sb in enumerate(sbs)], reps)
    only = sys.argv[4].split(",") if len(sys.argv) > 4 else None
    if only:
        dist = {k: v for k, v in dist.items() if any(o in k for o in only)}
    with torch.cuda.stream(sa):
        victim.graph.replay()
    torch.cuda.synchronize()
    ref = tuple(t.clone() for t in victim.outputs)
    for name, (gbs, reps) in dist.items():
        bad, c0 = 0, counters()
        for r in range(rounds):
            if gbs is not None:
                for sb, gb in zip(sbs, gbs):
                    with torch.cuda.stream(sb):
                        for _ in range(reps):
                            gb.replay()
            with torch.cuda.stream(sa):
                victim.graph.replay()
            torch.cuda.synchronize()
            bad += not all(torch.equal(a, b) for a, b in zip(victim.outputs, ref))
        c1 = counters()
        extra = ""
        if c0 is not None:
            extra = "; nms_mask self-check (victim and disturbers): %d near pairs, re-evaluation differs %d times" % (c1[0] - c0[0], c1[1] - c0[1])
        print("%-18s x %d stream(s) x %2d replays per round: %4d of %d victim sweeps differ%s" % (name, NB if gbs else 0, reps, bad, rounds, extra), flush=True)
