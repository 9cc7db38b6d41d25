"""Decode-only determinism soak (narrowing tools/soak_determinism.py): NS head outputs of the bf16 forecast_n3 model, one per stream; each stream
replays a small graph holding only fd_centerpoint_decode_packed on its own maps; every replay's packed rows must equal the first's.
    python tools/soak_decode.py [rounds] [streams] [fp32|bf16]"""
import sys

import torch

sys.path.insert(0, ".")
from futuredet_amd import build_detector, hip_ops  # noqa: E402
from futuredet_amd.configs import centerpoint_config  # noqa: E402
from futuredet_amd.synth import seeded_state_dict, synthetic_cloud, tame_box_dims  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 500
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dtype = torch.float32 if (len(sys.argv) > 3 and sys.argv[3] == "fp32") else torch.bfloat16
cfg = centerpoint_config("forecast_n3")
net = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
net.load_state_dict(tame_box_dims(seeded_state_dict(net, 7)), strict=False)
net = net.cuda().eval()
net.set_precision(dtype)
B = 2
streams = [torch.cuda.Stream() for _ in range(NS)]
graphs, outs, keep = [], [], []
with torch.no_grad():
    for s, st in enumerate(streams):
        clouds = [torch.from_numpy(synthetic_cloud(seed=10 * s + b, target_points=300000)).cuda() for b in range(B)]
        stage = {}
        net.__dict__["debug_taps"] = stage
        import futuredet_amd.detectors as D
        D._NO_GRAPH, old = True, D._NO_GRAPH
        net.forward_points(clouds, cfg.voxel_generator, padded="packed")
        D._NO_GRAPH = old
        net.__dict__["debug_taps"] = None
        torch.cuda.synchronize()
        preds = stage["preds"]
        keep.append(preds)
        with torch.cuda.stream(st):
            with hip_ops.workspace.scope(("soak", s)):
                net.bbox_head.predict_packed(preds, net.test_cfg)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=st):
                    out = net.bbox_head.predict_packed(preds, net.test_cfg)
        graphs.append(g)
        outs.append(out)
    torch.cuda.synchronize()
    first, bad = [None] * NS, 0
    for r in range(rounds):
        snaps = []
        for s, st in enumerate(streams):
            with torch.cuda.stream(st):
                graphs[s].replay()
                snaps.append((outs[s][0].clone(), outs[s][1].clone()))
        torch.cuda.synchronize()
        for s, snap in enumerate(snaps):
            if first[s] is None:
                first[s] = snap
            elif not (torch.equal(snap[0], first[s][0]) and torch.equal(snap[1], first[s][1])):
                bad += 1
                print("round %d stream %d differs: %d packed elements" % (r, s, int((snap[0] != first[s][0]).sum())))
    print("decode only, %s maps: %d rounds x %d streams, %d differing replays" % (str(dtype).split(".")[-1], rounds, NS, bad))
