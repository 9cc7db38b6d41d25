"""Do two large kernels from different streams overlap (does the tail of one fill with the next)?  Times 2 N launches of one sparse
conv layer on ONE stream against N + N on TWO streams (each a captured graph, replayed together).
usage: python tools/overlap_probe.py [--level 3] [--n 10]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from futuredet_amd import build_backbone, hip_ops  # noqa: E402
from futuredet_amd.synth import seeded_state_dict, synthetic_cloud  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--level", type=int, default=3)
ap.add_argument("--n", type=int, default=10)
ap.add_argument("--dtype", default="fp32")
args = ap.parse_args()
dev = torch.device("cuda")
dt = torch.float32 if args.dtype == "fp32" else torch.bfloat16
pts = torch.from_numpy(synthetic_cloud(0, 300000)).to(dev)
out = hip_ops.voxelize(pts, [0.075, 0.075, 0.2], [-54, -54, -5.0, 54, 54, 3.0], 10, 160000, want_voxels=False, want_mean=True, mean_stride=16, coor_cols=4)
m = int(out["num_voxels"].cpu()[0])
bb = build_backbone(dict(type="SpMiddleResNetFHD", num_input_features=5, ds_factor=8))
bb.load_state_dict(seeded_state_dict(bb, 7), strict=False)
bb = bb.to(dev).eval()
idx = bb.build_indexes(lambda i0: i0.mark(out["coors"][:m].contiguous()), 1, [1440, 1440, 40], dev)
C = [16, 32, 64, 128][args.level]
ix = idx[args.level]
w = torch.randn((27, C, C)) * (2.0 / (27 * C)) ** 0.5
wpk = hip_ops.pack_spconv_weight(w, dt).to(dev)
bias = torch.zeros(C, device=dev)
nbr = ix.rulebook(ix, [3, 3, 3], [1, 1, 1], [1, 1, 1])
xs = [torch.randn((ix.n, C), device=dev).to(dt) for _ in range(2)]
ys = [torch.empty_like(x) for x in xs]


def run(k, n):
    for _ in range(n):
        hip_ops.spconv_apply(xs[k], wpk, bias, nbr, ix.n, C, residual=xs[k], relu=True)


run(0, 3); run(1, 3)
torch.cuda.synchronize()
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
graphs = {}
for name, plan in (("one", [(0, 2 * args.n)]), ("a", [(0, args.n)]), ("b", [(1, args.n)])):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=streams[0]):
        for k, n in plan:
            run(k, n)
    graphs[name] = g
torch.cuda.synchronize()


def wall(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e6


def one():
    with torch.cuda.stream(streams[0]):
        graphs["one"].replay()


def two():
    with torch.cuda.stream(streams[0]):
        graphs["a"].replay()
    with torch.cuda.stream(streams[1]):
        graphs["b"].replay()


t1, t2 = wall(one), wall(two)
print("level %d (%d -> %d, %d rows) %s: %d launches on one stream %.0f us (%.1f each); %d + %d on two streams %.0f us (%.1f per launch): ratio %.3f"
      % (args.level, C, C, ix.n, args.dtype, 2 * args.n, t1, t1 / (2 * args.n), args.n, args.n, t2, t2 / (2 * args.n), t2 / t1))
