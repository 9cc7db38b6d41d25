"""Micro-benchmark of fd_spconv_apply on the real rulebooks of one synthetic cloud (for rocprofv3 / tuning).
usage: python tools/spconv_bench.py [--levels 0,1,2,3] [--dtype fp32] [--iters 20] [--points 300000] [--modes tiles,uniform,balanced]
       [--rpc 0,1,2]   (ranges per CU override; 0 = library default)"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from futuredet_amd import build_backbone, hip_ops  # noqa: E402
from futuredet_amd.synth import seeded_state_dict, synthetic_cloud  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--levels", default="0,1,2,3")
ap.add_argument("--dtype", default="fp32")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--points", type=int, default=300000)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--modes", default="tiles,uniform,balanced")
ap.add_argument("--rpc", default="0")
ap.add_argument("--tm", type=int, default=0, help="chunk rows override (64 | 128)")
ap.add_argument("--depth", type=int, default=0)
ap.add_argument("--ldspad", type=int, default=0, help="extra LDS bytes per workgroup (occupancy experiments; 1 = drop the 64->64 floor)")
ap.add_argument("--gp", default="0", help="bf16: comma list of bf16_gp knob values (0 = gather-pipeline kernel, -1 = the older kernels)")
ap.add_argument("--rg", default="0", help="bf16 gather pipeline: comma list of row groups per wave (0 = heuristic)")
ap.add_argument("--exp", type=int, default=0, help="bf16: experiment knob (timing only, results wrong)")
ap.add_argument("--bdepth", default="0", help="bf16 gather pipeline: comma list of ring depths (0 = heuristic)")
ap.add_argument("--win", default="0", help="bf16: comma list of bf16_win knob values (0 = LDS-window kernel where it applies, -1 = RING / RESIDENT kernels)")
args = ap.parse_args()
dev = torch.device("cuda")
dt = torch.float32 if args.dtype == "fp32" else torch.bfloat16
pts = torch.from_numpy(synthetic_cloud(args.seed, args.points)).to(dev)
out = hip_ops.voxelize(pts, [0.075, 0.075, 0.2], [-54, -54, -5.0, 54, 54, 3.0], 10, 160000, want_voxels=False, want_mean=True,
                       mean_stride=16, coor_cols=4)
m = int(out["num_voxels"].cpu()[0])
bb = build_backbone(dict(type="SpMiddleResNetFHD", num_input_features=5, ds_factor=8))
bb.load_state_dict(seeded_state_dict(bb, 7), strict=False)
bb = bb.to(dev).eval()
idx = bb.build_indexes(lambda i0: i0.mark(out["coors"][:m].contiguous()), 1, [1440, 1440, 40], dev)
MODE = {"tiles": "tiles", "uniform": False, "balanced": True}
hip_ops.set_tuning("v2_tm", args.tm)
hip_ops.set_tuning("v2_depth", args.depth)
hip_ops.set_tuning("v2_ldspad", args.ldspad)
hip_ops.set_tuning("bf16_nw", args.exp)
for lvl in [int(v) for v in args.levels.split(",")]:
    C = [16, 32, 64, 128][lvl]
    ix = idx[lvl]
    x = torch.randn((ix.n, C), device=dev).to(dt)
    w = torch.randn((27, C, C)) * (2.0 / (27 * C)) ** 0.5
    wpk = hip_ops.pack_spconv_weight(w, dt).to(dev)
    bias = torch.zeros(C, device=dev)
    ref = None
    variants = [(rpc, mode, 0, 0, 0) for rpc in [int(v) for v in args.rpc.split(",")] for mode in args.modes.split(",")]
    if args.dtype != "fp32":
        variants = [(0, "uniform", gp + 10 * wn, rg, bd) for wn in [int(v) for v in args.win.split(",")] for gp in [int(v) for v in args.gp.split(",")]
                    for rg in [int(v) for v in args.rg.split(",")]
                    for bd in [int(v) for v in args.bdepth.split(",")] if gp == 0 or (rg == int(args.rg.split(",")[0]) and bd == int(args.bdepth.split(",")[0]))]
    for rpc, mode, gp, rg, bd in variants:
        wn = -1 if gp <= -5 else 0   # (gp carries the bf16_win knob in its tens: 0 / -10)
        gp = gp - 10 * wn
        hip_ops.set_tuning("bf16_win", wn)
        hip_ops.set_tuning("v2_ranges_per_cu", rpc)
        hip_ops.set_tuning("bf16_gp", gp)
        hip_ops.set_tuning("bf16_rg", rg)
        hip_ops.set_tuning("bf16_depth", bd)
        if True:
            if mode == "tiles" and rpc != int(args.rpc.split(",")[0]):
                continue
            nbr = ix.rulebook(ix, [3, 3, 3], [1, 1, 1], [1, 1, 1])  # fresh tensor: the range table is cached on it
            pairs = int((nbr[:, :ix.n] >= 0).sum())
            for _ in range(3):
                y = hip_ops.spconv_apply(x, wpk, bias, nbr, ix.n, C, residual=x, relu=True, balanced=MODE[mode])
            torch.cuda.synchronize()
            if ref is None:
                ref = {}
            key = (gp, wn)
            if key not in ref:
                ref[key] = y.clone()
                for other in ([] if args.exp else ref.values()):  # two bf16 kernel families: same data, different summation trees
                    assert float((other.float() - y.float()).abs().max()) <= 2e-2 * float(other.float().abs().max()), "bf16 kernels disagree"
            assert torch.equal(ref[key], y), "work distribution changed the result"
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                y = hip_ops.spconv_apply(x, wpk, bias, nbr, ix.n, C, residual=x, relu=True, balanced=MODE[mode])
            e1.record()
            torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / args.iters
            s = 4 if dt == torch.float32 else 2
            bgs = s * pairs * 2 * C + 8 * pairs + s * 27 * C * C
            tag = "%-8s rpc=%d" % (mode, rpc) if args.dtype == "fp32" else "win=%d gp=%d rg=%d depth=%d" % (wn, gp, rg, bd)
            print("level %d C=%3d n=%6d pairs=%7d (%.1f/row) %s %s: %7.1f us  %6.1f TFLOP/s  B_gs %.0f GB/s" %
                  (lvl, C, ix.n, pairs, pairs / ix.n, args.dtype, tag, us, 2.0 * pairs * C * C / us / 1e6, bgs / us / 1e3), flush=True)
    hip_ops.set_tuning("v2_ranges_per_cu", 0)
    hip_ops.set_tuning("bf16_win", 0)
