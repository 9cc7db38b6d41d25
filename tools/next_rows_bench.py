"""Timings of the SURVEY 8(f) kernels on one MI355X (HIP events on the launch stream): sweep assembly, PointPillars
reader + scatter, forecast association.  Prints achieved GB/s against the algorithmic bytes where the kernel is HBM-bound."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from futuredet_amd import hip_ops  # noqa: E402
from futuredet_amd.synth import synthetic_cloud  # noqa: E402


def timed(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters  # us


def main():
    dev = "cuda"
    cloud = synthetic_cloud(seed=0, target_points=300000)
    R = len(cloud)
    # ---- sweep assembly: 10 sweeps, transforms on 9 of them
    rng = np.random.default_rng(0)
    raw = torch.from_numpy(np.concatenate([cloud[:, :4], np.zeros((R, 1), np.float32)], 1)).to(dev)
    rows = np.linspace(0, R, 11).astype(np.int64)
    mats = [None] + [np.eye(4) + rng.normal(0, 1e-3, (4, 4)) for _ in range(9)]
    desc = hip_ops.sweep_descriptors(rows, mats, [0.05 * s for s in range(10)], [s > 0 for s in range(10)])
    us = timed(lambda: hip_ops.assemble_sweeps(raw, desc))
    print("sweep assembly: %d rows  %.1f us  (%.0f GB/s of 2x20 B read + 20 B write per row; includes the descriptor upload)"
          % (R, us, 60.0 * R / us / 1e3))
    # ---- PointPillars reader + scatter on the pp grid
    vs, rg = [0.2, 0.2, 8.0], [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]
    pts = torch.from_numpy(cloud).to(dev)
    vox = hip_ops.voxelize(pts, vs, rg, 20, 60000, want_voxels=True, coor_cols=4)
    M = int(vox["num_voxels"].item())
    layers = [(torch.randn(32, 10, device=dev) * 0.1, torch.ones(32, device=dev), torch.zeros(32, device=dev)),
              (torch.randn(64, 64, device=dev) * 0.1, torch.ones(64, device=dev), torch.zeros(64, device=dev))]
    geom = (0.2, 0.2, 0.1 - 51.2, 0.1 - 51.2)
    us_v = timed(lambda: hip_ops.voxelize(pts, vs, rg, 20, 60000, want_voxels=True, coor_cols=4))
    us_e = timed(lambda: hip_ops.pillar_encode(vox["voxels"], vox["num_points"], vox["coors"], vox["num_voxels"], geom, layers))
    f = hip_ops.pillar_encode(vox["voxels"], vox["num_points"], vox["coors"], vox["num_voxels"], geom, layers)
    us_s = timed(lambda: hip_ops.pillar_scatter(f, vox["coors"], vox["num_voxels"], 1, 512, 512))
    print("pillars: %d pillars  voxelize(20 slots) %.1f us, reader %.1f us (%.0f GB/s of 400+256 B per pillar; %.1f GFLOP/s), "
          "scatter %.1f us (%.0f GB/s of the 64x512x512 canvas)" % (M, us_v, us_e, 656.0 * M / us_e / 1e3,
                                                                       2.0 * 20 * (10 * 32 + 32 * 64 + 32 * 64 / 20) * M / us_e / 1e3, us_s,
                                                                       64 * 512 * 512 * 4 / us_s / 1e3))
    # ---- forecast association: 7 steps x 83 boxes
    T, n = 7, 83
    c = torch.from_numpy(rng.uniform(-50, 50, (T, n, 3))).to(dev)
    v = torch.from_numpy(rng.normal(0, 3, (T, n, 3))).to(dev)
    cnt = torch.full((T,), n, dtype=torch.int32, device=dev)
    tm = torch.full((T - 1,), 0.5, dtype=torch.float64, device=dev)
    us = timed(lambda: hip_ops.forecast_chains(c, v, cnt, tm, 2.0))
    print("forecast association: T=%d x %d boxes  %.1f us per sweep (one workgroup; includes 7 output allocations)" % (T, n, us))


if __name__ == "__main__":
    t = time.time()
    main()
    print("done in %.1f s" % (time.time() - t))
