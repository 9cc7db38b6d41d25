import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import test_gpu_parity as T
from parity_util import match_rows
from futuredet_amd.synth import synthetic_cloud
cfg, net, onet = T._build_pair("forecast_n3", "pedestrian", voxel_size=(0.05, 0.05, 0.2), max_voxel_num=(300000, 400000))
cloud = synthetic_cloud(seed=0, target_points=500000)
v, c, n, obb, obev, want = T._oracle_run(cfg, onet, cloud)
for hipconv in (True, False):
    net.neck.use_hip_conv = net.bbox_head.use_hip_conv = hipconv
    net.invalidate_caches()
    with torch.no_grad():
        got = net.forward_points([T._dev(cloud)], cfg.voxel_generator, padded=False)[0]
    g, w = T._rows(got), T._rows(want)
    ug, uw = match_rows(g, w)
    print("hip conv", hipconv, "unmatched", len(ug), len(uw))
    for i in ug[:8]:
        d = np.abs(w[:, :2] - g[i, :2]).sum(1); j = int(d.argmin())
        rel = np.abs(g[i] - w[j]) / np.maximum(1, np.abs(w[j]))
        print("  got", np.round(g[i], 4).tolist()); print("  ref", np.round(w[j], 4).tolist()); print("  worst comp", int(rel.argmax()), float(rel.max()))
