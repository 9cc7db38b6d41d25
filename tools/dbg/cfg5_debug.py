import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import test_gpu_parity as T
from parity_util import match_rows, nms_layout
from futuredet_amd.synth import synthetic_cloud
from oracle import ops as oops
cfg, net, onet = T._build_pair("forecast_n3", "pedestrian", voxel_size=(0.05, 0.05, 0.2), max_voxel_num=(300000, 400000))
cloud = synthetic_cloud(seed=0, target_points=500000)
v, c, n, obb, obev, want = T._oracle_run(cfg, onet, cloud)
print("topk cut", want["topk_cut"])
with torch.no_grad():
    got = net.forward_points([T._dev(cloud)], cfg.voxel_generator, padded=False)[0]
    bb, x = T._hip_maps(net, cfg, v, c, n)
    hp = net.bbox_head(x, None)
    op = onet.bbox_head(obev, None)
    for k in op[0]:
        d = (hp[0][k].float().cpu() - op[0][k]).abs()
        print("head", k, "max abs err %.3e" % float(d.max()), "max |ref| %.3e" % float(op[0][k].abs().max()))
    # the oracle's predict on OUR head outputs
    mine = onet.bbox_head.predict({"metadata": [None]}, [{k: t.float().cpu() for k, t in hp[0].items()}], cfg.test_cfg)[0]
g, w, m = T._rows(got), T._rows(want), T._rows(mine)
print("got vs oracle-predict-on-our-head-outputs:", [len(a) for a in match_rows(g, m)], " oracle vs that:", [len(a) for a in match_rows(w, m)])
ug, uw = match_rows(g, w)
wrow = w[uw[0]]
# find our version of the want box among our raw candidates: decode near its cell
print("want row", np.round(wrow[[0, 1, 3, 4, 5, 8, 9]], 4).tolist())
from futuredet_amd import hip_ops
lab0 = g[g[:, 10] == 0]
wb = nms_layout(wrow[None, :9])
kb = nms_layout(lab0[:, :9])
host = oops.boxes_iou_bev(wb, kb)[0]
dev = hip_ops.boxes_iou_bev(torch.from_numpy(wb).cuda(), torch.from_numpy(kb).cuda()).cpu().numpy()[0]
o = np.argsort(-np.maximum(host, dev))[:6]
print("host IoU", np.round(host[o], 5).tolist()); print("dev  IoU", np.round(dev[o], 5).tolist())
print("boxes", np.round(kb[o], 3).tolist()); print("W'", np.round(wb, 3).tolist())
allh = oops.boxes_iou_bev(kb, kb); alld = hip_ops.boxes_iou_bev(torch.from_numpy(kb).cuda(), torch.from_numpy(kb).cuda()).cpu().numpy()
print("max |host-dev| IoU over kept x kept", float(np.abs(allh - alld).max()))
