"""Debug: per-level distance between the HIP bf16 backbone and oracle/bf16.py on one small cloud."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from futuredet_amd import build_detector
from futuredet_amd.configs import centerpoint_config
from futuredet_amd.synth import seeded_state_dict, synthetic_cloud, tame_box_dims
from oracle import model as omodel, ops as oops, bf16 as obf

cfg = centerpoint_config("forecast_n3")
net = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
sd = tame_box_dims(seeded_state_dict(net, 7))
net.load_state_dict(sd, strict=False)
net = net.cuda().eval()
net.set_precision(torch.bfloat16)
onet = omodel.VoxelNet(cfg.model["reader"], cfg.model["backbone"], cfg.model["neck"], cfg.model["bbox_head"], test_cfg=cfg.test_cfg).eval()
onet.load_state_dict(sd, strict=False)
cloud = synthetic_cloud(seed=0, target_points=int(sys.argv[1]) if len(sys.argv) > 1 else 30000)
vg = cfg.voxel_generator
v, c, n = oops.points_to_voxel(cloud, vg["voxel_size"], vg["range"], vg["max_points_in_voxel"], True, vg["max_voxel_num"][1])
grid = np.round((np.array(vg["range"][3:], np.float32) - np.array(vg["range"][:3], np.float32)) / np.array(vg["voxel_size"], np.float32)).astype(np.int64)
coors = torch.from_numpy(np.pad(c, ((0, 0), (1, 0))))
lv = {}
with torch.no_grad():
    feats = onet.reader(torch.from_numpy(v), torch.from_numpy(n))
    obb = obf.backbone(onet.backbone, feats, coors, 1, grid, levels=lv)
    hf = net.reader(torch.from_numpy(v).cuda().float(), torch.from_numpy(n).cuda())
    print("reader diff", float((hf.cpu() - feats).abs().max()))
    hbb, multi = net.backbone(hf, coors.cuda(), 1, [int(g) for g in grid])
for name in ("conv1", "conv2", "conv3", "conv4"):
    o = lv[name]
    oc = o.indices.numpy().astype(np.int64)
    of = o.features.numpy()
    h = multi[name]
    hc = h.indices.cpu().numpy().astype(np.int64)
    hfe = h.features.float().cpu().numpy()[:, : of.shape[1]]
    key = lambda q: ((q[:, 0] * 64 + q[:, 1]) * 4096 + q[:, 2]) * 4096 + q[:, 3]
    so, sh = np.argsort(key(oc)), np.argsort(key(hc))
    assert np.array_equal(oc[so], hc[sh]), name
    d = np.abs(hfe[sh] - of[so]) / np.maximum(1.0, np.abs(of[so]))
    print(name, "rows", len(oc), "max rel err %.4f" % d.max(), "mean %.2e" % d.mean(), "frac > 1e-2: %.2e" % (d > 1e-2).mean(), "max |ref| %.1f" % np.abs(of).max())
d = np.abs(hbb.float().cpu().numpy() - obb.numpy()) / np.maximum(1.0, np.abs(obb.numpy()))
print("bev max rel err %.4f frac>1e-2 %.2e max|ref| %.1f" % (d.max(), (d > 1e-2).mean(), np.abs(obb.numpy()).max()))
