import sys, numpy as np, torch
sys.path.insert(0, ".")
from futuredet_amd import build_detector
from futuredet_amd.configs import pointpillars_config
from futuredet_amd.synth import seeded_state_dict, synthetic_cloud
from oracle import model as omodel, ops as oops
cfg = pointpillars_config()
net = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
sd = seeded_state_dict(net, 9); net.load_state_dict(sd, strict=False); net = net.cuda().eval()
onet = omodel.PointPillars(cfg.model["reader"], cfg.model["backbone"], cfg.model["neck"], cfg.model["bbox_head"], test_cfg=cfg.test_cfg).eval()
onet.load_state_dict(sd, strict=False)
pts = synthetic_cloud(seed=2, target_points=30000)
v, c, n = oops.points_to_voxel(pts, cfg.voxel_generator["voxel_size"], cfg.voxel_generator["range"], 20, True, 60000)
c4 = np.concatenate([np.zeros((len(c), 1), np.int32), c], 1)
tv, tc, tn = torch.from_numpy(v), torch.from_numpy(c4), torch.from_numpy(n)
with torch.no_grad():
    of = onet.reader(tv, tn, tc); f = net.reader(tv.cuda(), tn.cuda(), tc.cuda())
    print("reader", float((f.cpu() - of).abs().max()), float(of.abs().max()))
    ocan = omodel.pillars_scatter(of, tc, 1, [512, 512, 1]); can = net.backbone(f, tc.cuda(), 1, [512, 512, 1])
    print("canvas", float((can.cpu() - ocan).abs().max()))
    ox = onet.neck(ocan); x = net.neck(can)
    print("neck", float((x.cpu() - ox).abs().max()), float(ox.abs().max()))
    x2 = net.neck.forward_modules(can)
    print("neck modules", float((x2.cpu() - ox).abs().max()))
    op = onet.bbox_head(ox); p = net.bbox_head(x)
    p2 = net.bbox_head.forward_modules(x2)
    for t in (0, 1, 6):
        for k in op[t]:
            print(t, k, float((p[t][k].float().cpu() - op[t][k]).abs().max()), float((p2[t][k].float().cpu() - op[t][k]).abs().max()), float(op[t][k].abs().max()))
