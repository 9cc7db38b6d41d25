"""Debug: which side is off on the RPN bf16 layers (oracle fp32 CPU conv vs device bf16 kernel vs float64)."""
import os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from futuredet_amd import build_detector
from futuredet_amd.configs import centerpoint_config
from futuredet_amd.dense_bf16 import RPNPlan
from futuredet_amd.synth import seeded_state_dict, synthetic_cloud, tame_box_dims
from oracle import model as omodel, ops as oops, bf16 as obf
import torch.nn.functional as F

cfg = centerpoint_config("forecast_n3")
net = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
sd = tame_box_dims(seeded_state_dict(net, 7)); net.load_state_dict(sd, strict=False)
net = net.cuda().eval(); net.set_precision(torch.bfloat16)
onet = omodel.VoxelNet(cfg.model["reader"], cfg.model["backbone"], cfg.model["neck"], cfg.model["bbox_head"], test_cfg=cfg.test_cfg).eval()
onet.load_state_dict(sd, strict=False)
cloud = synthetic_cloud(seed=0, target_points=300000)
vg = cfg.voxel_generator
v, c, n = oops.points_to_voxel(cloud, vg["voxel_size"], vg["range"], vg["max_points_in_voxel"], True, vg["max_voxel_num"][1])
grid = np.round((np.array(vg["range"][3:], np.float32) - np.array(vg["range"][:3], np.float32)) / np.array(vg["voxel_size"], np.float32)).astype(np.int64)
ex = dict(voxels=torch.from_numpy(v), coordinates=torch.from_numpy(np.pad(c, ((0, 0), (1, 0)))), num_points=torch.from_numpy(n), num_voxels=torch.tensor([len(n)]), shape=np.array([grid]), metadata=[None])
with obf.tracing() as tr:
    obf.run(onet, ex, cfg.test_cfg)
dn = [r for r in tr if r["kind"] == "dense"]
rp = RPNPlan(net.neck, torch.bfloat16)
k = 0
for i, stack in enumerate(rp.blocks):
    for j, conv in enumerate(stack):
        rec = dn[k]; k += 1
        x = rec["x"]
        y_dev = conv(x.cuda().to(torch.bfloat16).permute(0, 2, 3, 1).contiguous()).float().permute(0, 3, 1, 2).cpu()
        # float64 exact from the folded bf16 weights of the oracle
        m = rec["conv"]
        mods = list(onet.neck.blocks[i]._modules.values())
        mi = [q for q, mm in enumerate(mods) if mm is m][0]
        bn = mods[mi + 1]
        w, b = obf._fold_dense(m, bn)
        pad = 1
        y64 = F.conv2d(x.double(), w.double(), None, stride=m.stride, padding=pad) + b.double().view(1, -1, 1, 1)
        y64r = torch.relu(y64).float().to(torch.bfloat16).float()
        e_dev = float(((y_dev - y64r).abs() / y64r.abs().clamp(min=1)).max())
        e_ora = float(((rec["y"] - y64r).abs() / y64r.abs().clamp(min=1)).max())
        print("block %d conv %d  %s stride %s: device vs f64-rounded %.4f   oracle(fp32 cpu) vs f64-rounded %.4f   max|y| %.1f max|x| %.1f" % (i, j, tuple(w.shape), m.stride, e_dev, e_ora, float(y64r.abs().max()), float(x.abs().max())))
        if e_dev > 0.01:
            d = ((y_dev - y64r).abs() / y64r.abs().clamp(min=1))
            top = torch.topk(d.flatten(), 12)
            for val, ix in zip(top.values.tolist(), top.indices.tolist()):
                co, rem = divmod(ix, d.shape[2] * d.shape[3]); yy, xx = divmod(rem, d.shape[3])
                print("   err %.4f at co %d y %d x %d: dev %.5f f64 %.7f oracle %.5f" % (val, co, yy, xx, float(y_dev[0, co, yy, xx]), float(y64[0, co, yy, xx]), float(rec["y"][0, co, yy, xx])))
            print("   elements > 8e-3:", int((d > 8e-3).sum()), "of", d.numel())
    if i - rp.start >= 0:
        k += 1
