import faulthandler
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
faulthandler.dump_traceback_later(45, exit=True)
from futuredet_amd import hip_ops  # noqa: E402
from futuredet_amd.synth import synthetic_cloud  # noqa: E402

which = sys.argv[1]
pts = torch.from_numpy(synthetic_cloud(seed=3, target_points=40000)).cuda()
cap = torch.cuda.Stream()


def body():
    out = hip_ops.voxelize(pts, [0.075, 0.075, 0.2], [-54, -54, -5.0, 54, 54, 3.0], 10, 120000, want_voxels=False, want_mean=True,
                           mean_stride=16, coor_cols=4)
    if which == "vox":
        return out["num_voxels"]
    geoms = [((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (0, 1, 1)), ((3, 1, 1), (2, 1, 1), (0, 0, 0))]
    idx = hip_ops.build_pyramid(out["coors"], out["num_voxels"], 120000, 1, (41, 1440, 1440), geoms, pts.device, static=True)
    return torch.cat([ix.n_dev for ix in idx])


with torch.cuda.stream(cap):
    r = body()
torch.cuda.synchronize()
print("eager", r.cpu().tolist(), flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=cap):
    r = body()
for i in range(3):
    g.replay()
    torch.cuda.synchronize()
    print("replay", i, r.cpu().tolist(), flush=True)
