import cProfile, pstats, sys, torch
sys.path.insert(0, ".")
from futuredet_amd import build_detector
from futuredet_amd.configs import centerpoint_config
from futuredet_amd.synth import seeded_state_dict, synthetic_cloud
cfg = centerpoint_config("forecast_n0")
net = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
net.load_state_dict(seeded_state_dict(net, 7), strict=False)
net = net.cuda().eval(); net.set_precision(torch.float32)
cloud = [torch.from_numpy(synthetic_cloud(seed=0, target_points=300000)).cuda()]
for _ in range(5):
    net.forward_points(cloud, cfg.voxel_generator)[0].cpu()
pr = cProfile.Profile(); pr.enable()
for _ in range(30):
    net.forward_points(cloud, cfg.voxel_generator)[0].cpu()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(45)
