"""Instruction-class self-checks (tools/probes/alu_probe.hip) next to disturbers: stream A runs a probe kernel (every thread evaluates a
function twice from the same registers and counts disagreements) while streams B.. loop over the bf16 neck plan / fp32 neck plan / nothing.
    python tools/soak_alu.py [rounds] [disturber streams]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, ".")
import futuredet_amd.detectors as D  # noqa: E402
from futuredet_amd import build_detector  # noqa: E402
from futuredet_amd.configs import centerpoint_config  # noqa: E402
from futuredet_amd.synth import seeded_state_dict, synthetic_cloud, tame_box_dims  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 3
P = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes", "libalu_probe.so"))
P.alu_probe_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
cfg = centerpoint_config("forecast_n3")
net = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
net.load_state_dict(tame_box_dims(seeded_state_dict(net, 7)), strict=False)
net = net.cuda().eval()
sa = torch.cuda.Stream()
sbs = [torch.cuda.Stream() for _ in range(NB)]
names = ["fma chains", "packed fp32", "divisions", "compares + divergent branches", "LDS array, run-time indices", "atan2f", "v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32",
         "v_mul / v_add / v_sub", "pk_add + s_nop 0", "pk_add + s_nop 3", "pk_add + 1 VALU", "pk_add + 2 VALU", "pk_add plain", "pk_add distance 2", "pk_mul plain", "pk_mul neg", "pk_mul op_sel_hi:[1,0]", "pk_mul op_sel:[0,1] op_sel_hi:[1,0]",
         "pk_mul op_sel:[1,0] op_sel_hi:[0,1]", "pk_add op_sel:[0,1] op_sel_hi:[0,1]"]
dists = sys.argv[4].split(",") if len(sys.argv) > 4 else None
only = [int(m) for m in sys.argv[3].split(",")] if len(sys.argv) > 3 else list(range(len(names)))
with torch.no_grad():
    clouds = [torch.from_numpy(synthetic_cloud(seed=b, target_points=300000)).cuda() for b in range(2)]
    dist = {"nothing": None}
    for dt in (torch.bfloat16, torch.float32):
        net.set_precision(dt)
        stage = {}
        net.__dict__["debug_taps"] = stage
        D._NO_GRAPH = True
        net.forward_points(clouds, cfg.voxel_generator, padded="packed")
        net.__dict__["debug_taps"] = None
        torch.cuda.synchronize()
        bev = stage["bev"]
        gs = []
        for sb in sbs:
            with torch.cuda.stream(sb):
                net.neck(bev)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=sb):
                    keep = net.neck(bev)
            gs.append((g, keep, bev, net.neck._plan))  # (the plan owns the packed weights the graph reads: it must outlive the dtype switch below)
        dist["%s neck plan" % str(dt).split(".")[-1]] = gs
    torch.cuda.synchronize()
    mism = torch.zeros(1, dtype=torch.int64, device="cuda")
    sink = torch.zeros(4, device="cuda")
    for dname, gs in dist.items():
        if dists and not any(d in dname for d in dists):
            continue
        line = "%-22s" % dname
        for mode, nm in enumerate(names):
            if mode not in only:
                continue
            mism.zero_()
            torch.cuda.synchronize()
            for r in range(rounds):
                if gs is not None:
                    for sb, (g, _, _, _) in zip(sbs, gs):
                        with torch.cuda.stream(sb):
                            for _ in range(4):
                                g.replay()
                with torch.cuda.stream(sa):
                    for _ in range(6):  # ~16k threads x 24 double evaluations per launch, small workgroups coming and going like nms_mask's
                        P.alu_probe_run(mode, 128, 24, mism.data_ptr(), sink.data_ptr(), sa.cuda_stream)
                torch.cuda.synchronize()
            line += " | %s: %d" % (nm, int(mism.item()))
        print(line, flush=True)
