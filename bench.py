#!/usr/bin/env python
"""Benchmark of the FutureDet LiDAR hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic 10-sweep clouds already resident in HBM:
voxelize(+mean) -> sparse indexes/rulebooks -> 21 sparse convs -> densify -> RPN -> CenterHead -> decode + rotated
NMS -> detections copied to the host.  Workload at N=1: BASELINE.json configs[1] (forecast_n0 cars, one 300k-point
cloud, fp32).  Samples are independent, so ranks shard them with no data-path collective (weak scaling: every
rank processes --batch clouds per step); one fixed-shape all_gather of the detections closes the timed region.
Rank 0 prints ONE JSON line (metric, value, roofline of the dominant kernel = sparse conv apply, cpu_baseline).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--variant", default="forecast_n0", choices=["forecast_n0", "forecast_n3", "forecast_n3dtf", "pp_n3dtf"])
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--points", type=int, default=300000)
    ap.add_argument("--batch", type=int, default=1, help="clouds per rank per step")
    ap.add_argument("--channels-last", type=int, default=-1)
    ap.add_argument("--voxel-xy", type=float, default=0.075, help="x/y voxel size (0.05 = the finer grid of BASELINE configs[4])")
    ap.add_argument("--max-voxels", type=int, default=160000)
    ap.add_argument("--class-name", default="car")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stage-times", action="store_true", help="print per-stage GPU times to stderr")
    return ap.parse_args()


PROF_EVERY = 4


class SpconvProfiler(object):
    """backbone.profile_hook: brackets every fd_spconv_apply launch with events on the launch stream."""

    def __init__(self):
        self.records = []  # (tag, info, ev0, ev1)
        self.enabled = False

    def __call__(self, tag, info, fn):
        if not self.enabled:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        self.records.append((tag, info, e0, e1))
        return out


def algorithmic_bytes(info, pairs):
    # B_gs = s*P*(Cin+Cout) + 8*P + s*K*Cin*Cout
    s, K, cin, cout = info["s"], info["K"], info["cin"], info["cout"]
    return s * pairs * (cin + cout) + 8 * pairs + s * K * cin * cout


def cpu_baseline(cfg, sd, cloud):
    """The CPU oracle (our parity-checked restatement of the reference path: C/OpenMP voxelizer + spconv-1.0
    pair-list sparse conv, torch-CPU dense convs, reference decode + rotated NMS) on ONE cloud of the same
    workload, on the host cores of this box."""
    from oracle import model as omodel
    from oracle import ops as oops

    cores = min(os.cpu_count() or 1, 64)  # beyond ~64 threads the pair-list loops stop scaling (fork/join per tap)
    torch.set_num_threads(cores)
    oops.set_threads(cores)
    onet = omodel.VoxelNet(cfg.model["reader"], cfg.model["backbone"], cfg.model["neck"], cfg.model["bbox_head"],
                           test_cfg=cfg.test_cfg).eval()
    onet.load_state_dict(sd, strict=False)
    vg = cfg.voxel_generator
    grid = np.round((np.array(vg["range"][3:], np.float32) - np.array(vg["range"][:3], np.float32)) / np.array(vg["voxel_size"], np.float32))
    t_start = time.perf_counter()
    t_vox, n_done, n_vox, n_det = 0.0, 0, 0, 0
    # bounded sample: the bench cloud over and over for ~10 s of CPU work (at least 2, at most 8 sweeps)
    while n_done < 2 or (time.perf_counter() - t_start < 10.0 and n_done < 8):
        t0 = time.perf_counter()
        v, c, n = oops.points_to_voxel(cloud, vg["voxel_size"], vg["range"], vg["max_points_in_voxel"], True, vg["max_voxel_num"][1])
        t_vox += time.perf_counter() - t0
        ex = dict(voxels=torch.from_numpy(v), coordinates=torch.from_numpy(np.pad(c, ((0, 0), (1, 0)))), num_points=torch.from_numpy(n),
                  num_voxels=torch.tensor([len(n)]), shape=np.array([grid.astype(np.int64)]), metadata=[None])
        res = onet(ex)
        n_done += 1
        n_vox, n_det = len(n), len(res[0]["scores"])
    dt = time.perf_counter() - t_start
    return {"value": round(n_done / dt, 4), "unit": "sweeps/s", "cores": cores, "kind": "port",
            "sample": "%d passes over the bench cloud (%d pts, %d voxels, %d detections) through oracle/ in %.1f s "
                      "(voxelizer %.2f s/sweep single-thread; pair-list sparse conv on OpenMP, dense convs on torch-CPU, %d threads)"
                      % (n_done, len(cloud), n_vox, n_det, dt, t_vox / n_done, cores)}


def main():
    args = parse()
    from futuredet_amd import build as fbuild
    from futuredet_amd import build_detector, dist_infer, lib
    from futuredet_amd.configs import centerpoint_config
    from futuredet_amd.synth import seeded_state_dict, synthetic_cloud

    # FD_BENCH_ONE_DEVICE=1 (test hook for 1-GPU boxes): every rank uses cuda:0 and the collectives run over gloo, so the
    # world > 1 logic (sharded seeds, barriers, MAX over ranks, result gather, rank-0 print) can be exercised anywhere
    one_dev = bool(os.environ.get("FD_BENCH_ONE_DEVICE"))
    rank, world, local = dist_infer.init_from_env("gloo" if one_dev else "nccl")
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    if one_dev:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if rank == 0:
        fbuild.build()
    if world > 1:
        torch.distributed.barrier()
    lib.load()

    if args.variant == "pp_n3dtf":  # secondary line: the PointPillars configs; no sparse conv, no roofline object
        from futuredet_amd.configs import pointpillars_config
        cfg = pointpillars_config(args.class_name)
    else:
        cfg = centerpoint_config(args.variant, args.class_name, voxel_size=(args.voxel_xy, args.voxel_xy, 0.2),
                                 max_voxel_num=(min(120000, args.max_voxels), args.max_voxels))
    net = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    sd = seeded_state_dict(net, 7)
    net.load_state_dict(sd, strict=False)
    net = net.to(dev).eval()
    dtype = torch.float32 if args.dtype == "fp32" else torch.bfloat16
    net.set_precision(dtype, None if args.channels_last < 0 else bool(args.channels_last))
    prof = SpconvProfiler()
    net.backbone.profile_hook = prof
    is_pp = args.variant == "pp_n3dtf"

    # inputs resident in HBM before the timed region; every rank owns its own clouds (seeds by global sample id)
    host_clouds = [synthetic_cloud(seed=rank * args.batch + i, target_points=args.points) for i in range(args.batch)]
    clouds = [torch.from_numpy(c).to(dev) for c in host_clouds]
    bev = None
    if net.bbox_head.bev_map:
        side = int(round(108.0 / args.voxel_xy / 8))
        bev = torch.zeros((args.batch, 6, side, side), device=dev)

    stage_events = []

    def stage_hook(name):
        if prof.enabled and args.stage_times:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            stage_events.append((name, e))

    net.stage_hook = stage_hook

    def step():
        boxes, scores, labels, counts = net.forward_points(clouds, cfg.voxel_generator, bev_map=bev, padded=True)
        packed, cnt = dist_infer.pack_results(boxes, scores, labels, counts)
        return packed, cnt

    def sync_all():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            p, c = step()
            p.cpu()
        sync_all()
        t0 = time.perf_counter()
        for si in range(args.steps):
            # the per-launch HIP events of the roofline measurement are taken on every PROF_EVERY-th step of the timed
            # region: an event pair costs ~5 us of queue time per launch (21 launches per step), which would otherwise
            # be charged to every step of the headline number
            prof.enabled = (si % PROF_EVERY == 0)
            p, c = step()
            host_p, host_c = p.cpu(), c.cpu()  # detections on the host = end of a sweep
        if world > 1:
            full, fullc = dist_infer.gather_results(p, c)
        sync_all()
        dt = time.perf_counter() - t0
        prof.enabled = False
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    dt = float(t.item())

    sweeps = args.steps * args.batch * world
    out = {
        "metric": "sweeps/sec end-to-end (300k pts, 10-sweep voxel)", "value": round(sweeps / dt, 3), "unit": "sweeps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.dtype == "fp32" else "bf16",
        "data": "synthetic",
        "config": {"workload": "%s cars, %d-pt synthetic 10-sweep cloud x%d per GPU, %s+RPN+CenterHead, %s"
                               % (args.variant, len(host_clouds[0]), args.batch,
                                  "PointPillars(PillarFeatureNet+Scatter)" if is_pp else "VoxelNet+SpMiddleResNetFHD", args.dtype),
                   "parallelism": "sample-sharded x%d (no data-path collective)" % world, "detections_per_sweep": int(host_c[0].sum())},
    }
    if rank == 0 and is_pp:
        out["roofline"] = None
        if args.stage_times:
            st = {}
            for (n0, e0), (n1, e1) in zip(stage_events[:-1], stage_events[1:]):
                if n1 != "start":
                    st[n1] = st.get(n1, 0.0) + e0.elapsed_time(e1)
            n_prof = (args.steps + PROF_EVERY - 1) // PROF_EVERY
            print("[stage] GPU ms/step: " + ", ".join("%s=%.3f" % (k, v / n_prof) for k, v in st.items()), file=sys.stderr)
        print(json.dumps(out))
    elif rank == 0:
        # ---- roofline of the dominant kernel (sparse conv apply), from the events recorded in the timed region
        with torch.no_grad():
            ms = [(tag, info, e0.elapsed_time(e1)) for tag, info, e0, e1 in prof.records]
            n_prof = (args.steps + PROF_EVERY - 1) // PROF_EVERY  # instrumented steps
            per_step = len(ms) // max(n_prof, 1)
            pair_counts = [info["pairs"]() for _, info, _ in ms[:per_step]]
        tot_ms = sum(m for _, _, m in ms)
        tot_bytes = sum(algorithmic_bytes(info, pair_counts[i % per_step]) for i, (_, info, _) in enumerate(ms))
        tot_flops = sum(2.0 * pair_counts[i % per_step] * info["cin"] * info["cout"] for i, (_, info, _) in enumerate(ms))
        launches = len(ms)
        ach = tot_bytes / (tot_ms * 1e-3) / 1e9 if tot_ms > 0 else 0.0
        # HBM bytes per launch from the PMC passes (FETCH_SIZE / WRITE_SIZE collected in separate rocprofv3 --pmc runs
        # of this same command, gfx950 x2 read correction applied; see profiles/spconv_traffic.json) -- a profiler
        # cannot run inside the timed process, so the figure is looked up for the matching workload, else null
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "spconv_traffic.json")))
            traffic = tj.get("%s/%s/%d/b%d" % (args.variant, args.dtype, args.points, args.batch), {}).get("hbm_bytes_per_launch")
        except Exception:
            pass
        out["roofline"] = {"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                           "kernel": "spconv_f32/spconv_bf16 (fd_spconv_apply)", "launches_per_step": per_step,
                           "avg_launch_us": round(1e3 * tot_ms / max(launches, 1), 2),
                           "algorithmic_bytes_per_launch": int(tot_bytes / max(launches, 1)),
                           "pair_gflop_per_step": round(tot_flops / max(n_prof, 1) / 1e9, 2),
                           "spconv_ms_per_step": round(tot_ms / max(n_prof, 1), 3), "instrumented_steps": n_prof}
        if args.stage_times:
            st = {}
            for (n0, e0), (n1, e1) in zip(stage_events[:-1], stage_events[1:]):
                if n1 != "start":
                    st[n1] = st.get(n1, 0.0) + e0.elapsed_time(e1)
            print("[stage] GPU ms/step: " + ", ".join("%s=%.3f" % (k, v / n_prof) for k, v in st.items()), file=sys.stderr)
            agg = {}
            for i, (tag, info, m) in enumerate(ms):
                a = agg.setdefault((tag, info["n_out"]), [0.0, 0, 0])
                a[0] += m
                a[1] += 1
                a[2] = pair_counts[i % per_step]
            for (tag, n_out), (m, cnt, pairs) in agg.items():
                print("[stage] %-22s n_out=%7d pairs=%8d launches/step=%d avg=%.1f us" % (tag, n_out, pairs, cnt // n_prof, 1e3 * m / cnt),
                      file=sys.stderr)
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(cfg, sd, host_clouds[0])
            except Exception as e:  # the baseline is reported context, never a reason to lose the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "sweeps/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
