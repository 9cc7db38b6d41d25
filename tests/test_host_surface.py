"""-m "not gpu": the drop-in boundary (registry / config surface / C ABI / loud failure) and the multi-process
sharding logic over gloo."""
import glob
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

import futuredet_amd as fa
from futuredet_amd import Config, lib
from futuredet_amd.configs import centerpoint_config

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_CFG = "/root/reference/configs/centerpoint"


def _plain(v):
    if isinstance(v, dict):
        return {k: _plain(x) for k, x in v.items() if k != "logger"}
    if isinstance(v, (list, tuple)):
        return [_plain(x) for x in v]
    if isinstance(v, np.generic):
        return v.item()
    return v


def test_config_builder_matches_parsed_reference_configs():
    g = json.load(open(os.path.join(REPO, "tests", "golden", "configs.json")))
    for fname, variant, cls in [("nusc_centerpoint_forecast_n0_detection.py", "forecast_n0", "car"),
                                ("nusc_centerpoint_forecast_n3_detection.py", "forecast_n3", "car"),
                                ("nusc_centerpoint_forecast_n3dtf_detection.py", "forecast_n3dtf", "car"),
                                ("nusc_centerpoint_forecast_n3dtfm_detection.py", "forecast_n3dtfm", "car"),
                                ("nusc_centerpoint_pedestrian_forecast_n0_detection.py", "forecast_n0", "pedestrian"),
                                ("nusc_centerpoint_pedestrian_forecast_n3_detection.py", "forecast_n3", "pedestrian")]:
        c = centerpoint_config(variant, cls)
        for k in ("model", "test_cfg", "voxel_generator", "timesteps", "tasks", "class_names"):
            assert _plain(c[k]) == g[fname][k], (fname, k)
    from futuredet_amd.configs import pointpillars_config
    for fname, cls in [("nusc_centerpoint_pp_forecast_n3dtf_detection.py", "car"),
                       ("nusc_centerpoint_pp_pedestrian_forecast_n3dtf_detection.py", "pedestrian")]:
        c = pointpillars_config(cls)
        for k in ("model", "test_cfg", "voxel_generator", "timesteps", "tasks", "class_names"):
            assert _plain(c[k]) == g[fname][k], (fname, k)


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason="reference configs only exist in the build container")
def test_reference_config_files_load_unchanged():
    g = json.load(open(os.path.join(REPO, "tests", "golden", "configs.json")))
    files = sorted(glob.glob(os.path.join(REF_CFG, "*.py")))
    assert len(files) == 10
    for f in files:
        cfg = Config.fromfile(f)
        want = g[os.path.basename(f)]
        for k in ("model", "test_cfg", "voxel_generator", "timesteps", "assigner"):
            assert _plain(cfg[k]) == want[k], (f, k)
        assert cfg.test_cfg.nms.nms_pre_max_size == 1000 and cfg.TWO_STAGE is False
        if cfg.model.type == "VoxelNet":
            net = fa.build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
            assert type(net).__name__ == "VoxelNet"
        else:
            net = fa.build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
            assert type(net).__name__ == "PointPillars" and "reader.pfn_layers.1.linear.weight" in net.state_dict()


def test_registry_contract():
    from futuredet_amd.registry import Registry, build_from_cfg

    R = Registry("thing")

    @R.register_module
    class A(object):
        def __init__(self, x, y=2):
            self.x, self.y = x, y

    with pytest.raises(KeyError):
        R.register_module(A)
    a = build_from_cfg(dict(type="A", x=1), R, dict(y=5))
    assert (a.x, a.y) == (1, 5)
    with pytest.raises(KeyError):
        build_from_cfg(dict(type="B"), R)
    with pytest.raises(TypeError):
        R.register_module(3)
    for name in ("VoxelNet", "SingleStageDetector", "PointPillars", "TwoStageDetector"):
        assert fa.DETECTORS.get(name) is not None
    assert fa.READERS.get("VoxelFeatureExtractorV3") and fa.BACKBONES.get("SpMiddleResNetFHD") and fa.NECKS.get("RPN") and fa.HEADS.get("CenterHead")
    for name in ("LoadPointCloudFromFile", "LoadPointCloudAnnotations", "Preprocess", "Voxelization", "AssignLabel", "Reformat", "DoubleFlip", "Empty"):
        assert fa.PIPELINES.get(name) is not None


def test_det3d_alias_names():
    from futuredet_amd import compat

    assert compat.install_det3d_alias()
    from det3d.models import build_detector  # noqa: F401
    from det3d.torchie import Config as C2
    from det3d.utils.config_tool import get_downsample_factor

    assert C2 is Config
    cfg = centerpoint_config("forecast_n0")
    assert get_downsample_factor(cfg.model) == 8
    # the import block of the reference's inference entry point (tools/dist_test.py:19-32), minus third-party modules
    from det3d import torchie  # noqa: F401
    from det3d.datasets import build_dataloader, build_dataset  # noqa: F401
    from det3d.torchie.apis import (batch_processor, build_optimizer, get_root_logger, init_dist, set_random_seed,  # noqa: F401
                                    train_detector)
    from det3d.torchie.trainer import get_dist_info, load_checkpoint  # noqa: F401
    from det3d.torchie.trainer.utils import all_gather, synchronize

    assert get_dist_info() == (0, 1) and all_gather({"a": 1}) == [{"a": 1}]
    synchronize()
    with pytest.raises(NotImplementedError):
        train_detector()
    with pytest.raises(NotImplementedError):
        batch_processor(None, {}, True)


def test_state_dict_keys_match_reference(golden):
    """Checkpoint compatibility: parameter / buffer names and shapes equal the reference modules' (goldens hold the
    reference's own key lists)."""
    g = golden("backbone.npz")
    bb = fa.build_backbone(dict(type="SpMiddleResNetFHD", num_input_features=5, ds_factor=8))
    assert sorted(bb.state_dict().keys()) == list(g["keys"])
    shapes = {k: str(tuple(v.shape)) for k, v in bb.state_dict().items()}
    assert [shapes[k] for k in sorted(shapes)] == list(g["shapes"])
    d = golden("dense_nets.npz")
    import logging
    rpn = fa.build_neck(dict(type="RPN", layer_nums=[2, 2], ds_layer_strides=[1, 2], ds_num_filters=[16, 32], us_layer_strides=[1, 2],
                             us_num_filters=[32, 32], num_input_features=24, logger=logging.getLogger("RPN")))
    assert sorted(rpn.state_dict().keys()) == list(d["rpn_keys"])
    for name, T, dense, ff, classify in (("n0", 1, False, False, False), ("n3", 7, False, False, False), ("n3dtf", 7, True, True, False),
                                         ("cls3", 3, False, False, True), ("rev3", 3, False, False, False), ("sp7", 7, False, False, False),
                                         ("wide7", 7, False, False, False)):
        kw = dict(type="CenterHead", in_channels=64, tasks=[dict(num_class=1, class_names=["car"])], dataset="nuscenes",
                  weight=0.25, code_weights=[1.0] * 10,
                  common_heads={"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2), "vel": (2, 2)},
                  share_conv_channel=64, dcn_head=False, timesteps=T, two_stage=False, reverse=name == "rev3", sparse=name == "sp7",
                  dense=dense, bev_map=False, forecast_feature=ff, classify=False, wide_head=name == "wide7")
        if classify:  # the reference constructor's DEFAULT (center_head.py:253): built without the keyword, as the golden was
            del kw["classify"]
        head = fa.build_head(kw)
        assert head.classify == classify
        assert sorted(head.state_dict().keys()) == list(d["head_%s_keys" % name])


def test_dense_modules_match_reference_golden_on_cpu(golden):
    """The nn.Module stacks of RPN / CenterHead (what fixes the state-dict keys; run for host tensors and in training mode)
    reproduce the reference outputs on the CPU -- including the constructor's default ``classify`` head.  On the device, eval
    mode runs the convolution plan and nothing else (tests/test_gpu_parity.py)."""
    import logging

    from futuredet_amd.synth import seeded_state_dict

    d = golden("dense_nets.npz")
    rpn = fa.build_neck(dict(type="RPN", layer_nums=[2, 2], ds_layer_strides=[1, 2], ds_num_filters=[16, 32], us_layer_strides=[1, 2],
                             us_num_filters=[32, 32], num_input_features=24, logger=logging.getLogger("RPN"))).eval()
    rpn.load_state_dict(seeded_state_dict(rpn, 11), strict=False)
    x = torch.from_numpy(d["rpn_x"])
    with torch.no_grad():
        y_fold, y_mod = rpn(x), rpn.forward_modules(x)
    np.testing.assert_allclose(y_fold.numpy(), d["rpn_y"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(y_mod.numpy(), d["rpn_y"], rtol=1e-4, atol=1e-4)
    for name, T, dense, ff, classify in (("n0", 1, False, False, False), ("n3", 7, False, False, False), ("n3dtf", 7, True, True, False),
                                         ("cls3", 3, False, False, True), ("rev3", 3, False, False, False), ("sp7", 7, False, False, False),
                                         ("wide7", 7, False, False, False)):
        head = fa.build_head(dict(type="CenterHead", in_channels=64, tasks=[dict(num_class=1, class_names=["car"])], dataset="nuscenes",
                                  weight=0.25, code_weights=[1.0] * 10,
                                  common_heads={"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2), "vel": (2, 2)},
                                  share_conv_channel=64, dcn_head=False, timesteps=T, two_stage=False, reverse=name == "rev3", sparse=name == "sp7",
                                  dense=dense, bev_map=False, forecast_feature=ff, classify=classify, wide_head=name == "wide7")).eval()
        head.load_state_dict(seeded_state_dict(head, 12), strict=False)
        with torch.no_grad():
            preds = head(torch.from_numpy(d["rpn_y"]))
        for ti, pd in enumerate(preds):
            for k, v in pd.items():
                np.testing.assert_allclose(v.numpy(), d["head_%s_t%d_%s" % (name, ti, k)], rtol=1e-3, atol=2e-4)


def test_c_abi_exports_every_declared_symbol():
    """The shared library loads (no GPU needed) and exports exactly what include/futuredet_hip.h declares."""
    from futuredet_amd import build

    build.build()
    L = lib.load()
    hdr = open(os.path.join(REPO, "include", "futuredet_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(fd_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(lib.SIGNATURES), declared ^ set(lib.SIGNATURES)
    nm = subprocess.check_output(["nm", "-D", "--defined-only", lib.LIB_PATH]).decode()
    exported = set(re.findall(r" T (fd_[a-z0-9_]+)", nm))
    assert declared <= exported, declared - exported
    assert L.fd_abi_version() == 8
    assert L.fd_index_num_cols(2, 180, 180) == 2 * 23 * 23 * 64
    assert L.fd_voxelize_workspace_bytes(1000, 100) > 0 and L.fd_nms_workspace_bytes(1000) >= 1000 * 16 * 8


def test_ctypes_structs_mirror_the_header(tmp_path):
    """The structs that cross the C ABI by pointer (fd_decode_cfg, fd_forecast_buffers, fd_index_level): the ctypes mirrors of lib.py hold
    the header's members in the header's order and gcc gives them the same size -- a member added on one side only would shift every
    field behind it without any symbol going missing."""
    import ctypes

    hdr = open(os.path.join(REPO, "include", "futuredet_hip.h")).read()
    bare = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    pairs = {"fd_decode_cfg": lib.DecodeCfg, "fd_forecast_buffers": lib.ForecastBuffers, "fd_index_level": lib.IndexLevel}
    prog = ['#include <stdio.h>', '#include "futuredet_hip.h"', "int main(void) {"]
    for cname, mirror in pairs.items():
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), bare, flags=re.S).group(1)
        members = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):  # "float out_size_factor, voxel_x" / "const void *words" / "float center_range[6]"
                m = re.search(r"([A-Za-z_][A-Za-z0-9_]*)\s*(\[(\d+)\])?\s*$", part.strip())
                members.append((m.group(1), int(m.group(3)) if m.group(3) else None))
        mine = [(n, getattr(t, "_length_", None)) for n, t in mirror._fields_]
        assert mine == members, (cname, mine, members)
        prog.append('    printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
    prog += ["    return 0;", "}"]
    src = tmp_path / "sizes.c"
    src.write_text("\n".join(prog) + "\n")
    exe = str(tmp_path / "sizes")
    subprocess.check_call(["gcc", "-I", os.path.join(REPO, "include"), "-o", exe, str(src)])
    sizes = dict(line.split() for line in subprocess.check_output([exe]).decode().splitlines())
    for cname, mirror in pairs.items():
        assert int(sizes[cname]) == ctypes.sizeof(mirror), (cname, sizes[cname], ctypes.sizeof(mirror))


def test_circular_nms_configuration_follows_the_reference():
    """test_cfg.circular_nms on the host side (no device work): min_radius is indexed by the output step like the reference's
    ``test_cfg.min_radius[task_id]`` (center_head.py:609,724) -- a scalar is a TypeError there and here; steps that share a decode group
    share its radius or the head decodes a group per step; the decode configuration takes the kernels' maximal candidate cut."""
    from futuredet_amd import hip_ops
    from futuredet_amd.heads import CenterHead

    assert CenterHead._group_radius({"min_radius": [2.0] * 7}, [0] * 7, 1) == [2.0]
    assert CenterHead._group_radius({"min_radius": [1.0, 1.0, 3.0, 3.0]}, [0, 0, 1, 1], 2) == [1.0, 3.0]
    assert CenterHead._group_radius({"min_radius": [1.0, 2.0]}, [0, 0], 1) is None
    with pytest.raises(TypeError):
        CenterHead._group_radius({"min_radius": 2}, [0], 1)
    with pytest.raises(IndexError):
        CenterHead._group_radius({"min_radius": [2.0]}, [0, 0], 1)
    assert CenterHead._circular({"circular_nms": True}) and not CenterHead._circular({})
    with pytest.raises(NotImplementedError):
        CenterHead._circular({"per_class_nms": True})
    test_cfg = dict(post_center_limit_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0], nms=dict(nms_pre_max_size=1000, nms_post_max_size=83, nms_iou_threshold=0.2),
                    score_threshold=0.1, pc_range=[-54, -54], out_size_factor=8, voxel_size=[0.075, 0.075])
    c = hip_ops.make_decode_cfg(180, 180, test_cfg)
    assert (c.nms_kind, c.n_radius, c.nms_pre_max) == (0, 0, 1000)
    c = hip_ops.make_decode_cfg(180, 180, test_cfg, group_radius=[4.0, 0.85])
    assert (c.nms_kind, c.n_radius, c.nms_pre_max) == (1, 2, hip_ops.CIRCLE_PRE_MAX) and list(c.circle_radius)[:3] == [4.0, np.float32(0.85), 0.0]
    with pytest.raises(ValueError):
        hip_ops.make_decode_cfg(180, 180, test_cfg, group_radius=[1.0] * 17)


def test_no_packed_fp32_instruction_selects_the_high_half_of_src1(tmp_path):
    """Static guard for the round-6 determinism defect (profiles/round6_determinism_soak.txt): on MI355X a packed-fp32 instruction whose
    op_sel swizzles src1 (``v_pk_mul_f32 / v_pk_add_f32 ... op_sel:[0,1]``) returned wrong values in lanes 48-63 while bf16 dense-convolution
    waves of another stream shared the compute unit (tools/soak_alu.py, modes 19 / 21: thousands of disagreements per run; plain, neg and
    op_sel_hi forms: none).  The SLP vectoriser emits that form for the rotated-IoU geometry; fd_decode.hip and fd_pillars.hip are built with
    -fno-slp-vectorize, and NO kernel of the library may contain the form: the device code of the built library is disassembled and searched."""
    import shutil

    from futuredet_amd import build

    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.isfile(objdump):
        pytest.skip("llvm-objdump not found")
    build.build()
    so = str(tmp_path / "lib.so")
    shutil.copy(lib.LIB_PATH, so)
    subprocess.check_output([objdump, "--offloading", so], stderr=subprocess.STDOUT)  # writes lib.so.<n>.hipv4-amdgcn-amd-amdhsa--gfx950 next to it
    objs = [str(tmp_path / f) for f in os.listdir(str(tmp_path)) if f.endswith("gfx950")]
    assert len(objs) == len(build.SOURCES), objs
    packed, bad = 0, []
    for o in objs:
        for line in subprocess.check_output([objdump, "-d", o]).decode().splitlines():
            m = re.search(r"\bv_pk_(mul|add|fma)_f32\b.*", line)
            if m:
                packed += 1
                if "op_sel:[" in m.group(0):
                    bad.append(m.group(0).strip())
    assert packed > 1000, "the disassembly should show the epilogues' packed additions (%d found)" % packed
    assert not bad, "%d packed-fp32 instructions with an op_sel swizzle, e.g. %s" % (len(bad), bad[:3])


def test_product_path_fails_loudly_without_gpu_tensors():
    from futuredet_amd import hip_ops
    from futuredet_amd.lib import FutureDetHipError

    with pytest.raises(FutureDetHipError):
        hip_ops.voxelize(torch.zeros((10, 5)), [0.1] * 3, [0, 0, 0, 1, 1, 1], 5, 10)
    with pytest.raises(FutureDetHipError):
        hip_ops.boxes_iou_bev(torch.zeros((2, 7)), torch.zeros((2, 7)))
    src = open(os.path.join(REPO, "futuredet_amd", "hip_ops.py")).read() + open(os.path.join(REPO, "futuredet_amd", "detectors.py")).read()
    assert "oracle" not in src, "the product path must never import the oracle"


def test_weight_packing_layout():
    """fd_spconv_pack_weight is host code: check the fragment order documented in fd_spconv.hip."""
    import ctypes

    L = lib.load()
    K, cin, cout = 3, 32, 16
    w = np.arange(K * cin * cout, dtype=np.float32).reshape(K, cin, cout)
    out = np.zeros(K * cin * cout, np.float32)
    assert L.fd_spconv_pack_weight(w.ctypes.data_as(ctypes.c_void_p), K, cin, cout, 0, out.ctypes.data_as(ctypes.c_void_p)) == 0
    out = out.reshape(K, cin // 16, cout // 16, 64, 4)
    for k, c, nb, lane, j in [(0, 0, 0, 0, 0), (2, 1, 0, 37, 3), (1, 0, 0, 63, 2)]:
        assert out[k, c, nb, lane, j] == w[k, 16 * c + 4 * (lane >> 4) + j, 16 * nb + (lane & 15)]
    assert L.fd_spconv_pack_weight(w.ctypes.data_as(ctypes.c_void_p), K, 20, cout, 0, out.ctypes.data_as(ctypes.c_void_p)) != 0
    assert b"multiples of 16" in L.fd_last_error()


def test_collate_prefixes_batch_index():
    from futuredet_amd.collate import collate_kitti_multi

    ex = [dict(voxels=np.zeros((3, 10, 5), np.float32), coordinates=np.ones((3, 3), np.int32), num_points=np.ones(3, np.int32),
               num_voxels=np.array([3]), shape=np.array([1440, 1440, 40]), metadata={"token": i}) for i in range(2)]
    b = collate_kitti_multi(ex)
    assert b["coordinates"].shape == (6, 4) and b["coordinates"][:, 0].tolist() == [0, 0, 0, 1, 1, 1]
    assert b["voxels"].shape == (6, 10, 5) and b["num_voxels"].tolist() == [3, 3] and b["shape"].shape == (2, 3)


def _dist_worker(rank, world, port, n_samples, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from futuredet_amd import dist_infer

    r, w, _ = dist_infer.init_from_env("gloo")
    mine = dist_infer.shard_indices(n_samples, r, w)
    S, post = 7, 83
    boxes = torch.zeros((len(mine), S, post, 9))
    scores = torch.zeros((len(mine), S, post))
    labels = torch.zeros((len(mine), S, post), dtype=torch.int64)
    counts = torch.zeros((len(mine), S), dtype=torch.int32)
    for li, gi in enumerate(mine):
        counts[li] = (gi % post) + 1
        boxes[li] = float(gi)
        scores[li] = gi / 100.0
        labels[li] = torch.arange(S).view(S, 1)
    packed, cnt = dist_infer.pack_results(boxes, scores, labels, counts)
    full, fullc = dist_infer.gather_results(packed, cnt, n_samples)
    res = dist_infer.unpack_results(full, fullc)
    ok = len(res) == n_samples
    for gi, d in enumerate(res):
        k = (gi % post) + 1
        ok &= d["box3d_lidar"].shape == (S * k, 9) and bool((d["box3d_lidar"] == float(gi)).all())
        ok &= d["label_preds"].tolist() == [s for s in range(S) for _ in range(k)]
    # the reference-contract helpers (trainer/utils.py:100-155): barrier + picklable-object gather, one entry per rank
    dist_infer.synchronize()
    objs = dist_infer.all_gather({"rank": r, "tokens": mine})
    ok &= dist_infer.get_dist_info() == (r, w) and [o["rank"] for o in objs] == list(range(w)) and objs[r]["tokens"] == mine
    q.put((rank, ok, mine))
    torch.distributed.destroy_process_group()


def test_two_process_gloo_shard_and_gather():
    """N>1 path on CPU: DistributedSampler-style rank-strided shards + one fixed-shape all_gather, world_size 2."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29000 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_dist_worker, args=(r, 2, port, 5, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert out[0][1] and out[1][1]
    assert out[0][2] == [0, 2, 4] and out[1][2] == [1, 3, 0]  # wrap-padded like DistributedSampler(shuffle=False)


def test_synthetic_cloud_is_deterministic_and_shaped():
    from futuredet_amd.synth import synthetic_cloud

    a, b = synthetic_cloud(3, 20000), synthetic_cloud(3, 20000)
    assert np.array_equal(a, b) and a.dtype == np.float32 and a.shape[1] == 5
    assert 15000 < len(a) < 25000 and set(np.round(np.unique(a[:, 4]) / 0.05).astype(int)) == set(range(10))


def test_sweep_descriptor_layout_and_visit_order(golden):
    """struct fd_sweep_desc <-> numpy dtype, and the reference's seeded visit order (loading.py:128-129)."""
    from futuredet_amd import hip_ops, loading

    assert hip_ops.SWEEP_DESC.itemsize == 16 * 8 + 8 + 8 + 4 + 4
    assert [hip_ops.SWEEP_DESC.fields[k][1] for k in ("m", "row_begin", "row_end", "time", "flags")] == [0, 128, 136, 144, 148]
    order = loading.sweep_visit_order(9, 10)
    assert sorted(order) == list(range(9)) and order == [int(i) for i in np.random.default_rng(0).choice(9, 9, replace=False)]
    m = np.arange(16.0).reshape(4, 4).astype(np.float32)
    d = hip_ops.sweep_descriptors([0, 5, 9], [None, m], [0.0, 0.1], [False, True])
    assert d["flags"].tolist() == [0, 3] and d["row_end"].tolist() == [5, 9]
    assert d["time"][1] == np.float32(0.1) and np.array_equal(d["m"][1], np.arange(16.0))


def test_load_point_cloud_class_resolves_and_rejects_other_datasets():
    from futuredet_amd import PIPELINES, build_from_cfg

    stage = build_from_cfg(dict(type="LoadPointCloudFromFile", dataset="NuScenesDataset"), PIPELINES)
    assert stage.type == "NuScenesDataset"
    with pytest.raises(NotImplementedError):
        build_from_cfg(dict(type="LoadPointCloudFromFile", dataset="WaymoDataset"), PIPELINES)({"lidar": {}}, {})


def test_hip_ops_exposes_every_wrapper():
    """Guards the Python front end against accidental deletions: every op family has its wrapper."""
    from futuredet_amd import hip_ops

    for name in ("voxelize", "SparseIndex", "build_pyramid", "ranges_for", "rows_permute", "pack_spconv_weight", "spconv_apply",
                 "densify", "pack_conv2d_weight", "conv2d_nhwc_bf16", "pack_conv2d_weight_f32", "conv2d_nhwc_f32", "pack_conv2d_weight_wino", "conv2d_wino_nhwc_f32", "conv2d_shuffle_nhwc_f32", "conv2d_grouped_nhwc_f32", "make_decode_cfg", "centerpoint_decode", "rotated_nms",
                 "boxes_iou_bev", "sweep_descriptors", "assemble_sweeps", "pillar_encode", "pillar_scatter",
                 "forecast_chains", "det_to_global_boxes", "forecast_groups", "set_tuning"):
        assert hasattr(hip_ops, name), name


def _replica_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from futuredet_amd import build_detector, dist_infer
    from futuredet_amd.synth import seeded_state_dict

    dist_infer.init_from_env("gloo")
    cfg = centerpoint_config("forecast_n3")
    net = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    net.load_state_dict(seeded_state_dict(net, 100 + rank), strict=False)  # every rank starts from DIFFERENT weights
    cs = dist_infer.sync_replicas(net, src=0, check=True)
    ref = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    ref.load_state_dict(seeded_state_dict(ref, 100), strict=False)
    same = all(torch.equal(a, b) for a, b in zip(net.state_dict().values(), ref.state_dict().values()))
    # a replica that drifts afterwards must fail the check (every rank sees the mismatch and raises)
    if rank == 1:
        with torch.no_grad():
            next(net.parameters()).add_(1.0)
    import torch.distributed as dist
    mine = torch.tensor([float(sum(p.double().sum() for p in net.parameters()))], dtype=torch.float64)
    allc = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allc, mine)
    caught = float(allc[0]) != float(allc[1])
    q.put((rank, bool(same), bool(caught), float(cs)))
    torch.distributed.destroy_process_group()


def test_replicas_take_rank0_weights_and_are_checked():
    """VERDICT r2 missing #2: the reference replicates weights through the DDP constructor (tools/dist_test.py:177-188).
    dist_infer.sync_replicas: two gloo ranks start from different seeds; afterwards both hold rank 0's weights bit for bit and
    report the same checksum; a later drift of one replica shows up in a checksum exchange."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31000 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_replica_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
    assert out[0][1] and out[1][1], "both ranks must hold rank 0's weights"
    assert out[0][3] == out[1][3], "equal checksums after the broadcast"
    assert out[0][2] and out[1][2], "a drifting replica must change the exchanged checksum"


def test_build_lock_and_cpu_pinning_helpers():
    from futuredet_amd import build as fbuild
    from futuredet_amd import dist_infer

    assert os.path.isfile(fbuild.build_locked())  # up to date here: returns the library path without compiling
    if hasattr(os, "sched_getaffinity"):
        before = sorted(os.sched_getaffinity(0))
        try:
            mine = dist_infer.pin_rank_cpus(1, 2)
            if len(before) >= 2:
                assert mine == before[len(before) // 2: 2 * (len(before) // 2)] and sorted(os.sched_getaffinity(0)) == mine
        finally:
            os.sched_setaffinity(0, before)
            torch.set_num_threads(max(1, min(len(before), 64)))


def test_rank_cpu_slices_follow_the_gpus_numa_nodes():
    """8 ranks on a two-socket box (CPUs 0-63 + 128-191 on node 0, 64-127 + 192-255 on node 1; GPUs 0-3 on node 0, 4-7 on node 1):
    every rank gets 32 CPUs of ITS GPU's node, slices are disjoint and cover the box; unknown topology = contiguous eighths."""
    from futuredet_amd import dist_infer

    cpus = list(range(256))
    node_cpus = {0: dist_infer._parse_cpulist("0-63,128-191"), 1: dist_infer._parse_cpulist("64-127,192-255\n")}
    nodes = [0, 0, 0, 0, 1, 1, 1, 1]
    sl = dist_infer.rank_cpu_slices(cpus, 8, nodes, node_cpus)
    assert all(len(s) == 32 for s in sl) and sorted(c for s in sl for c in s) == cpus
    for r, s in enumerate(sl):
        assert set(s) <= set(node_cpus[nodes[r]])
    assert dist_infer.rank_cpu_slices(cpus, 8) == [cpus[r * 32:(r + 1) * 32] for r in range(8)]
    # one GPU with an unknown node keeps its contiguous slice; a restricted affinity mask is honoured
    sl = dist_infer.rank_cpu_slices(list(range(64)), 2, [0, None], {0: list(range(0, 16))})
    assert sl == [list(range(0, 16)), list(range(32, 64))]
    assert dist_infer.gpu_numa_node(0, sysfs="/nonexistent") is None


def test_bench_refuses_more_ranks_than_devices():
    """`python bench.py --gpus N` without a launcher starts its own ranks (bench.self_launch); with fewer devices than ranks it must
    refuse with a message instead of running one rank and printing n_gpus = 1.  No GPU here: 0 devices < 2."""
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "FD_BENCH_ONE_DEVICE")}
    if __import__("torch").cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than two devices")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 2 and "device(s) are visible" in out.stderr and "n_gpus" not in out.stdout


def test_bench_cpu_baseline_reports_the_all_core_sample():
    """bench.cpu_baseline runs the oracle on every logical CPU of the host in-process (the sparse conv is one OpenMP region over output
    rows: no fork / join per tap) and matches the rows it is given against the oracle's detections."""
    import bench
    from futuredet_amd.configs import centerpoint_config
    from futuredet_amd.synth import seeded_state_dict, synthetic_cloud, tame_box_dims

    cfg = centerpoint_config("forecast_n0")
    net = fa.build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    sd = tame_box_dims(seeded_state_dict(net, 7))
    cloud = synthetic_cloud(seed=0, target_points=3000)
    out, par = bench.cpu_baseline(cfg, sd, cloud, np.zeros((0, 11), np.float32), budget_s=(0.5, 0.5))
    assert out["cores"] in (os.cpu_count(), 64) and out["host_cpu_count"] == os.cpu_count() and out["kind"] == "port"
    assert out["value"] > 0 and out["value_all_cores"] > 0 and out["value_all_cores_passes"] >= 3
    assert par["gpu_rows"] == 0 and par["unmatched"] == par["oracle_rows"]


def test_load_checkpoint_follows_the_reference_contract(tmp_path):
    """det3d/torchie/trainer/checkpoint.py:122-173: a {"state_dict": {"module.<reference key>": tensor}} file -- what the
    reference's DDP training run saves -- loads into the detector with the wrapper prefix stripped; a bare OrderedDict loads too;
    a missing file is an IOError, a dict without "state_dict" a RuntimeError, remote names are refused with the reason."""
    import collections

    from futuredet_amd import build_detector
    from futuredet_amd.configs import centerpoint_config
    from futuredet_amd.detectors import load_checkpoint
    from futuredet_amd.synth import seeded_state_dict

    cfg = centerpoint_config("forecast_n3", "car")
    src = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    sd = seeded_state_dict(src, 11)
    src.load_state_dict(sd, strict=False)
    full = src.state_dict()
    # reference key names (scn.py / rpn.py / center_head.py module layout)
    for k, shape in (("backbone.conv_input.0.weight", (3, 3, 3, 5, 16)), ("backbone.conv1.0.conv1.bias", (16,)), ("backbone.conv4.0.weight", (3, 3, 3, 64, 128)),
                     ("neck.blocks.0.1.weight", (128, 256, 3, 3)), ("bbox_head.shared_conv.0.weight", (64, 512, 3, 3)),
                     ("bbox_head.tasks.0.hm.0.weight", (64, 64, 3, 3))):
        assert k in full and tuple(full[k].shape) == shape, (k, tuple(full[k].shape) if k in full else None)
    wrapped = {"meta": {"epoch": 20}, "state_dict": collections.OrderedDict(("module." + k, v.clone()) for k, v in full.items())}
    f1 = str(tmp_path / "epoch_20.pth")
    torch.save(wrapped, f1)
    dst = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    ck = load_checkpoint(dst, f1, map_location="cpu", strict=True)
    assert ck["meta"]["epoch"] == 20
    got = dst.state_dict()
    assert set(got) == set(full) and all(torch.equal(got[k], full[k]) for k in full)

    f2 = str(tmp_path / "bare.pth")
    torch.save(collections.OrderedDict(full), f2)
    dst2 = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    load_checkpoint(dst2, f2)
    assert all(torch.equal(dst2.state_dict()[k], full[k]) for k in full)

    class Wrapper(torch.nn.Module):  # what DataParallel / DDP look like to load_checkpoint: the model sits in .module
        def __init__(self, m):
            super().__init__()
            self.module = m

    dst3 = Wrapper(build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg))
    load_checkpoint(dst3, f1)
    assert torch.equal(dst3.module.state_dict()["neck.blocks.0.1.weight"], full["neck.blocks.0.1.weight"])

    with pytest.raises(IOError):
        load_checkpoint(dst, str(tmp_path / "nope.pth"))
    f3 = str(tmp_path / "junk.pth")
    torch.save({"weights": 1}, f3)
    with pytest.raises(RuntimeError):
        load_checkpoint(dst, f3)
    with pytest.raises(NotImplementedError):
        load_checkpoint(dst, "torchvision://resnet50")
    # the constructor's pretrained= path goes through the same function (single_stage.py:29-36)
    net = build_detector(dict(cfg.model, pretrained=f1), train_cfg=None, test_cfg=cfg.test_cfg)
    assert torch.equal(net.state_dict()["bbox_head.tasks.0.hm.0.weight"], full["bbox_head.tasks.0.hm.0.weight"])
