"""SpMiddleResNetFHD (det3d/models/backbones/scn.py:37-176) on the HIP sparse-conv kernels.

Module / parameter names reproduce the reference so checkpoints load unchanged
(conv_input.0.weight [3,3,3,5,16], conv1.0.conv1.{weight,bias}, conv1.0.bn1.*, conv2.0.weight, ...).
Two execution paths, both on the HIP kernels:
  * generic: the spconv-surface modules one by one (SparseSequential / SparseBasicBlock.forward), used in
    training mode or when a caller drives sub-modules directly;
  * fused (eval): all five sparse indexes are built first (one host read of the five active counts), every
    conv runs as one fd_spconv_apply launch with BatchNorm1d folded into weight/bias and ReLU / residual
    fused in the epilogue, features stay channel-padded and in index order, then fd_densify.
"""
import numpy as np
import torch
from torch import nn

from . import hip_ops, sparse as spconv
from .nn_utils import build_norm_layer
from .registry import BACKBONES
from .sparse import SparseConv3d, SubMConv3d


def conv3x3(in_planes, out_planes, stride=1, indice_key=None, bias=True):
    return spconv.SubMConv3d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=bias,
                             indice_key=indice_key)


class SparseBasicBlock(spconv.SparseModule):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, norm_cfg=None, downsample=None, indice_key=None):
        super().__init__()
        if norm_cfg is None:
            norm_cfg = dict(type="BN1d", eps=1e-3, momentum=0.01)
        bias = norm_cfg is not None  # scn.py:54 -- always True, so block convs carry a bias
        self.conv1 = conv3x3(inplanes, planes, stride, indice_key=indice_key, bias=bias)
        self.bn1 = build_norm_layer(norm_cfg, planes)[1]
        self.relu = nn.ReLU()
        self.conv2 = conv3x3(planes, planes, indice_key=indice_key, bias=bias)
        self.bn2 = build_norm_layer(norm_cfg, planes)[1]
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x
        out = self.conv1(x)
        out.features = self.relu(self.bn1(out.features))
        out = self.conv2(out)
        out.features = self.bn2(out.features)
        if self.downsample is not None:
            identity = self.downsample(x)
        out.features = self.relu(out.features + identity.features)
        return out


@BACKBONES.register_module
class SpMiddleResNetFHD(nn.Module):
    def __init__(self, num_input_features=128, norm_cfg=None, name="SpMiddleResNetFHD", **kwargs):
        super().__init__()
        self.name = name
        if norm_cfg is None:
            norm_cfg = dict(type="BN1d", eps=1e-3, momentum=0.01)
        S = spconv.SparseSequential
        self.conv_input = S(SubMConv3d(num_input_features, 16, 3, bias=False, indice_key="res0"),
                            build_norm_layer(norm_cfg, 16)[1], nn.ReLU(inplace=True))
        self.conv1 = S(SparseBasicBlock(16, 16, norm_cfg=norm_cfg, indice_key="res0"),
                       SparseBasicBlock(16, 16, norm_cfg=norm_cfg, indice_key="res0"))
        self.conv2 = S(SparseConv3d(16, 32, 3, 2, padding=1, bias=False), build_norm_layer(norm_cfg, 32)[1],
                       nn.ReLU(inplace=True), SparseBasicBlock(32, 32, norm_cfg=norm_cfg, indice_key="res1"),
                       SparseBasicBlock(32, 32, norm_cfg=norm_cfg, indice_key="res1"))
        self.conv3 = S(SparseConv3d(32, 64, 3, 2, padding=1, bias=False), build_norm_layer(norm_cfg, 64)[1],
                       nn.ReLU(inplace=True), SparseBasicBlock(64, 64, norm_cfg=norm_cfg, indice_key="res2"),
                       SparseBasicBlock(64, 64, norm_cfg=norm_cfg, indice_key="res2"))
        self.conv4 = S(SparseConv3d(64, 128, 3, 2, padding=[0, 1, 1], bias=False), build_norm_layer(norm_cfg, 128)[1],
                       nn.ReLU(inplace=True), SparseBasicBlock(128, 128, norm_cfg=norm_cfg, indice_key="res3"),
                       SparseBasicBlock(128, 128, norm_cfg=norm_cfg, indice_key="res3"))
        self.extra_conv = S(SparseConv3d(128, 128, (3, 1, 1), (2, 1, 1), bias=False), build_norm_layer(norm_cfg, 128)[1],
                            nn.ReLU())
        self.compute_dtype = torch.float32   # torch.bfloat16 selects the bf16 MFMA path (fp32 accumulate)
        self.dense_channels_last = True
        self.dense_dtype = None              # dtype of the BEV map handed to the neck (default: compute dtype)
        self.profile_hook = None             # callable(tag, algorithmic_bytes, flops, fn) used by bench.py

    # ------------------------------------------------------------------------------------------------ generic
    def forward_generic(self, voxel_features, coors, batch_size, input_shape):
        sparse_shape = np.array(input_shape[::-1]) + [1, 0, 0]  # scn.py:151
        ret = spconv.SparseConvTensor(voxel_features, coors.int(), sparse_shape, batch_size)
        x = self.conv_input(ret)
        x_conv1 = self.conv1(x)
        x_conv2 = self.conv2(x_conv1)
        x_conv3 = self.conv3(x_conv2)
        x_conv4 = self.conv4(x_conv3)
        ret = self.extra_conv(x_conv4).dense()
        N, C, D, H, W = ret.shape
        ret = ret.view(N, C * D, H, W)
        return ret, {"conv1": x_conv1, "conv2": x_conv2, "conv3": x_conv3, "conv4": x_conv4}

    # ------------------------------------------------------------------------------------------------ fused
    def _stages(self):
        """[(strided conv or None, its bn), [blocks]] per level, in execution order."""
        return [(self.conv_input[0], self.conv_input[1], [self.conv1[0], self.conv1[1]], True),
                (self.conv2[0], self.conv2[1], [self.conv2[3], self.conv2[4]], False),
                (self.conv3[0], self.conv3[1], [self.conv3[3], self.conv3[4]], False),
                (self.conv4[0], self.conv4[1], [self.conv4[3], self.conv4[4]], False),
                (self.extra_conv[0], self.extra_conv[1], [], False)]

    def build_indexes(self, mark_fn, batch_size, input_shape, device, voxels=None, static=False, expected=None, row_caps=None):
        """Builds the five sparse indexes.  ``mark_fn(index0)`` marks the input voxels.  One host read.  With
        ``voxels = (coors [B*n_max,4], nvox [B], n_max)`` the whole pyramid is built by two library calls; with
        ``static=True`` there is no host read at all: every level is sized by its row capacity and the counts stay on the
        device (hip_ops.build_pyramid)."""
        D, H, W = [int(v) for v in (np.array(input_shape[::-1]) + [1, 0, 0])]
        if voxels is not None:
            geoms = [self._stages()[lvl][0].geometry() for lvl in range(1, 5)]
            return hip_ops.build_pyramid(voxels[0], voxels[1], voxels[2], batch_size, (D, H, W), geoms, device, static=static,
                                         expected=expected, row_caps=row_caps)
        assert not static, "static indexes are built from the voxelizer's device output (voxels=...)"
        counts = torch.zeros((5,), dtype=torch.int32, device=device)
        idx = [hip_ops.SparseIndex(batch_size, D, H, W, device)]
        mark_fn(idx[0])
        idx[0].scan(counts[0:1])
        for lvl in range(1, 5):
            conv = self._stages()[lvl][0]
            ks, st, pd = conv.geometry()
            nxt = idx[-1].downsample(ks, st, pd)
            nxt.scan(counts[lvl:lvl + 1])
            idx.append(nxt)
        host = counts.cpu()  # the only synchronisation of the backbone
        for i, ix in enumerate(idx):
            ix.finalize(int(host[i]))
        return idx

    def run_fused(self, idx, feats0, dense_out=None, fork=False):
        """feats0: [n0, 16] features already in index order (channel padded).  Returns (dense BEV, per-level
        (features, index)).  ``fork``: the rulebooks (and work-range tables) of levels 1-4 only need the indexes, not the features:
        they are built on a side stream while the first level's convolutions run (inside a captured sweep this forks the graph;
        a sweep's latency loses ~0.1 ms of front-end kernels that used to sit between the convolutions)."""
        dt = self.compute_dtype
        stages = self._stages()
        rb_cache = {}

        def rulebook(src, dst, conv):
            ks, st, pd = conv.geometry()
            key = (id(src), id(dst), tuple(ks), tuple(st), tuple(pd))
            if key not in rb_cache:
                rb_cache[key] = src.rulebook(dst, ks, st, pd)
            return rb_cache[key]

        def run(conv, bn, x, src, dst, relu, residual=None):
            wpk, bias, cin_p, cout_p = conv.packed_weight(dt, bn)
            assert x.shape[1] == cin_p, (x.shape, cin_p)
            nbr = rulebook(src, dst, conv)
            fn = lambda: hip_ops.spconv_apply(x, wpk, bias, nbr, dst.n, cout_p, residual=residual, relu=relu)  # noqa: E731
            if self.profile_hook is not None:
                s = 4 if dt == torch.float32 else 2
                K = nbr.shape[0]
                pairs = lambda: int((nbr[:, : dst.n] >= 0).sum().item())  # noqa: E731
                return self.profile_hook("spconv_%dx%d_K%d" % (cin_p, cout_p, K),
                                         dict(s=s, K=K, cin=cin_p, cout=cout_p, n_in=src.n, n_out=dst.n, pairs=pairs), fn)
            return fn()

        main, side = None, None
        if fork and feats0.is_cuda:
            main = torch.cuda.current_stream(feats0.device)
            pool = self.__dict__.setdefault("_side_streams", {})
            side = pool.get(main.cuda_stream)
            if side is None:  # one side stream per launch stream: sweeps in flight on different streams fork independently
                side = pool[main.cuda_stream] = torch.cuda.Stream(device=feats0.device)
            side.wait_stream(main)  # the indexes are complete
            with torch.cuda.stream(side):
                made = []
                for lvl, (conv, bn, blocks, is_subm_in) in enumerate(stages):
                    if lvl == 0:
                        continue
                    src_l, dst_l = (idx[lvl] if is_subm_in else idx[lvl - 1]), idx[lvl]
                    made.append(rulebook(src_l, dst_l, conv))
                    for blk in blocks:
                        nbr = rulebook(dst_l, dst_l, blk.conv1)
                        made.append(nbr)
                        cp = spconv.pad_channels(blk.conv1.out_channels)
                        if dt == torch.float32 and nbr.shape[0] == 27 and \
                                hip_ops._lib.load().fd_spconv_wants_balanced_ranges(cp, cp, 0):
                            r = hip_ops.ranges_for(nbr, cp, cp)  # cached on the rulebook tensor; spconv_apply finds it there
                            if r[0] is not None:
                                made.append(r[0])
                for t in made:  # allocated on the side stream, consumed by kernels of the main stream
                    t.record_stream(main)
        x = feats0
        levels = {}
        for lvl, (conv, bn, blocks, is_subm_in) in enumerate(stages):
            src = idx[lvl] if is_subm_in else idx[lvl - 1]
            dst = idx[lvl]
            if lvl == 1 and side is not None:
                main.wait_stream(side)  # join: the first level's convolutions are issued, the other levels' rulebooks are needed now
            x = run(conv, bn, x, src, dst, relu=True)
            for blk in blocks:
                y = run(blk.conv1, blk.bn1, x, dst, dst, relu=True)
                x = run(blk.conv2, blk.bn2, y, dst, dst, relu=True, residual=x)
            levels[lvl] = (x, dst)
        bev = hip_ops.densify(x, idx[4], out_dtype=self.dense_dtype or dt, channels_last=self.dense_channels_last, out=dense_out)
        return bev, levels

    def forward(self, voxel_features, coors, batch_size, input_shape):
        if self.training:
            return self.forward_generic(voxel_features, coors, batch_size, input_shape)
        coors = coors.int().contiguous()
        dev = voxel_features.device
        idx = self.build_indexes(lambda i0: i0.mark(coors), batch_size, input_shape, dev)
        row_of = idx[0].lookup(coors)
        feats0 = hip_ops.rows_permute(voxel_features.float().contiguous(), row_of,
                                      spconv.pad_channels(voxel_features.shape[1]), self.compute_dtype, n_rows=idx[0].n)
        bev, levels = self.run_fused(idx, feats0)
        multi = {}
        for name, lvl, c in (("conv1", 0, 16), ("conv2", 1, 32), ("conv3", 2, 64), ("conv4", 3, 128)):
            f, ix = levels[lvl]
            multi[name] = spconv.SparseConvTensor(f[:, :c], None, ix.spatial_shape, batch_size, _index=ix)
        return bev, multi


@BACKBONES.register_module
class PointPillarsScatter(nn.Module):
    """det3d/models/readers/pillar_encoder.py:167-221: pillar rows -> [B, C, ny, nx] pseudo image (fd_pillar_scatter:
    cell = y*nx + x as in :204-214; written NCHW, or channels-last for the bf16 convolution path)."""

    def __init__(self, num_input_features=64, norm_cfg=None, name="PointPillarsScatter", **kwargs):
        super().__init__()
        self.name = "PointPillarsScatter"
        self.nchannels = num_input_features
        self.dense_channels_last = True

    def forward(self, voxel_features, coords, batch_size, input_shape, n_dev=None):
        self.nx = int(input_shape[0])
        self.ny = int(input_shape[1])
        return hip_ops.pillar_scatter(voxel_features, coords.int().contiguous(), n_dev, batch_size, self.ny, self.nx,
                                      channels_last=self.dense_channels_last)
