"""spconv-1.0 module surface on the HIP kernels.

The reference's backbone (det3d/models/backbones/scn.py:2-3,13-21,37,98-165) is written against spconv 1.0:
``SparseConvTensor(features, indices, spatial_shape, batch_size)``, ``SubMConv3d`` / ``SparseConv3d`` with the
keyword names (in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
indice_key=None), weight ``Parameter(*kernel_size, Cin, Cout)``, ``SparseSequential``, ``.dense()``.  This module
keeps that surface; underneath, an active set is a ``hip_ops.SparseIndex`` (bitmap + prefix, rows spatially
sorted) and a convolution is one ``fd_spconv_apply`` launch on an output-stationary rulebook.

Row order: spconv leaves the output row order unspecified (it differs between its own CPU and GPU paths);
here rows are always in index order.  ``SparseConvTensor`` re-orders the features / indices it is given once,
at construction, so ``.features`` and ``.indices`` always agree.
"""
import math

import numpy as np
import torch
from torch import nn

from . import hip_ops

CH_ALIGN = 16  # MFMA K granularity of fd_spconv_apply; narrower inputs are zero padded


def _triple(v):
    if isinstance(v, (list, tuple, np.ndarray)):
        assert len(v) == 3
        return [int(x) for x in v]
    return [int(v)] * 3


def pad_channels(c):
    return max(CH_ALIGN, (c + CH_ALIGN - 1) // CH_ALIGN * CH_ALIGN)


class SparseConvTensor(object):
    def __init__(self, features, indices, spatial_shape, batch_size, grid=None, _index=None):
        self.spatial_shape = [int(v) for v in spatial_shape]
        self.batch_size = int(batch_size)
        self.indice_dict = {}
        self.grid = grid
        if _index is not None:  # internal: already in index order
            self.index = _index
            self.features = features
            return
        indices = indices.int().contiguous()
        D, H, W = self.spatial_shape
        idx = hip_ops.SparseIndex(self.batch_size, D, H, W, features.device)
        n_dev = torch.zeros((1,), dtype=torch.int32, device=features.device)
        idx.mark(indices)
        idx.scan(n_dev)
        idx.finalize(int(n_dev.cpu()[0]))
        row_of = idx.lookup(indices)
        self.index = idx
        self._true_channels = features.shape[1]
        self.features = hip_ops.rows_permute(features.float().contiguous(), row_of, features.shape[1], torch.float32,
                                             n_rows=idx.n)

    @property
    def indices(self):
        return self.index.coords

    def find_indice_pair(self, key):
        return self.indice_dict.get(key) if key is not None else None

    def dense(self, channels_first=True):
        feats = self.features.contiguous()
        out = hip_ops.densify(feats, self.index)  # [B, C*D, H, W], channel = c*D + d
        B, C, D = self.batch_size, feats.shape[1], self.index.D
        out = out.view(B, C, D, self.index.H, self.index.W)
        if not channels_first:
            out = out.permute(0, 2, 3, 4, 1).contiguous()
        return out


class SparseModule(nn.Module):
    pass


class SparseConvolution(SparseModule):
    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, subm=False, indice_key=None):
        super().__init__()
        assert ndim == 3 and groups == 1 and _triple(dilation) == [1, 1, 1]
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = _triple(kernel_size)
        self.stride = _triple(stride)
        self.padding = _triple(padding)
        self.subm = subm
        self.indice_key = indice_key
        self.weight = nn.Parameter(torch.Tensor(*self.kernel_size, in_channels, out_channels))
        if bias:
            self.bias = nn.Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()
        self._packed = {}
        self.register_load_state_dict_post_hook(lambda m, keys: m._packed.clear())

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = self.in_channels * int(np.prod(self.kernel_size))
            bound = 1 / math.sqrt(fan_in)
            nn.init.uniform_(self.bias, -bound, bound)

    def geometry(self):
        """(ksize, stride, pad) as the kernels use them: SubM forces stride 1 / pad k//2 (spconv 1.0)."""
        if self.subm:
            return self.kernel_size, [1, 1, 1], [k // 2 for k in self.kernel_size]
        return self.kernel_size, self.stride, self.padding

    def packed_weight(self, dtype, bn=None):
        """Fragment-ordered weights (+ fp32 bias) with the eval-mode BatchNorm ``bn`` that follows the conv folded
        in; channel counts are padded to multiples of 16.  Cached per (dtype, bn, device) and re-derived whenever one
        of the source tensors changed (their ``_version`` counters: load_state_dict, init, optimizer step, BN statistics)."""
        key = (dtype, id(bn), self.weight.device)
        ver = self.weight._version + (self.bias._version if self.bias is not None else 0)
        if bn is not None:
            ver += bn.weight._version + bn.bias._version + bn.running_mean._version + bn.running_var._version
        hit = self._packed.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1]
        K = int(np.prod(self.kernel_size))
        w = self.weight.detach().float().reshape(K, self.in_channels, self.out_channels)
        b = self.bias.detach().float() if self.bias is not None else None
        if bn is not None:
            from .nn_utils import fold_bn

            w, b = fold_bn(w, b, bn, 2)
        cin_p, cout_p = pad_channels(self.in_channels), pad_channels(self.out_channels)
        wp = torch.zeros((K, cin_p, cout_p), dtype=torch.float32, device=w.device)
        wp[:, : self.in_channels, : self.out_channels] = w
        bp = None
        if b is not None:
            bp = torch.zeros((cout_p,), dtype=torch.float32, device=w.device)
            bp[: self.out_channels] = b
            bp = bp.contiguous()
        packed = (hip_ops.pack_spconv_weight(wp, dtype), bp, cin_p, cout_p)
        self._packed[key] = (ver, packed)
        return packed

    def rulebook_for(self, x):
        data = x.find_indice_pair(self.indice_key)
        if data is not None:
            return data
        ks, st, pd = self.geometry()
        if self.subm:
            out_index = x.index
        else:
            out_index = x.index.downsample(ks, st, pd)
            n_dev = torch.zeros((1,), dtype=torch.int32, device=x.features.device)
            out_index.scan(n_dev)
            out_index.finalize(int(n_dev.cpu()[0]))
        nbr = x.index.rulebook(out_index, ks, st, pd)
        data = (out_index, nbr)
        if self.indice_key is not None:
            x.indice_dict[self.indice_key] = data
        return data

    def forward(self, x):
        assert isinstance(x, SparseConvTensor)
        if self.training and torch.is_grad_enabled() and self.weight.requires_grad:
            raise NotImplementedError("the HIP sparse convolution is inference-only (no autograd): call under "
                                      "torch.no_grad() / in eval mode; training is outside this path (SURVEY 2)")
        out_index, nbr = self.rulebook_for(x)
        wpk, bias, cin_p, cout_p = self.packed_weight(torch.float32)
        feats = x.features
        if feats.shape[1] != cin_p:
            feats = torch.nn.functional.pad(feats, (0, cin_p - feats.shape[1]))
        out = hip_ops.spconv_apply(feats.contiguous(), wpk, bias, nbr, out_index.n, cout_p)
        if cout_p != self.out_channels:
            out = out[:, : self.out_channels].contiguous()
        y = SparseConvTensor(out, None, out_index.spatial_shape, x.batch_size, grid=x.grid, _index=out_index)
        y.indice_dict = x.indice_dict
        return y


class SparseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias,
                         indice_key=indice_key)


class SubMConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, subm=True,
                         indice_key=indice_key)


class SparseSequential(SparseModule):
    """SparseModules see the tensor, plain modules see ``.features`` (spconv 1.0 modules.py)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for i, m in enumerate(args):
            self.add_module(str(i), m)
        for k, m in kwargs.items():
            self.add_module(k, m)

    def __getitem__(self, idx):
        return list(self._modules.values())[idx]

    def __len__(self):
        return len(self._modules)

    def add(self, module, name=None):
        self.add_module(str(len(self._modules)) if name is None else name, module)

    def forward(self, x):
        for m in self._modules.values():
            if isinstance(m, SparseModule):
                x = m(x)
            elif isinstance(x, SparseConvTensor):
                if x.features.shape[0] != 0:
                    x.features = m(x.features)
            else:
                x = m(x)
        return x
