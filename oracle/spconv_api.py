"""spconv-1.0 Python surface restated over the CPU oracle.  TEST INFRASTRUCTURE ONLY.

spconv 1.0 (github.com/neeharperi/spconv; README.md:26,31,59 of the reference) is not in the
image.  This module restates the part of its public surface that the reference touches
(det3d/models/backbones/scn.py:2-3,13-21,37,98-165): SparseConvTensor, SparseModule,
SparseSequential, SubMConv3d, SparseConv3d with spconv's keyword names
(in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
indice_key=None), weight Parameter of shape (*kernel_size, Cin, Cout), rulebook cached per
indice_key, `.dense()` -> [B,C,D,H,W].  It is used (a) by tests/golden/make_golden.py, registered
under the name ``spconv`` so the reference's unmodified scn.py builds SpMiddleResNetFHD on it and
pins the backbone topology, and (b) by oracle/model.py.
"""
import math

import numpy as np
import torch
from torch import nn

from . import ops


def _triple(v):
    if isinstance(v, (list, tuple)):
        assert len(v) == 3
        return [int(x) for x in v]
    return [int(v)] * 3


class SparseConvTensor(object):
    def __init__(self, features, indices, spatial_shape, batch_size, grid=None):
        self.features = features
        self.indices = indices
        self.spatial_shape = [int(v) for v in spatial_shape]
        self.batch_size = batch_size
        self.indice_dict = {}
        self.grid = grid

    def find_indice_pair(self, key):
        if key is None:
            return None
        return self.indice_dict.get(key)

    def dense(self, channels_first=True):
        out = ops.dense(self.features.detach().numpy(), self.indices.numpy(), self.batch_size, self.spatial_shape)
        out = torch.from_numpy(out)
        if not channels_first:
            out = out.permute(0, 2, 3, 4, 1).contiguous()
        return out


class SparseModule(nn.Module):
    pass


class SparseConvolution(SparseModule):
    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, subm=False, output_padding=0, transposed=False, inverse=False,
                 indice_key=None):
        super().__init__()
        assert ndim == 3 and groups == 1 and not transposed and not inverse
        assert _triple(dilation) == [1, 1, 1]
        self.in_channels = in_channels
        self.out_channels = out_channels
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

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = self.in_channels * int(np.prod(self.kernel_size))
            bound = 1 / math.sqrt(fan_in)
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x):
        assert isinstance(x, SparseConvTensor)
        data = x.find_indice_pair(self.indice_key)
        if data is not None:
            out_idx, pairs, pnum, out_shape = data
        else:
            out_idx, pairs, pnum, out_shape = ops.rulebook(
                x.indices.numpy(), x.spatial_shape, self.kernel_size, self.stride, self.padding, self.subm)
            if self.indice_key is not None:
                x.indice_dict[self.indice_key] = (out_idx, pairs, pnum, out_shape)
        feats = ops.indice_conv(x.features.detach().numpy(), self.weight.detach().numpy(),
                                None if self.bias is None else self.bias.detach().numpy(),
                                pairs, pnum, out_idx.shape[0])
        out = SparseConvTensor(torch.from_numpy(feats), torch.from_numpy(out_idx), out_shape, x.batch_size)
        out.indice_dict = x.indice_dict
        out.grid = x.grid
        return out


class SparseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, indice_key=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias,
                         indice_key=indice_key)


class SubMConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, indice_key=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias,
                         subm=True, indice_key=indice_key)


class SparseSequential(SparseModule):
    """Applies SparseModules to the tensor and plain modules to ``.features`` (spconv 1.0 modules.py)."""

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
                if x.indices.shape[0] != 0:
                    x.features = m(x.features)
            else:
                x = m(x)
        return x
