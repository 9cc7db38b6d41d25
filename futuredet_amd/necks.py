"""RPN neck (det3d/models/necks/rpn.py:22-159).  Same constructor, same state_dict keys
(blocks.{i}.{1,4,...}.weight, deblocks.{i}.0.weight, ...).  In eval mode the stack runs with BatchNorm folded
into the convolutions (ZeroPad2d merged into the conv padding) on the hand-written MFMA convolutions (dense_bf16.py:
NHWC, bf16 or fp32, concat and transposed convolution written in place).  That plan is the ONLY eval-mode device path: a stack
it cannot take (a channel count that is not a multiple of the kernels' granule) raises instead of running anywhere else.
Training mode and host tensors (the CPU tests of the state-dict surface) run the plain nn.Module stack, forward_modules."""
import logging

import numpy as np
import torch
from torch import nn

from .nn_utils import Sequential, build_norm_layer, weights_version
from .registry import NECKS



def _drop_caches(module, incompatible_keys=None):
    module._plan = None
    module.__dict__.pop("_wv_tensors", None)


@NECKS.register_module
class RPN(nn.Module):
    def __init__(self, layer_nums, ds_layer_strides, ds_num_filters, us_layer_strides, us_num_filters,
                 num_input_features, norm_cfg=None, name="rpn", logger=None, **kwargs):
        super().__init__()
        # constructor keywords of det3d/models/necks/rpn.py:24-36; the module layout below (blocks.<i>.<j>, deblocks.<i>.<j>) is what
        # fixes the state-dict keys of the reference's checkpoints
        self.stage_strides, self.stage_filters, self.stage_layers = list(ds_layer_strides), list(ds_num_filters), list(layer_nums)
        self.up_strides, self.up_filters = list(us_layer_strides), list(us_num_filters)
        self.norm_cfg = norm_cfg if norm_cfg is not None else dict(type="BN", eps=1e-3, momentum=0.01)
        n_stage, n_up = len(self.stage_layers), len(self.up_strides)
        if not (len(self.stage_strides) == len(self.stage_filters) == n_stage and len(self.up_filters) == n_up <= n_stage):
            raise ValueError("RPN: ds_layer_strides / ds_num_filters / layer_nums need one entry per stage, us_* one per up-sampled stage")
        self.first_up = n_stage - n_up  # stages in front of the first one that has a deblock
        # every deblock must bring its stage to ONE common resolution (their outputs are concatenated)
        scales = {float(self.up_strides[j]) / float(np.prod(self.stage_strides[: self.first_up + j + 1])) for j in range(n_up)}
        if len(scales) > 1:
            raise ValueError("RPN: the deblocks end at different resolutions %s" % sorted(scales))
        blocks, deblocks = [], []
        cin = num_input_features
        for i in range(n_stage):
            blocks.append(self._make_layer(cin, self.stage_filters[i], self.stage_layers[i], stride=self.stage_strides[i]))
            cin = self.stage_filters[i]
            j = i - self.first_up
            if j >= 0:
                up, cout = self.up_strides[j], self.up_filters[j]
                if up > 1:  # transposed convolution up, or (stride < 1, the PointPillars neck) a strided convolution down
                    conv = nn.ConvTranspose2d(cin, cout, up, stride=up, bias=False)
                else:
                    down = int(round(1.0 / up))
                    conv = nn.Conv2d(cin, cout, down, stride=down, bias=False)
                deblocks.append(Sequential(conv, build_norm_layer(self.norm_cfg, cout)[1], nn.ReLU()))
        self.blocks = nn.ModuleList(blocks)
        self.deblocks = nn.ModuleList(deblocks)
        self.compute_dtype = torch.float32
        self._plan = None
        self.register_load_state_dict_post_hook(_drop_caches)
        (logger or logging.getLogger("RPN")).info("Finish RPN Initialization")

    @property
    def downsample_factor(self):
        factor = np.prod(self.stage_strides)
        if self.up_strides:
            factor /= self.up_strides[-1]
        return factor

    def _make_layer(self, inplanes, planes, num_blocks, stride=1):
        block = Sequential(nn.ZeroPad2d(1), nn.Conv2d(inplanes, planes, 3, stride=stride, bias=False),
                           build_norm_layer(self.norm_cfg, planes)[1], nn.ReLU())
        for _ in range(num_blocks):
            block.add(nn.Conv2d(planes, planes, 3, padding=1, bias=False))
            block.add(build_norm_layer(self.norm_cfg, planes)[1])
            block.add(nn.ReLU())
        return block

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight)

    def forward_modules(self, x):
        ups = []
        for i in range(len(self.blocks)):
            x = self.blocks[i](x)
            if i >= self.first_up:
                ups.append(self.deblocks[i - self.first_up](x))
        if len(ups) > 0:
            x = torch.cat(ups, dim=1)
        return x

    invalidate_caches = _drop_caches

    def _apply(self, fn, *a, **kw):  # .to() / .cuda() / .float(): derived weights live on the old device / dtype
        _drop_caches(self)
        return super()._apply(fn, *a, **kw)

    def forward(self, x):
        if self.training or not x.is_cuda:
            return self.forward_modules(x)
        if self.compute_dtype not in (torch.bfloat16, torch.float32):
            raise ValueError("RPN: compute_dtype must be float32 or bfloat16, got %s" % (self.compute_dtype,))
        # hand-written MFMA convolutions on NHWC activations (bf16 or fp32); returned as an NCHW-shaped view of the NHWC buffer
        ver = (weights_version(self), self.compute_dtype)
        if self._plan is None or self._plan[0] != ver:
            from .dense_bf16 import RPNPlan

            self._plan = (ver, RPNPlan(self, self.compute_dtype))  # raises ValueError for a stack the kernels do not take
        xin = x.to(self.compute_dtype).permute(0, 2, 3, 1).contiguous()  # no copy for a channels-last BEV map
        return self._plan[1](xin).permute(0, 3, 1, 2)
