"""bf16 execution plan of the RPN neck + CenterHead on the hand-written MFMA convolution (fd_conv2d_nhwc_bf16).

Activations are NHWC bf16 tensors; BatchNorm is folded; the RPN concat is written in place (channel offsets), the
2x2 stride-2 transposed convolution runs as four 1x1 convolutions writing interleaved pixels, and the six branches
of a SepHead run as one 64->384 convolution followed by one block-diagonal 384->(9+2T) convolution.
Built lazily from the torch modules (so state_dict loading is unchanged) and cached until the next load.
"""
import torch
from torch import nn

from . import hip_ops
from .nn_utils import fold_stack


class _Conv(object):
    def __init__(self, w_oihw, bias, stride, relu):
        self.cout, self.cin, self.ks, _ = w_oihw.shape
        self.stride, self.relu = stride, relu
        self.wpk = hip_ops.pack_conv2d_weight(w_oihw)
        self.bias = bias.float().contiguous() if bias is not None else None

    def __call__(self, x, out=None, co_off=0, **kw):
        return hip_ops.conv2d_nhwc_bf16(x, self.wpk, self.bias, self.cout, self.ks, self.stride, self.relu, out=out,
                                        co_off=co_off, **kw)


def _convs_from_stack(modules):
    out = []
    for f in fold_stack(modules, torch.float32, False):
        assert not f.transposed
        assert (f.weight.shape[-1] == 3 and f.padding == 1) or (f.weight.shape[-1] == 1 and f.padding == 0), \
            "only 3x3 pad 1 / 1x1 pad 0 convolutions appear in the RPN / head"
        out.append(_Conv(f.weight, f.bias, f.stride, f.relu))
    return out


class RPNPlan(object):
    def __init__(self, rpn):
        self.blocks = [_convs_from_stack(b._modules.values()) for b in rpn.blocks]
        self.start = rpn._upsample_start_idx
        self.deblocks = []
        for d in rpn.deblocks:
            mods = list(d._modules.values())
            f = fold_stack(mods, torch.float32, False)[0]
            if f.transposed:  # ConvTranspose2d(k = s): weight [Cin, Cout, k, k] -> k*k 1x1 convs on interleaved pixels
                k = f.weight.shape[-1]
                assert f.stride == k and f.padding == 0
                subs = [(_Conv(f.weight[:, :, dy, dx].t().contiguous()[:, :, None, None], f.bias, 1, f.relu), dy, dx)
                        for dy in range(k) for dx in range(k)]
                self.deblocks.append(("up", k, subs, f.weight.shape[1]))
            else:
                assert f.stride == 1, "strided down-sampling deblocks do not occur in the shipped configs"
                self.deblocks.append(("conv", 1, _Conv(f.weight, f.bias, 1, f.relu), f.weight.shape[0]))
        self.cout_total = sum(d[3] for d in self.deblocks)

    def __call__(self, x):  # x [B,H,W,C] bf16
        ups = None
        co = 0
        for i, stack in enumerate(self.blocks):
            for conv in stack:
                x = conv(x)
            j = i - self.start
            if j >= 0:
                kind, k, op, cout = self.deblocks[j]
                B, H, W, _ = x.shape
                if ups is None:
                    ups = torch.empty((B, H * k, W * k, self.cout_total), dtype=torch.bfloat16, device=x.device)
                assert ups.shape[1] == H * k and ups.shape[2] == W * k
                if kind == "conv":
                    op(x, out=ups, co_off=co)
                else:
                    for sub, dy, dx in op:
                        sub(x, out=ups, co_off=co, osy=k, osx=k, ooy=dy, oox=dx)
                co += cout
        return ups


class HeadPlan(object):
    """Standard CenterHead (no bev_map / forecast_feature): shared conv, then per task one fused first conv and one
    block-diagonal final conv."""

    def __init__(self, head):
        assert not head.bev_map and not head.forecast_feature
        self.shared = _convs_from_stack(head.shared_conv)
        self.tasks = []
        for task in head.tasks:
            names = list(task.heads)
            firsts, finals = [], []
            for h in names:
                st = fold_stack(list(getattr(task, h)._modules.values()), torch.float32, False)
                assert len(st) == 2
                firsts.append(st[0])
                finals.append(st[1])
            w1 = torch.cat([f.weight for f in firsts], 0)
            b1 = torch.cat([f.bias for f in firsts], 0)
            hc = firsts[0].weight.shape[0]
            couts = [f.weight.shape[0] for f in finals]
            w2 = torch.zeros((sum(couts), hc * len(names), 3, 3), dtype=torch.float32, device=w1.device)
            o = 0
            for i, f in enumerate(finals):
                w2[o:o + couts[i], i * hc:(i + 1) * hc] = f.weight
                o += couts[i]
            b2 = torch.cat([f.bias for f in finals], 0)
            self.tasks.append((_Conv(w1, b1, 1, True), _Conv(w2, b2, 1, False), names, couts))

    def __call__(self, x):  # x [B,H,W,C] bf16 -> list of dicts of NCHW float32 tensors
        for conv in self.shared:
            x = conv(x)
        rets = []
        for c1, c2, names, couts in self.tasks:
            z = c2(c1(x)).permute(0, 3, 1, 2).float()  # [B, sum(couts), H, W]
            d, o = {}, 0
            for name, c in zip(names, couts):
                d[name] = z[:, o:o + c]
                o += c
            rets.append(d)
        return rets
