"""Small module helpers that define the reference's state_dict key names, plus BN folding.

build_norm_layer : det3d/models/utils/norm.py:59-108  ("BN"->BatchNorm2d, "BN1d"->BatchNorm1d, default eps 1e-5)
Sequential.add() : det3d/models/utils/misc.py:22-95   (children named "0","1",...)
"""
import torch
from torch import nn

_NORMS = {"BN": ("bn", nn.BatchNorm2d), "BN1d": ("bn1d", nn.BatchNorm1d), "GN": ("gn", nn.GroupNorm)}


def build_norm_layer(cfg, num_features, postfix=""):
    assert isinstance(cfg, dict) and "type" in cfg
    cfg_ = dict(cfg)
    kind = cfg_.pop("type")
    if kind not in _NORMS:
        raise KeyError("Unrecognized norm type {}".format(kind))
    abbr, cls = _NORMS[kind]
    requires_grad = cfg_.pop("requires_grad", True)
    cfg_.setdefault("eps", 1e-5)
    if kind != "GN":
        layer = cls(num_features, **cfg_)
    else:
        layer = cls(num_channels=num_features, **cfg_)
    for p in layer.parameters():
        p.requires_grad = requires_grad
    return abbr + str(postfix), layer


class Sequential(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        for i, m in enumerate(args):
            self.add_module(str(i), m)
        for k, m in kwargs.items():
            if k in self._modules:
                raise ValueError("name exists.")
            self.add_module(k, m)

    def __getitem__(self, idx):
        if not (-len(self) <= idx < len(self)):
            raise IndexError("index {} is out of range".format(idx))
        return list(self._modules.values())[idx]

    def __len__(self):
        return len(self._modules)

    def add(self, module, name=None):
        if name is None:
            name = str(len(self._modules))
            if name in self._modules:
                raise KeyError("name exists")
        self.add_module(name, module)

    def forward(self, x):
        for m in self._modules.values():
            x = m(x)
        return x


def kaiming_init(module, mode="fan_out", nonlinearity="relu", bias=0):
    nn.init.kaiming_normal_(module.weight, mode=mode, nonlinearity=nonlinearity)
    if getattr(module, "bias", None) is not None:
        nn.init.constant_(module.bias, bias)


def weights_version(*modules):
    """Cheap change detector for derived-weight caches (folded BN, packed fragments, captured graphs).  Parameters and
    buffers bump ``_version`` on every in-place update -- optimizer steps, ``init_weights``, ``load_state_dict``, BN
    running statistics in train() -- so the sum over a module's tensors changes whenever one of them does (versions
    only grow).  A write through ``p.data`` bypasses the counter: call ``invalidate_caches()`` after one."""
    v = 0
    for m in modules:
        ts = m.__dict__.get("_wv_tensors")
        if ts is None:
            ts = list(m.parameters()) + list(m.buffers())
            m.__dict__["_wv_tensors"] = ts
        for t in ts:
            v += t._version
    return v


def bn_affine(bn, double=False):
    """Eval-mode BatchNorm as y = x * scale + shift.  Evaluated in float64 and rounded once (``double`` returns the float64
    values for a caller that folds them into weights): a float32 sqrt / divide differs by an ulp between devices, and one ulp
    of a folded weight that sits next to a bf16 rounding boundary becomes a whole bf16 ulp of that weight -- found by the
    teacher-forced bf16 test on one output channel of one RPN layer."""
    scale = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
    shift = bn.bias.detach().double() - bn.running_mean.detach().double() * scale
    return (scale, shift) if double else (scale.float(), shift.float())


def fold_bn(w, b, bn, out_axis):
    """conv weight ``w`` (float, output channels along ``out_axis``) and optional bias -> float32 (w', b') with the eval-mode
    BatchNorm ``bn`` folded in: w' = w * scale, b' = b * scale + shift, formed in float64 and rounded to float32 once."""
    scale, shift = bn_affine(bn, double=True)
    shape = [1] * w.dim()
    shape[out_axis] = -1
    w2 = (w.detach().double() * scale.view(shape)).float()
    b2 = ((b.detach().double() * scale if b is not None else torch.zeros_like(scale)) + shift).float()
    return w2, b2


class FoldedConv(object):
    """One conv (+folded BN) (+ReLU) of a dense stack: the folded float32 weight / bias and the geometry, as the convolution
    plan (dense_bf16.py) packs them.  (tools/torch_dense_ab.py runs the same records through F.conv2d for A/B timing.)"""

    def __init__(self, weight, bias, stride, padding, relu, transposed=False):
        self.weight, self.bias, self.stride, self.padding, self.relu, self.transposed = \
            weight, bias, stride, padding, relu, transposed


def fold_stack(modules, dtype, channels_last):
    """[ZeroPad2d?, Conv2d|ConvTranspose2d, BatchNorm2d?, ReLU?]* -> [FoldedConv]; ZeroPad2d(p) becomes conv padding."""
    out = []
    mods = list(modules)
    i = 0
    pend_pad = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.ZeroPad2d):
            pend_pad = int(m.padding[0])
            i += 1
            continue
        assert isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)), type(m)
        transposed = isinstance(m, nn.ConvTranspose2d)
        w = m.weight.detach().float()
        b = m.bias.detach().float() if m.bias is not None else None
        j = i + 1
        if j < len(mods) and isinstance(mods[j], nn.BatchNorm2d):
            w, b = fold_bn(w, b, mods[j], 1 if transposed else 0)
            j += 1
        relu = j < len(mods) and isinstance(mods[j], nn.ReLU)
        if relu:
            j += 1
        pad = int(m.padding[0]) + pend_pad
        pend_pad = 0
        w = w.to(dtype)
        if channels_last:
            w = w.contiguous(memory_format=torch.channels_last)
        out.append(FoldedConv(w, b.to(dtype) if b is not None else None, int(m.stride[0]), pad, relu, transposed))
        i = j
    return out
