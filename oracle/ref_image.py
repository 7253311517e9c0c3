"""BioViL image encoder restated with F.conv2d / F.batch_norm / F.max_pool2d (CPU fp32).
TEST INFRASTRUCTURE ONLY.

Follows `health_multimodal/image/model/resnet.py:25-47` (stem -> maxpool -> layer1..4, returns the
layer-4 map), `model.py:141-154` (`ImageModel.forward`: trunk -> projector -> spatial mean),
`model.py:197-205` (`ImageEncoder.forward`) and `modules.py:29-47` (projector: 1x1 conv without bias,
BatchNorm2d, ReLU, 1x1 conv with bias).  The trunk is torchvision 0.10 `ResNet(Bottleneck,[3,4,6,3])`
("v1.5": stride on the 3x3, expansion 4, BN eps 1e-5, downsample = 1x1 stride-s conv + BN on the first
block of every stage).  torchvision is not installed in the build image, so this restatement is
"parity unpinned" for the trunk; the projector is pinned against the reference's own
`modules.MLP` in `oracle/gen_golden.py`.

BatchNorm runs in eval mode (running statistics) by default: the only mode the reference ever uses
the encoder in (`chexpert-get-embedding.py:41-42`).
"""
from __future__ import annotations

from typing import Optional, Dict, List, Tuple

import torch
import torch.nn.functional as F

P = Dict[str, torch.Tensor]
LAYERS = (3, 4, 6, 3)
PLANES = (64, 128, 256, 512)
BN_EPS = 1e-5


class ReluPolicy:
    """ReLU with optionally imposed decisions.

    A ReLU network's gradient is discontinuous where a pre-activation is exactly at zero: two correct fp32
    implementations whose forward values differ in the last bit can take different sides of the kink for a handful of
    the ~5M activations of a ResNet-50 pass, and that single 0/1 decision then shifts every upstream gradient at the
    1e-3 level.  For gradient parity the oracle can therefore be run with the decisions (`masks`, in execution
    order: stem, then relu1/relu2/relu_out of every bottleneck, then the projector) taken from the implementation under
    test; `flips` / `max_flip_rel` then report how many decisions differ from the oracle's own and how close to zero
    (relative to the layer's max |pre-activation|) those pre-activations are."""

    def __init__(self, masks=None):
        self.masks, self.i, self.flips, self.count, self.max_flip_rel = masks, 0, 0, 0, 0.0
        # The stem max-pool is the same kind of discontinuity: where the two largest values of a 3x3 window are within rounding
        # of each other, two correct implementations route the window's gradient to different pixels, and because a weight gradient
        # is a sum of ~sqrt(n)-cancelling terms ONE such re-routing moves the stem's weight gradient by ~1/sqrt(n) (0.6 % at batch
        # 2).  `masks.pool_taps` (winning tap 3*r + s per pooled output, NCHW) imposes the implementation's winners;
        # `pool_flips` / `pool_max_gap` report how many differ from the oracle's own and by how much the values differ there.
        self.pool_taps = getattr(masks, "pool_taps", None)
        self.pool_flips, self.pool_max_gap = 0, 0.0

    def __call__(self, z: torch.Tensor) -> torch.Tensor:
        if self.masks is None:
            return F.relu(z)
        m = self.masks[self.i].to(z.device)
        self.i += 1
        diff = (z > 0) != m
        self.count += m.numel()
        if diff.any():
            self.flips += int(diff.sum())
            self.max_flip_rel = max(self.max_flip_rel, float(z[diff].abs().max() / z.abs().max()))
        return z * m.to(z.dtype)


    def maxpool(self, x: torch.Tensor) -> torch.Tensor:
        """nn.MaxPool2d(3, stride 2, padding 1) of the stem (resnet.py:37), optionally with imposed winners."""
        if self.pool_taps is None:
            return F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
        N, C, H, W = x.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        neg = torch.finfo(x.dtype).min
        win = F.unfold(F.pad(x, (1, 1, 1, 1), value=neg), kernel_size=3, stride=2).view(N, C, 9, Ho, Wo)
        taps = self.pool_taps.to(torch.int64).unsqueeze(2)
        out = win.gather(2, taps).squeeze(2)
        with torch.no_grad():
            best, arg = win.max(dim=2)
            diff = arg != taps.squeeze(2)
            self.pool_flips = int(diff.sum())
            if diff.any():
                self.pool_max_gap = float(((best - out)[diff]).abs().max() / x.abs().max())
        return out


_PLAIN = ReluPolicy()


# BatchNorm mode of the restatement: None = eval (running statistics, the reference's every use of the encoder); a float = train
# mode with that momentum (`torch.nn.BatchNorm2d`, training=True: what `ImageModel.train()` selects, model.py:119,131-139).
_BN_MOMENTUM: Optional[float] = None


class bn_training:
    """`with bn_training(0.1): ...` runs the restatement with train-mode BatchNorm: batch statistics in the forward (and their terms
    in autograd's backward), running statistics in the parameter dict updated in place."""

    def __init__(self, momentum: float = 0.1):
        self.momentum = momentum

    def __enter__(self):
        global _BN_MOMENTUM
        self._old, _BN_MOMENTUM = _BN_MOMENTUM, self.momentum
        return self

    def __exit__(self, *exc):
        global _BN_MOMENTUM
        _BN_MOMENTUM = self._old
        return False


def _bn(p: P, name: str, x: torch.Tensor, training: bool = False) -> torch.Tensor:
    if _BN_MOMENTUM is not None:
        return F.batch_norm(x, p[name + ".running_mean"], p[name + ".running_var"], p[name + ".weight"], p[name + ".bias"],
                            training=True, momentum=_BN_MOMENTUM, eps=BN_EPS)
    return F.batch_norm(x, p[name + ".running_mean"], p[name + ".running_var"], p[name + ".weight"],
                        p[name + ".bias"], training=training, eps=BN_EPS)


def bottleneck(p: P, pre: str, x: torch.Tensor, stride: int, has_down: bool, relu=_PLAIN) -> torch.Tensor:
    idt = x
    o = relu(_bn(p, pre + "bn1", F.conv2d(x, p[pre + "conv1.weight"])))
    o = relu(_bn(p, pre + "bn2", F.conv2d(o, p[pre + "conv2.weight"], stride=stride, padding=1)))
    o = _bn(p, pre + "bn3", F.conv2d(o, p[pre + "conv3.weight"]))
    if has_down:
        idt = _bn(p, pre + "downsample.1", F.conv2d(x, p[pre + "downsample.0.weight"], stride=stride))
    return relu(o + idt)


def resnet50_trunk(p: P, x: torch.Tensor, prefix: str = "encoder.encoder.", collect: List = None, relu=_PLAIN) -> torch.Tensor:
    x = F.conv2d(x, p[prefix + "conv1.weight"], stride=2, padding=3)
    x = relu(_bn(p, prefix + "bn1", x))
    x = relu.maxpool(x) if hasattr(relu, "maxpool") else F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    if collect is not None:
        collect.append(x)
    for li, (nblk, planes) in enumerate(zip(LAYERS, PLANES), start=1):
        for b in range(nblk):
            stride = 2 if (b == 0 and li > 1) else 1
            x = bottleneck(p, f"{prefix}layer{li}.{b}.", x, stride, has_down=(b == 0), relu=relu)
        if collect is not None:
            collect.append(x)
    return x


def projector(p: P, patch: torch.Tensor, prefix: str = "projector.model.", relu=_PLAIN) -> torch.Tensor:
    h = F.conv2d(patch, p[prefix + "0.weight"])
    h = relu(_bn(p, prefix + "1", h))
    return F.conv2d(h, p[prefix + "3.weight"], p[prefix + "3.bias"])


def image_model_forward(p: P, x: torch.Tensor, collect: List = None, relu=_PLAIN) -> torch.Tensor:
    """`ImageModel.forward` (model.py:141-154): returns the projected global embedding [B,128]."""
    patch = resnet50_trunk(p, x, collect=collect, relu=relu)
    proj = projector(p, patch, relu=relu)
    return proj.mean(dim=(2, 3))


def image_param_shapes(joint: int = 128) -> Tuple[Dict[str, tuple], Dict[str, tuple]]:
    """(parameters, buffers) name -> shape, in torchvision/BioViL state-dict naming."""
    prm: Dict[str, tuple] = {}
    buf: Dict[str, tuple] = {}

    def bn(name, c):
        prm[name + ".weight"] = (c,)
        prm[name + ".bias"] = (c,)
        buf[name + ".running_mean"] = (c,)
        buf[name + ".running_var"] = (c,)
        buf[name + ".num_batches_tracked"] = ()

    e = "encoder.encoder."
    prm[e + "conv1.weight"] = (64, 3, 7, 7)
    bn(e + "bn1", 64)
    inpl = 64
    for li, (nblk, planes) in enumerate(zip(LAYERS, PLANES), start=1):
        for b in range(nblk):
            pre = f"{e}layer{li}.{b}."
            prm[pre + "conv1.weight"] = (planes, inpl, 1, 1)
            bn(pre + "bn1", planes)
            prm[pre + "conv2.weight"] = (planes, planes, 3, 3)
            bn(pre + "bn2", planes)
            prm[pre + "conv3.weight"] = (planes * 4, planes, 1, 1)
            bn(pre + "bn3", planes * 4)
            if b == 0:
                prm[pre + "downsample.0.weight"] = (planes * 4, inpl, 1, 1)
                bn(pre + "downsample.1", planes * 4)
            inpl = planes * 4
    # torchvision's ResNet also owns an (unused here) fc layer; BioViL checkpoints carry it
    prm[e + "fc.weight"] = (1000, 2048)
    prm[e + "fc.bias"] = (1000,)
    pj = "projector.model."
    prm[pj + "0.weight"] = (joint, 2048, 1, 1)
    bn(pj + "1", joint)
    prm[pj + "3.weight"] = (joint, joint, 1, 1)
    prm[pj + "3.bias"] = (joint,)
    return prm, buf
