"""BioViL image encoder (ResNet-50 trunk + projector + spatial mean) forward and hand-written backward on the cxrk
kernels, exposed as one `torch.autograd.Function`.

Reference arithmetic: `health_multimodal/image/model/resnet.py:34-47` (stem, max-pool, layer1..4),
torchvision 0.10 `Bottleneck` (1x1 -> 3x3(stride) -> 1x1, BN after each, residual add, ReLU; 1x1-stride downsample
+ BN on the first block of a stage), `model.py:141-145` (trunk -> projector -> mean over (H, W)) and
`modules.py:43-47` (projector).  BatchNorm uses its running statistics — the only mode the reference runs the
encoder in (`chexpert-get-embedding.py:41-42`) — but gamma/beta (and every conv weight) still receive gradients:

    y = conv(x, w)*s + t,  s = gamma*rsqrt(var+eps),  t = beta - mean*s
    dx = conv^T(dy, w*s);  dw = s * wgrad(x, dy);  dbeta = sum(dy);  dgamma = sum(dy * (y_bn - beta)) / gamma

Working layout: activations NHWC, filters [Ko][R][S][C] (the parameters are kept in torch `channels_last` memory
format, so state-dict shapes stay OIHW).  The stem's 3 input channels are zero-padded to 4 (16-byte loads).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import kernels as K

LAYERS = (3, 4, 6, 3)
PLANES = (64, 128, 256, 512)
BN_EPS = 1e-5


class ConvSpec:
    __slots__ = ("conv", "bn", "cin", "cpad", "cout", "k", "stride", "pad", "widx", "off")

    def __init__(self, conv, bn, cin, cout, k, stride, pad):
        self.conv, self.bn, self.cin, self.cout, self.k, self.stride, self.pad = conv, bn, cin, cout, k, stride, pad
        self.cpad = (cin + 3) // 4 * 4


def resnet50_specs(prefix: str = "encoder.encoder.", joint: int = 128) -> Tuple[List[ConvSpec], List[dict]]:
    """Conv+BN units in execution order and the block wiring."""
    specs: List[ConvSpec] = [ConvSpec(prefix + "conv1", prefix + "bn1", 3, 64, 7, 2, 3)]
    blocks: List[dict] = []
    inpl = 64
    for li, (nblk, planes) in enumerate(zip(LAYERS, PLANES), start=1):
        for b in range(nblk):
            pre = f"{prefix}layer{li}.{b}."
            stride = 2 if (b == 0 and li > 1) else 1
            blk = {"c1": len(specs)}
            specs.append(ConvSpec(pre + "conv1", pre + "bn1", inpl, planes, 1, 1, 0))
            blk["c2"] = len(specs)
            specs.append(ConvSpec(pre + "conv2", pre + "bn2", planes, planes, 3, stride, 1))
            blk["c3"] = len(specs)
            specs.append(ConvSpec(pre + "conv3", pre + "bn3", planes, planes * 4, 1, 1, 0))
            blk["ds"] = None
            if b == 0:
                blk["ds"] = len(specs)
                specs.append(ConvSpec(pre + "downsample.0", pre + "downsample.1", inpl, planes * 4, 1, stride, 0))
            blocks.append(blk)
            inpl = planes * 4
    specs.append(ConvSpec("projector.model.0", "projector.model.1", 2048, joint, 1, 1, 0))
    return specs, blocks


def param_names(specs: Sequence[ConvSpec]) -> List[str]:
    names: List[str] = []
    for s in specs:
        names += [s.conv + ".weight", s.bn + ".weight", s.bn + ".bias"]
    return names + ["projector.model.3.weight", "projector.model.3.bias"]


def buffer_names(specs: Sequence[ConvSpec]) -> List[str]:
    names: List[str] = []
    for s in specs:
        names += [s.bn + ".running_mean", s.bn + ".running_var"]
    return names


class _Fold:
    """Per-forward folded filters and BN vectors (one flat buffer each)."""

    def __init__(self, specs: Sequence[ConvSpec], device):
        tot_w = sum(s.cout * s.k * s.k * s.cpad for s in specs)
        tot_c = sum(s.cout for s in specs)
        self.w = torch.empty(tot_w, dtype=torch.float32, device=device)
        self.vec = torch.empty(3, tot_c, dtype=torch.float32, device=device)
        self.woff, self.coff = [], []
        a = b = 0
        for s in specs:
            self.woff.append(a)
            self.coff.append(b)
            a += s.cout * s.k * s.k * s.cpad
            b += s.cout

    def ws(self, i, s):
        return self.w[self.woff[i]: self.woff[i] + s.cout * s.k * s.k * s.cpad]

    def scale(self, i, s):
        return self.vec[0, self.coff[i]: self.coff[i] + s.cout]

    def shift(self, i, s):
        return self.vec[1, self.coff[i]: self.coff[i] + s.cout]

    def rstd(self, i, s):
        return self.vec[2, self.coff[i]: self.coff[i] + s.cout]


def _filter_rsc(w: torch.Tensor) -> torch.Tensor:
    """OIHW parameter -> its [Ko][R][S][C] memory (no copy when the parameter is channels_last)."""
    v = w.permute(0, 2, 3, 1)
    return v if v.is_contiguous() else v.contiguous()


def _conv(i, s, fold, x, residual, relu, N, H, W):
    Ho = (H + 2 * s.pad - s.k) // s.stride + 1
    Wo = (W + 2 * s.pad - s.k) // s.stride + 1
    y = torch.empty(N, Ho, Wo, s.cout, dtype=torch.float32, device=x.device)
    K.conv_fwd(x, fold.ws(i, s), fold.shift(i, s), residual, y, N, H, W, s.cpad, s.cout, s.k, s.k, s.stride, s.pad, relu)
    return y


def _forward(specs, blocks, p: Sequence[torch.Tensor], bufs: Sequence[torch.Tensor], x: torch.Tensor, save: bool,
             want_patch: bool):
    N, C, H, W = x.shape
    if C != 3:
        raise ValueError(f"ImageModel expects 3-channel input (ExpandChannels, transforms.py:12-38), got {C}")
    dev = x.device
    fold = _Fold(specs, dev)
    for i, s in enumerate(specs):
        K.bn_fold(_filter_rsc(p[3 * i]), p[3 * i + 1], p[3 * i + 2], bufs[2 * i], bufs[2 * i + 1], BN_EPS, s.cout, s.k * s.k,
                  s.cin, s.cpad, fold.ws(i, s), fold.scale(i, s), fold.shift(i, s), fold.rstd(i, s))
    acts: Dict[str, torch.Tensor] = {}
    x0 = K.nchw_to_nhwc(x, 4)
    stem = _conv(0, specs[0], fold, x0, None, True, N, H, W)
    pooled, idx = K.maxpool_fwd(stem)
    cur = pooled
    h, w = cur.shape[1], cur.shape[2]
    binfo = []
    for blk in blocks:
        s1, s2, s3 = specs[blk["c1"]], specs[blk["c2"]], specs[blk["c3"]]
        o1 = _conv(blk["c1"], s1, fold, cur, None, True, N, h, w)
        o2 = _conv(blk["c2"], s2, fold, o1, None, True, N, h, w)
        h2, w2 = o2.shape[1], o2.shape[2]
        if blk["ds"] is not None:
            idt = _conv(blk["ds"], specs[blk["ds"]], fold, cur, None, False, N, h, w)
        else:
            idt = cur
        out = _conv(blk["c3"], s3, fold, o2, idt, True, N, h2, w2)
        if save:
            binfo.append((cur, o1, o2, out, idt if blk["ds"] is not None else None, h, w, h2, w2))
        cur, h, w = out, h2, w2
    ip = len(specs) - 1
    pj1 = _conv(ip, specs[ip], fold, cur, None, True, N, h, w)
    w3, b3 = p[3 * len(specs)], p[3 * len(specs) + 1]
    pj2 = K.linear_fwd(pj1.view(N * h * w, -1), w3.reshape(w3.shape[0], -1), b3)
    emb = K.spatial_mean_fwd(pj2.view(N, h * w, -1))
    patch = pj2.view(N, h, w, -1) if want_patch else None
    state = (fold, x0, stem, idx, pooled, binfo, cur, pj1, (N, H, W, h, w)) if save else None
    return emb, patch, state


def _unit_bwd(i, s, fold, p, bufs, x, dy, y, sub, N, H, W, grads, sums=None):
    """Parameter gradients of conv+BN unit i.  x: its input [N,H,W,cpad]; dy: masked gradient w.r.t. its BN output;
    (y - sub) equals the BN output wherever dy != 0 (y = post-ReLU output, sub = the residual that was added).
    `sums` = (sum dy, sum dy*(y_bn - beta)) when the data-gradient kernel that produced dy already reduced them."""
    if sums is None:
        sums = torch.empty(2, s.cout, dtype=torch.float32, device=dy.device)
        K.bn_bwd_reduce(dy, y, sub, p[3 * i + 2], sums[0], sums[1])
    w = _filter_rsc(p[3 * i])
    dw = torch.empty_like(w)
    dg, db = torch.empty_like(p[3 * i + 1]), torch.empty_like(p[3 * i + 2])
    K.conv_bwd_params(x, dy, w, fold.scale(i, s), fold.rstd(i, s), bufs[2 * i], sums[0], p[3 * i + 1], sums[1], dw, dg, db,
                      False, N, H, W, s.cin, s.cpad, s.cout, s.k, s.k, s.stride, s.pad)
    grads[3 * i] = dw.permute(0, 3, 1, 2)  # logical OIHW, channels_last strides (matches the parameter)
    grads[3 * i + 1], grads[3 * i + 2] = dg, db


def _dgrad(i, s, fold, dy, residual, relu_src, N, H, W, bn=None):
    """Data gradient of unit i.  bn = (sub, beta, beta2): also reduce, in the same epilogue, the BN-backward channel
    sums of the unit(s) that produced `relu_src` -> returns (dx, sums[3,C])."""
    dx = torch.empty(N, H, W, s.cpad, dtype=torch.float32, device=dy.device)
    if bn is None:
        K.conv_bwd_data(dy, fold.ws(i, s), residual, relu_src, dx, N, H, W, s.cpad, s.cout, s.k, s.k, s.stride, s.pad)
        return dx
    sums = torch.empty(3, s.cpad, dtype=torch.float32, device=dy.device)
    K.conv_bwd_data_bnsum(dy, fold.ws(i, s), residual, relu_src, dx, N, H, W, s.cpad, s.cout, s.k, s.k, s.stride, s.pad,
                          bn[0], bn[1], bn[2], sums)
    return dx, sums


def _backward(specs, blocks, p, bufs, state, demb: torch.Tensor, dpatch: Optional[torch.Tensor]):
    fold, x0, stem, idx, pooled, binfo, last, pj1, (N, H, W, h, w) = state
    grads: List[Optional[torch.Tensor]] = [None] * len(p)
    ns = len(specs)
    ip = ns - 1
    w3 = p[3 * ns]
    w3m = w3.reshape(w3.shape[0], -1)
    J = w3m.shape[0]
    if demb is not None:
        dpj2 = K.spatial_mean_bwd(demb.contiguous(), h * w).view(N * h * w, J)
        if dpatch is not None:
            dpj2 = dpj2 + dpatch.reshape(N * h * w, J)
    else:
        dpj2 = dpatch.reshape(N * h * w, J).contiguous()
    pj1m = pj1.view(N * h * w, -1)
    grads[3 * ns] = K.linear_bwd_weight(dpj2, pj1m, torch.empty_like(w3m)).view(w3.shape)
    grads[3 * ns + 1] = K.colsum(dpj2, torch.empty_like(p[3 * ns + 1]))
    g = K.linear_bwd_data(dpj2, w3m, aux=pj1m, auxmode=K.AUX_RELU_MASK).view(N, h, w, -1)
    _unit_bwd(ip, specs[ip], fold, p, bufs, last, g, pj1, None, N, h, w, grads)

    def beta(i):
        return p[3 * i + 2]

    def block_bn(bi):
        """(sub, beta, beta2) for the gradient w.r.t. block bi's output: its conv3 unit (y_bn = out - identity) and, when
        the identity is a downsample unit, that unit too (y_bn = identity)."""
        blk = blocks[bi]
        idt = binfo[bi][4] if blk["ds"] is not None else binfo[bi][0]
        return (idt, beta(blk["c3"]), beta(blk["ds"]) if blk["ds"] is not None else None)

    nb = len(blocks)
    g, gs = _dgrad(ip, specs[ip], fold, g, None, last, N, h, w, bn=block_bn(nb - 1))
    for bi in reversed(range(nb)):
        blk = blocks[bi]
        cur, o1, o2, out, idt, hi, wi, h2, w2 = binfo[bi]
        s1, s2, s3 = specs[blk["c1"]], specs[blk["c2"]], specs[blk["c3"]]
        # out = relu(bn3(conv3(o2)) + identity): g (already masked by out > 0) is dy of bn3 and of the downsample BN;
        # their channel sums gs were reduced by the kernel that produced g
        _unit_bwd(blk["c3"], s3, fold, p, bufs, o2, g, out, idt if idt is not None else cur, N, h2, w2, grads, sums=gs[0:2])
        d2, q2 = _dgrad(blk["c3"], s3, fold, g, None, o2, N, h2, w2, bn=(None, beta(blk["c2"]), None))
        _unit_bwd(blk["c2"], s2, fold, p, bufs, o1, d2, o2, None, N, hi, wi, grads, sums=q2[0:2])
        d1, q1 = _dgrad(blk["c2"], s2, fold, d2, None, o1, N, hi, wi, bn=(None, beta(blk["c1"]), None))
        del d2
        _unit_bwd(blk["c1"], s1, fold, p, bufs, cur, d1, o1, None, N, hi, wi, grads, sums=q1[0:2])
        if blk["ds"] is not None:
            sd = specs[blk["ds"]]
            _unit_bwd(blk["ds"], sd, fold, p, bufs, cur, g, idt, None, N, hi, wi, grads, sums=torch.stack([gs[0], gs[2]]))
            res = _dgrad(blk["ds"], sd, fold, g, None, None, N, hi, wi)
        else:
            res = g
        binfo_prev_bn = block_bn(bi - 1) if bi > 0 else None
        if binfo_prev_bn is not None:
            g, gs = _dgrad(blk["c1"], s1, fold, d1, res, cur, N, hi, wi, bn=binfo_prev_bn)
        else:
            g, gs = _dgrad(blk["c1"], s1, fold, d1, res, cur, N, hi, wi), None
        del d1, res
        binfo[bi] = None
    ds = K.maxpool_bwd(g, idx, stem, True)
    _unit_bwd(0, specs[0], fold, p, bufs, x0, ds, stem, None, N, H, W, grads)
    return grads


def relu_decisions(state) -> List[torch.Tensor]:
    """The 0/1 decision of every ReLU of a forward pass, in execution order (stem, relu1/relu2/relu_out per
    bottleneck, projector), as NCHW bool tensors on the CPU.  Used by the parity tests: gradients of a ReLU network
    are only comparable between two fp32 implementations under identical decisions (see oracle/ref_image.ReluPolicy)."""
    fold, x0, stem, idx, pooled, binfo, cur, pj1, _ = state
    acts = [stem]
    for b in binfo:
        acts += [b[1], b[2], b[3]]
    acts.append(pj1)
    return [(a > 0).permute(0, 3, 1, 2).contiguous().cpu() for a in acts]


_capture: Optional[list] = None


class capture_relu_decisions:
    """Test hook: `with capture_relu_decisions() as cap:` makes every grad-enabled image-encoder forward inside the block
    append its ReLU decisions (`relu_decisions`, CPU bool tensors) to `cap`.  Nothing is kept outside the block, and the
    saved-activation state itself is never referenced from module level (it lives on the autograd node only)."""

    def __enter__(self):
        global _capture
        self._old, _capture = _capture, []
        return _capture

    def __exit__(self, *exc):
        global _capture
        _capture = self._old
        return False


class ImageEncodeFn(torch.autograd.Function):
    """(x[N,3,H,W], meta, *params, *buffers) -> (global embedding [N,J], projected patch embeddings NHWC or None)."""

    @staticmethod
    def forward(ctx, x, meta, *tensors):
        specs, blocks, n_params, want_patch = meta
        ctx.set_materialize_grads(False)
        params, bufs = tensors[:n_params], tensors[n_params:]
        save = any(t.requires_grad for t in params)
        p = [t.detach() for t in params]
        b = [t.detach() for t in bufs]
        emb, patch, state = _forward(specs, blocks, p, b, x.detach(), save, want_patch)
        if save:
            ctx.state, ctx.p, ctx.b, ctx.meta = state, p, b, meta
            if _capture is not None:
                _capture.append(relu_decisions(state))
        ctx.mark_non_differentiable(*[])
        if patch is None:
            patch = emb.new_empty(0)
        return emb, patch

    @staticmethod
    def backward(ctx, demb, dpatch):
        specs, blocks, n_params, want_patch = ctx.meta
        if dpatch is not None and dpatch.numel() == 0:
            dpatch = None
        grads = _backward(specs, blocks, ctx.p, ctx.b, ctx.state, demb, dpatch)
        ctx.state = None
        return (None, None) + tuple(grads) + (None,) * len(ctx.b)
