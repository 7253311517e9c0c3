"""BioViL image encoder (ResNet-50 trunk + projector + spatial mean) forward and hand-written backward on the cxrk
kernels, exposed as one `torch.autograd.Function`.

Reference arithmetic: `health_multimodal/image/model/resnet.py:34-47` (stem, max-pool, layer1..4),
torchvision 0.10 `Bottleneck` (1x1 -> 3x3(stride) -> 1x1, BN after each, residual add, ReLU; 1x1-stride downsample
+ BN on the first block of a stage), `model.py:141-145` (trunk -> projector -> mean over (H, W)) and
`modules.py:43-47` (projector).  Eval-mode BatchNorm (running statistics) is the only mode the reference runs the encoder in
(`chexpert-get-embedding.py:41-42`) and the one everything below is organised around — gamma/beta (and every conv weight) still
receive gradients:

    y = conv(x, w)*s + t,  s = gamma*rsqrt(var+eps),  t = beta - mean*s
    dx = conv^T(dy, w*s);  dw = s * wgrad(x, dy);  dbeta = sum(dy);  dgamma = rstd * (<w, wgrad(x, dy)> - mean*sum(dy))

(dgamma = sum dy*xhat written through the raw weight gradient: nothing of the forward has to be re-read for it and gamma is
never divided by; measured against the direct sum on this network: < 1e-6 of the tensor maximum, scripts/exp_dgamma.py.)

Train-mode BatchNorm (`ImageModel.train()`, the state the reference's constructor leaves the model in, `model.py:119`) runs the same
GEMM kernels with an identity fold: batch statistics, normalisation + ReLU, running-statistics update and the batch-statistics terms
of the backward are separate kernels around them (`_conv_bn_train`, `through_bn` in `_backward`, csrc/bn_train.hip).

Working layout: activations NHWC, filters [Ko][R][S][C] (the parameters are kept in torch `channels_last` memory
format, so state-dict shapes stay OIHW).  The stem's 3 input channels are zero-padded to 4 (16-byte loads).

Two storage modes, chosen by the library's contraction precision (`_lib.get_precision()`):
  fp32        activations, folded filters and gradients are fp32 tensors; ReLU masks are taken from the sign of the saved
              activation (exact-fp32 MFMA mainloop).
  split_bf16  every activation / gradient / folded filter that feeds a contraction is a `kernels.Planes` tensor (bf16 hi + lo
              planes, 4 bytes per element) written by the producing kernel's epilogue, so the MFMA mainloops load operands with
              no conversion work; ReLU decisions are saved as bit masks (1 bit per element) by the forward epilogues and read
              by the data-gradient epilogues.  The stem reads the fp32 image (fp32 gather, split on the fly), in the forward
              and in its weight gradient.
Parameter gradients are written (accumulated) straight into `param.grad` when that exists with the parameter's own memory
layout — the flat gradient buffer of `optim._FlatOptimizer` — so autograd has nothing to add afterwards.
"""
from __future__ import annotations

import os

from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from . import kernels as K
from .gradsink import GradSink
from .kernels import Planes

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


def _planes_mode() -> bool:
    return _lib.get_precision() == "split_bf16"


class _Fold:
    """Per-forward folded filters and BN vectors (one flat buffer each).  In planes mode the folded filters of every unit but
    the stem live in ONE [2, total] bf16 buffer (unit i = a column slice of both planes)."""

    def __init__(self, specs: Sequence[ConvSpec], device, pl: bool):
        self.pl = pl
        sizes = [s.cout * s.k * s.k * s.cpad for s in specs]
        tot_c = sum(s.cout for s in specs)
        self.vec = torch.empty(3, tot_c, dtype=torch.float32, device=device)
        self.woff, self.coff = [], []
        a = b = 0
        for s, n in zip(specs, sizes):
            self.woff.append(a)
            self.coff.append(b)
            a += (n + 7) // 8 * 8
            b += s.cout
        self.sizes = sizes
        self.train = False      # train-mode BatchNorm (batch statistics): filters folded with an identity BatchNorm, see `_conv_bn_train`
        self.tstate = None      # train mode with gradients: {unit: (raw convolution output z, batch mean, batch rstd)} for the backward
        if pl:
            self.wp = torch.empty(2, a, dtype=torch.bfloat16, device=device)
            self.w = torch.empty(sizes[0], dtype=torch.float32, device=device)    # the stem's filters stay fp32
        else:
            self.w = torch.empty(a, dtype=torch.float32, device=device)

    def ws(self, i, s):
        """folded filter of unit i: fp32 [Ko*R*S*cpad] (fp32 mode, and the stem always) or Planes [Ko, R*S*cpad]"""
        n = self.sizes[i]
        if self.pl and i > 0:
            return Planes(self.wp[:, self.woff[i]: self.woff[i] + n].view(2, s.cout, n // s.cout))
        return self.w[:n] if self.pl else self.w[self.woff[i]: self.woff[i] + n]

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


def _out_hw(s, H, W):
    return (H + 2 * s.pad - s.k) // s.stride + 1, (W + 2 * s.pad - s.k) // s.stride + 1


def _conv(i, s, fold, x, residual, relu, N, H, W, pl, want_mask=True):
    """unit i forward -> (y, mask).  planes mode: y Planes, mask = ReLU decision bits (uint8 [N*Ho*Wo, Ko/8]) when relu;
    fp32 mode: y fp32, mask None (the backward reads the sign of y)."""
    Ho, Wo = _out_hw(s, H, W)
    dev = fold.vec.device
    if not pl:
        y = torch.empty(N, Ho, Wo, s.cout, dtype=torch.float32, device=dev)
        K.conv_fwd(x, fold.ws(i, s), fold.shift(i, s), residual, y, N, H, W, s.cpad, s.cout, s.k, s.k, s.stride, s.pad, relu)
        return y, None
    y = Planes.empty(N, Ho, Wo, s.cout, device=dev)
    mask = torch.empty(N * Ho * Wo, s.cout // 8, dtype=torch.uint8, device=dev) if (relu and want_mask) else None
    K.conv_fwd_pl(x, fold.ws(i, s), fold.shift(i, s), residual, y, mask, N, H, W, s.cpad, s.cout, s.k, s.k, s.stride, s.pad, relu)
    return y, mask


def _conv_bn_train(i, s, fold, p, bufs, x, residual, relu, N, H, W, pl, want_mask, momentum):
    """Unit i in TRAIN mode (`torch.nn.BatchNorm2d`, training=True): z = conv(x, w) by the GEMM kernel (identity fold), batch
    statistics over all N*Ho*Wo pixels (one pass: per-block shifted sums merged with Chan's formula), y = relu(gamma (z - mean) rstd
    + beta + residual), running statistics updated with `momentum` (unbiased variance).  -> (y, mask)"""
    z, _ = _conv(i, s, fold, x, None, False, N, H, W, pl, want_mask=False)
    C = s.cout
    rows = z.numel() // C
    mean, var = K.colstats(z)                                        # one pass; biased variance, as the forward uses it
    scale, shift, rstd = K.bn_train_fwd_coeffs(mean, var, p[3 * i + 1], p[3 * i + 2], BN_EPS, rows, momentum, bufs[2 * i], bufs[2 * i + 1])
    y, mask = K.bn_apply(z, scale, shift, residual, relu, want_mask and pl)
    if fold.tstate is not None:
        fold.tstate[i] = (z, mean, rstd)
    return y, mask


def _forward(specs, blocks, p: Sequence[torch.Tensor], bufs: Sequence[torch.Tensor], x: torch.Tensor, save: bool,
             want_patch: bool, stages: Optional[list] = None, keep_stem: bool = False, bn_momentum: Optional[float] = None):
    """bn_momentum: None = eval-mode BatchNorm (running statistics folded into the filters, the reference's only use of the encoder);
    a float = train-mode BatchNorm with that momentum (batch statistics; `ImageModel.train()`)."""
    N, C, H, W = x.shape
    if C != 3:
        raise ValueError(f"ImageModel expects 3-channel input (ExpandChannels, transforms.py:12-38), got {C}")
    dev = x.device
    pl = _planes_mode()
    fold = _Fold(specs, dev, pl)
    train = bn_momentum is not None
    fold.train = train
    if train and save:
        fold.tstate = {}
    ident = {}
    for i, s in enumerate(specs):
        if train:   # identity BatchNorm: the GEMM kernels then produce the raw convolution output (scale 1, shift 0)
            if s.cout not in ident:
                one, zero = torch.ones(s.cout, device=dev), torch.zeros(s.cout, device=dev)
                ident[s.cout] = (one, zero, zero, one - BN_EPS)        # gamma, beta, mean, var: rsqrt(var + eps) = 1
            bn = ident[s.cout]
        else:
            bn = (p[3 * i + 1], p[3 * i + 2], bufs[2 * i], bufs[2 * i + 1])
        args = (_filter_rsc(p[3 * i]), *bn, BN_EPS, s.cout, s.k * s.k, s.cin, s.cpad)
        if pl and i > 0:
            K.bn_fold_pl(*args, fold.ws(i, s), fold.scale(i, s), fold.shift(i, s), fold.rstd(i, s))
        else:
            K.bn_fold(*args, fold.ws(i, s), fold.scale(i, s), fold.shift(i, s), fold.rstd(i, s))

    def unit(i, s, fold, x, residual, relu, N, H, W, pl, want_mask=True):   # conv + BatchNorm (+ residual, ReLU) of unit i in the mode of this pass
        if train:
            return _conv_bn_train(i, s, fold, p, bufs, x, residual, relu, N, H, W, pl, want_mask, bn_momentum)
        return _conv(i, s, fold, x, residual, relu, N, H, W, pl, want_mask)

    x0 = K.nchw_to_nhwc(x, 4)
    stem, _ = unit(0, specs[0], fold, x0, None, True, N, H, W, pl, want_mask=False)   # its ReLU mask = sign of the pooled value
    Hs, Ws = stem.shape[1], stem.shape[2]
    pooled, idx = K.maxpool_fwd_pl(stem) if pl else K.maxpool_fwd(stem)
    if pl and not keep_stem:
        stem = None                       # planes mode: the max-pool backward needs only `pooled`
    cur = pooled
    h, w = cur.shape[1], cur.shape[2]
    if stages is not None:
        stages.append(cur)
    binfo = []
    for bi, blk in enumerate(blocks):
        s1, s2, s3 = specs[blk["c1"]], specs[blk["c2"]], specs[blk["c3"]]
        o1, m1 = unit(blk["c1"], s1, fold, cur, None, True, N, h, w, pl)
        o2, m2 = unit(blk["c2"], s2, fold, o1, None, True, N, h, w, pl)
        h2, w2 = o2.shape[1], o2.shape[2]
        if blk["ds"] is not None:
            idt, _ = unit(blk["ds"], specs[blk["ds"]], fold, cur, None, False, N, h, w, pl)
        else:
            idt = cur
        out, m3 = unit(blk["c3"], s3, fold, o2, idt, True, N, h2, w2, pl)
        if save:
            # fp32 mode keeps `out` for its sign; planes mode keeps the three bit masks instead (out lives on as the next `cur`)
            binfo.append((cur, o1, o2, out if not pl else None, h, w, h2, w2, m1, m2, m3))
        cur, h, w = out, h2, w2
        if stages is not None and (bi + 1 == len(blocks) or blocks[bi + 1]["ds"] is not None):
            stages.append(cur)
    ip = len(specs) - 1
    pj1, mp = unit(ip, specs[ip], fold, cur, None, True, N, h, w, pl)
    w3, b3 = p[3 * len(specs)], p[3 * len(specs) + 1]
    w3m = w3.reshape(w3.shape[0], -1)
    if pl:
        w3p = K.split_planes(w3m)
        pj2 = K.linear_fwd_pl(pj1.view(N * h * w, pj1.shape[-1]), w3p, b3)
    else:
        w3p = None
        pj2 = K.linear_fwd(pj1.view(N * h * w, -1), w3m, b3)
    emb = K.spatial_mean_fwd(pj2.view(N, h * w, -1))
    patch = pj2.view(N, h, w, -1) if want_patch else None
    state = (fold, x0, stem, idx, pooled, binfo, cur, pj1, mp, w3p, (N, H, W, Hs, Ws, h, w), pl) if save else None
    return emb, patch, state


def _unit_params_bwd(i, s, fold, p, bufs, x, dy, sumdy, N, H, W, sink: GradSink, pl):
    """Parameter gradients of conv+BN unit i.  x: its input; dy: masked gradient w.r.t. its BN output; sumdy = sum of dy over
    pixels ([cout], reduced by the kernel that produced dy)."""
    w = _filter_rsc(p[3 * i])
    if fold.train:   # train-mode BatchNorm: dy is dz (gradient w.r.t. the raw convolution output), gamma / beta gradients are already
        #              written by `through_bn`; the kernel's own gamma / beta outputs go to scratch
        gw, acc = sink.dst(3 * i)
        dg = torch.empty(s.cout, dtype=torch.float32, device=w.device)
        db = torch.empty(s.cout, dtype=torch.float32, device=w.device)
    else:
        trip = [sink.dst(3 * i + k) for k in range(3)]
        if len({acc for _, acc in trip}) > 1:          # one accumulate switch per launch: all three direct, or all three fresh
            trip = [sink.dst(3 * i + k, force_fresh=True) for k in range(3)]
        (gw, acc), (dg, _), (db, _) = trip
    if not acc and not gw.permute(0, 2, 3, 1).is_contiguous():     # fresh tensor: give it the filter's [Ko][R][S][C] memory
        gw = torch.empty_like(w).permute(0, 3, 1, 2)
        sink.ret[3 * i] = gw
    dw = gw.permute(0, 2, 3, 1)
    args = (w, fold.scale(i, s), fold.rstd(i, s), bufs[2 * i], sumdy, dw, dg, db, acc, N, H, W)
    if pl and i > 0:
        K.conv_bwd_params_pl(x, dy, *args, s.cpad, s.cout, s.k, s.k, s.stride, s.pad)
    else:
        K.conv_bwd_params(x, dy, *args, s.cin, s.cpad, s.cout, s.k, s.k, s.stride, s.pad)


def _dgrad(i, s, fold, dy, residual, relu, N, H, W, pl, want_sums, residual_s2=False):
    """Data gradient of unit i, masked by `relu` (planes mode: the bit mask of the tensor the gradient flows into; fp32 mode:
    that tensor itself).  want_sums: also the column sums of the result (-> (dx, sums[C])).  residual_s2 (planes mode): the
    residual is the compact gradient of a stride-2 projection shortcut (kernels.conv1x1_s2_bwd_data_compact_pl)."""
    dev = fold.vec.device
    sums = torch.empty(s.cpad, dtype=torch.float32, device=dev) if want_sums else None
    if pl:
        dx = Planes.empty(N, H, W, s.cpad, device=dev)
        K.conv_bwd_data_pl(dy, fold.ws(i, s), residual, relu, dx, N, H, W, s.cpad, s.cout, s.k, s.k, s.stride, s.pad, sums,
                           residual_s2=residual_s2)
    else:
        dx = torch.empty(N, H, W, s.cpad, dtype=torch.float32, device=dev)
        K.conv_bwd_data(dy, fold.ws(i, s), residual, relu, dx, N, H, W, s.cpad, s.cout, s.k, s.k, s.stride, s.pad, sums)
    return (dx, sums) if want_sums else dx


def stage_of_param(name: str) -> str:
    """Stage tag of an image-encoder parameter, in the order the backward completes them: "head" (projector + layer4),
    "layer3", "layer2", "stem" (layer1 + the stem)."""
    if name.startswith("projector.") or ".layer4." in name:
        return "head"
    if ".layer3." in name:
        return "layer3"
    if ".layer2." in name:
        return "layer2"
    return "stem"


def _backward(specs, blocks, p, bufs, state, demb: Optional[torch.Tensor], dpatch: Optional[torch.Tensor], sink: GradSink,
              on_grads_ready=None):
    """`on_grads_ready(tag)` (optional) is called on the current stream when every parameter gradient of a stage has been written
    in place: "head" after the projector and layer4, "layer3", "layer2", and "stem" (layer1 + stem) at the end — only while no
    gradient of the stage had to be returned to autograd as a fresh tensor."""
    fold, x0, stem, idx, pooled, binfo, last, pj1, mp, w3p, (N, H, W, Hs, Ws, h, w), pl = state
    def params_bwd(i, s_, x_, dy_, sum_, N_, H_, W_, pl_):
        # (Running these launches on a side stream, beside the data-gradient chain they depend on but which does not depend on them,
        #  was measured in round 3: 158.74 against 158.66 / 158.94 ms per step, bit-identical results — the GPU is not idle at kernel
        #  tails, the step is the sum of its kernels' times.  Not kept.)
        _unit_params_bwd(i, s_, fold, p, bufs, x_, dy_, sum_, N_, H_, W_, sink, pl_)

    def report(stage):
        if on_grads_ready is not None:
            on_grads_ready(stage)

    tstate = fold.tstate

    def through_bn(i, s_, dy_, sum_):
        """TRAIN-mode BatchNorm of unit i: the gradient w.r.t. its output (already masked by the unit's ReLU) and its column sums ->
        the gradient w.r.t. the raw convolution output, dz = gamma rstd (dy - mean(dy) - xhat mean(dy xhat)); writes dgamma = sum dy
        xhat and dbeta = sum dy.  Eval mode: the identity (the running statistics are constants folded into the filters)."""
        if tstate is None:
            return dy_, sum_
        z, mean, rstd = tstate.pop(i)
        C = s_.cout
        rows = z.numel() // C
        (dg, a1), (db, a2) = sink.dst(3 * i + 1), sink.dst(3 * i + 2)
        if a1 != a2:
            (dg, a1), (db, a2) = sink.dst(3 * i + 1, True), sink.dst(3 * i + 2, True)
        dot = K.coldot(dy_, z, mean)                                # sum dy (z - mean): centred before it is summed
        A, B, Cc = K.bn_train_bwd_coeffs(p[3 * i + 1], mean, rstd, sum_, dot, rows, dg, db, a1)
        return K.bn_train_dz(dy_, z, A, B, Cc), torch.zeros(C, dtype=torch.float32, device=dev)   # sum of dz over the pixels is 0

    ns = len(specs)
    ip = ns - 1
    w3 = p[3 * ns]
    w3m = w3.reshape(w3.shape[0], -1)
    J = w3m.shape[0]
    dev = w3.device
    P = h * w
    # ---- projector head: pj2 = pj1 @ w3^T + b3, emb = mean over the P patches
    gw3, acc3 = sink.dst(3 * ns)
    gb3, accb3 = sink.dst(3 * ns + 1)
    gw3m = gw3.permute(0, 2, 3, 1).reshape(J, -1) if gw3.dim() == 4 else gw3.reshape(J, -1)   # [J, C, 1, 1] is [J, C] in memory
    assert gw3m.data_ptr() == gw3.data_ptr()
    if pl:
        if demb is not None:
            dpj2 = K.spatial_mean_bwd_pl(demb.contiguous(), P, add=dpatch.reshape(N, P, J) if dpatch is not None else None).view(N * P, J)
        else:
            dpj2 = K.split_planes(dpatch.reshape(N * P, J).contiguous())
        pj1m = pj1.view(N * P, pj1.shape[-1])
        K.linear_bwd_weight_pl(dpj2, pj1m, gw3m, accumulate=acc3)
        K.colsum_pl(dpj2, gb3, accumulate=accb3)
        g = K.linear_bwd_data_pl(dpj2, w3p, maskin=mp, out_planes=True)
        sum_p = K.colsum_pl(g, torch.empty(g.shape[1], dtype=torch.float32, device=dev))
        g = g.view(N, h, w, g.shape[1])
    else:
        if demb is not None:
            dpj2 = K.spatial_mean_bwd(demb.contiguous(), P).view(N * P, J)
            if dpatch is not None:
                dpj2 = dpj2 + dpatch.reshape(N * P, J)
        else:
            dpj2 = dpatch.reshape(N * P, J).contiguous()
        pj1m = pj1.view(N * P, -1)
        K.linear_bwd_weight(dpj2, pj1m, gw3m, accumulate=acc3)
        K.colsum(dpj2, gb3, accumulate=accb3)
        g = K.linear_bwd_data(dpj2, w3m, aux=pj1m, auxmode=K.AUX_RELU_MASK)
        sum_p = K.colsum(g, torch.empty(g.shape[1], dtype=torch.float32, device=dev))
        g = g.view(N, h, w, -1)
    g, sum_p = through_bn(ip, specs[ip], g, sum_p)
    params_bwd(ip, specs[ip], last, g, sum_p, N, h, w, pl)

    nb = len(blocks)

    def relu_of_block_out(bi):
        """what masks a gradient flowing into block bi's output: its bit mask (planes) or the output itself (fp32)"""
        return binfo[bi][10] if pl else binfo[bi][3]

    g, gs = _dgrad(ip, specs[ip], fold, g, None, relu_of_block_out(nb - 1), N, h, w, pl, True)
    for bi in reversed(range(nb)):
        blk = blocks[bi]
        cur, o1, o2, out, hi, wi, h2, w2, m1, m2, m3 = binfo[bi]
        s1, s2, s3 = specs[blk["c1"]], specs[blk["c2"]], specs[blk["c3"]]
        # out = relu(bn3(conv3(o2)) + identity): g (already masked by out > 0) is dy of bn3 and of the downsample BN; its
        # channel sums gs were reduced by the kernel that produced g
        g3, gs3 = through_bn(blk["c3"], s3, g, gs)       # (eval mode: g itself; the downsample BatchNorm below sees the same g)
        params_bwd(blk["c3"], s3, o2, g3, gs3, N, h2, w2, pl)
        d2, q2 = _dgrad(blk["c3"], s3, fold, g3, None, m2 if pl else o2, N, h2, w2, pl, True)
        del g3
        d2, q2 = through_bn(blk["c2"], s2, d2, q2)
        params_bwd(blk["c2"], s2, o1, d2, q2, N, hi, wi, pl)
        d1, q1 = _dgrad(blk["c2"], s2, fold, d2, None, m1 if pl else o1, N, hi, wi, pl, True)
        del d2
        d1, q1 = through_bn(blk["c1"], s1, d1, q1)
        params_bwd(blk["c1"], s1, cur, d1, q1, N, hi, wi, pl)
        if blk["ds"] is not None:
            sd = specs[blk["ds"]]
            gd, gsd = through_bn(blk["ds"], sd, g, gs)
            params_bwd(blk["ds"], sd, cur, gd, gsd, N, hi, wi, pl)
            res = None
            if pl and sd.k == 1 and sd.stride == 2 and sd.pad == 0 and os.environ.get("CXRK_S2RES", "1") != "0":
                # a stride-2 projection: its data gradient is non-zero at the even pixels only -> compact [N, hi/2, wi/2, C], added
                # by the epilogue of conv1's data gradient at those pixels (instead of zero-filling, writing and re-reading a
                # full-resolution tensor that is 3/4 zeros: 12.8 -> 2.8 GB per step at batch 1024)
                res = K.conv1x1_s2_bwd_data_compact_pl(gd, fold.ws(blk["ds"], sd), N, hi, wi, sd.cpad, sd.cout)
                res_s2 = res is not None
            if res is None:
                res = _dgrad(blk["ds"], sd, fold, gd, None, None, N, hi, wi, pl, False)
                res_s2 = False
            del gd
        else:
            res, res_s2 = g, False
        if bi > 0:
            g, gs = _dgrad(blk["c1"], s1, fold, d1, res, relu_of_block_out(bi - 1), N, hi, wi, pl, True, residual_s2=res_s2)
        else:   # the block input is the max-pool output: its ReLU (the stem's) is applied by the max-pool backward
            g, gs = _dgrad(blk["c1"], s1, fold, d1, res, None, N, hi, wi, pl, False, residual_s2=res_s2), None
        del d1, res
        binfo[bi] = None
        if on_grads_ready is not None and blk["ds"] is not None and bi > 0:
            # the first block of layer4 / layer3 / layer2 is done: every gradient from its first unit onwards is complete
            stage = {LAYERS[0] + LAYERS[1] + LAYERS[2]: "head", LAYERS[0] + LAYERS[1]: "layer3", LAYERS[0]: "layer2"}[bi]
            if all(r is None for r in sink.ret[3 * blk["c1"]:]):
                report(stage)
    if pl:
        ds = K.maxpool_bwd_pl(g, idx, pooled, Hs, Ws)
    else:
        ds = K.maxpool_bwd(g, idx, stem, True)
    if _debug is not None:
        _debug["ds"], _debug["x0"] = ds, x0
    sum_s = K.colsum(ds.view(-1, ds.shape[-1]), torch.empty(ds.shape[-1], dtype=torch.float32, device=dev))
    ds, sum_s = through_bn(0, specs[0], ds, sum_s)
    params_bwd(0, specs[0], x0, ds, sum_s, N, H, W, False)   # fp32 operands (the image), split on the fly in split_bf16 mode
    if on_grads_ready is not None and all(r is None for r in sink.ret):
        report("stem")
    return sink.ret


class Decisions(list):
    """ReLU decisions of a forward pass (a list, in execution order) + the max-pool winners (`pool_taps`, NCHW uint8)."""
    pool_taps: Optional[torch.Tensor] = None


def relu_decisions(state) -> List[torch.Tensor]:
    """The 0/1 decision of every ReLU of a forward pass, in execution order (stem, relu1/relu2/relu_out per
    bottleneck, projector), as NCHW bool tensors on the CPU.  Used by the parity tests: gradients of a ReLU network
    are only comparable between two fp32 implementations under identical decisions (see oracle/ref_image.ReluPolicy).
    Planes mode keeps the decisions as bit masks; the stem's come from its output, which `capture_relu_decisions` makes the
    forward keep for this purpose."""
    fold, x0, stem, idx, pooled, binfo, cur, pj1, mp, w3p, dims, pl = state
    N, H, W, Hs, Ws, h, w = dims

    def nchw(t):
        return (t > 0).permute(0, 3, 1, 2).contiguous().cpu()

    taps = idx.permute(0, 3, 1, 2).contiguous().cpu()
    if not pl:
        acts = [stem]
        for b in binfo:
            acts += [b[1], b[2], b[3]]
        acts.append(pj1)
        out = Decisions(nchw(a) for a in acts)
        out.pool_taps = taps
        return out
    if stem is None:
        raise RuntimeError("relu_decisions: the stem output was not kept (run the forward inside capture_relu_decisions())")
    out = Decisions([nchw(stem.float())])
    out.pool_taps = taps
    for b in binfo:
        _, _, _, _, hi, wi, h2, w2, m1, m2, m3 = b
        for m, hh, ww in ((m1, hi, wi), (m2, h2, w2), (m3, h2, w2)):
            C = m.shape[1] * 8
            out.append(K.unpack_mask(m, C).view(N, hh, ww, C).permute(0, 3, 1, 2).contiguous())
    C = mp.shape[1] * 8
    out.append(K.unpack_mask(mp, C).view(N, h, w, C).permute(0, 3, 1, 2).contiguous())
    return out


def _pack_bits(a: torch.Tensor) -> torch.Tensor:
    """bool [rows, C] (device) -> the ReLU bit-mask layout of the epilogues: uint8 [rows, C/8], bit c % 8 of byte c / 8."""
    rows, C = a.shape
    w = 1 << torch.arange(8, device=a.device, dtype=torch.int32)
    return (a.view(rows, C // 8, 8).to(torch.int32) * w).sum(-1).to(torch.uint8)


def device_decisions(state) -> dict:
    """Every 0/1 decision of a saved forward pass, kept ON THE DEVICE in the layout the planes backward reads: per bottleneck the
    three ReLU bit masks, the projector's, the max-pool winners and the stem's ReLU decision at the winning input.  Works on the
    state of either storage mode; `impose_decisions_` writes such a set into a planes-mode state.  Test / bench instrumentation
    (full-size cross-precision gradient check): two correct forwards differ in the last bits, so a few 1e-5 of the decisions
    differ, and a gradient is only comparable under equal decisions."""
    fold, x0, stem, idx, pooled, binfo, cur, pj1, mp, w3p, dims, pl = state
    if pl:
        blocks = [(b[8], b[9], b[10]) for b in binfo]
        proj = mp
        stem_pos = pooled.t[0].float() > 0
    else:
        blocks = [tuple(_pack_bits((b[k] > 0).view(-1, b[k].shape[-1])) for k in (1, 2, 3)) for b in binfo]
        proj = _pack_bits((pj1 > 0).view(-1, pj1.shape[-1]))
        stem_pos = pooled > 0
    return {"blocks": [tuple(m.clone() for m in t) for t in blocks], "proj": proj.clone(), "pool_taps": idx.clone(), "stem_pos": stem_pos}


def count_decision_differences(a: dict, b: dict) -> dict:
    """How many decisions differ between two `device_decisions` sets (population counts of the XOR-ed masks)."""
    def bits(x, y):
        d = (x ^ y).to(torch.int32)
        n = 0
        for k in range(8):
            n += int(((d >> k) & 1).sum())
        return n
    relu = sum(bits(x, y) for ta, tb in zip(a["blocks"], b["blocks"]) for x, y in zip(ta, tb)) + bits(a["proj"], b["proj"])
    total = sum(x.numel() * 8 for ta in a["blocks"] for x in ta) + a["proj"].numel() * 8
    return {"relu": relu + int((a["stem_pos"] != b["stem_pos"]).sum()), "relu_total": total + a["stem_pos"].numel(),
            "pool_taps": int((a["pool_taps"] != b["pool_taps"]).sum()), "pool_total": a["pool_taps"].numel()}


def impose_decisions_(node, dec: dict) -> None:
    """Overwrite the decisions a planes-mode forward saved for its backward (`node` = the `grad_fn` of its output) with `dec`:
    the backward then differentiates the function the OTHER forward selected.  The stem's ReLU decision is read by the max-pool
    backward from the sign of the pooled activation's hi plane; it gets a stand-in tensor that carries the imposed signs, the
    pooled activation itself (the weight-gradient operand of layer1.0) stays untouched."""
    fold, x0, stem, idx, pooled, binfo, cur, pj1, mp, w3p, dims, pl = node.state
    if not pl:
        raise RuntimeError("impose_decisions_: only the planes (split_bf16) backward reads stored decisions")
    for b, t in zip(binfo, dec["blocks"]):
        for m, src in zip((b[8], b[9], b[10]), t):
            m.copy_(src)
    mp.copy_(dec["proj"])
    idx.copy_(dec["pool_taps"])
    sign = Planes(torch.zeros_like(pooled.t))
    sign.t[0].copy_(dec["stem_pos"].to(torch.bfloat16))
    node.state = (fold, x0, stem, idx, sign, binfo, cur, pj1, mp, w3p, dims, pl)


@torch.no_grad()
def calibrate_batchnorm_(specs, blocks, params, bufs, x: torch.Tensor) -> None:
    """Set the running statistics of every BatchNorm of the encoder to the statistics of its input over the batch `x` (what one
    train-mode pass with momentum 1 would leave behind; `torch.nn.BatchNorm2d` semantics: unbiased variance), unit by unit in
    execution order on the encoder's own kernels: the raw convolution output of a unit is obtained by folding an identity
    BatchNorm, then the unit is re-run with its new statistics to feed the next one.  Used to give SYNTHETIC weights the property
    every trained checkpoint has — statistics that match the activations — without which a random ResNet-50 in eval mode maps all
    images to nearly the same embedding (bench.py; the reference only ever loads trained BioViL weights, model.py:117-118)."""
    import math
    N, C, H, W = x.shape
    dev = x.device
    pl = _planes_mode()
    fold = _Fold(specs, dev, pl)
    p = [t.detach() for t in params]
    b = [t.detach() for t in bufs]
    k = math.sqrt(1.0 + BN_EPS)          # the identity fold scales by rsqrt(1 + eps)

    def fold_unit(i, s, gamma, beta, mean, var):
        args = (_filter_rsc(p[3 * i]), gamma, beta, mean, var, BN_EPS, s.cout, s.k * s.k, s.cin, s.cpad)
        fn = K.bn_fold_pl if (pl and i > 0) else K.bn_fold
        fn(*args, fold.ws(i, s), fold.scale(i, s), fold.shift(i, s), fold.rstd(i, s))

    def unit(i, xin, residual, relu, h, w, want_mask=True):
        s = specs[i]
        one, zero = torch.ones(s.cout, device=dev), torch.zeros(s.cout, device=dev)
        fold_unit(i, s, one, zero, zero, one)
        raw, _ = _conv(i, s, fold, xin, None, False, N, h, w, pl, want_mask=False)
        r = raw.view(-1, s.cout)                       # [pixels, channels], fp32 or planes
        rows = r.shape[0]
        # two-pass batch statistics on the path's own reductions; the identity fold scaled the output by 1 / k
        K.colsum(r, b[2 * i], alpha=1.0 / rows)                                       # mean of raw
        K.colvar(r, b[2 * i], b[2 * i + 1], alpha=k * k / max(1, rows - 1))           # unbiased variance of k * raw
        K.scale_mask(b[2 * i], alpha=k, out=b[2 * i])                                 # mean of k * raw
        del raw, r
        fold_unit(i, s, p[3 * i + 1], p[3 * i + 2], b[2 * i], b[2 * i + 1])
        return _conv(i, s, fold, xin, residual, relu, N, h, w, pl, want_mask=want_mask)[0]

    stem = unit(0, K.nchw_to_nhwc(x, 4), None, True, H, W, want_mask=False)
    cur = (K.maxpool_fwd_pl(stem) if pl else K.maxpool_fwd(stem))[0]
    del stem
    h, w = cur.shape[1], cur.shape[2]
    for blk in blocks:
        o1 = unit(blk["c1"], cur, None, True, h, w)
        o2 = unit(blk["c2"], o1, None, True, h, w)
        h2, w2 = o2.shape[1], o2.shape[2]
        idt = unit(blk["ds"], cur, None, False, h, w) if blk["ds"] is not None else cur
        cur, h, w = unit(blk["c3"], o2, idt, True, h2, w2), h2, w2
    unit(len(specs) - 1, cur, None, True, h, w)


_capture: Optional[list] = None
_debug: Optional[dict] = None      # diagnostics (scripts/exp_grad_err.py): set to a dict to receive the stem's gradient tensors


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
        specs, blocks, n_params, want_patch = meta[:4]
        ctx.set_materialize_grads(False)
        params, bufs = tensors[:n_params], tensors[n_params:]
        save = any(t.requires_grad for t in params)
        p = [t.detach() for t in params]
        b = [t.detach() for t in bufs]
        momentum = meta[5] if len(meta) > 5 else None     # None: eval-mode BatchNorm; float: train mode (ImageModel.train())
        emb, patch, state = _forward(specs, blocks, p, b, x.detach(), save, want_patch, keep_stem=_capture is not None, bn_momentum=momentum)
        if save:
            ctx.state, ctx.p, ctx.b, ctx.meta, ctx.params = state, p, b, meta, params
            if _capture is not None:
                _capture.append(relu_decisions(state))
        if patch is None:
            patch = emb.new_empty(0)
        return emb, patch

    @staticmethod
    def backward(ctx, demb, dpatch):
        specs, blocks, n_params, want_patch = ctx.meta[:4]
        hook = ctx.meta[4] if len(ctx.meta) > 4 else None    # per-call `on_grads_ready` (ImageModel.grad_ready_hook at forward time)
        if dpatch is not None and dpatch.numel() == 0:
            dpatch = None
        sink = GradSink(ctx.params)
        grads = _backward(specs, blocks, ctx.p, ctx.b, ctx.state, demb, dpatch, sink, hook)
        ctx.state = None
        return (None, None) + tuple(grads) + (None,) * len(ctx.b)


@torch.no_grad()
def forward_stages(specs, blocks, params, bufs, x: torch.Tensor) -> List[torch.Tensor]:
    """Diagnostic (parity tests): the trunk's stage outputs [max-pooled stem, layer1, layer2, layer3, layer4] as fp32 NCHW
    tensors, from the same kernels `ImageEncodeFn` runs."""
    stages: list = []
    _forward(specs, blocks, [t.detach() for t in params], [t.detach() for t in bufs], x, False, False, stages=stages)
    out = []
    for t in stages:
        f = t.float() if isinstance(t, Planes) else t
        out.append(K.nhwc_to_nchw(f.contiguous()))
    return out


@torch.no_grad()
def project_patches(specs, params, bufs, patch_nchw: torch.Tensor) -> torch.Tensor:
    """The projector alone (`modules.MLP`, reference modules.py:29-47): trunk patch embeddings [N,2048,h,w] -> projected patch
    embeddings [N,J,h,w], on the same kernels the encoder uses (fixture check against the reference's own MLP)."""
    N, C, h, w = patch_nchw.shape
    pl = _planes_mode()
    ip = len(specs) - 1
    s = specs[ip]
    p = [t.detach() for t in params]
    bufs = [t.detach() for t in bufs]
    dev = patch_nchw.device
    fold = _Fold(specs, dev, pl)
    args = (_filter_rsc(p[3 * ip]), p[3 * ip + 1], p[3 * ip + 2], bufs[2 * ip], bufs[2 * ip + 1], BN_EPS, s.cout, 1, s.cin, s.cpad)
    (K.bn_fold_pl if pl else K.bn_fold)(*args, fold.ws(ip, s), fold.scale(ip, s), fold.shift(ip, s), fold.rstd(ip, s))
    x = K.nchw_to_nhwc(patch_nchw.contiguous(), C)
    if pl:
        x = K.split_planes(x)
    pj1, _ = _conv(ip, s, fold, x, None, True, N, h, w, pl, want_mask=False)
    w3, b3 = p[3 * len(specs)], p[3 * len(specs) + 1]
    w3m = w3.reshape(w3.shape[0], -1)
    if pl:
        pj2 = K.linear_fwd_pl(pj1.view(N * h * w, pj1.shape[-1]), K.split_planes(w3m), b3)
    else:
        pj2 = K.linear_fwd(pj1.view(N * h * w, -1), w3m, b3)
    return K.nhwc_to_nchw(pj2.view(N, h, w, -1))
