"""Autograd-enabled operators of the hot path (each a `torch.autograd.Function` over the cxrk kernels).

    l2_normalize                 F.normalize(x, dim=1)            modelling_cxrbert.py:138-139, vlp/inference_engine.py:51
    linear / mlp_adapter         nn.Linear, models.myMLP          models.py:7-26
    pairwise_cosine_similarity   torchmetrics call                Trainer.py:1682-1704
    group_mean                   prompt-embedding mean            Trainer.py:1665-1666
    posneg_bce_loss              pos-neg logits + BCEWithLogits   Trainer.py:575-583, ZERO_JOINT_BOUNDS.py:36
    infonce_loss                 north-star contrastive head (not in the reference), data-parallel aware
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import kernels as K


class _L2Norm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps):
        xhat, norm = K.l2norm_fwd(x.detach(), eps)
        ctx.save_for_backward(xhat, norm)
        return xhat

    @staticmethod
    def backward(ctx, g):
        xhat, norm = ctx.saved_tensors
        return K.l2norm_bwd(g, xhat, norm), None


def l2_normalize(x: torch.Tensor, eps: float = 1e-12) -> torch.Tensor:
    """F.normalize(x, p=2, dim=1) for a 2-D (or 1-D) tensor."""
    if x.dim() == 1:
        return _L2Norm.apply(x.unsqueeze(0), eps).squeeze(0)
    return _L2Norm.apply(x, eps)


class _Linear(torch.autograd.Function):
    """y = act(x W^T + b); act in {none, relu}.  Saves x and (for relu) y."""

    @staticmethod
    def forward(ctx, x, w, b, act):
        xd, wd = x.detach(), w.detach()
        if xd.dim() != 2:
            raise ValueError(f"linear: expected a 2-D input, got {tuple(xd.shape)}")
        xd = xd.contiguous()
        y = K.linear_fwd(xd, wd, None if b is None else b.detach(), act=act)
        ctx.act = act
        ctx.has_bias = b is not None
        ctx.save_for_backward(xd, wd, y if act == K.ACT_RELU else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        dy = dy.contiguous()
        if ctx.act == K.ACT_RELU:
            # mask the incoming gradient by the ReLU of this layer: dz = dy * (y > 0)
            dz = K.scale_mask(dy, mask_src=y)
        else:
            dz = dy
        dx = K.linear_bwd_data(dz, w) if ctx.needs_input_grad[0] else None
        dw = K.linear_bwd_weight(dz, x, torch.empty_like(w)) if ctx.needs_input_grad[1] else None
        db = K.colsum(dz, torch.empty(w.shape[0], dtype=torch.float32, device=w.device)) if ctx.has_bias else None
        return dx, dw, db, None


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, act: int = K.ACT_NONE):
    return _Linear.apply(x, weight, bias, act)


class _MLPAdapter(torch.autograd.Function):
    """models.myMLP: Linear(128,256) -> ReLU -> Linear(256,128), one Function so the ReLU mask rides the dgrad GEMM."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        xd = x.detach().contiguous()
        h = K.linear_fwd(xd, w1.detach(), b1.detach(), act=K.ACT_RELU)
        y = K.linear_fwd(h, w2.detach(), b2.detach())
        ctx.save_for_backward(xd, h, w1.detach(), w2.detach())
        return y

    @staticmethod
    def backward(ctx, dy):
        x, h, w1, w2 = ctx.saved_tensors
        dy = dy.contiguous()
        dev = dy.device
        dw2 = K.linear_bwd_weight(dy, h, torch.empty_like(w2))
        db2 = K.colsum(dy, torch.empty(w2.shape[0], dtype=torch.float32, device=dev))
        dh = K.linear_bwd_data(dy, w2, aux=h, auxmode=K.AUX_RELU_MASK)
        dw1 = K.linear_bwd_weight(dh, x, torch.empty_like(w1))
        db1 = K.colsum(dh, torch.empty(w1.shape[0], dtype=torch.float32, device=dev))
        dx = K.linear_bwd_data(dh, w1) if ctx.needs_input_grad[0] else None
        return dx, dw1, db1, dw2, db2


def mlp_adapter(x, w1, b1, w2, b2):
    return _MLPAdapter.apply(x, w1, b1, w2, b2)


class _PairwiseCosine(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y):
        xd, yd = x.detach().contiguous(), y.detach().contiguous()
        cosv, xn, yn = K.pairwise_cosine_fwd(xd, yd)
        ctx.save_for_backward(xd, yd, cosv, xn, yn)
        return cosv

    @staticmethod
    def backward(ctx, dcos):
        x, y, cosv, xn, yn = ctx.saved_tensors
        dx, dy = K.pairwise_cosine_bwd(x, y, cosv, dcos.contiguous(), xn, yn, need_dx=ctx.needs_input_grad[0])
        return dx, (dy if ctx.needs_input_grad[1] else None)


def pairwise_cosine_similarity(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """torchmetrics.functional.pairwise_cosine_similarity(x, y): [B,D] x [P,D] -> [B,P] (no epsilon)."""
    return _PairwiseCosine.apply(x, y)


class _PairwiseCosineMax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, groups):
        xd, yd = x.detach().contiguous(), y.detach().contiguous()
        cosv, xn, yn, mx, mean, arg = K.pairwise_cosine_max_fwd(xd, yd, groups)
        ctx.save_for_backward(xd, yd, cosv, xn, yn, arg)
        ctx.mark_non_differentiable(mean, arg)
        return mx, mean, arg

    @staticmethod
    def backward(ctx, dmax, _dmean, _darg):
        x, y, cosv, xn, yn, arg = ctx.saved_tensors
        dx, dy = K.pairwise_cosine_max_bwd(x, y, cosv, dmax, arg, xn, yn, need_dx=ctx.needs_input_grad[0])
        return dx, (dy if ctx.needs_input_grad[1] else None), None


def pairwise_cosine_max(x: torch.Tensor, y: torch.Tensor, groups: int = 1):
    """MAX_EMB scoring (`Trainer.py:1691-1693`) for `groups` prompt sets at once: x [B,D], y [groups*Pg, D] ->
    (max over each set's prompts [B,groups], their mean [B,groups] (logged only, not differentiable), winner index)."""
    return _PairwiseCosineMax.apply(x, y, groups)


class _GroupMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, groups, n):
        ctx.gn = (groups, n)
        return K.group_mean_fwd(x.detach(), groups, n)

    @staticmethod
    def backward(ctx, g):
        groups, n = ctx.gn
        return K.group_mean_bwd(g, groups, n), None, None


def group_mean(x: torch.Tensor, groups: int, n: int) -> torch.Tensor:
    """x [groups*n, D] -> [groups, D]: mean over the n prompts of each group."""
    return _GroupMean.apply(x, groups, n)


class _PosNegBCE(torch.autograd.Function):
    """cos [B,2C] -> scalar BCE-with-logits of (cos_pos - cos_neg), mean (scale 1) or sum (scale B*C) over the B*C logits; the
    gradient is computed in the same pass."""

    @staticmethod
    def forward(ctx, cosv, labels, diff, scale):
        logits, dcos, loss = K.bce_posneg_fwd_bwd(cosv.detach().contiguous(), labels, diff=diff, need_grad=True)
        ctx.save_for_backward(dcos)
        ctx.scale = float(scale)
        ctx.mark_non_differentiable(logits)
        if scale != 1.0:
            loss = K.scale_mask(loss.reshape(1), alpha=float(scale)).reshape(())
        return loss, logits

    @staticmethod
    def backward(ctx, gloss, _glogits):
        (dcos,) = ctx.saved_tensors
        return K.scale_mask(dcos, alpha_dev=gloss.reshape(()).contiguous(), alpha=ctx.scale), None, None, None


def posneg_bce_loss(cosv: torch.Tensor, labels: torch.Tensor, diff: bool = True, reduction: str = "mean") -> Tuple[torch.Tensor, torch.Tensor]:
    """(loss, logits).  cos column 2c = positive prompt of class c, 2c+1 = negative.  `reduction`: "mean" (`nn.BCEWithLogitsLoss()`,
    the reference's criterion, ZERO_JOINT_BOUNDS.py:36) or "sum"."""
    if reduction not in ("mean", "sum"):
        raise ValueError(f"posneg_bce_loss: reduction must be 'mean' or 'sum', got {reduction!r}")
    n_logits = cosv.shape[0] * (cosv.shape[1] // 2)
    return _PosNegBCE.apply(cosv, labels, diff, 1.0 if reduction == "mean" else float(n_logits))


# --------------------------------------------------------------------------------------------------------------
# InfoNCE (north star)
# --------------------------------------------------------------------------------------------------------------

def _all_gather_rows(t: torch.Tensor, group) -> torch.Tensor:
    """[rows, ...] contiguous on every rank -> [world * rows, ...], rank-major.  One collective, no staging copy on RCCL."""
    import torch.distributed as dist
    ws = dist.get_world_size(group)
    assert t.is_contiguous()
    out = torch.empty((ws * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    if t.is_cuda and dist.get_backend(group) == "gloo":
        # gloo has no GPU all-gather: stage through the host (rehearsals of the N>1 path on one GPU; RCCL takes the line below)
        host = torch.empty(out.shape, dtype=t.dtype)
        dist.all_gather_into_tensor(host, t.detach().cpu(), group=group)
        out.copy_(host)
        return out
    dist.all_gather_into_tensor(out, t, group=group)
    return out


class _InfoNCE(torch.autograd.Function):
    """loss = ( CE(S, diag) + CE(S^T, diag) ) / 2 with S = I_hat T_hat^T / tau over the GLOBAL batch.

    Data-parallel form (one process per GPU): rank r owns B rows of I and T.  One all-gather of the normalised
    [B, 2D] embeddings and one of the [B, 2] log-sum-exps is all the communication: every rank then holds what it
    needs to form the exact gradient of the global-mean loss w.r.t. its own rows — no reduce-scatter of embedding
    gradients and no second matmul exchange.
    """

    @staticmethod
    def forward(ctx, img, txt, temperature, group):
        import torch.distributed as dist
        dist_on = group is not None or (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)
        img, txt = img.detach().contiguous(), txt.detach().contiguous()
        B, D = img.shape
        if tuple(txt.shape) != (B, D):
            raise ValueError(f"infonce_loss: image embeddings are {tuple(img.shape)} but text embeddings {tuple(txt.shape)}")
        world_ = dist.get_world_size(group) if dist_on else 1
        if (B * world_) % 4 != 0 or D % 4 != 0:
            # the backward GEMMs contract over the global batch with 16-byte row accesses; fail HERE, before any collective
            # is issued, not inside loss.backward() with the other ranks already waiting in the gradient all-reduce
            raise ValueError(f"infonce_loss: global batch {B * world_} (= {B} rows x {world_} ranks) and embedding size {D} must be "
                             f"multiples of 4 (drop or pad the ragged last batch)")
        if dist_on:
            # image and text halves of ONE [B, 2D] send buffer, written in place by the normalisation kernels; the gathered
            # [Bg, 2D] buffer is read by the GEMMs through its column halves (row stride 2D): no cat / contiguous copies
            rank, world = dist.get_rank(group), dist.get_world_size(group)
            send = torch.empty(B, 2 * D, dtype=torch.float32, device=img.device)
            ih, inorm = K.l2norm_fwd(img, out=send[:, :D])
            th, tnorm = K.l2norm_fwd(txt, out=send[:, D:])
            both = _all_gather_rows(send, group)                           # [Bg, 2D]
            ih_all, th_all = both[:, :D], both[:, D:]
        else:
            rank, world = 0, 1
            ih, inorm = K.l2norm_fwd(img)
            th, tnorm = K.l2norm_fwd(txt)
            ih_all, th_all = ih, th
        Bg = B * world
        off = rank * B
        inv_tau = 1.0 / float(temperature)
        S1 = torch.empty(B, Bg, dtype=torch.float32, device=img.device)     # local images x all texts
        S2 = torch.empty(B, Bg, dtype=torch.float32, device=img.device)     # local texts  x all images
        K.gemm(ih, th_all, S1, B, Bg, D, False, True, alpha=inv_tau)
        K.gemm(th, ih_all, S2, B, Bg, D, False, True, alpha=inv_tau)
        loss = torch.zeros((), dtype=torch.float32, device=img.device)
        lse1, _ = K.infonce_row_lse(S1, off, loss_out=loss, loss_scale=0.5 / Bg, loss_accumulate=True)
        lse2, _ = K.infonce_row_lse(S2, off, loss_out=loss, loss_scale=0.5 / Bg, loss_accumulate=True)
        if dist_on:
            dist.all_reduce(loss, group=group)
            lse1_all, lse2_all = _all_gather_rows(lse1, group), _all_gather_rows(lse2, group)   # [Bg] each (4 KiB per rank)
        else:
            lse1_all, lse2_all = lse1, lse2
        ctx.save_for_backward(S1, S2, lse1, lse2, lse1_all, lse2_all, ih, th, ih_all, th_all, inorm, tnorm)
        ctx.meta = (off, inv_tau, Bg)
        return loss

    @staticmethod
    def backward(ctx, gloss):
        S1, S2, lse1, lse2, lse1_all, lse2_all, ih, th, ih_all, th_all, inorm, tnorm = ctx.saved_tensors
        off, inv_tau, Bg = ctx.meta
        B, D = ih.shape
        # dL/dS_ij = (softmax_row + softmax_col - 2*delta) / (2 Bg); fold 1/tau and the upstream scalar into alpha
        K.infonce_grad_inplace(S1, off, lse1, lse2_all)
        K.infonce_grad_inplace(S2, off, lse2, lse1_all)
        c = inv_tau * 0.5 / Bg
        dih = torch.empty(B, D, dtype=torch.float32, device=ih.device)
        dth = torch.empty(B, D, dtype=torch.float32, device=ih.device)
        K.gemm(S1, th_all, dih, B, D, Bg, False, False, alpha=c)
        K.gemm(S2, ih_all, dth, B, D, Bg, False, False, alpha=c)
        di = K.l2norm_bwd(dih, ih, inorm)
        dt = K.l2norm_bwd(dth, th, tnorm)
        g = gloss.reshape(()).contiguous()
        return K.scale_mask(di, alpha_dev=g, out=di), K.scale_mask(dt, alpha_dev=g, out=dt), None, None


def infonce_loss(img_emb: torch.Tensor, txt_emb: torch.Tensor, temperature: float = 0.07, group=None) -> torch.Tensor:
    """Symmetric InfoNCE over the global batch (all ranks of `group`, or the default group when initialised)."""
    return _InfoNCE.apply(img_emb, txt_emb, temperature, group)


@torch.no_grad()
def similarity_logits(img_emb: torch.Tensor, txt_emb: torch.Tensor, temperature: float = 1.0) -> torch.Tensor:
    """normalize(img) @ normalize(txt).T / tau — the zero-shot score matrix (trash/lower_bound_mcs.py:79-117)."""
    ih, _ = K.l2norm_fwd(img_emb.contiguous())
    th, _ = K.l2norm_fwd(txt_emb.contiguous())
    out = torch.empty(ih.shape[0], th.shape[0], dtype=torch.float32, device=ih.device)
    return K.gemm(ih, th, out, ih.shape[0], th.shape[0], ih.shape[1], False, True, alpha=1.0 / temperature)
