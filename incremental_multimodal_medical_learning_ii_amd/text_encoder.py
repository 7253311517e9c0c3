"""CXR-BERT forward + hand-written backward on the cxrk kernels, exposed as one `torch.autograd.Function`.

Arithmetic follows HuggingFace `BertForMaskedLM` as the reference drives it
(`health_multimodal/text/model/modelling_cxrbert.py:87-99`: `hidden_states[-1][:, 0, :]` -> `BertProjectionHead`,
`:43-49`), post-LN, erf-GELU, LayerNorm eps = config.layer_norm_eps (1e-12), additive key mask, dropout inactive.
The MLM head the reference computes and discards on this path (`:87-95`) is not computed here.

Q, K and V projections run as ONE [3H, H] GEMM per layer: `fuse_qkv_` re-points the three nn.Parameter tensors
of a layer at consecutive slices of one buffer (state-dict names and values unchanged).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import kernels as K

LAYER_KEYS = (
    "attention.self.query.weight", "attention.self.query.bias",
    "attention.self.key.weight", "attention.self.key.bias",
    "attention.self.value.weight", "attention.self.value.bias",
    "attention.output.dense.weight", "attention.output.dense.bias",
    "attention.output.LayerNorm.weight", "attention.output.LayerNorm.bias",
    "intermediate.dense.weight", "intermediate.dense.bias",
    "output.dense.weight", "output.dense.bias",
    "output.LayerNorm.weight", "output.LayerNorm.bias",
)
EMB_KEYS = ("bert.embeddings.word_embeddings.weight", "bert.embeddings.position_embeddings.weight",
            "bert.embeddings.token_type_embeddings.weight", "bert.embeddings.LayerNorm.weight",
            "bert.embeddings.LayerNorm.bias")
HEAD_KEYS = ("cls_projection_head.dense_to_hidden.weight", "cls_projection_head.dense_to_hidden.bias",
             "cls_projection_head.LayerNorm.weight", "cls_projection_head.LayerNorm.bias",
             "cls_projection_head.dense_to_output.weight", "cls_projection_head.dense_to_output.bias")


def param_names(n_layers: int) -> List[str]:
    names = list(EMB_KEYS)
    for i in range(n_layers):
        names += [f"bert.encoder.layer.{i}.{k}" for k in LAYER_KEYS]
    return names + list(HEAD_KEYS)


def _fused(w_q: torch.Tensor, w_k: torch.Tensor, w_v: torch.Tensor) -> Optional[torch.Tensor]:
    """Return the [3H, ...] tensor the three slices live in, if they are consecutive in memory."""
    n = w_q.numel() * w_q.element_size()
    st = w_q.untyped_storage().data_ptr()
    if (w_q.is_contiguous() and w_k.is_contiguous() and w_v.is_contiguous()
            and w_k.untyped_storage().data_ptr() == st and w_v.untyped_storage().data_ptr() == st
            and w_k.data_ptr() == w_q.data_ptr() + n and w_v.data_ptr() == w_k.data_ptr() + n):
        shape = (3 * w_q.shape[0],) + tuple(w_q.shape[1:])
        return torch.as_strided(w_q, shape, w_q.stride(), w_q.storage_offset())
    return None


@torch.no_grad()
def fuse_qkv_(q: torch.nn.Parameter, k: torch.nn.Parameter, v: torch.nn.Parameter) -> None:
    """Make q/k/v (weights or biases) consecutive views of one buffer; values are preserved."""
    if _fused(q.data, k.data, v.data) is not None:
        return
    buf = torch.cat([q.data.reshape(q.shape[0], -1), k.data.reshape(k.shape[0], -1), v.data.reshape(v.shape[0], -1)], 0)
    h = q.shape[0]
    q.data = buf[0:h].view(q.shape)
    k.data = buf[h:2 * h].view(k.shape)
    v.data = buf[2 * h:3 * h].view(v.shape)


class _Saved:
    __slots__ = ("x", "qkv", "probs", "ctx", "xhat1", "rstd1", "a", "u_pre", "u", "xhat2", "rstd2")


def _forward(p: Sequence[torch.Tensor], ids: torch.Tensor, mask: Optional[torch.Tensor], n_layers: int, n_heads: int,
             eps: float, save: bool, cls_only: bool = False):
    """cls_only: the caller consumes only the projected CLS embedding (`get_projected_text_embeddings`,
    modelling_cxrbert.py:117-141 -> `hidden_states[-1][:, 0, :]`).  Everything in the LAST layer after the attention is
    row-wise, so it runs on the N CLS rows instead of N*L tokens (output projection, both LayerNorms, the FFN: 75 % of that
    layer's GEMM work); the returned `last` is then [N, H] (the CLS rows)."""
    N, L = ids.shape
    word, pos, typ, eg, eb = p[0:5]
    H = word.shape[1]
    dH = H // n_heads
    if L > pos.shape[0]:
        raise ValueError(f"sequence length {L} exceeds max_position_embeddings {pos.shape[0]}")
    x, xhat0, rstd0 = K.embed_ln_fwd(ids.reshape(-1), word, pos, typ[0], eg, eb, eps, L)
    saved: List[_Saved] = []
    for i in range(n_layers):
        (wq, bq, wk, bk, wv, bv, wo, bo, g1, b1, wi, bi, wo2, bo2, g2, b2) = p[5 + 16 * i: 5 + 16 * (i + 1)]
        wqkv, bqkv = _fused(wq, wk, wv), _fused(bq, bk, bv)
        if wqkv is None or bqkv is None:
            raise RuntimeError("q/k/v parameters are not fused; call CXRBertModel.prepare_() after moving the model")
        qkv = K.linear_fwd(x, wqkv, bqkv)
        ctx, probs = K.attn_fwd(qkv, mask, N, L, n_heads, dH, save_probs=save)
        rows_cls = cls_only and i == n_layers - 1
        if rows_cls:   # row 0 of every sequence: [N, H] views with row stride L*H
            t1 = K.linear_fwd(ctx.view(N, L * H)[:, :H], wo, bo, residual=x.view(N, L * H)[:, :H])
        else:
            t1 = K.linear_fwd(ctx, wo, bo, residual=x)
        a, xhat1, rstd1 = K.residual_ln_fwd(t1, None, g1, b1, eps, save=save)
        u_pre = torch.empty(t1.shape[0], wi.shape[0], dtype=torch.float32, device=x.device) if save else None
        u = K.linear_fwd(a, wi, bi, act=K.ACT_GELU, preact_out=u_pre)
        t2 = K.linear_fwd(u, wo2, bo2, residual=a)
        xn, xhat2, rstd2 = K.residual_ln_fwd(t2, None, g2, b2, eps, save=save)
        if save:
            s = _Saved()
            s.x, s.qkv, s.probs, s.ctx, s.xhat1, s.rstd1, s.a, s.u_pre, s.u, s.xhat2, s.rstd2 = \
                x, qkv, probs, ctx, xhat1, rstd1, a, u_pre, u, xhat2, rstd2
            saved.append(s)
        x = xn
    wdh, bdh, gh, bh, wdo, bdo = p[5 + 16 * n_layers:]
    cls = x if cls_only else x.view(N, L * H)[:, :H]
    h1_pre = torch.empty(N, wdh.shape[0], dtype=torch.float32, device=x.device)
    h1 = K.linear_fwd(cls, wdh, bdh, act=K.ACT_GELU, preact_out=h1_pre)
    h2, xhat_h, rstd_h = K.residual_ln_fwd(h1, None, gh, bh, 1e-12)
    proj = K.linear_fwd(h2, wdo, bdo)
    return proj, x, (xhat0, rstd0, saved, h1_pre, h2, xhat_h, rstd_h)


def _backward(p: Sequence[torch.Tensor], ids: torch.Tensor, n_layers: int, n_heads: int, state, last, dproj, dlast,
              need: Sequence[bool], cls_only: bool = False):
    """Returns the list of parameter gradients (same order as `p`).  `last` is the final hidden state [T,H]
    ([N,H], the CLS rows, when cls_only: the last layer's row-wise part then runs on those rows only)."""
    N, L = ids.shape
    xhat0, rstd0, saved, h1_pre, h2, xhat_h, rstd_h = state
    word, pos, typ, eg, eb = p[0:5]
    H = word.shape[1]
    T = N * L
    dev = word.device
    grads: List[Optional[torch.Tensor]] = [None] * len(p)

    def new(like):
        return torch.empty_like(like)

    base = 5 + 16 * n_layers
    wdh, bdh, gh, bh, wdo, bdo = p[base:]
    R = N if cls_only else T          # rows of the last layer's row-wise part
    if dlast is not None:
        dx = dlast.reshape(R, H).contiguous().clone()
    else:
        dx = torch.zeros(R, H, dtype=torch.float32, device=dev)
    if dproj is not None:
        dproj = dproj.contiguous()
        grads[base + 4] = K.linear_bwd_weight(dproj, h2, new(wdo))
        grads[base + 5] = K.colsum(dproj, new(bdo))
        dh2 = K.linear_bwd_data(dproj, wdo)
        dgh, dbh = new(gh), new(bh)
        dh1 = K.residual_ln_bwd(dh2, xhat_h, rstd_h, gh, dgh, dbh)
        grads[base + 2], grads[base + 3] = dgh, dbh
        dh1p = K.gelu_bwd(dh1, h1_pre)
        cls = last if cls_only else last.view(N, L * H)[:, :H]   # last_hidden_state[:, 0, :] (modelling_cxrbert.py:98-99)
        grads[base + 0] = K.linear_bwd_weight(dh1p, cls, new(wdh))
        grads[base + 1] = K.colsum(dh1p, new(bdh))
        K.linear_bwd_data(dh1p, wdh, out=dx if cls_only else dx.view(N, L * H)[:, :H], accumulate=True)
    else:
        for j in range(6):
            grads[base + j] = torch.zeros_like(p[base + j])

    for i in reversed(range(n_layers)):
        o = 5 + 16 * i
        (wq, bq, wk, bk, wv, bv, wo, bo, g1, b1, wi, bi, wo2, bo2, g2, b2) = p[o:o + 16]
        s = saved[i]
        dg2, db2 = new(g2), new(b2)
        dt2 = K.residual_ln_bwd(dx, s.xhat2, s.rstd2, g2, dg2, db2)
        grads[o + 14], grads[o + 15] = dg2, db2
        grads[o + 12] = K.linear_bwd_weight(dt2, s.u, new(wo2))
        grads[o + 13] = K.colsum(dt2, new(bo2))
        du = K.linear_bwd_data(dt2, wo2, aux=s.u_pre, auxmode=K.AUX_GELU_GRAD)
        grads[o + 10] = K.linear_bwd_weight(du, s.a, new(wi))
        grads[o + 11] = K.colsum(du, new(bi))
        da = K.linear_bwd_data(du, wi, residual=dt2)
        dg1, db1 = new(g1), new(b1)
        dt1 = K.residual_ln_bwd(da, s.xhat1, s.rstd1, g1, dg1, db1)
        grads[o + 8], grads[o + 9] = dg1, db1
        rows_cls = cls_only and i == n_layers - 1
        if rows_cls:   # dt1 holds the CLS rows only: its context rows are row 0 of every sequence, all other rows get no gradient
            grads[o + 6] = K.linear_bwd_weight(dt1, s.ctx.view(N, L * H)[:, :H], new(wo))
            grads[o + 7] = K.colsum(dt1, new(bo))
            dctx = torch.zeros(T, H, dtype=torch.float32, device=dev)
            K.linear_bwd_data(dt1, wo, out=dctx.view(N, L * H)[:, :H])
        else:
            grads[o + 6] = K.linear_bwd_weight(dt1, s.ctx, new(wo))
            grads[o + 7] = K.colsum(dt1, new(bo))
            dctx = K.linear_bwd_data(dt1, wo)
        dqkv = K.attn_bwd(s.qkv, s.probs, dctx, N, L, n_heads, H // n_heads)
        wqkv = _fused(wq, wk, wv)
        dwqkv = K.linear_bwd_weight(dqkv, s.x, torch.empty(3 * H, H, dtype=torch.float32, device=dev))
        dbqkv = K.colsum(dqkv, torch.empty(3 * H, dtype=torch.float32, device=dev))
        grads[o + 0], grads[o + 2], grads[o + 4] = dwqkv[0:H], dwqkv[H:2 * H], dwqkv[2 * H:3 * H]
        grads[o + 1], grads[o + 3], grads[o + 5] = dbqkv[0:H], dbqkv[H:2 * H], dbqkv[2 * H:3 * H]
        if rows_cls:   # the residual branch x -> t1 exists for the CLS rows only
            dx = K.linear_bwd_data(dqkv, wqkv)
            dx.view(N, L * H)[:, :H].add_(dt1)
        else:
            dx = K.linear_bwd_data(dqkv, wqkv, residual=dt1)
        saved[i] = None  # free this layer's activations early

    deg, deb = new(eg), new(eb)
    demb = K.residual_ln_bwd(dx, xhat0, rstd0, eg, deg, deb)
    grads[3], grads[4] = deg, deb
    if need[0]:
        dword = torch.zeros_like(word)
        K.embed_bwd(ids.reshape(-1), demb, dword)
        grads[0] = dword
    if need[1]:
        dpos = torch.zeros_like(pos)
        K.colsum(demb.view(N, L * H), dpos.view(-1)[: L * H])
        grads[1] = dpos
    if need[2]:
        dtyp = torch.zeros_like(typ)
        K.colsum(demb, dtyp[0])
        grads[2] = dtyp
    return grads


class CXRBertEncodeFn(torch.autograd.Function):
    """(ids, mask, cfg, cls_only, *params) -> (cls_projected_embedding [N,P], last_hidden_state [N,L,H], or [N,1,H] = its
    CLS rows when cls_only)."""

    @staticmethod
    def forward(ctx, ids, mask, n_layers, n_heads, eps, cls_only, *params):
        ctx.set_materialize_grads(False)
        save = any(t.requires_grad for t in params)
        p = [t.detach() for t in params]
        proj, last, state = _forward(p, ids, mask, n_layers, n_heads, eps, save, cls_only)
        if save:
            ctx.state = state
            ctx.cfg = (n_layers, n_heads, cls_only)
            ctx.ids = ids
            ctx.last = last
            ctx.params = p
            ctx.need = [t.requires_grad for t in params]
        N, L = ids.shape
        return proj, last.view(N, 1 if cls_only else L, -1)

    @staticmethod
    def backward(ctx, dproj, dlast):
        n_layers, n_heads, cls_only = ctx.cfg
        grads = _backward(ctx.params, ctx.ids, n_layers, n_heads, ctx.state, ctx.last, dproj, dlast, ctx.need, cls_only)
        ctx.state = None
        ctx.last = None
        return (None, None, None, None, None, None) + tuple(g if n else None for g, n in zip(grads, ctx.need))


def encode(params: Sequence[torch.Tensor], ids: torch.Tensor, mask: Optional[torch.Tensor], n_layers: int,
           n_heads: int, eps: float = 1e-12, cls_only: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    if ids.dtype != torch.int64:
        ids = ids.to(torch.int64)
    if mask is not None and mask.dtype != torch.int64:
        mask = mask.to(torch.int64)
    ids = ids.contiguous()
    mask = mask.contiguous() if mask is not None else None
    return CXRBertEncodeFn.apply(ids, mask, n_layers, n_heads, eps, bool(cls_only), *params)
