"""CXR-BERT forward + hand-written backward on the cxrk kernels, exposed as one `torch.autograd.Function`.

Arithmetic follows HuggingFace `BertForMaskedLM` as the reference drives it
(`health_multimodal/text/model/modelling_cxrbert.py:87-99`: `hidden_states[-1][:, 0, :]` -> `BertProjectionHead`,
`:43-49`), post-LN, erf-GELU, LayerNorm eps = config.layer_norm_eps (1e-12), additive key mask, dropout inactive.
The MLM head the reference computes and discards on this path (`:87-95`) is not computed here.

Q, K and V projections run as ONE [3H, H] GEMM per layer: `fuse_qkv_` re-points the three nn.Parameter tensors
of a layer at consecutive slices of one buffer (state-dict names and values unchanged).

Two storage modes, chosen by the library's contraction precision (`_lib.get_precision()`), as in `image_encoder`:
  fp32        every tensor fp32, exact-fp32 MFMA mainloop.
  split_bf16  every tensor that feeds a GEMM (layer inputs, attention context, GELU output, all back-propagated gradients that
              are GEMM operands, and the weights, split once per forward) is a `kernels.Planes` tensor written by the producing
              kernel (LayerNorm, attention, GEMM epilogue), so the mainloops load operands with no conversion work.  Tensors
              only LayerNorm / attention read (qkv, the pre-LN sums, pre-GELU values, attention probabilities) stay fp32.
Parameter gradients are accumulated straight into `param.grad` (see `gradsink`).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from . import kernels as K
from .gradsink import GradSink
from .kernels import Planes

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


def _planes_mode() -> bool:
    return _lib.get_precision() == "split_bf16"


class _Saved:
    __slots__ = ("x", "qkv", "probs", "ctx", "xhat1", "rstd1", "a", "u_pre", "u", "xhat2", "rstd2")


# ---- storage-format dispatch of the GEMM family (x / dy Planes <=> planes mode) ---------------------------------------------
def _lin(x, w, bias, pl, **kw):
    return K.linear_fwd_pl(x, w, bias, **kw) if pl else K.linear_fwd(x, w, bias, **{k: v for k, v in kw.items() if k != "out_planes"})


def _dgrad(dy, w, pl, **kw):
    return K.linear_bwd_data_pl(dy, w, **kw) if pl else K.linear_bwd_data(dy, w, **{k: v for k, v in kw.items() if k != "out_planes"})


def _wgrad(dy, x, dw, acc, pl):
    return K.linear_bwd_weight_pl(dy, x, dw, accumulate=acc) if pl else K.linear_bwd_weight(dy, x, dw, accumulate=acc)


def _cls_rows(t, N, L, H):
    """row 0 of every sequence of a [N*L, H] tensor: an [N, H] view with row stride L*H"""
    if isinstance(t, Planes):
        return Planes(t.t.view(2, N, L * H)[:, :, :H])
    return t.view(N, L * H)[:, :H]


def _weights(p: Sequence[torch.Tensor], n_layers: int, pl: bool):
    """GEMM weights of the forward in the storage format of the mode: (per-layer (wqkv, wo, wi, wo2), (wdh, wdo), bqkv list).
    Planes mode splits every matrix once per forward (133 M parameters: 1 GB of traffic, ~0.2 ms) into one bf16 buffer."""
    layers, bq = [], []
    mats = []
    for i in range(n_layers):
        (wq, bqi, wk, bk, wv, bv, wo, bo, g1, b1, wi, bi, wo2, bo2, g2, b2) = p[5 + 16 * i: 5 + 16 * (i + 1)]
        wqkv, bqkv = _fused(wq, wk, wv), _fused(bqi, bk, bv)
        if wqkv is None or bqkv is None:
            raise RuntimeError("q/k/v parameters are not fused; call CXRBertModel.prepare_() after moving the model")
        mats += [wqkv, wo, wi, wo2]
        bq.append(bqkv)
    wdh, wdo = p[5 + 16 * n_layers], p[5 + 16 * n_layers + 4]
    mats += [wdh, wdo]
    if pl:
        tot = sum((m.numel() + 7) // 8 * 8 for m in mats)
        buf = torch.empty(2, tot, dtype=torch.bfloat16, device=mats[0].device)
        o, outm = 0, []
        for m in mats:
            n = m.numel()
            dst = Planes(buf[:, o:o + n].view(2, m.shape[0], m.shape[1]))
            K.split_planes(m, out=dst)
            outm.append(dst)
            o += (n + 7) // 8 * 8
        mats = outm
    for i in range(n_layers):
        layers.append(tuple(mats[4 * i: 4 * i + 4]))
    return layers, (mats[-2], mats[-1]), bq


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
    pl = _planes_mode()
    wl, (wdh_w, wdo_w), bq = _weights(p, n_layers, pl)
    x, xhat0, rstd0 = K.embed_ln_fwd(ids.reshape(-1), word, pos, typ[0], eg, eb, eps, L, out_planes=pl)
    saved: List[_Saved] = []
    for i in range(n_layers):
        (wq, bqi, wk, bk, wv, bv, wo, bo, g1, b1, wi, bi, wo2, bo2, g2, b2) = p[5 + 16 * i: 5 + 16 * (i + 1)]
        wqkv_w, wo_w, wi_w, wo2_w = wl[i]
        qkv = _lin(x, wqkv_w, bq[i], pl)                                   # fp32: only the attention kernel reads it
        ctx, probs = K.attn_fwd(qkv, mask, N, L, n_heads, dH, save_probs=save, out_planes=pl)
        rows_cls = cls_only and i == n_layers - 1
        if rows_cls:   # row 0 of every sequence: [N, H] views with row stride L*H
            t1 = _lin(_cls_rows(ctx, N, L, H), wo_w, bo, pl, residual=_cls_rows(x, N, L, H))
        else:
            t1 = _lin(ctx, wo_w, bo, pl, residual=x)
        a, xhat1, rstd1 = K.residual_ln_fwd(t1, None, g1, b1, eps, save=save, out_planes=pl)
        u_pre = torch.empty(t1.shape[0], wi.shape[0], dtype=torch.float32, device=t1.device) if save else None
        u = _lin(a, wi_w, bi, pl, act=K.ACT_GELU, preact_out=u_pre, out_planes=True)
        t2 = _lin(u, wo2_w, bo2, pl, residual=a)
        xn, xhat2, rstd2 = K.residual_ln_fwd(t2, None, g2, b2, eps, save=save, out_planes=pl)
        if save:
            s = _Saved()
            s.x, s.qkv, s.probs, s.ctx, s.xhat1, s.rstd1, s.a, s.u_pre, s.u, s.xhat2, s.rstd2 = \
                x, qkv, probs, ctx, xhat1, rstd1, a, u_pre, u, xhat2, rstd2
            saved.append(s)
        x = xn
    wdh, bdh, gh, bh, wdo, bdo = p[5 + 16 * n_layers:]
    cls = x if cls_only else _cls_rows(x, N, L, H)
    h1_pre = torch.empty(N, wdh.shape[0], dtype=torch.float32, device=word.device)
    h1 = _lin(cls, wdh_w, bdh, pl, act=K.ACT_GELU, preact_out=h1_pre)     # fp32: LayerNorm input
    h2, xhat_h, rstd_h = K.residual_ln_fwd(h1, None, gh, bh, 1e-12, out_planes=pl)
    proj = _lin(h2, wdo_w, bdo, pl)
    return proj, x, (xhat0, rstd0, saved, h1_pre, h2, xhat_h, rstd_h, wl, (wdh_w, wdo_w), pl)


def _backward(p: Sequence[torch.Tensor], ids: torch.Tensor, n_layers: int, n_heads: int, state, last, dproj, dlast,
              sink: GradSink, cls_only: bool = False) -> None:
    """Writes every parameter gradient through `sink` (same order as `p`).  `last` is the final hidden state [T,H]
    ([N,H], the CLS rows, when cls_only: the last layer's row-wise part then runs on those rows only)."""
    N, L = ids.shape
    xhat0, rstd0, saved, h1_pre, h2, xhat_h, rstd_h, wl, (wdh_w, wdo_w), pl = state
    word, pos, typ, eg, eb = p[0:5]
    H = word.shape[1]
    T = N * L
    dev = word.device
    need = sink.need

    def op(t):   # an fp32 gradient that is about to be a GEMM operand
        return K.split_planes(t.contiguous()) if pl else t

    base = 5 + 16 * n_layers
    R = N if cls_only else T          # rows of the last layer's row-wise part
    if dlast is not None:
        dx = dlast.reshape(R, H).contiguous().clone()
    else:
        dx = torch.zeros(R, H, dtype=torch.float32, device=dev)
    if dproj is not None:
        dproj = dproj.contiguous()
        dpo = op(dproj)
        g, acc = sink.dst(base + 4)
        _wgrad(dpo, h2, g, acc, pl)
        g, acc = sink.dst(base + 5)
        K.colsum(dproj, g, accumulate=acc)
        dh2 = _dgrad(dpo, wdo_w, pl)
        (dgh, a1), (dbh, a2) = sink.dst(base + 2), sink.dst(base + 3)
        if a1 != a2:
            (dgh, a1), (dbh, a2) = sink.dst(base + 2, True), sink.dst(base + 3, True)
        dh1 = K.residual_ln_bwd(dh2, xhat_h, rstd_h, p[base + 2], dgh, dbh, accumulate=a1)
        dh1p = K.gelu_bwd(dh1, h1_pre)
        dh1o = op(dh1p)
        cls = last if cls_only else _cls_rows(last, N, L, H)   # last_hidden_state[:, 0, :] (modelling_cxrbert.py:98-99)
        g, acc = sink.dst(base + 0)
        _wgrad(dh1o, cls, g, acc, pl)
        g, acc = sink.dst(base + 1)
        K.colsum(dh1p, g, accumulate=acc)
        _dgrad(dh1o, wdh_w, pl, out=dx if cls_only else dx.view(N, L * H)[:, :H], accumulate=True)
    else:
        for j in range(6):
            sink.ret[base + j] = torch.zeros_like(p[base + j])

    for i in reversed(range(n_layers)):
        o = 5 + 16 * i
        (wq, bq, wk, bk, wv, bv, wo, bo, g1, b1, wi, bi, wo2, bo2, g2, b2) = p[o:o + 16]
        wqkv_w, wo_w, wi_w, wo2_w = wl[i]
        s = saved[i]

        def ln_bwd(dy, xhat, rstd, gamma, jg, jbias):
            """LayerNorm backward; the bias gradient of the dense layer in front of it (parameter jbias) is the column sum of the
            result and comes out of the same kernel."""
            (dg, a1), (db, a2) = sink.dst(jg), sink.dst(jg + 1)
            if a1 != a2:
                (dg, a1), (db, a2) = sink.dst(jg, True), sink.dst(jg + 1, True)
            bg, bacc = sink.dst(jbias)
            return K.residual_ln_bwd(dy, xhat, rstd, gamma, dg, db, accumulate=a1, out_planes=pl, dxsum=bg, dxsum_accumulate=bacc)

        def wb(dy, x_in, jw, bias: bool = True):   # weight (+ bias) gradient of a dense layer
            g, acc = sink.dst(jw)
            _wgrad(dy, x_in, g, acc, pl)
            if bias:
                g, acc = sink.dst(jw + 1)
                K.colsum(dy, g, accumulate=acc)

        dt2 = ln_bwd(dx, s.xhat2, s.rstd2, g2, o + 14, o + 13)
        wb(dt2, s.u, o + 12, bias=False)
        if pl:   # the FFN-up bias gradient = column sums of du, reduced in the epilogue that writes du
            bg, bacc = sink.dst(o + 11)
            du = _dgrad(dt2, wo2_w, pl, aux=s.u_pre, auxmode=K.AUX_GELU_GRAD, out_planes=True, colsum=bg, colsum_accumulate=bacc)
            wb(du, s.a, o + 10, bias=False)
        else:
            du = _dgrad(dt2, wo2_w, pl, aux=s.u_pre, auxmode=K.AUX_GELU_GRAD, out_planes=True)
            wb(du, s.a, o + 10)
        da = _dgrad(du, wi_w, pl, residual=dt2)
        dt1 = ln_bwd(da, s.xhat1, s.rstd1, g1, o + 8, o + 7)
        rows_cls = cls_only and i == n_layers - 1
        if rows_cls:   # dt1 holds the CLS rows only: its context rows are row 0 of every sequence, all other rows get no gradient
            wb(dt1, _cls_rows(s.ctx, N, L, H), o + 6, bias=False)
            dctx = torch.zeros(T, H, dtype=torch.float32, device=dev)
            _dgrad(dt1, wo_w, pl, out=dctx.view(N, L * H)[:, :H])
        else:
            wb(dt1, s.ctx, o + 6, bias=False)
            dctx = _dgrad(dt1, wo_w, pl)
        dqkv = K.attn_bwd(s.qkv, s.probs, dctx, N, L, n_heads, H // n_heads, out_planes=pl)
        # fused q/k/v gradients: one [3H, H] GEMM into the three (adjacent) .grad slices when they are adjacent too
        gq = [getattr(sink.params[o + j], "grad", None) for j in (0, 2, 4)]
        gb = [getattr(sink.params[o + j], "grad", None) for j in (1, 3, 5)]
        gw_f = _fused(*gq) if all(t is not None and t.dtype == torch.float32 for t in gq) else None
        gb_f = _fused(*gb) if all(t is not None and t.dtype == torch.float32 for t in gb) else None
        if gw_f is not None and gb_f is not None:
            _wgrad(dqkv, s.x, gw_f, True, pl)
            K.colsum(dqkv, gb_f, accumulate=True)
        else:
            dwqkv = torch.empty(3 * H, H, dtype=torch.float32, device=dev)
            dbqkv = torch.empty(3 * H, dtype=torch.float32, device=dev)
            _wgrad(dqkv, s.x, dwqkv, False, pl)
            K.colsum(dqkv, dbqkv)
            for j in range(3):
                sink.ret[o + 2 * j], sink.ret[o + 2 * j + 1] = dwqkv[j * H:(j + 1) * H], dbqkv[j * H:(j + 1) * H]
        if rows_cls:   # the residual branch x -> t1 exists for the CLS rows only
            dx = _dgrad(dqkv, wqkv_w, pl)
            if pl:
                K.planes_add_rows(dt1, dx.view(N, L * H)[:, :H])
            else:
                dx.view(N, L * H)[:, :H].add_(dt1)
        else:
            dx = _dgrad(dqkv, wqkv_w, pl, residual=dt1)
        saved[i] = None  # free this layer's activations early

    (deg, a1), (deb, a2) = sink.dst(3), sink.dst(4)
    if a1 != a2:
        (deg, a1), (deb, a2) = sink.dst(3, True), sink.dst(4, True)
    demb = K.residual_ln_bwd(dx, xhat0, rstd0, eg, deg, deb, accumulate=a1)
    if need[0]:
        dword, _ = sink.dst(0, zero=True)
        K.embed_bwd(ids.reshape(-1), demb, dword)
    if need[1]:
        dpos, acc = sink.dst(1, zero=True)
        K.colsum(demb.view(N, L * H), dpos.view(-1)[: L * H], accumulate=acc)
    if need[2]:
        dtyp, acc = sink.dst(2, zero=True)
        K.colsum(demb, dtyp[0], accumulate=acc)


class CXRBertEncodeFn(torch.autograd.Function):
    """(ids, mask, cfg, cls_only, *params) -> (cls_projected_embedding [N,P], last_hidden_state [N,L,H], or [N,1,H] = its
    CLS rows when cls_only; an empty tensor when the caller does not want it)."""

    @staticmethod
    def forward(ctx, ids, mask, n_layers, n_heads, eps, cls_only, want_last, on_grads_ready, *params):
        ctx.set_materialize_grads(False)
        ctx.on_grads_ready = on_grads_ready
        save = any(t.requires_grad for t in params)
        p = [t.detach() for t in params]
        proj, last, state = _forward(p, ids, mask, n_layers, n_heads, eps, save, cls_only)
        if save:
            ctx.state = state
            ctx.cfg = (n_layers, n_heads, cls_only)
            ctx.ids = ids
            ctx.last = last
            ctx.p = p
            ctx.params = params
            ctx.need = [t.requires_grad for t in params]
        N, L = ids.shape
        if not want_last:
            return proj, proj.new_empty(0)
        last_f = last.float() if isinstance(last, Planes) else last
        return proj, last_f.view(N, 1 if cls_only else L, -1)

    @staticmethod
    def backward(ctx, dproj, dlast):
        n_layers, n_heads, cls_only = ctx.cfg
        if dlast is not None and dlast.numel() == 0:
            dlast = None
        sink = GradSink(ctx.params, ctx.need)
        _backward(ctx.p, ctx.ids, n_layers, n_heads, ctx.state, ctx.last, dproj, dlast, sink, cls_only)
        ctx.state = None
        ctx.last = None
        res = sink.result()
        hook = ctx.on_grads_ready
        if hook is not None and all(r is None for r in res):
            hook("text")   # every gradient of this encoder already sits in its `.grad` view: the data-parallel step starts reducing them now
        return (None, None, None, None, None, None, None, None) + res


def encode(params: Sequence[torch.Tensor], ids: torch.Tensor, mask: Optional[torch.Tensor], n_layers: int,
           n_heads: int, eps: float = 1e-12, cls_only: bool = False, want_last: bool = True,
           on_grads_ready=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """`on_grads_ready(tag)`: optional callable bound to THIS call (it travels on the autograd node, not in a module global); the
    backward calls it with "text" on the stream it runs on once every parameter gradient of the call has been written in place
    (`gradsink`).  contrastive.JointContrastiveTrainer starts the encoder's gradient all-reduce from it."""
    if ids.dtype != torch.int64:
        ids = ids.to(torch.int64)
    if mask is not None and mask.dtype != torch.int64:
        mask = mask.to(torch.int64)
    ids = ids.contiguous()
    mask = mask.contiguous() if mask is not None else None
    return CXRBertEncodeFn.apply(ids, mask, n_layers, n_heads, eps, bool(cls_only), bool(want_last), on_grads_ready, *params)
