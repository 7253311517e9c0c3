"""CXR-BERT restated in plain torch functional ops (CPU fp32).  TEST INFRASTRUCTURE ONLY.

Follows `health_multimodal/text/model/modelling_cxrbert.py:28-49` (projection head),
`:70-115` (forward: last hidden state's CLS row -> head) and `:117-141`
(`get_projected_text_embeddings`, optional `F.normalize(dim=1)`), on top of HuggingFace
`BertForMaskedLM` semantics as configured by `configuration_cxrbert.py:11-22`
(post-LN encoder, erf-GELU, LayerNorm eps 1e-12, additive attention mask, absolute
position embeddings, token_type 0).  Dropout is inactive (the engine asserts eval mode,
`text/inference_engine.py:63`).  The unused MLM head (`:87-95`) is only computed on request.
Parameters are passed as a dict keyed by the reference's state-dict names.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

P = Dict[str, torch.Tensor]


def bert_embeddings(p: P, ids: torch.Tensor, eps: float = 1e-12) -> torch.Tensor:
    L = ids.shape[1]
    x = p["bert.embeddings.word_embeddings.weight"][ids]
    x = x + p["bert.embeddings.position_embeddings.weight"][:L][None]
    x = x + p["bert.embeddings.token_type_embeddings.weight"][0][None, None]
    return F.layer_norm(x, (x.shape[-1],), p["bert.embeddings.LayerNorm.weight"],
                        p["bert.embeddings.LayerNorm.bias"], eps)


def bert_layer(p: P, i: int, x: torch.Tensor, add_mask: torch.Tensor, n_heads: int,
               eps: float = 1e-12) -> torch.Tensor:
    pre = f"bert.encoder.layer.{i}."
    N, L, H = x.shape
    d = H // n_heads
    q = F.linear(x, p[pre + "attention.self.query.weight"], p[pre + "attention.self.query.bias"])
    k = F.linear(x, p[pre + "attention.self.key.weight"], p[pre + "attention.self.key.bias"])
    v = F.linear(x, p[pre + "attention.self.value.weight"], p[pre + "attention.self.value.bias"])
    q = q.view(N, L, n_heads, d).transpose(1, 2)
    k = k.view(N, L, n_heads, d).transpose(1, 2)
    v = v.view(N, L, n_heads, d).transpose(1, 2)
    s = q @ k.transpose(-1, -2) / math.sqrt(d) + add_mask
    pr = torch.softmax(s, dim=-1)
    ctx = (pr @ v).transpose(1, 2).reshape(N, L, H)
    a = F.linear(ctx, p[pre + "attention.output.dense.weight"], p[pre + "attention.output.dense.bias"])
    x = F.layer_norm(a + x, (H,), p[pre + "attention.output.LayerNorm.weight"],
                     p[pre + "attention.output.LayerNorm.bias"], eps)
    u = F.linear(x, p[pre + "intermediate.dense.weight"], p[pre + "intermediate.dense.bias"])
    u = F.gelu(u)  # erf form
    o = F.linear(u, p[pre + "output.dense.weight"], p[pre + "output.dense.bias"])
    return F.layer_norm(o + x, (H,), p[pre + "output.LayerNorm.weight"], p[pre + "output.LayerNorm.bias"], eps)


def projection_head(p: P, cls: torch.Tensor) -> torch.Tensor:
    """`BertProjectionHead.forward`, modelling_cxrbert.py:43-49."""
    h = F.linear(cls, p["cls_projection_head.dense_to_hidden.weight"], p["cls_projection_head.dense_to_hidden.bias"])
    h = F.gelu(h)
    h = F.layer_norm(h, (h.shape[-1],), p["cls_projection_head.LayerNorm.weight"],
                     p["cls_projection_head.LayerNorm.bias"], 1e-12)
    return F.linear(h, p["cls_projection_head.dense_to_output.weight"], p["cls_projection_head.dense_to_output.bias"])


def cxrbert_last_hidden(p: P, ids: torch.Tensor, mask: torch.Tensor, n_layers: int, n_heads: int) -> torch.Tensor:
    x = bert_embeddings(p, ids)
    add_mask = (1.0 - mask[:, None, None, :].to(x.dtype)) * torch.finfo(x.dtype).min
    for i in range(n_layers):
        x = bert_layer(p, i, x, add_mask, n_heads)
    return x


def cxrbert_projected(p: P, ids: torch.Tensor, mask: torch.Tensor, n_layers: int = 12, n_heads: int = 12,
                      normalize: bool = False) -> torch.Tensor:
    """`CXRBertModel.get_projected_text_embeddings`, modelling_cxrbert.py:117-141."""
    h = cxrbert_last_hidden(p, ids, mask, n_layers, n_heads)
    e = projection_head(p, h[:, 0, :])
    return F.normalize(e, dim=1) if normalize else e


def mlm_logits(p: P, hidden: torch.Tensor) -> torch.Tensor:
    """HF `BertOnlyMLMHead`: transform(dense, gelu, LN) -> decoder tied to word embeddings + bias."""
    h = F.linear(hidden, p["cls.predictions.transform.dense.weight"], p["cls.predictions.transform.dense.bias"])
    h = F.gelu(h)
    h = F.layer_norm(h, (h.shape[-1],), p["cls.predictions.transform.LayerNorm.weight"],
                     p["cls.predictions.transform.LayerNorm.bias"], 1e-12)
    return F.linear(h, p["bert.embeddings.word_embeddings.weight"], p["cls.predictions.bias"])


def cxrbert_param_shapes(vocab: int = 30522, hidden: int = 768, n_layers: int = 12, inter: int = 3072,
                         max_pos: int = 512, proj: int = 128, type_vocab: int = 2,
                         with_mlm_head: bool = False) -> Dict[str, tuple]:
    """State-dict names/shapes of the parameters the hot path reads."""
    s: Dict[str, tuple] = {
        "bert.embeddings.word_embeddings.weight": (vocab, hidden),
        "bert.embeddings.position_embeddings.weight": (max_pos, hidden),
        "bert.embeddings.token_type_embeddings.weight": (type_vocab, hidden),
        "bert.embeddings.LayerNorm.weight": (hidden,),
        "bert.embeddings.LayerNorm.bias": (hidden,),
    }
    for i in range(n_layers):
        pre = f"bert.encoder.layer.{i}."
        for nm in ("attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense"):
            s[pre + nm + ".weight"] = (hidden, hidden)
            s[pre + nm + ".bias"] = (hidden,)
        s[pre + "attention.output.LayerNorm.weight"] = (hidden,)
        s[pre + "attention.output.LayerNorm.bias"] = (hidden,)
        s[pre + "intermediate.dense.weight"] = (inter, hidden)
        s[pre + "intermediate.dense.bias"] = (inter,)
        s[pre + "output.dense.weight"] = (hidden, inter)
        s[pre + "output.dense.bias"] = (hidden,)
        s[pre + "output.LayerNorm.weight"] = (hidden,)
        s[pre + "output.LayerNorm.bias"] = (hidden,)
    if with_mlm_head:
        s["cls.predictions.bias"] = (vocab,)
        s["cls.predictions.transform.dense.weight"] = (hidden, hidden)
        s["cls.predictions.transform.dense.bias"] = (hidden,)
        s["cls.predictions.transform.LayerNorm.weight"] = (hidden,)
        s["cls.predictions.transform.LayerNorm.bias"] = (hidden,)
    s["cls_projection_head.dense_to_hidden.weight"] = (proj, hidden)
    s["cls_projection_head.dense_to_hidden.bias"] = (proj,)
    s["cls_projection_head.LayerNorm.weight"] = (proj,)
    s["cls_projection_head.LayerNorm.bias"] = (proj,)
    s["cls_projection_head.dense_to_output.weight"] = (proj, proj)
    s["cls_projection_head.dense_to_output.bias"] = (proj,)
    return s
