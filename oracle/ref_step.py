"""Whole-step restatements on CPU (fp32).  TEST INFRASTRUCTURE ONLY.

* `adapter_step`  — the reference's real train step (`Trainer.py:537-601`, T-ref in SURVEY.md §0):
  image adapter -> 10 prompt vectors (text adapter + mean over 4 prompts, `Trainer.py:1657-1680`)
  -> pairwise cosine (`:1682-1704`) -> pos-neg logits (`:575`) -> BCEWithLogits(mean) -> Adam.
* `joint_step`    — the north-star superset step (T-ns): both encoders in-loop with gradients,
  InfoNCE, Adam on every parameter.  No reference oracle exists for it; this is autograd over the
  restated modules.
* `eval_scores`   — `Trainer.val/test` scoring (`Trainer.py:797-837,1016-1047`).
* `weight_reset`  — `Trainer.myIncremental` (`Trainer.py:1556-1587`).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from . import ref_image, ref_loss, ref_text


def mlp_adapter(p: Dict[str, torch.Tensor], x: torch.Tensor, prefix: str = "layer.") -> torch.Tensor:
    """`models.myMLP.forward` (models.py:7-15): Linear(128,256) -> ReLU -> Linear(256,128);
    `myLinearModel` (models.py:18-26) when only `layer.0` exists."""
    h = F.linear(x, p[prefix + "0.weight"], p[prefix + "0.bias"])
    if prefix + "2.weight" in p:
        h = F.linear(F.relu(h), p[prefix + "2.weight"], p[prefix + "2.bias"])
    return h


def prompt_vectors(text_adapter: Optional[Dict[str, torch.Tensor]], bert_out: torch.Tensor) -> torch.Tensor:
    """`Trainer.bert_forward_mean` (Trainer.py:1657-1680) for all classes at once.
    bert_out [2C, n_prompts, 128] (row 2c = positive prompts of class c, 2c+1 = negative) -> [2C,128]."""
    g, n, d = bert_out.shape
    e = bert_out.reshape(g * n, d)
    if text_adapter is not None:
        e = mlp_adapter(text_adapter, e)
    return e.reshape(g, n, d).mean(dim=1)


def adapter_logits(img_ad, txt_ad, embs, bert_out, n_cols: Optional[int] = None, diff: bool = True):
    x = mlp_adapter(img_ad, embs) if img_ad is not None else embs
    pv = prompt_vectors(txt_ad, bert_out)
    pos, neg = pv[0::2], pv[1::2]
    if n_cols is not None:
        pos, neg = pos[:n_cols], neg[:n_cols]
    return ref_loss.posneg_logits(x, pos, neg, diff)


def adapter_logits_max_emb(img_ad, txt_ad, embs, bert_out, diff: bool = True):
    """The same logits with `MAX_EMB` (`Trainer.py:1664-1666,1691-1693`): the prompt embeddings of a class are NOT averaged; the
    cosine against each prompt is taken and the maximum over the class's prompts is the class's (positive / negative) score."""
    x = mlp_adapter(img_ad, embs) if img_ad is not None else embs
    g, n, d = bert_out.shape
    e = bert_out.reshape(g * n, d)
    if txt_ad is not None:
        e = mlp_adapter(txt_ad, e)
    mx, _, _ = ref_loss.pairwise_cosine_max(x, e, g)          # [B, 2C]
    return (mx[:, 0::2] - mx[:, 1::2]) if diff else mx[:, 0::2]


def adapter_step(img_ad: Dict[str, torch.Tensor], txt_ad: Dict[str, torch.Tensor], embs, labels, bert_out,
                 optimizer: torch.optim.Optimizer, n_cols: Optional[int] = None):
    """One reference train step.  Parameters in the dicts must be leaf tensors owned by `optimizer`."""
    optimizer.zero_grad()
    logits = adapter_logits(img_ad, txt_ad, embs, bert_out, n_cols)
    lab = labels if n_cols is None else labels[:, :n_cols]
    loss = ref_loss.bce_with_logits_mean(logits, lab)
    loss.backward()
    optimizer.step()
    return loss.detach(), logits.detach()


def eval_scores(img_ad, txt_ad, embs, bert_out, pred_diff: bool = False):
    """`Trainer.val` scoring (Trainer.py:797-837): y_score = (pos+1)/2 (or (pos-neg+2)/4 when
    PRED_LOGIT_DIFF), y_pred = argmax([neg, pos])."""
    with torch.no_grad():
        x = mlp_adapter(img_ad, embs) if img_ad is not None else embs
        pv = prompt_vectors(txt_ad, bert_out)
        cp = ref_loss.pairwise_cosine_similarity(x, pv[0::2])
        cn = ref_loss.pairwise_cosine_similarity(x, pv[1::2])
        score = (cp - cn + 2) / 4 if pred_diff else (cp + 1) / 2
        pred = (cp > cn).to(torch.float32)  # argmax over [neg, pos]; ties -> index 0 (neg)
        return score, pred, cp - cn


def weight_reset(new: torch.Tensor, old: torch.Tensor, threshold: float) -> Tuple[torch.Tensor, int]:
    """`Trainer.myIncremental` per tensor (Trainer.py:1562-1572): restore entries whose |new-old| is
    below min + thr*(max-min)."""
    diff = (new - old).abs()
    to_reset = diff.min() + threshold * (diff.max() - diff.min())
    mask = diff < to_reset
    out = torch.where(mask, old, new)
    return out, int(mask.sum())


def joint_forward(img_p, txt_p, images, ids, mask, temperature: float, n_layers: int = 12, n_heads: int = 12,
                  relu=ref_image._PLAIN):
    ie = ref_image.image_model_forward(img_p, images, relu=relu)
    te = ref_text.cxrbert_projected(txt_p, ids, mask, n_layers, n_heads, normalize=False)
    loss, s = ref_loss.infonce(ie, te, temperature)
    return loss, s, ie, te


def joint_step(img_p, txt_p, images, ids, mask, temperature, optimizer, n_layers: int = 12, n_heads: int = 12,
               relu=ref_image._PLAIN):
    optimizer.zero_grad()
    loss, s, ie, te = joint_forward(img_p, txt_p, images, ids, mask, temperature, n_layers, n_heads, relu)
    loss.backward()
    optimizer.step()
    return loss.detach()
