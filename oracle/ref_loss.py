"""Similarity + loss heads restated on CPU (fp32).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

from typing import Tuple

import torch
import torch.nn.functional as F


def pairwise_cosine_similarity(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """torchmetrics.functional.pairwise_cosine_similarity(x, y) as called at `Trainer.py:1688-1692`:
    rows of x and y divided by their L2 norm (no epsilon), then x @ y.T; no diagonal zeroing when y
    is given.  torchmetrics is an unpinned third-party dependency absent from this image, so this
    is restated from its published formula ("parity unpinned"); it equals the reference's commented
    legacy form `F.normalize(x, dim=-1) @ F.normalize(y).T` (`Trainer.py:1684-1686`) away from zero rows."""
    xn = x / torch.linalg.norm(x, ord=2, dim=1, keepdim=True)
    yn = y / torch.linalg.norm(y, ord=2, dim=1, keepdim=True)
    return xn @ yn.T


def pairwise_cosine_max(x: torch.Tensor, y: torch.Tensor, groups: int = 1):
    """MAX_EMB branch of `Trainer.myCosineSimilarity` (`Trainer.py:1691-1693`) for `groups` prompt sets stacked in y
    ([groups*Pg, D], set g = rows g*Pg..): cosines -> (max over the set, mean over the set, winner index), each [B, groups]."""
    res = pairwise_cosine_similarity(x, y).reshape(x.shape[0], groups, -1)
    mx, idx = torch.max(res, dim=2)
    return mx, torch.mean(res, dim=2), idx


def similarity_map(patches: torch.Tensor, text: torch.Tensor, sigma: float = 1.5) -> torch.Tensor:
    """`vlp/inference_engine.py:94-108`: patches [h,w,D], text [1,D] -> scipy-gaussian-smoothed <patch, text> map [h,w]."""
    from scipy import ndimage
    h, w, d = patches.shape
    raw = (patches.reshape(h * w, d) @ text.reshape(d, 1)).reshape(h, w).numpy()
    return torch.from_numpy(ndimage.gaussian_filter(raw, sigma=(sigma, sigma), order=0))


def similarity_to_image_size(sim: torch.Tensor, width: int, height: int, resize_size, crop_size, interpolation: str = "nearest"):
    """`vlp/inference_engine.py:110-155` in numpy-free torch: stretch the patch grid over the crop's footprint in original
    pixels (or the whole image when nothing was cropped), NaN outside."""
    import math
    g = sim[None, None]
    ac = False if interpolation in ("linear", "bilinear", "bicubic", "trilinear") else None
    if crop_size is None:
        return F.interpolate(g, size=(height, width), mode=interpolation, align_corners=ac)[0, 0].numpy()
    side = int(crop_size * min(height, width) / resize_size) if resize_size is not None else crop_size
    m = F.interpolate(g, size=(side, side), mode=interpolation, align_corners=ac)[0, 0]
    mw, mh = width - side, height - side
    return F.pad(m, (math.floor(mw / 2), math.ceil(mw / 2), math.floor(mh / 2), math.ceil(mh / 2)), value=float("nan")).numpy()


def posneg_logits(new_embs: torch.Tensor, pos: torch.Tensor, neg: torch.Tensor, diff: bool = True) -> torch.Tensor:
    """`Trainer.py:557-577`: logits[:, i] = cos(emb, pos_i) - cos(emb, neg_i) (or pos only).
    pos/neg: [C,128] prompt-mean vectors."""
    cp = pairwise_cosine_similarity(new_embs, pos)
    if not diff:
        return cp
    return cp - pairwise_cosine_similarity(new_embs, neg)


def bce_with_logits_mean(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """nn.BCEWithLogitsLoss() default (mean over B*C): `ZERO_JOINT_BOUNDS.py:36`, `Trainer.py:582`."""
    return F.binary_cross_entropy_with_logits(logits, labels)


def infonce(img: torch.Tensor, txt: torch.Tensor, temperature: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """North-star head (NOT in the reference): L2-normalise (F.normalize, eps 1e-12),
    S = I_hat @ T_hat.T / tau, loss = (CE(S, diag) + CE(S.T, diag)) / 2.  Returns (loss, S)."""
    i = F.normalize(img, dim=1)
    t = F.normalize(txt, dim=1)
    s = i @ t.T / temperature
    tgt = torch.arange(s.shape[0])
    loss = 0.5 * (F.cross_entropy(s, tgt) + F.cross_entropy(s.T, tgt))
    return loss, s


def zero_shot_scores(img_emb: torch.Tensor, text_mean: torch.Tensor) -> torch.Tensor:
    """`vlp/inference_engine.py:45-55` batched as `trash/lower_bound_mcs.py:79-117`:
    normalize(img) @ normalize(mean-of-prompts).T -> [B,C]."""
    return F.normalize(img_emb, dim=-1) @ F.normalize(text_mean, dim=-1).T
