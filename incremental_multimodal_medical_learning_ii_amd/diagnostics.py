"""Measurement helpers shared by `bench.py` and the `-m gpu` tests (instrumentation, not part of the step).

`imposed_decision_gradient_errors` answers the question a free-running cross-precision comparison cannot: do the split-bf16
(bf16x3) backward kernels reproduce the exact-fp32 gradients at a given batch?  A ReLU network's gradient is discontinuous in its
forward values; the two contraction precisions differ by ~1e-5 in the forward, so a few 1e-5 of the ReLU decisions (and some
max-pool winners) differ, and every flipped decision moves upstream gradients at the 1e-3 level.  The
fp32 forward's decisions are therefore captured on the device and imposed on the split-bf16 pass before its backward runs: both
backwards then differentiate the same piecewise-linear function.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch

from . import _lib
from . import image_encoder as IE


def imposed_decision_gradient_errors(model, images: torch.Tensor, cot: torch.Tensor):
    """(imposed, free, flips, emb_err): `imposed` / `free` = {parameter name: (max |diff| / max |ref|, ||diff|| / ||ref||)} of the
    split-bf16 image-encoder gradients of sum(embedding * cot) against the exact-fp32 ones, with the fp32 forward's decisions
    imposed / free-running; `flips` = how many decisions differed (`image_encoder.count_decision_differences`); `emb_err` = max
    relative difference of the embeddings.  Parameter `.grad`s are left as None (a flat optimiser re-attaches its views in
    `zero_grad`)."""
    named = [(n, p) for n, p in model.named_parameters() if not n.startswith("encoder.encoder.fc.")]

    def grads() -> Dict[str, torch.Tensor]:
        out = {n: p.grad.detach().clone() for n, p in named}
        for _, p in named:
            p.grad = None
        return out

    for _, p in named:
        p.grad = None
    old = _lib.get_precision()
    try:
        _lib.set_precision("fp32")
        e32 = model(images)
        dec32 = IE.device_decisions(e32.grad_fn.state)
        (e32 * cot).sum().backward()
        g32 = grads()
        _lib.set_precision("split_bf16")
        es = model(images)                                       # free-running: its own decisions
        flips = IE.count_decision_differences(dec32, IE.device_decisions(es.grad_fn.state))
        (es * cot).sum().backward()
        g_free = grads()
        es = model(images)
        IE.impose_decisions_(es.grad_fn, dec32)
        (es * cot).sum().backward()
        g_imp = grads()
    finally:
        _lib.set_precision(old)

    def errs(g) -> Dict[str, Tuple[float, float]]:
        return {n: (float((g[n] - g32[n]).abs().max() / g32[n].abs().max().clamp_min(1e-30)),
                    float((g[n] - g32[n]).norm() / g32[n].norm().clamp_min(1e-30))) for n, _ in named}
    emb_err = float((es.detach() - e32.detach()).abs().max() / e32.detach().abs().max())
    return errs(g_imp), errs(g_free), flips, emb_err
