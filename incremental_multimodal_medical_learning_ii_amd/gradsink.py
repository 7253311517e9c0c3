"""Where the hand-written backward passes put parameter gradients.

PyTorch's autograd adds every gradient a `Function.backward` returns into `param.grad` with one elementwise kernel per tensor
(AccumulateGrad) — 367 launches per training step on this path in round 1.  The encoders instead hand the kernels the
parameter's own `.grad` memory (the flat gradient buffer of `optim._FlatOptimizer`) and let them ACCUMULATE into it, which is
exactly autograd's semantics; `backward` then returns None for those inputs.  A parameter whose `.grad` is missing or laid out
differently gets a fresh tensor that is returned to autograd as usual."""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch


def own_layout(g: Optional[torch.Tensor], p: torch.Tensor) -> bool:
    return g is not None and g.dtype == torch.float32 and g.device == p.device and g.shape == p.shape and g.stride() == p.stride()


def dense(t: torch.Tensor) -> bool:
    """the elements occupy one gap-free block in the parameter's own order (contiguous, or channels_last for filters)"""
    return t.is_contiguous() or (t.dim() == 4 and t.permute(0, 2, 3, 1).is_contiguous())


class GradSink:
    def __init__(self, params: Sequence[torch.Tensor], need: Optional[Sequence[bool]] = None):
        self.params = params
        self.need = list(need) if need is not None else [True] * len(params)
        self.ret: List[Optional[torch.Tensor]] = [None] * len(params)

    def dst(self, j: int, force_fresh: bool = False, zero: bool = False):
        """(tensor the kernels write, accumulate flag).  zero: a fresh tensor must start at zero (scatter-add targets)."""
        p = self.params[j]
        g = getattr(p, "grad", None)
        if not force_fresh and own_layout(g, p) and dense(g):
            return g, True
        t = torch.zeros_like(p) if zero else torch.empty_like(p)
        self.ret[j] = t
        return t, False

    def result(self) -> tuple:
        return tuple(g if n else None for g, n in zip(self.ret, self.need))
