"""Optimisers on the cxrk kernels: drop-in for `optim.Adam(params, lr=lr)` / `optim.SGD(params, lr=lr)` as
constructed in the reference's `Trainer.__init__` (Trainer.py:172-178; torch defaults betas (0.9,0.999), eps 1e-8).

All parameters are re-pointed at slices of ONE flat fp32 buffer (and their `.grad` at slices of a flat gradient
buffer), so an optimiser step is a single HBM-bound kernel launch (28 B/parameter) and a data-parallel gradient
all-reduce is a handful of large contiguous RCCL calls instead of one per tensor.  Parameters that were adjacent in
memory before (the fused q/k/v projections) stay adjacent; the physical layout of each tensor (e.g. channels_last
conv filters) is preserved.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch

from . import kernels as K


class _FlatOptimizer:
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float):
        plist: List[torch.nn.Parameter] = []
        seen = set()
        for p in params:
            if id(p) not in seen and p.requires_grad:
                seen.add(id(p))
                plist.append(p)
        if not plist:
            raise ValueError("optimizer got an empty parameter list")
        dev = plist[0].device  # step() raises for non-GPU parameters (the kernels have no CPU fallback)
        for p in plist:
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError("all parameters must be fp32 on one device")
        self.params = plist
        self.param_groups = [{"params": plist, "lr": lr}]
        # Layout of the flat buffers: parameter-list order, except that tensors sharing one storage (the fused q/k/v
        # projections) stay together in storage order.  It must NOT depend on allocation addresses: every data-parallel
        # rank has to lay its buffers out identically, or the flat all-reduce adds gradients of different tensors
        # (tests/test_dist_gpu.py caught exactly that with an address-sorted layout).
        groups: dict = {}
        for i, p in enumerate(plist):
            groups.setdefault(p.untyped_storage().data_ptr(), []).append(i)
        order = [i for idxs in groups.values() for i in sorted(idxs, key=lambda j: plist[j].storage_offset())]
        offs, total = {}, 0
        for i in order:
            offs[i] = total
            total += (plist[i].numel() + 3) // 4 * 4
        self.numel = total
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self._views = []
        with torch.no_grad():
            for i, p in enumerate(plist):
                o, n = offs[i], p.numel()
                if not _dense(p):
                    p.data = p.data.contiguous()
                pv = self.flat_p[o:o + n].as_strided(p.shape, p.stride())
                pv.copy_(p.data)
                p.data = pv
                gv = self.flat_g[o:o + n].as_strided(p.shape, p.stride())
                p.grad = gv
                self._views.append(gv)
        self.steps = 0

    def zero_grad(self, set_to_none: bool = False) -> None:
        """Zero the flat gradient buffer.  `set_to_none` is ignored: gradients must stay views of the flat buffer."""
        self.flat_g.zero_()
        for p, gv in zip(self.params, self._views):
            if p.grad is not gv:
                p.grad = gv

    def _sync_grads(self) -> None:
        for p, gv in zip(self.params, self._views):
            if p.grad is None:
                continue  # never received a gradient this step: its slice of flat_g is zero
            if p.grad.data_ptr() != gv.data_ptr():
                gv.copy_(p.grad)
                p.grad = gv

    def grad_span(self, params) -> Optional[tuple]:
        """(lo, hi) element range of the flat gradient buffer that holds exactly the gradients of `params`, or None when they
        do not form one gap-free range (then only the whole buffer can be reduced at once)."""
        ids = {id(p) for p in params}
        base = self.flat_g.data_ptr()
        lo, hi, tot = None, 0, 0
        for p, gv in zip(self.params, self._views):
            if id(p) in ids:
                o = (gv.data_ptr() - base) // 4
                n = (p.numel() + 3) // 4 * 4
                lo = o if lo is None else min(lo, o)
                hi, tot = max(hi, o + n), tot + n
        return (lo, hi) if lo is not None and hi - lo == tot else None

    def all_reduce_span(self, lo: int, hi: int, group=None, bucket_bytes: int = 256 << 20) -> list:
        """Start the sum of flat_g[lo:hi] over the group (async, ordered after the work already queued on the CURRENT stream);
        returns the work handles for `all_reduce_grads(skip=..., pending=...)` to wait on."""
        import torch.distributed as dist
        step = max(1, bucket_bytes // 4)
        return [dist.all_reduce(self.flat_g[o:min(hi, o + step)], group=group, async_op=True) for o in range(lo, hi, step)]

    def all_reduce_grads(self, group=None, bucket_bytes: int = 256 << 20, average: bool = False, skip=None,
                         pending: Optional[list] = None, pre_scale: Optional[float] = None) -> None:
        """Sum (or average) the flat gradient buffer over the data-parallel group in large contiguous buckets.  `skip` = element
        range(s) already being reduced by `all_reduce_span` (one (lo, hi) tuple or a list of them; their handles in `pending`).
        `pre_scale`: multiply this rank's gradients by it before the sum (shards of unequal size: rows / global rows)."""
        import torch.distributed as dist
        self._sync_grads()
        n = self.flat_g.numel()
        if pre_scale is not None:
            if skip:
                raise ValueError("all_reduce_grads: pre_scale cannot be combined with ranges that are already being reduced")
            K.scale_mask(self.flat_g, alpha=float(pre_scale), out=self.flat_g)
        spans = _complement(skip, n)
        works = list(pending or [])
        for lo, hi in spans:
            if hi > lo:
                works += self.all_reduce_span(lo, hi, group, bucket_bytes)
        for w in works:
            w.wait()
        if average:
            K.scale_mask(self.flat_g, alpha=1.0 / dist.get_world_size(group), out=self.flat_g)

    @property
    def lr(self) -> float:
        return float(self.param_groups[0]["lr"])


def _complement(skip, n: int) -> list:
    """[0, n) minus the (lo, hi) range(s) in `skip` (which must not overlap), as a sorted list of ranges."""
    if not skip:
        return [(0, n)]
    ranges = sorted([tuple(skip)] if isinstance(skip[0], int) else [tuple(r) for r in skip])
    out, pos = [], 0
    for lo, hi in ranges:
        if lo < pos or hi > n or hi < lo:
            raise ValueError(f"all_reduce_grads: skip ranges {ranges} overlap or leave the buffer of {n} elements")
        if lo > pos:
            out.append((pos, lo))
        pos = hi
    if pos < n:
        out.append((pos, n))
    return out


def _dense(p: torch.Tensor) -> bool:
    """True when the tensor's elements occupy one gap-free block (any dim permutation)."""
    return p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))


class Adam(_FlatOptimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        super().__init__(params, lr)
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        self.flat_m = torch.zeros_like(self.flat_p)
        self.flat_v = torch.zeros_like(self.flat_p)

    @torch.no_grad()
    def step(self, grad_scale: float = 1.0) -> None:
        self._sync_grads()
        self.steps += 1
        K.adam_fused(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.lr, self.betas[0], self.betas[1], self.eps,
                     self.weight_decay, self.steps, grad_scale)

    def state_dict(self):
        return {"step": self.steps, "exp_avg": self.flat_m, "exp_avg_sq": self.flat_v, "lr": self.lr}


class SGD(_FlatOptimizer):
    def __init__(self, params, lr: float = 1e-3, weight_decay: float = 0.0):
        super().__init__(params, lr)
        self.weight_decay = weight_decay

    @torch.no_grad()
    def step(self, grad_scale: float = 1.0) -> None:
        self._sync_grads()
        self.steps += 1
        K.sgd(self.flat_p, self.flat_g, self.lr, self.weight_decay, grad_scale)
