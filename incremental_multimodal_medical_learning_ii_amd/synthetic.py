"""Deterministic synthetic weights and inputs (no network, no checkpoints).

The BioViL / CXR-BERT checkpoints and the CheXpert data the reference uses
(`health_multimodal/image/model/model.py:27-33`, `Trainer.py:221-235`) are not
reachable offline, so every test, the oracle and `bench.py` fill tensors from a
*name-keyed* rule: the values of a tensor depend only on its state-dict name and
shape, never on a framework's RNG stream or on construction order.  The same
rule therefore regenerates identical weights in the survey container (where the
golden fixtures are produced against the reference's own modules) and on the
GPU box.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Tuple

import numpy as np
import torch


def _rng(name: str) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64(zlib.crc32(name.encode("utf-8"))))


def _normal(name: str, shape: Tuple[int, ...]) -> np.ndarray:
    return _rng(name).standard_normal(size=tuple(shape), dtype=np.float32)


def rule_tensor(name: str, shape: Iterable[int]) -> torch.Tensor:
    """Value of the parameter / buffer called `name` with `shape` (fp32, CPU).

    Kinds are recognised from the state-dict suffix:
      * BatchNorm `running_var`      -> 0.5 + |n|*0.5            (positive, non-trivial)
      * BatchNorm `running_mean`     -> 0.1 n
      * `num_batches_tracked`        -> 0
      * norm `weight` (1-D)          -> 1 + 0.1 n  (x0.5 for the last BN of a bottleneck: tames residual growth)
      * any `bias` (1-D)             -> 0.05 n
      * embeddings                   -> 0.05 n
      * conv weight (4-D)            -> n * sqrt(2 / fan_in)
      * linear weight (2-D)          -> n * sqrt(1 / fan_in)
    """
    shape = tuple(int(s) for s in shape)
    if name.endswith("num_batches_tracked"):
        return torch.zeros(shape, dtype=torch.int64)
    if name.endswith("position_ids"):
        return torch.arange(shape[-1], dtype=torch.int64).reshape(shape)
    n = _normal(name, shape)
    if name.endswith("running_var"):
        out = 0.5 + 0.5 * np.abs(n)
    elif name.endswith("running_mean"):
        out = 0.1 * n
    elif name.endswith("bias"):
        out = 0.05 * n
    elif "embeddings" in name and len(shape) == 2 and "LayerNorm" not in name:
        out = 0.05 * n
    elif len(shape) == 1:  # norm weight
        out = 1.0 + 0.1 * n
        if name.endswith("bn3.weight"):
            out = 0.5 * out
    elif len(shape) == 4:
        fan_in = shape[1] * shape[2] * shape[3]
        out = n * np.sqrt(2.0 / fan_in)
    elif len(shape) == 2:
        out = n * np.sqrt(1.0 / shape[1])
    else:
        out = n
    return torch.from_numpy(np.ascontiguousarray(out, dtype=np.float32))


@torch.no_grad()
def fill_module_(module: torch.nn.Module, prefix: str = "") -> torch.nn.Module:
    """Overwrite every parameter and buffer of `module` in place with `rule_tensor(prefix+name)`.

    Tied parameters (e.g. the MLM decoder tied to the word embeddings) are visited once, under the
    first name `named_parameters()` reports, exactly as `state_dict` round-trips them.
    """
    seen = set()
    for name, p in list(module.named_parameters()) + list(module.named_buffers()):
        if id(p) in seen:
            continue
        seen.add(id(p))
        v = rule_tensor(prefix + name, p.shape)
        p.copy_(v.to(dtype=p.dtype, device=p.device))
    return module


def rule_state_dict(names_shapes: Dict[str, Tuple[int, ...]], prefix: str = "") -> Dict[str, torch.Tensor]:
    return {k: rule_tensor(prefix + k, s) for k, s in names_shapes.items()}


# ----------------------------------------------------------------------------------------------
# synthetic inputs
# ----------------------------------------------------------------------------------------------

def synthetic_images(batch: int, size: int = 224, seed: int = 27) -> torch.Tensor:
    """[B,3,size,size] fp32 in [0,1): one grayscale plane replicated to 3 channels, no mean/std
    normalisation — what the reference feeds its encoder (`DataRetrieval.py:175-180`,
    `health_multimodal/image/data/transforms.py:12-38`)."""
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(batch, 1, size, size, generator=g, dtype=torch.float32)
    return u.repeat_interleave(3, dim=1).contiguous()


def synthetic_tokens(batch: int, seq_len: int = 32, vocab: int = 30522, seed: int = 28,
                     ragged: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """int64 ids [B,L] uniform over the vocabulary and an attention mask (all ones, or right-padded
    with lengths ~U{8..L} when `ragged`)."""
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(0, vocab, (batch, seq_len), generator=g, dtype=torch.int64)
    mask = torch.ones(batch, seq_len, dtype=torch.int64)
    if ragged:
        lens = torch.randint(min(8, seq_len), seq_len + 1, (batch,), generator=g)
        mask = (torch.arange(seq_len)[None, :] < lens[:, None]).to(torch.int64)
        ids = ids * mask
    return ids, mask


def synthetic_adapter_batch(batch: int, n_classes: int = 5, n_prompts: int = 4, dim: int = 128,
                            seed: int = 29):
    """T-ref inputs: pre-computed image embeddings [B,128], multi-hot labels [B,5] (p=0.3) and the
    frozen CXR-BERT outputs for 5 classes x (pos,neg) x 4 prompts: [10,4,128]
    (`Trainer.py:537-575`, `DataRetrieval.py:183-237`)."""
    g = torch.Generator().manual_seed(seed)
    embs = torch.randn(batch, dim, generator=g)
    labels = (torch.rand(batch, n_classes, generator=g) < 0.3).float()
    bert_out = torch.randn(2 * n_classes, n_prompts, dim, generator=g)
    return embs, labels, bert_out
