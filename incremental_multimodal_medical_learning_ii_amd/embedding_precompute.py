"""Batched image-embedding pre-compute — the producer of the `TensorDataset([N,128],[N,5])` files the reference's
training loops consume.  Reference: `chexpert-get-embedding.py:57-113` (frozen eval-mode encoder,
512x512 un-normalised images, batch size 1, a checkpoint file every 5000 images, then a final one) and
`CSV_reformatting/glue_dataset.py:33-38` (chunks glued back into one dataset).

Here the frozen `ImageModel` runs at a large batch on the HIP forward kernels; chunks are written as plain tensor
dicts `{"embs": [n,128], "labels": [n,5]}` (loadable with `weights_only=True`, which is what `Trainer._preprocessing`
uses) instead of pickled `TensorDataset` objects."""
from __future__ import annotations

import os
from typing import Iterable, List, Optional, Tuple

import torch


@torch.no_grad()
def compute_embeddings(image_model: torch.nn.Module, batches: Iterable[Tuple[torch.Tensor, torch.Tensor]],
                       out_dir: Optional[str] = None, checkpoint_interval: int = 5000,
                       final_name: str = "embeddings_dataset_final_old.pt") -> Tuple[torch.Tensor, torch.Tensor]:
    """Encode every (images [b,3,H,W] in [0,1], labels [b,5]) batch; returns (embs [N,128], labels [N,5]) on the CPU.
    With `out_dir`, also writes `embeddings_dataset_<count>.pt` every `checkpoint_interval` images and the glued
    `final_name` at the end (same directory layout as the reference's `embeddingDataset/<split>/<variant>/`)."""
    image_model.train(mode=False, my_freeze=True) if hasattr(image_model, "prepare_") else image_model.eval()
    image_model.eval()
    device = next(image_model.parameters()).device
    embs: List[torch.Tensor] = []
    labs: List[torch.Tensor] = []
    pend_e: List[torch.Tensor] = []
    pend_l: List[torch.Tensor] = []
    count = pending = 0
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
    for images, labels in batches:
        e = image_model(images.to(device, non_blocking=True))
        pend_e.append(e)
        pend_l.append(labels.to(device))
        count += images.shape[0]
        pending += images.shape[0]
        if pending >= checkpoint_interval:
            ce, cl = torch.cat(pend_e).cpu(), torch.cat(pend_l).cpu().float()
            embs.append(ce)
            labs.append(cl)
            if out_dir:
                torch.save({"embs": ce, "labels": cl}, os.path.join(out_dir, f"embeddings_dataset_{count}.pt"))
            pend_e, pend_l, pending = [], [], 0
    if pend_e:
        embs.append(torch.cat(pend_e).cpu())
        labs.append(torch.cat(pend_l).cpu().float())
    all_e, all_l = torch.cat(embs), torch.cat(labs)
    if out_dir:
        torch.save({"embs": all_e, "labels": all_l}, os.path.join(out_dir, final_name))
    return all_e, all_l


def synthetic_image_batches(n: int, batch: int, size: int = 512, seed: int = 27):
    """Synthetic stand-in for the CheXpert loader: replicated-grayscale images in [0,1), Bernoulli(0.3) labels."""
    from .synthetic import synthetic_images
    g = torch.Generator().manual_seed(seed + 1)
    for start in range(0, n, batch):
        b = min(batch, n - start)
        yield synthetic_images(b, size, seed=seed + start), (torch.rand(b, 5, generator=g) < 0.3).float()
