#!/usr/bin/env python3
"""Which learning rate / weight synthesis keeps the joint step of bench.py in a regime with LIVE gradients?

Round 2's bench ran Adam at lr 1e-4 on `synthetic.fill_module_` weights: one step collapses every embedding onto one direction
(loss = ln B exactly, InfoNCE cotangents ~ 0).  This script prints, for the plain fill and for BatchNorm statistics calibrated on a
sample batch (ImageModel.calibrate_batchnorm_): the spread of the initial embeddings (off-diagonal cosine), the norm of the loss
gradient w.r.t. the embeddings, and the loss trace of a few steps per learning rate.

    python scripts/exp_bench_regime.py [--batch 256] [--steps 12]
"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from bench import structured_images  # noqa: E402
from incremental_multimodal_medical_learning_ii_amd import functional as Fh  # noqa: E402
from incremental_multimodal_medical_learning_ii_amd import synthetic as syn  # noqa: E402
from incremental_multimodal_medical_learning_ii_amd.contrastive import JointContrastiveTrainer  # noqa: E402
from incremental_multimodal_medical_learning_ii_amd.health_multimodal.image.model import get_biovil_resnet  # noqa: E402
from incremental_multimodal_medical_learning_ii_amd.health_multimodal.text import CXRBertConfig, CXRBertModel  # noqa: E402


def cos_stats(e):
    n = torch.nn.functional.normalize(e.double(), dim=1)
    c = n @ n.T
    off = c[~torch.eye(len(c), dtype=torch.bool, device=c.device)]
    return float(off.mean()), float(off.std())


def head_grads(tr, images, ids, mask):
    with torch.no_grad():
        ie = tr.image_model(images)
        te = tr.text_model.get_projected_text_embeddings(ids, mask, normalize_embeddings=False)
    ie, te = ie.clone().requires_grad_(True), te.clone().requires_grad_(True)
    loss = Fh.infonce_loss(ie, te, tr.temperature)
    gi, gt = torch.autograd.grad(loss, (ie, te))
    return float(loss), ie.detach(), te.detach(), float(gi.norm()), float(gt.norm())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--lrs", default="1e-4,1e-5,1e-6,1e-7")
    args = ap.parse_args()
    dev = torch.device("cuda")
    B = args.batch
    im = get_biovil_resnet(None).eval()
    tm = CXRBertModel(CXRBertConfig()).eval()
    syn.fill_module_(im)
    syn.fill_module_(tm)
    tr = JointContrastiveTrainer(im.to(dev), tm.to(dev), lr=1e-4, temperature=0.07)
    batches = []
    for j in range(4):
        img = structured_images(B, 224, seed=27 + 101 * j).to(dev)
        ids, mask = syn.synthetic_tokens(B, 32, seed=28 + 101 * j)
        batches.append((img, ids.to(dev), mask.to(dev)))
    opt = tr.optimizer
    for variant in ("plain fill", "calibrated BN"):
        if variant == "calibrated BN":
            tr.image_model.calibrate_batchnorm_(batches[0][0][:64])
        p0 = opt.flat_p.clone()
        loss, ie, te, gi, gt = head_grads(tr, *batches[0])
        print(f"== {variant}: loss {loss:.4f} (ln B = {math.log(B):.4f}); off-diagonal cosine image {cos_stats(ie)} text {cos_stats(te)}; "
              f"|dL/dI| {gi:.3e} |dL/dT| {gt:.3e}; |I| row norm {float(ie.norm(dim=1).mean()):.3e} |T| {float(te.norm(dim=1).mean()):.3e}", flush=True)
        for lr in [float(s) for s in args.lrs.split(",")]:
            opt.flat_p.copy_(p0)
            opt.flat_m.zero_()
            opt.flat_v.zero_()
            opt.steps = 0
            opt.param_groups[0]["lr"] = lr
            trace = [float(tr.step(*batches[i % 4])) for i in range(args.steps)]
            _, ie2, te2, gi2, gt2 = head_grads(tr, *batches[0])
            print(f"   lr {lr:g}: loss {' '.join(f'{x:.4f}' for x in trace)} | after: cos image {cos_stats(ie2)[0]:.4f} text {cos_stats(te2)[0]:.4f} "
                  f"|dL/dI| {gi2:.3e} |dL/dT| {gt2:.3e}", flush=True)
        opt.flat_p.copy_(p0)


if __name__ == "__main__":
    main()
