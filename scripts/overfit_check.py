#!/usr/bin/env python3
"""Does the joint step TRAIN?  Overfit one small batch of (image, text) pairs with the full ResNet-50 + a small CXR-BERT and print the
loss trace: a symmetric InfoNCE over B pairs starts near ln B and must fall towards 0 as the encoders memorise the pairing.
    python scripts/overfit_check.py [--batch 32] [--steps 300] [--lr 3e-5] [--precision split_bf16]"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import structured_images  # noqa: E402
from incremental_multimodal_medical_learning_ii_amd import _lib, synthetic as syn  # noqa: E402
from incremental_multimodal_medical_learning_ii_amd.contrastive import JointContrastiveTrainer  # noqa: E402
from incremental_multimodal_medical_learning_ii_amd.health_multimodal.image.model import get_biovil_resnet  # noqa: E402
from incremental_multimodal_medical_learning_ii_amd.health_multimodal.text import CXRBertConfig, CXRBertModel  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--lr", type=float, default=3e-5)
ap.add_argument("--size", type=int, default=64)
ap.add_argument("--precision", default="split_bf16")
args = ap.parse_args()
_lib.set_precision(args.precision)
dev = "cuda"
im = get_biovil_resnet(None).eval()
tm = CXRBertModel(CXRBertConfig(vocab_size=2048, hidden_size=128, num_attention_heads=2, intermediate_size=256, num_hidden_layers=2,
                                max_position_embeddings=32)).eval()
syn.fill_module_(im)
syn.fill_module_(tm)
im.to(dev)
images = structured_images(args.batch, args.size, seed=7).to(dev)
im.calibrate_batchnorm_(images)
ids, mask = syn.synthetic_tokens(args.batch, 16, vocab=2048, seed=8)
tr = JointContrastiveTrainer(im, tm.to(dev), lr=args.lr, temperature=0.07)
trace = [float(tr.step(images, ids.to(dev), mask.to(dev))) for _ in range(args.steps)]
print(f"ln B = {math.log(args.batch):.4f}; loss every {max(1, args.steps // 20)} steps:",
      " ".join(f"{x:.3f}" for x in trace[:: max(1, args.steps // 20)]), f"final {trace[-1]:.4f}")
