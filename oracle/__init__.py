"""CPU oracle — TEST INFRASTRUCTURE ONLY.

A plain PyTorch-CPU fp32 restatement of the reference's hot path
(SURVEY.md §8a): ResNet-50 `ImageModel`, `CXRBertModel` + projection head,
`myMLP` adapters, torchmetrics-style pairwise cosine, pos-neg BCE-with-logits,
Adam, and the north-star InfoNCE head.  Every function cites the reference
file:line it follows.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg
may import this package, and only as the checker / the timed CPU baseline.
The product package (`incremental_multimodal_medical_learning_ii_amd`) never
imports it and has no CPU fallback.

Pinning status:
  * text encoder, projection head      pinned against the reference's own `CXRBertModel`
                                       (by-path import in the build container; fixtures G1)
  * adapters (`models.myMLP`)          pinned against the reference's `models.py` (fixture G2)
  * image projector (`modules.MLP`)    pinned against the reference's `modules.py` (fixture G3)
  * ResNet-50 trunk                    parity unpinned: torchvision 0.10 is absent and the reference
                                       holds no golden vectors; restated from `resnet.py:25-47` and
                                       torchvision's ResNet-50 v1.5 Bottleneck definition
  * pairwise cosine                    parity unpinned: torchmetrics absent/unpinned; restated from its
                                       published formula, cross-checked with the reference's commented
                                       legacy form (`Trainer.py:1684-1686`)
  * InfoNCE head, encoder backward     not in the reference (north-star superset): oracle = autograd
                                       over this restatement
"""
