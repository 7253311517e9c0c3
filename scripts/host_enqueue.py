"""How far ahead of the GPU is the host?  Times the ENQUEUE of one joint step (no synchronisation) against its GPU time.
If enqueue time approaches the step time the step is launch-bound and the kernel sequence should be captured in a hipGraph.
usage: host_enqueue.py [batch]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from incremental_multimodal_medical_learning_ii_amd import _lib, synthetic as syn
from incremental_multimodal_medical_learning_ii_amd.contrastive import JointContrastiveTrainer
from incremental_multimodal_medical_learning_ii_amd.health_multimodal.image.model import get_biovil_resnet
from incremental_multimodal_medical_learning_ii_amd.health_multimodal.text import CXRBertConfig, CXRBertModel

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
_lib.set_precision("split_bf16")
im, tm = get_biovil_resnet(None).eval(), CXRBertModel(CXRBertConfig()).eval()
syn.fill_module_(im); syn.fill_module_(tm)
tr = JointContrastiveTrainer(im.cuda(), tm.cuda(), lr=1e-4)
images = syn.synthetic_images(B, 224, seed=1).cuda()
ids, mask = syn.synthetic_tokens(B, 32, seed=2)
ids, mask = ids.cuda(), mask.cuda()
for _ in range(3):
    tr.step(images, ids, mask)
torch.cuda.synchronize()
enq, tot = [], []
for _ in range(5):
    t0 = time.perf_counter()
    tr.step(images, ids, mask)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
print(f"batch {B}: host enqueue {sorted(enq)[2]:.1f} ms per step, step (enqueue + drain) {sorted(tot)[2]:.1f} ms")
