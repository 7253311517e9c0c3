"""Diagnostic: image-encoder gradients in exact-fp32 mode against the CPU oracle under the same decisions, every tensor, twice."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from incremental_multimodal_medical_learning_ii_amd import _lib, synthetic as syn, image_encoder as IE
from incremental_multimodal_medical_learning_ii_amd.health_multimodal.image.model import get_biovil_resnet
from oracle import ref_image
_lib.set_precision(sys.argv[1] if len(sys.argv) > 1 else "fp32")
model = get_biovil_resnet(None); syn.fill_module_(model)
sd_cpu = {k: v.clone() for k, v in model.state_dict().items()}
model.to("cuda").eval()
x = syn.synthetic_images(2, 224, seed=27)
probe = torch.randn(2, 128, generator=torch.Generator().manual_seed(3))
ref = None
from incremental_multimodal_medical_learning_ii_amd import kernels as K
for rep in range(3):
    model.zero_grad(set_to_none=True)
    # poison the caching allocator's free blocks and drop the scratch buffer: every torch.empty() now returns NaNs, so a kernel
    # that reads what nothing wrote shows up as NaN gradients
    K._ws.bufs.clear()
    if os.environ.get("DIAG_BIG_WS"):      # as in a long test session: one large scratch buffer full of old data
        K.workspace(1 << 29, torch.device("cuda", 0)).fill_(1.0e6)
    junk = [torch.full((1 << 26,), float("nan"), device="cuda") for _ in range(8)] + [torch.full((n,), float("nan"), device="cuda") for n in (1 << 10, 1 << 14, 1 << 18, 1 << 22) for _ in range(16)]
    del junk
    with IE.capture_relu_decisions() as cap:
        emb = model(x.to("cuda"))
    (emb * probe.to("cuda")).sum().backward()
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().float().cpu().clone() for k, p in model.named_parameters() if p.grad is not None}
    if ref is None:
        p = {k: v.detach().clone() for k, v in sd_cpu.items()}
        for k, v in p.items():
            if v.is_floating_point() and "running" not in k and ".fc." not in k:
                v.requires_grad_(True)
        pol = ref_image.ReluPolicy(cap[0])
        e = ref_image.image_model_forward(p, x, relu=pol)
        (e * probe).sum().backward()
        ref = {k: v.grad for k, v in p.items() if v.requires_grad}
    bad = []
    for k, v in ref.items():
        err = float((grads[k] - v).abs().max() / v.abs().max().clamp_min(1e-30))
        if not torch.isfinite(grads[k]).all():
            err = float("nan")
        if not err < 1e-3:
            bad.append((k, err))
    print(f"rep {rep}: {len(bad)} of {len(ref)} tensors off by > 1e-3:", bad[:12], flush=True)
