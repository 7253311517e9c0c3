"""Conditioning study (DESIGN §9 / VERDICT r1 item 4): BatchNorm gamma gradient taken from the weight gradient,
dgamma = rstd * (<w, dW_raw> - mean * sum(dy)), against the direct form sum(dy * (y_bn - beta)) / gamma, on the real ResNet-50
at the parity-test size (batch 2) and at a larger batch, in both contraction precisions.  Reference = direct form in exact fp32.
Prints, per configuration, the worst per-tensor error (max |diff| / max |ref|) over the 53 BN weight tensors."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from incremental_multimodal_medical_learning_ii_amd import _lib, image_encoder as IE, synthetic as syn
from incremental_multimodal_medical_learning_ii_amd.health_multimodal.image.model import get_biovil_resnet


def grads(B, mode, wgrad, size=224):
    _lib.set_precision(mode)
    IE.DGAMMA_FROM_WGRAD = wgrad
    m = get_biovil_resnet(None)
    syn.fill_module_(m)
    m.to("cuda").eval()
    x = syn.synthetic_images(B, size, seed=27).to("cuda")
    probe = torch.from_numpy(syn._normal("g3.probe", (2, 128))).to("cuda").repeat((B + 1) // 2, 1)[:B]
    (m(x) * probe).sum().backward()
    return {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None and p.dim() == 1 and "bn" in n or "downsample.1" in n or n == "projector.model.1.weight"}


for B in (2, 32):
    ref = grads(B, "fp32", False)
    for mode in ("fp32", "split_bf16"):
        for wg in (False, True):
            g = grads(B, mode, wg)
            worst = ("", 0.0)
            errs = []
            for n, r in ref.items():
                if not n.endswith("weight"):
                    continue
                e = float((g[n] - r).abs().max() / r.abs().max().clamp_min(1e-30))
                errs.append(e)
                if e > worst[1]:
                    worst = (n, e)
            errs.sort()
            print(f"B={B:3d} {mode:10s} dgamma_from_wgrad={wg!s:5s} worst {worst[1]:.3e} ({worst[0]}), median {errs[len(errs)//2]:.3e}, "
                  f">1e-3: {sum(e > 1e-3 for e in errs)}/{len(errs)}", flush=True)
