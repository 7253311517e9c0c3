"""Diagnostic (GPU): per-stage forward error and per-tensor gradient error of the HIP image encoder vs the CPU oracle
in fp32 and fp64."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from incremental_multimodal_medical_learning_ii_amd import synthetic as syn, image_encoder as IE, kernels as K
from incremental_multimodal_medical_learning_ii_amd.health_multimodal.image.model import get_biovil_resnet
from oracle import ref_image

size = int(sys.argv[1]) if len(sys.argv) > 1 else 224
model = get_biovil_resnet(None); syn.fill_module_(model); model.eval()
x = syn.synthetic_images(2, size, seed=27)
probe = torch.from_numpy(syn._normal("g3.probe", (2, 128)))

def oracle(dtype):
    p = {k: (v.detach().clone().to(dtype) if v.dtype == torch.float32 else v.clone()) for k, v in model.state_dict().items()}
    for k, v in p.items():
        if v.is_floating_point() and "running" not in k and ".fc." not in k:
            v.requires_grad_(True)
    coll = []
    emb = ref_image.image_model_forward(p, x.to(dtype), collect=coll)
    (emb * probe.to(dtype)).sum().backward()
    return emb.detach(), [c.detach() for c in coll], {k: v.grad for k, v in p.items() if v.requires_grad}

e32, c32, g32 = oracle(torch.float32)
e64, c64, g64 = oracle(torch.float64)
rel = lambda a, b: float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max())
print("oracle fp32 vs fp64: emb", rel(e32, e64), "stages", [f"{rel(a, b):.1e}" for a, b in zip(c32, c64)])
model.cuda().prepare_()
params, bufs = model._tensors()
specs, blocks = model._specs, model._blocks
p = [t.detach() for t in params]; b = [t.detach() for t in bufs]
emb, _, state = IE._forward(specs, blocks, p, b, x.cuda(), True, False)
fold, x0, stem, idx, pooled, binfo, cur, pj1, _ = state
stages = [pooled] + [binfo[i][3] for i in (2, 6, 12, 15)]
print("hip vs fp64: emb", rel(emb, e64), "stages", [f"{rel(K.nhwc_to_nchw(a), b):.1e}" for a, b in zip(stages, c64)])
print("hip per-block out vs fp64-propagated? (only stage ends available)")
grads = IE._backward(specs, blocks, p, b, state, probe.cuda(), None)
names = IE.param_names(specs)
errs = []
for n, g in zip(names, grads):
    if n in g64:
        errs.append((rel(g, g64[n]), rel(g32[n], g64[n]), n))
errs.sort(reverse=True)
print("worst HIP grads vs fp64 (hip_err, cpu32_err, name):")
for e in errs[:15]:
    print(f"  {e[0]:.2e} {e[1]:.2e} {e[2]}")
import statistics
print("median hip err", statistics.median(e[0] for e in errs), "median cpu32 err", statistics.median(e[1] for e in errs))
print("in execution order (name, hip_err, cpu32_err):")
byname = {e[2]: e for e in errs}
for n in names:
    if n in byname and (n.endswith("conv1.weight") or n.endswith("conv2.weight") or n.endswith("conv3.weight") or "downsample.0" in n or n.endswith(".bias") and "bn" in n or "projector" in n):
        print(f"  {byname[n][0]:.2e} {byname[n][1]:.2e} {n}")
