"""Per-tensor gradient error of the image encoder against the CPU oracle run under the device's ReLU decisions
(the check of tests/test_models_gpu.py::test_image_model_forward_backward), all tensors listed, both precisions."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from incremental_multimodal_medical_learning_ii_amd import _lib, image_encoder as IE, synthetic as syn
from incremental_multimodal_medical_learning_ii_amd.health_multimodal.image.model import get_biovil_resnet
from oracle import ref_image

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g3_image.npz"))
probe = torch.from_numpy(g["probe"]).repeat((B + 1) // 2, 1)[:B]
for mode in ("fp32", "split_bf16"):
    _lib.set_precision(mode)
    model = get_biovil_resnet(None)
    syn.fill_module_(model)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model.to("cuda").eval()
    x = syn.synthetic_images(B, 224, seed=27)
    IE._debug = {}
    with IE.capture_relu_decisions() as cap:
        emb = model(x.to("cuda"))
    (emb * probe.to("cuda")).sum().backward()
    dbg, IE._debug = IE._debug, None
    p = {k: v.detach().clone() for k, v in sd.items()}
    for k, v in p.items():
        if v.is_floating_point() and "running" not in k and ".fc." not in k:
            v.requires_grad_(True)
    pol = ref_image.ReluPolicy(cap[0])
    coll = []
    e = ref_image.image_model_forward(p, x, relu=pol, collect=coll)
    stem_out = None
    (e * probe).sum().backward()
    # the stem's weight gradient recomputed on the host in fp64 from the DEVICE's ds: separates the wgrad kernel from errors in ds
    import torch.nn.functional as F
    ds = dbg["ds"].permute(0, 3, 1, 2).double().cpu()                       # gradient w.r.t. the stem's BN output (masked)
    w1 = p["encoder.encoder.conv1.weight"].detach().double()
    xin = x.double().requires_grad_(False)
    wz = torch.zeros_like(w1).requires_grad_(True)
    z = F.conv2d(xin, wz, stride=2, padding=3)
    (z * ds).sum().backward()
    rs = 1.0 / torch.sqrt(p["encoder.encoder.bn1.running_var"].double() + 1e-5)
    sc = (p["encoder.encoder.bn1.weight"].detach().double() * rs)[:, None, None, None]
    dw_host = wz.grad * sc
    gd = dict(model.named_parameters())["encoder.encoder.conv1.weight"].grad.detach().double().cpu()
    go = p["encoder.encoder.conv1.weight"].grad.double()
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    print(f"   stem dW: device vs host-fp64(device ds) {rel(gd, dw_host):.2e} | host-fp64(device ds) vs oracle {rel(dw_host, go):.2e} | "
          f"cancellation sum|ds*x| / |sum ds*x| ~ {float((ds.abs().sum()) * 0.5 / go.abs().max()):.1f}")
    named = dict(model.named_parameters())
    rows = []
    for k, v in p.items():
        if v.requires_grad and v.grad is not None:
            gd = named[k].grad.detach().float().cpu()
            rows.append((float((gd - v.grad).abs().max() / v.grad.abs().max().clamp_min(1e-30)), k))
    rows.sort(reverse=True)
    print(f"   max-pool winners differing from the oracle's own: {pol.pool_flips} (max value gap {pol.pool_max_gap:.1e})")
    print(f"== {mode} B={B}: emb err {float((emb.cpu() - e).abs().max() / e.abs().max()):.2e}; flips {pol.flips}/{pol.count} max_flip_rel {pol.max_flip_rel:.1e}")
    for er, k in rows[:12]:
        print(f"   {er:.3e}  {k}")
    print(f"   median {rows[len(rows)//2][0]:.2e}; > 1e-3: {sum(r[0] > 1e-3 for r in rows)}/{len(rows)}", flush=True)
