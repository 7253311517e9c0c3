"""`__graft_entry__.smoke()`: one small joint image+text contrastive training step on cuda:0, checked against the
CPU oracle (loss, gradients under the same ReLU decisions, parameters after the fused Adam step)."""
from __future__ import annotations

import torch


def run_smoke(verbose: bool = True) -> None:
    """Both contraction precisions of the library: split-bf16 (what bench.py reports) and exact fp32."""
    if not torch.cuda.is_available():
        raise RuntimeError("smoke() needs an MI355X (cuda:0)")
    from . import _lib
    old = _lib.get_precision()
    try:
        for mode in ("split_bf16", "fp32"):
            _lib.set_precision(mode)
            _run_one(mode, verbose)
    finally:
        _lib.set_precision(old)


def _run_one(mode: str, verbose: bool) -> None:
    from . import image_encoder as IE
    from . import synthetic as syn
    from .contrastive import JointContrastiveTrainer
    from .health_multimodal.image.model import get_biovil_resnet
    from .health_multimodal.text import CXRBertConfig, CXRBertModel
    from oracle import ref_image, ref_step  # checker only

    dev = torch.device("cuda:0")
    B, L, tau = 4, 16, 0.07
    cfg = CXRBertConfig(vocab_size=300, hidden_size=128, num_attention_heads=2, intermediate_size=256,
                        num_hidden_layers=2, max_position_embeddings=32)
    tm, im = CXRBertModel(cfg).eval(), get_biovil_resnet(None).eval()
    syn.fill_module_(tm)
    syn.fill_module_(im)
    images = syn.synthetic_images(B, 64, seed=3)
    ids, mask = syn.synthetic_tokens(B, L, vocab=300, seed=4, ragged=True)
    ip = {k: v.detach().clone() for k, v in im.state_dict().items()}
    tp = {k: v.detach().clone() for k, v in tm.state_dict().items()}

    tr = JointContrastiveTrainer(im.to(dev), tm.to(dev), lr=1e-4, temperature=tau)
    tr.optimizer.zero_grad()
    with IE.capture_relu_decisions() as cap:
        loss = tr.forward_loss(images.to(dev), ids.to(dev), mask.to(dev))
    masks = cap[0]
    loss.backward()
    g_img = dict(im.named_parameters())["encoder.encoder.layer2.0.conv2.weight"].grad.detach().cpu().clone()
    g_txt = dict(tm.named_parameters())["bert.encoder.layer.0.intermediate.dense.weight"].grad.detach().cpu().clone()
    tr.optimizer.step()
    torch.cuda.synchronize()

    leaves = []
    for d in (ip, tp):
        for k, v in d.items():
            if v.dtype == torch.float32 and "running" not in k and not k.startswith("cls.predictions") and ".fc." not in k:
                v.requires_grad_(True)
                leaves.append(v)
    opt = torch.optim.Adam(leaves, lr=1e-4)
    pol = ref_image.ReluPolicy(masks)
    loss_ref = ref_step.joint_step(ip, tp, images, ids, mask, tau, opt, n_layers=2, n_heads=2, relu=pol)

    def rel(a, b):
        return float((a.float().cpu() - b.float().cpu()).abs().max() / b.float().abs().max().clamp_min(1e-30))

    e_loss = abs(loss.item() - loss_ref.item()) / abs(loss_ref.item())
    e_gi = rel(g_img, ip["encoder.encoder.layer2.0.conv2.weight"].grad)
    e_gt = rel(g_txt, tp["bert.encoder.layer.0.intermediate.dense.weight"].grad)
    def rel_q(a, b):
        """Adam's first step moves a weight by ~lr*sign(g): an entry whose gradient is ~0 can take the other sign in two correct
        implementations, so the parameter check is on the 99.9th percentile of the error, not on its maximum."""
        d = (a.float().cpu() - b.float().cpu()).abs().flatten()
        q = d.kthvalue(max(1, int(0.999 * d.numel()))).values if d.numel() > 1 else d.max()
        return float(q / b.float().abs().max().clamp_min(1e-30))

    e_p = max(rel_q(im.state_dict()[k], v) for k, v in ip.items() if v.requires_grad)
    if verbose:
        print(f"[smoke {mode}] loss hip={loss.item():.6f} oracle={loss_ref.item():.6f} rel={e_loss:.2e}; grad rel err image={e_gi:.2e} "
              f"text={e_gt:.2e}; params after Adam rel={e_p:.2e}; relu decisions flipped={pol.flips}/{pol.count}")
    assert e_loss < 1e-3 and e_gi < 1e-3 and e_gt < 1e-3 and e_p < 1e-3, (e_loss, e_gi, e_gt, e_p)
