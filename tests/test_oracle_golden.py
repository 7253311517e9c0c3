"""The CPU oracle against the committed golden fixtures (tests/golden/*.npz).  G1 (text), G2 (adapters) and the
projector part of G3 were produced by the reference's own modules (oracle/gen_golden.py), so this pins the
restatement; the ResNet trunk, pairwise cosine and InfoNCE fixtures are the restatement's own ("parity unpinned")."""
import os

import numpy as np
import pytest
import torch

from incremental_multimodal_medical_learning_ii_amd import synthetic as syn
from oracle import ref_image, ref_loss, ref_step, ref_text


def T(x):
    return torch.from_numpy(np.asarray(x))


def rel(a, b):
    a, b = torch.as_tensor(a).float(), torch.as_tensor(b).float()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def test_g1_text_tiny_vs_reference_outputs(golden_dir):
    g = np.load(f"{golden_dir}/g1_text_tiny.npz")
    sd = {k[3:]: T(g[k]) for k in g.files if k.startswith("w::")}
    ids = T(g["ids"])
    for tag in ("full", "ragged"):
        mask = T(g["mask_" + tag])
        h = ref_text.cxrbert_last_hidden(sd, ids, mask, 2, 4)
        assert rel(h[:, 0], g["last_hidden_" + tag][:, 0]) < 1e-5
        assert rel(ref_text.projection_head(sd, h[:, 0]), g["proj_" + tag]) < 1e-5
        assert rel(ref_text.mlm_logits(sd, h)[:, 0], g["mlm_logits_cls_" + tag]) < 1e-5
    # backward: autograd over the restatement == gradients of the reference model
    for v in sd.values():
        if v.dtype == torch.float32:
            v.requires_grad_(True)
    proj = ref_text.cxrbert_projected(sd, ids, T(g["mask_ragged"]), 2, 4)
    (proj * T(g["probe"])).sum().backward()
    for name in ("bert.encoder.layer.0.attention.self.query.weight", "bert.embeddings.word_embeddings.weight",
                 "cls_projection_head.dense_to_hidden.weight", "bert.encoder.layer.1.output.LayerNorm.bias"):
        assert rel(sd[name].grad, g["g::" + name]) < 1e-4, name


def test_g1_text_full_config_rule_weights(golden_dir):
    g = np.load(f"{golden_dir}/g1_text_full.npz")
    shapes = ref_text.cxrbert_param_shapes()
    p = syn.rule_state_dict(shapes)
    ids = T(g["ids"])
    for tag, mask in (("full", torch.ones(4, 32, dtype=torch.int64)), ("ragged", T(g["mask_ragged"]))):
        assert rel(ref_text.cxrbert_projected(p, ids, mask), g["proj_" + tag]) < 1e-4


def test_g2_adapter_steps(golden_dir):
    g = np.load(f"{golden_dir}/g2_adapter_step.npz")
    ip = {k.split("image_adapter.")[1]: T(g[k]).clone().requires_grad_(True) for k in g.files if k.startswith("w0::image_adapter.")}
    tp = {k.split("text_adapter.")[1]: T(g[k]).clone().requires_grad_(True) for k in g.files if k.startswith("w0::text_adapter.")}
    opt = torch.optim.Adam(list(tp.values()) + list(ip.values()), lr=1e-4)
    embs, labels, bert_out = T(g["embs"]), T(g["labels"]), T(g["bert_out"])
    for step in (1, 2, 3):
        loss, logits = ref_step.adapter_step(ip, tp, embs, labels, bert_out, opt)
        assert abs(float(loss) - float(g[f"loss_step{step}"])) < 1e-6
        if step == 1:
            assert rel(logits, g["logits_step1"]) < 1e-5
    for k, v in ip.items():
        assert rel(v, g["w3::image_adapter." + k]) < 1e-5
    assert 0.68 < float(g["loss_step1"]) < 0.70  # ~ln 2, the reference's own first-step train/Loss
    sc, pr, _ = ref_step.eval_scores({k: v.detach() for k, v in ip.items()}, {k: v.detach() for k, v in tp.items()}, embs, bert_out)
    assert rel(sc, g["eval_score"]) < 1e-5 and torch.equal(pr, T(g["eval_pred"]))


def test_g3_image(golden_dir):
    g = np.load(f"{golden_dir}/g3_image.npz")
    prm, buf = ref_image.image_param_shapes()
    p = {k: syn.rule_tensor(k, s) for k, s in {**prm, **buf}.items()}
    assert rel(ref_image.projector(p, T(g["proj_patch_in"])), g["proj_patch_out"]) < 1e-5  # reference modules.MLP output
    emb = ref_image.image_model_forward(p, syn.synthetic_images(2, 224, seed=27))
    assert rel(emb, g["emb"]) < 1e-5


def test_g7_trunk_vs_independent_resnet(golden_dir):
    """G7: stage outputs and probed gradients of the ResNet-50 trunk as computed by an INDEPENDENT implementation of the architecture
    (HuggingFace transformers.ResNetModel, torchvision-v1.5 stride placement, same name-keyed weights; oracle/gen_golden.py::gen_g7).
    The reference's own torchvision 0.10 is absent, so this is the pin available for `oracle/ref_image.resnet50_trunk`."""
    g = np.load(f"{golden_dir}/g7_trunk_hf.npz")
    prm, buf = ref_image.image_param_shapes()
    p = {k: syn.rule_tensor(k, s) for k, s in {**prm, **buf}.items()}
    for k, v in p.items():
        if v.dtype == torch.float32:
            v.requires_grad_("running" not in k)
    coll = []
    patch = ref_image.resnet50_trunk(p, syn.synthetic_images(2, 224, seed=27), collect=coll)
    assert len(coll) == 5
    for i, c in enumerate(coll):
        assert abs(float(c.double().sum()) - float(g[f"stage{i}_sum"])) <= 1e-6 * float(g[f"stage{i}_absmean"]) * c.numel(), i
        assert rel(c[:, :4, :3, :3], g[f"stage{i}_corner"]) < 1e-6, i
    (patch * T(g["probe"])).sum().backward()
    checked = 0
    for k in g.files:
        if k.startswith("g::"):
            ref = T(g[k])
            got = p[k[3:]].grad.flatten()[: ref.numel()].reshape(ref.shape) if ref.numel() != p[k[3:]].numel() else p[k[3:]].grad
            assert rel(got, ref) < 1e-5, k
            assert abs(float(p[k[3:]].grad.double().norm()) - float(g["gnorm::" + k[3:]])) / float(g["gnorm::" + k[3:]]) < 1e-6, k
            checked += 1
    assert checked == 8


def test_g4_g5_heads(golden_dir):
    g = np.load(f"{golden_dir}/g4_infonce.npz")
    for tau in (1.0, 0.07):
        loss, s = ref_loss.infonce(T(g["I"]), T(g["T"]), tau)
        assert abs(float(loss) - float(g[f"loss_tau{tau}"])) < 1e-6 and rel(s, g[f"S_tau{tau}"]) < 1e-6
    # cosine restatement == the reference's commented legacy form F.normalize(x) @ F.normalize(y).T (Trainer.py:1684-1686)
    x, y = T(g["I"]), T(g["T"])[:5]
    legacy = torch.nn.functional.normalize(x, dim=-1) @ torch.nn.functional.normalize(y, dim=-1).T
    assert rel(ref_loss.pairwise_cosine_similarity(x, y), legacy) < 1e-6
    z = np.load(f"{golden_dir}/g5_zeroshot.npz")
    sc = ref_loss.zero_shot_scores(T(z["img"]), T(z["txt"]).mean(1))
    assert rel(sc, z["scores"]) < 1e-6 and torch.equal(sc.argmax(1), T(z["argmax"]))


def test_cosine_and_infonce_against_independent_implementations():
    """The two heads the reference holds no fixture for, against implementations that share no code with the restatement:
    `pairwise_cosine_similarity` (torchmetrics is absent) vs scikit-learn's `cosine_similarity` (a pinned dependency of the reference,
    requirements.txt), incl. its gradient by central differences in float64; `infonce` vs a numpy / scipy float64 evaluation of
    (CE(S, diag) + CE(S^T, diag)) / 2 and its autograd gradient vs the closed form (softmax_row + softmax_col - 2 I) / (2 B tau)."""
    from scipy.special import logsumexp, softmax
    from sklearn.metrics.pairwise import cosine_similarity
    g = torch.Generator().manual_seed(11)
    x, y = torch.randn(37, 128, generator=g), torch.randn(10, 128, generator=g) * 3.0
    assert rel(ref_loss.pairwise_cosine_similarity(x, y), torch.from_numpy(cosine_similarity(x.numpy(), y.numpy()))) < 1e-6
    xd = x.double().requires_grad_(True)
    w = torch.randn(37, 10, generator=g).double()
    (ref_loss.pairwise_cosine_similarity(xd, y.double()) * w).sum().backward()
    i, j, h = 5, 77, 1e-6
    xp, xm = x.double().numpy().copy(), x.double().numpy().copy()
    xp[i, j] += h
    xm[i, j] -= h
    fd = ((cosine_similarity(xp, y.double().numpy()) - cosine_similarity(xm, y.double().numpy())) * w.numpy()).sum() / (2 * h)
    assert abs(float(xd.grad[i, j]) - fd) < 1e-6 * max(1.0, abs(fd))
    for tau in (1.0, 0.07):
        I, Tt = torch.randn(24, 128, generator=g).requires_grad_(True), torch.randn(24, 128, generator=g)
        loss, S = ref_loss.infonce(I, Tt, tau)
        In = I.detach().double().numpy() / np.linalg.norm(I.detach().double().numpy(), axis=1, keepdims=True)
        Tn = Tt.double().numpy() / np.linalg.norm(Tt.double().numpy(), axis=1, keepdims=True)
        Sn = In @ Tn.T / tau
        ref = 0.5 * ((logsumexp(Sn, axis=1) - np.diag(Sn)).mean() + (logsumexp(Sn, axis=0) - np.diag(Sn)).mean())
        assert abs(float(loss) - ref) < 2e-6 * max(1.0, abs(ref)) and np.abs(S.detach().numpy() - Sn).max() < 1e-4
        loss.backward()
        dS = (softmax(Sn, axis=1) + softmax(Sn, axis=0) - 2 * np.eye(24)) / (2 * 24)          # dL/dS
        dIn = dS @ Tn / tau                                                                   # dL/d(normalised I)
        nI = np.linalg.norm(I.detach().double().numpy(), axis=1, keepdims=True)
        dI = (dIn - In * (In * dIn).sum(1, keepdims=True)) / nI                               # through x / |x|
        assert np.abs(I.grad.numpy() - dI).max() < 1e-5 * np.abs(dI).max()


def test_g6_similarity_map_and_max_emb(golden_dir):
    """G6: the similarity-map vectors are outputs of the reference's own `_get_similarity_map_from_embeddings` /
    `convert_similarity_to_image_size`; the oracle restatement and the product's host-side resize must reproduce them exactly."""
    from incremental_multimodal_medical_learning_ii_amd.health_multimodal.vlp import ImageTextInferenceEngine as E
    g = np.load(os.path.join(golden_dir, "g6_simmap_maxemb.npz"))
    sim = ref_loss.similarity_map(T(g["patches"]), T(g["text"]))
    assert torch.equal(sim, T(g["sim"]))
    for i in range(4):
        w, h, rs, cs = (int(v) for v in g[f"resize{i}_args"])
        rs, cs = (None if rs < 0 else rs), (None if cs < 0 else cs)
        mode = str(g[f"resize{i}_mode"])
        exp = g[f"resize{i}_out"]
        assert exp.shape == (h, w)
        assert np.array_equal(ref_loss.similarity_to_image_size(sim, w, h, rs, cs, mode), exp, equal_nan=True)
        got = E.convert_similarity_to_image_size(sim, width=w, height=h, resize_size=rs, crop_size=cs, interpolation=mode)
        assert np.array_equal(got, exp, equal_nan=True), (i, mode)
        side = None if cs is None else (cs if rs is None else int(cs * min(w, h) / rs))
        assert bool(np.isnan(exp).any()) == (side is not None and (side < w or side < h))   # NaN border only where a crop hid pixels
    x, y = T(g["x"]).requires_grad_(True), T(g["y"]).requires_grad_(True)
    mx, mean, idx = ref_loss.pairwise_cosine_max(x, y, 10)
    assert rel(mx, T(g["max"])) < 1e-6 and rel(mean, T(g["mean"])) < 1e-6 and torch.equal(idx.int(), T(g["argmax"]))
    full = ref_loss.pairwise_cosine_similarity(x, y).reshape(48, 10, 4)
    assert torch.equal(mx, full.amax(2)) and bool((mx >= mean).all())
    (mx * T(g["dmax"])).sum().backward()
    assert rel(x.grad, T(g["dx"])) < 1e-6 and rel(y.grad, T(g["dy"])) < 1e-6


def test_weight_reset_oracle_edge_cases():
    new, old = torch.tensor([1.0, 2.0, 3.0, 4.0]), torch.tensor([1.0, 2.5, 3.0, 0.0])
    out, n = ref_step.weight_reset(new, old, 0.0)   # threshold 0 -> nothing strictly below the minimum diff
    assert n == 0 and torch.equal(out, new)
    out, n = ref_step.weight_reset(new, old, 1.0)   # everything strictly below the max is restored
    assert n == 3 and out[3] == 4.0
