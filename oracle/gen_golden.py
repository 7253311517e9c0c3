"""Generate tests/golden/*.npz.  TEST INFRASTRUCTURE ONLY — run in the build container, where
/root/reference exists; the GPU box only ever sees the arrays this writes.

    python -m oracle.gen_golden            # from the repo root

Where the reference's own Python modules can be imported (by file path, with empty stub packages
so the torchvision-importing `__init__`s are bypassed — SURVEY.md §8c) their outputs are the expected
values and the restatement in `oracle/` is asserted against them first:
  G1  `health_multimodal/text/model/modelling_cxrbert.py`  CXRBertModel          (text encoder + head)
  G2  `models.py`                                           myMLP, myLinearModel  (adapters) + torch BCE/Adam
  G3  `health_multimodal/image/model/modules.py`            MLP                   (image projector)
The ResNet-50 trunk cannot be checked against the reference's own torchvision 0.10 (absent): G7 checks the restatement against
an independent third-party implementation of the same architecture (transformers.ResNetModel) instead.  Pairwise cosine
(torchmetrics absent) and InfoNCE (not in the reference) are produced by the restatement alone: "parity unpinned".
No reference source text is stored — only inputs, weights the build's own rule generated, and outputs.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
REF = os.environ.get("CXR_REFERENCE_ROOT", "/root/reference")
OUT = os.path.join(REPO, "tests", "golden")

from incremental_multimodal_medical_learning_ii_amd import synthetic as syn  # noqa: E402
from oracle import ref_image, ref_loss, ref_step, ref_text  # noqa: E402


def _load_by_path(modname: str, path: str):
    spec = importlib.util.spec_from_file_location(modname, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference_text():
    for pkg in ("health_multimodal", "health_multimodal.text", "health_multimodal.text.model"):
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = []  # mark as package
            sys.modules[pkg] = m
    base = os.path.join(REF, "health_multimodal", "text", "model")
    cfg = _load_by_path("health_multimodal.text.model.configuration_cxrbert", os.path.join(base, "configuration_cxrbert.py"))
    mdl = _load_by_path("health_multimodal.text.model.modelling_cxrbert", os.path.join(base, "modelling_cxrbert.py"))
    return cfg, mdl


def np_(t):
    return t.detach().cpu().numpy().copy()  # copy: parameters are updated in place after being recorded


def relerr(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def gen_g1():
    cfg_m, mdl_m = load_reference_text()
    torch.manual_seed(27)
    # (i) tiny config, weights stored
    cfg = cfg_m.CXRBertConfig(vocab_size=128, hidden_size=64, num_attention_heads=4, intermediate_size=256,
                              num_hidden_layers=2, max_position_embeddings=64, projection_size=128)
    model = mdl_m.CXRBertModel(cfg).eval()
    syn.fill_module_(model)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    ids, _ = syn.synthetic_tokens(4, 32, vocab=128, seed=28)
    mask_full = torch.ones(4, 32, dtype=torch.int64)
    _, mask_rag = syn.synthetic_tokens(4, 32, vocab=128, seed=31, ragged=True)
    outs = {}
    for tag, mask in (("full", mask_full), ("ragged", mask_rag)):
        with torch.no_grad():
            r = model.forward(ids, mask, output_cls_projected_embedding=True, return_dict=False)
        last_hidden, logits, proj = r[0], r[1], r[2]
        mine_h = ref_text.cxrbert_last_hidden(sd, ids, mask, 2, 4)
        mine_p = ref_text.projection_head(sd, mine_h[:, 0, :])
        mine_l = ref_text.mlm_logits(sd, mine_h)
        # padded positions' hidden states are don't-care downstream of the mask only for CLS; compare CLS rows + proj
        assert relerr(mine_h[:, 0], last_hidden[:, 0]) < 2e-5, relerr(mine_h[:, 0], last_hidden[:, 0])
        assert relerr(mine_p, proj) < 2e-5, relerr(mine_p, proj)
        assert relerr(mine_l[:, 0], logits[:, 0]) < 2e-5
        print(f"G1 tiny/{tag}: restatement vs reference CXRBertModel rel err hidden {relerr(mine_h[:, 0], last_hidden[:, 0]):.2e} "
              f"proj {relerr(mine_p, proj):.2e}")
        outs[f"last_hidden_{tag}"] = np_(last_hidden)
        outs[f"proj_{tag}"] = np_(proj)
        outs[f"mlm_logits_cls_{tag}"] = np_(logits[:, 0])
    # gradient of a scalar probe through the reference model (pins the backward of the text encoder)
    model.zero_grad()
    r = model.forward(ids, mask_rag, output_cls_projected_embedding=True, return_dict=False)
    probe = torch.from_numpy(syn._normal("g1.probe", (4, 128)))
    (r[2] * probe).sum().backward()
    grads = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    np.savez_compressed(os.path.join(OUT, "g1_text_tiny.npz"), ids=np_(ids), mask_full=np_(mask_full),
                        mask_ragged=np_(mask_rag), probe=np_(probe),
                        **{"w::" + k: np_(v) for k, v in sd.items()},
                        **{"g::" + k: np_(v) for k, v in grads.items()}, **outs)

    # (ii) full 12-layer config: weights from the name-keyed rule, only inputs/outputs stored
    cfg = cfg_m.CXRBertConfig()
    model = mdl_m.CXRBertModel(cfg).eval()
    syn.fill_module_(model)
    sd = model.state_dict()
    ids, _ = syn.synthetic_tokens(4, 32, seed=28)
    _, mask_rag = syn.synthetic_tokens(4, 32, seed=31, ragged=True)
    outs = {}
    for tag, mask in (("full", torch.ones(4, 32, dtype=torch.int64)), ("ragged", mask_rag)):
        with torch.no_grad():
            r = model.forward(ids, mask, output_cls_projected_embedding=True, return_dict=False)
        mine = ref_text.cxrbert_projected(sd, ids, mask, 12, 12)
        assert relerr(mine, r[2]) < 5e-5, relerr(mine, r[2])
        print(f"G1 full/{tag}: restatement vs reference rel err {relerr(mine, r[2]):.2e}")
        outs[f"proj_{tag}"] = np_(r[2])
        outs[f"cls_hidden_{tag}"] = np_(r[0][:, 0])
    np.savez_compressed(os.path.join(OUT, "g1_text_full.npz"), ids=np_(ids), mask_ragged=np_(mask_rag), **outs)


def gen_g2():
    ref_models = _load_by_path("ref_models", os.path.join(REF, "models.py"))
    torch.manual_seed(27)
    img_ad, txt_ad = ref_models.myMLP(), ref_models.myMLP()
    syn.fill_module_(img_ad, "image_adapter.")
    syn.fill_module_(txt_ad, "text_adapter.")
    embs, labels, bert_out = syn.synthetic_adapter_batch(64)
    crit = torch.nn.BCEWithLogitsLoss()
    opt = torch.optim.Adam(list(txt_ad.parameters()) + list(img_ad.parameters()), lr=1e-4)  # order: Trainer.py:151,164
    # restated twin
    ip = {k: v.detach().clone().requires_grad_(True) for k, v in img_ad.state_dict().items()}
    tp = {k: v.detach().clone().requires_grad_(True) for k, v in txt_ad.state_dict().items()}
    opt2 = torch.optim.Adam(list(tp.values()) + list(ip.values()), lr=1e-4)
    rec = {"embs": np_(embs), "labels": np_(labels), "bert_out": np_(bert_out)}
    for k, v in img_ad.state_dict().items():
        rec["w0::image_adapter." + k] = np_(v)
    for k, v in txt_ad.state_dict().items():
        rec["w0::text_adapter." + k] = np_(v)
    for step in range(1, 4):
        # reference-side step written the way Trainer.train does it (Trainer.py:541-585)
        opt.zero_grad()
        new_embs = img_ad(embs)
        logits = torch.empty(64, 5)
        for c in range(5):
            pos = txt_ad(bert_out[2 * c]).mean(dim=0)
            neg = txt_ad(bert_out[2 * c + 1]).mean(dim=0)
            ps = ref_loss.pairwise_cosine_similarity(new_embs, pos.reshape(1, -1))
            ns = ref_loss.pairwise_cosine_similarity(new_embs, neg.reshape(1, -1))
            logits[:, c] = ps.flatten() - ns.flatten()
        loss = crit(logits, labels)
        loss.backward()
        if step == 1:
            rec["logits_step1"] = np_(logits)
            for k, p in img_ad.named_parameters():
                rec["g1::image_adapter." + k] = np_(p.grad)
            for k, p in txt_ad.named_parameters():
                rec["g1::text_adapter." + k] = np_(p.grad)
        opt.step()
        l2, lg2 = ref_step.adapter_step(ip, tp, embs, labels, bert_out, opt2)
        assert abs(float(l2) - float(loss)) < 1e-6
        rec[f"loss_step{step}"] = np.float32(loss.item())
        if step in (1, 3):
            for k, v in img_ad.state_dict().items():
                assert relerr(ip[k].detach(), v) < 1e-5
                rec[f"w{step}::image_adapter." + k] = np_(v)
            for k, v in txt_ad.state_dict().items():
                assert relerr(tp[k].detach(), v) < 1e-5
                rec[f"w{step}::text_adapter." + k] = np_(v)
    # eval scoring + class-incremental column subset + dense adapter
    sc, pr, dl = ref_step.eval_scores({k: v.detach() for k, v in ip.items()}, {k: v.detach() for k, v in tp.items()}, embs, bert_out)
    rec["eval_score"], rec["eval_pred"] = np_(sc), np_(pr)
    lin = ref_models.myLinearModel()
    syn.fill_module_(lin, "dense_adapter.")
    rec["dense_out"] = np_(lin(embs))
    print("G2 adapters: losses", [float(rec[f"loss_step{s}"]) for s in (1, 2, 3)])
    np.savez_compressed(os.path.join(OUT, "g2_adapter_step.npz"), **rec)


def gen_g3():
    modules = _load_by_path("ref_image_modules", os.path.join(REF, "health_multimodal", "image", "model", "modules.py"))
    prm, buf = ref_image.image_param_shapes()
    p = {k: syn.rule_tensor(k, s) for k, s in {**prm, **buf}.items()}
    # projector pinned against the reference's modules.MLP
    mlp = modules.MLP(input_dim=2048, output_dim=128, hidden_dim=128, use_1x1_convs=True).eval()
    mlp.load_state_dict({k[len("projector."):]: v for k, v in p.items() if k.startswith("projector.")})
    patch = torch.from_numpy(syn._normal("g3.patch", (2, 2048, 7, 7)))
    with torch.no_grad():
        ref_out = mlp(patch)
    mine = ref_image.projector(p, patch)
    assert relerr(mine, ref_out) < 1e-5
    print(f"G3 projector: restatement vs reference modules.MLP rel err {relerr(mine, ref_out):.2e}")
    x = syn.synthetic_images(2, 224, seed=27)
    coll = []
    for k in p:
        if p[k].dtype == torch.float32:
            p[k].requires_grad_(not ("running" in k))
    emb = ref_image.image_model_forward(p, x, collect=coll)
    probe = torch.from_numpy(syn._normal("g3.probe", (2, 128)))
    (emb * probe).sum().backward()
    rec = {"emb": np_(emb), "probe": np_(probe), "proj_patch_in": np_(patch), "proj_patch_out": np_(ref_out)}
    for i, c in enumerate(coll):
        rec[f"stage{i}_absmean"] = np.float32(c.abs().mean().item())
        rec[f"stage{i}_sum"] = np.float64(c.double().sum().item())
        rec[f"stage{i}_corner"] = np_(c[:, :4, :3, :3])
    # gradient fingerprints: per-tensor L2 norm for every parameter + a few full small tensors
    for k, v in p.items():
        if v.requires_grad and v.grad is not None:
            rec["gnorm::" + k] = np.float64(v.grad.double().norm().item())
    for k in ("encoder.encoder.bn1.weight", "encoder.encoder.bn1.bias", "encoder.encoder.layer4.2.bn3.weight",
              "projector.model.3.bias", "projector.model.1.weight", "encoder.encoder.layer1.0.bn2.bias"):
        rec["g::" + k] = np_(p[k].grad)
    rec["g::encoder.encoder.conv1.weight"] = np_(p["encoder.encoder.conv1.weight"].grad)
    rec["g::encoder.encoder.layer3.4.conv2.weight[:8]"] = np_(p["encoder.encoder.layer3.4.conv2.weight"].grad[:8])
    np.savez_compressed(os.path.join(OUT, "g3_image.npz"), **rec)


def gen_g4_g5():
    rec = {}
    I = torch.from_numpy(syn._normal("g4.I", (32, 128))).requires_grad_(True)
    T = torch.from_numpy(syn._normal("g4.T", (32, 128))).requires_grad_(True)
    rec["I"], rec["T"] = np_(I), np_(T)
    for tau in (1.0, 0.07):
        I.grad = T.grad = None
        loss, s = ref_loss.infonce(I, T, tau)
        loss.backward()
        tag = f"tau{tau}"
        rec["S_" + tag], rec["loss_" + tag] = np_(s), np.float32(loss.item())
        rec["dI_" + tag], rec["dT_" + tag] = np_(I.grad), np_(T.grad)
    np.savez_compressed(os.path.join(OUT, "g4_infonce.npz"), **rec)
    img = torch.from_numpy(syn._normal("g5.img", (64, 128)))
    txt = torch.from_numpy(syn._normal("g5.txt", (5, 4, 128)))  # 5 CheXpert classes x 4 prompts
    sc = ref_loss.zero_shot_scores(img, txt.mean(dim=1))
    np.savez_compressed(os.path.join(OUT, "g5_zeroshot.npz"), img=np_(img), txt=np_(txt), scores=np_(sc),
                        argmax=np_(sc.argmax(dim=1)))
    print("G4/G5 written")


def load_reference_vlp():
    """`health_multimodal/vlp/inference_engine.py` by file path.  Its two package imports are only used in type annotations
    (`ImageInferenceEngine`, `TextInferenceEngine`); the packages themselves import torchvision / transformers at module level
    and cannot load here, so empty namespaces carrying those two names stand in for them."""
    for pkg, name in (("health_multimodal", None), ("health_multimodal.image", "ImageInferenceEngine"),
                      ("health_multimodal.text", "TextInferenceEngine"), ("health_multimodal.vlp", None)):
        m = sys.modules.get(pkg)
        if m is None:
            m = types.ModuleType(pkg)
            m.__path__ = []
            sys.modules[pkg] = m
        if name and not hasattr(m, name):
            setattr(m, name, type(name, (), {}))
    return _load_by_path("health_multimodal.vlp.inference_engine", os.path.join(REF, "health_multimodal", "vlp", "inference_engine.py"))


def gen_g6():
    """G6: patch-wise similarity map (`vlp/inference_engine.py:94-155`, expected values from the reference's own static
    methods) and the MAX_EMB cosine head (`Trainer.py:1691-1693`; torchmetrics absent -> restatement only, parity unpinned)."""
    rec = {}
    vlp = load_reference_vlp().ImageTextInferenceEngine
    pat = torch.nn.functional.normalize(torch.from_numpy(syn._normal("g6.patches", (15, 15, 128))), dim=-1)
    txt = torch.nn.functional.normalize(torch.from_numpy(syn._normal("g6.text", (1, 128))), dim=-1)
    sim = vlp._get_similarity_map_from_embeddings(pat, txt)
    mine = ref_loss.similarity_map(pat, txt)
    assert torch.equal(sim, mine), relerr(mine, sim)
    rec["patches"], rec["text"], rec["sim"] = np_(pat), np_(txt), np_(sim)
    cases = [(640, 512, 512, 480, "nearest"), (300, 420, 512, 480, "bilinear"), (97, 131, None, None, "nearest"),
             (500, 500, None, 448, "bicubic")]
    for i, (w, h, rs, cs, mode) in enumerate(cases):
        exp = vlp.convert_similarity_to_image_size(sim, width=w, height=h, resize_size=rs, crop_size=cs, interpolation=mode)
        got = ref_loss.similarity_to_image_size(sim, w, h, rs, cs, mode)
        assert np.array_equal(exp, got, equal_nan=True)
        rec[f"resize{i}_args"] = np.array([w, h, -1 if rs is None else rs, -1 if cs is None else cs])
        rec[f"resize{i}_mode"] = np.array(mode)
        rec[f"resize{i}_out"] = exp
    x = torch.from_numpy(syn._normal("g6.x", (48, 128))).requires_grad_(True)
    y = torch.from_numpy(syn._normal("g6.y", (10 * 4, 128))).requires_grad_(True)   # 5 classes x (pos, neg) x 4 prompts
    y.data[5] = y.data[4]   # a tie inside one set: torch.max keeps the first
    mx, mean, idx = ref_loss.pairwise_cosine_max(x, y, 10)
    wgt = torch.from_numpy(syn._normal("g6.w", (48, 10)))
    (mx * wgt).sum().backward()
    rec.update(x=np_(x), y=np_(y), max=np_(mx), mean=np_(mean), argmax=np_(idx).astype(np.int32), dmax=np_(wgt), dx=np_(x.grad),
               dy=np_(y.grad))
    np.savez_compressed(os.path.join(OUT, "g6_simmap_maxemb.npz"), **rec)
    print("G6 written")


def hf_resnet50_with_rule_weights(p):
    """An INDEPENDENT implementation of the trunk's architecture: HuggingFace `transformers.ResNetModel` (bottleneck ResNet-50,
    stride on the 3x3 convolution = torchvision's v1.5 layout that `resnet.py:73-80` instantiates), loaded with the same name-keyed
    weights under its own parameter names.  torchvision 0.10 itself is absent here, so this is the closest
    third-party pin available for the ResNet-50 restatement in oracle/ref_image.py."""
    from transformers import ResNetConfig, ResNetModel
    cfg = ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[256, 512, 1024, 2048], depths=[3, 4, 6, 3], layer_type="bottleneck",
                       hidden_act="relu", downsample_in_first_stage=False, downsample_in_bottleneck=False)
    model = ResNetModel(cfg).eval()
    sd = {}

    def bn(dst, src):
        for suf in ("weight", "bias", "running_mean", "running_var"):
            sd[f"{dst}.{suf}"] = p[f"{src}.{suf}"].detach().clone()

    pre = "encoder.encoder."
    sd["embedder.embedder.convolution.weight"] = p[pre + "conv1.weight"].detach().clone()
    bn("embedder.embedder.normalization", pre + "bn1")
    for li, nblk in enumerate(ref_image.LAYERS, start=1):
        for b in range(nblk):
            src, dst = f"{pre}layer{li}.{b}.", f"encoder.stages.{li - 1}.layers.{b}."
            for k in (1, 2, 3):
                sd[f"{dst}layer.{k - 1}.convolution.weight"] = p[f"{src}conv{k}.weight"].detach().clone()
                bn(f"{dst}layer.{k - 1}.normalization", f"{src}bn{k}")
            if b == 0:
                sd[f"{dst}shortcut.convolution.weight"] = p[f"{src}downsample.0.weight"].detach().clone()
                bn(f"{dst}shortcut.normalization", f"{src}downsample.1")
    res = model.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys and all("num_batches_tracked" in k for k in res.missing_keys), (res.unexpected_keys, res.missing_keys[:5])
    return model


def gen_g7():
    """G7: the ResNet-50 trunk against an independent implementation (see `hf_resnet50_with_rule_weights`): stage outputs and the
    gradients of a probe through the whole trunk."""
    prm, buf = ref_image.image_param_shapes()
    p = {k: syn.rule_tensor(k, s) for k, s in {**prm, **buf}.items()}
    model = hf_resnet50_with_rule_weights(p)
    x = syn.synthetic_images(2, 224, seed=27)
    for q in model.parameters():
        q.requires_grad_(True)
    out = model(x, output_hidden_states=True)
    hs = list(out.hidden_states)                       # [pooled stem, stage1, stage2, stage3, stage4]
    assert len(hs) == 5 and hs[-1].shape == (2, 2048, 7, 7)
    probe = torch.from_numpy(syn._normal("g7.probe", (2, 2048, 7, 7)))
    (hs[-1] * probe).sum().backward()
    # the restatement, same weights, same input, same probe
    for k in p:
        if p[k].dtype == torch.float32:
            p[k].requires_grad_("running" not in k)
    coll = []
    patch = ref_image.resnet50_trunk(p, x, collect=coll)
    (patch * probe).sum().backward()
    worst = max(relerr(a, b) for a, b in zip(coll, hs))
    hf_named = dict(model.named_parameters())
    pairs = {"encoder.encoder.conv1.weight": "embedder.embedder.convolution.weight",
             "encoder.encoder.bn1.weight": "embedder.embedder.normalization.weight",
             "encoder.encoder.layer1.0.conv1.weight": "encoder.stages.0.layers.0.layer.0.convolution.weight",
             "encoder.encoder.layer2.0.downsample.0.weight": "encoder.stages.1.layers.0.shortcut.convolution.weight",
             "encoder.encoder.layer3.2.conv2.weight": "encoder.stages.2.layers.2.layer.1.convolution.weight",
             "encoder.encoder.layer3.2.bn2.bias": "encoder.stages.2.layers.2.layer.1.normalization.bias",
             "encoder.encoder.layer4.2.conv3.weight": "encoder.stages.3.layers.2.layer.2.convolution.weight",
             "encoder.encoder.layer4.2.bn3.weight": "encoder.stages.3.layers.2.layer.2.normalization.weight"}
    gworst = max(relerr(p[a].grad, hf_named[b].grad) for a, b in pairs.items())
    print(f"G7 trunk: restatement vs transformers.ResNetModel: stage outputs rel err {worst:.2e}, probed gradients {gworst:.2e}")
    assert worst < 1e-5 and gworst < 1e-4
    rec = {"probe": np_(probe)}
    for i, c in enumerate(hs):
        rec[f"stage{i}_absmean"] = np.float32(c.abs().mean().item())
        rec[f"stage{i}_sum"] = np.float64(c.double().sum().item())
        rec[f"stage{i}_corner"] = np_(c[:, :4, :3, :3])
    for a, b in pairs.items():
        g = hf_named[b].grad
        rec["g::" + a] = np_(g if g.numel() <= 40000 else g.flatten()[:40000])
        rec["gnorm::" + a] = np.float64(g.double().norm().item())
    np.savez_compressed(os.path.join(OUT, "g7_trunk_hf.npz"), **rec)
    print("G7 written")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(os.cpu_count() or 1)
    which = sys.argv[1:] or ["g1", "g2", "g3", "g45", "g6", "g7"]
    if "g1" in which:
        gen_g1()
    if "g2" in which:
        gen_g2()
    if "g3" in which:
        gen_g3()
    if "g45" in which:
        gen_g4_g5()
    if "g6" in which:
        gen_g6()
    if "g7" in which:
        gen_g7()
