"""`Trainer` drop-in (reference Trainer.py) on the MI355X against the CPU oracle: joint / class-incremental /
data-incremental loops, val/test scoring, weight-reset continual learning, save/load."""
import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("precision")]

from incremental_multimodal_medical_learning_ii_amd import Trainer as TR  # noqa: E402
from incremental_multimodal_medical_learning_ii_amd import synthetic as syn  # noqa: E402
from incremental_multimodal_medical_learning_ii_amd.DataRetrieval import CHEXPERT_COMPETITION_CLASSES, create_prompts  # noqa: E402
from incremental_multimodal_medical_learning_ii_amd.health_multimodal.text import (CXRBertConfig, CXRBertModel, SyntheticTokenizer,  # noqa: E402
                                                                                  TextInferenceEngine)
from oracle import ref_step  # noqa: E402

DEV = "cuda"


def _engine():
    cfg = CXRBertConfig(vocab_size=2048, hidden_size=128, num_attention_heads=2, intermediate_size=256,
                        num_hidden_layers=2, max_position_embeddings=32)
    model = CXRBertModel(cfg)
    syn.fill_module_(model)
    return TextInferenceEngine(SyntheticTokenizer(2048), model.eval().to(DEV))


def _trainer(tmp_path, lr=1e-3):
    classes = list(CHEXPERT_COMPETITION_CLASSES)
    prompts = create_prompts(classes)
    writer = TR.ScalarWriter(str(tmp_path / "run"))
    tr = TR.Trainer(False, prompts, classes, "standard", lr, DEV, writer, bert_encoder=_engine())
    syn.fill_module_(tr.image_adapter, "image_adapter.")
    syn.fill_module_(tr.text_adapter, "text_adapter.")
    return tr, classes, prompts


def _oracle_state(tr, classes, prompts):
    ip = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in tr.image_adapter.state_dict().items()}
    tp = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in tr.text_adapter.state_dict().items()}
    bert = []
    for c in classes:
        for kind in ("positive", "negative"):
            bert.append(tr.bert_encoder.get_embeddings_from_prompt(prompts[c][kind], normalize=False, verbose=False).cpu())
    return ip, tp, torch.stack(bert)  # [10,4,128]


def test_joint_train_val_test_match_oracle(tmp_path):
    tr, classes, prompts = _trainer(tmp_path)
    ip, tp, bert_out = _oracle_state(tr, classes, prompts)
    opt = torch.optim.Adam(list(tp.values()) + list(ip.values()), lr=1e-3)
    train, val, test = TR.Trainer.synthetic_loaders(3 * 96, 200, 100, batch_size=96, shuffle=False)
    crit = nn.BCEWithLogitsLoss()
    tr.train(train, crit, epoch=1)
    ref_losses = [float(ref_step.adapter_step(ip, tp, e, l, bert_out, opt)[0]) for e, l in train]
    got = [v for _, v, _ in tr.writer.scalars("train/Loss")]
    assert len(got) == 3 and max(abs(a - b) for a, b in zip(got, ref_losses)) < 1e-5, (got, ref_losses)
    assert 0.6 < got[0] < 0.8            # ~ln 2, like the reference's own first logged train/Loss
    for k, v in tr.image_adapter.state_dict().items():
        assert float((v.cpu() - ip[k].detach()).abs().max() / ip[k].detach().abs().max()) < 1e-4, k
    for k, v in tr.text_adapter.state_dict().items():
        assert float((v.cpu() - tp[k].detach()).abs().max() / tp[k].detach().abs().max()) < 1e-4, k
    # scoring: (pos+1)/2 and argmax([neg,pos]) (Trainer.py:825-836)
    e, l = next(iter(val))
    y_true, y_pred, y_score = tr._eval_loop([(e, l)], crit, 1, "val")
    sc, pr, _ = ref_step.eval_scores({k: v.detach() for k, v in ip.items()}, {k: v.detach() for k, v in tp.items()}, e, bert_out)
    assert np.abs(y_score - sc.numpy()).max() < 1e-5 and (y_pred == pr.numpy()).mean() > 0.999
    m = tr.val(val, crit, epoch=1, epochs=1)
    assert {"Accuracy", "F1-macro score", "F1-weighted score"} <= set(m)
    assert "Accuracy" in tr.test(test, crit, epoch=1, epochs=1)
    # reduction="sum" runs on the fused loss + gradient kernel too: loss and the update against torch
    tr2, _, _ = _trainer(tmp_path / "b", lr=1e-5)
    ip2, tp2, _ = _oracle_state(tr2, classes, prompts)
    opt2 = torch.optim.Adam(list(tp2.values()) + list(ip2.values()), lr=1e-5)
    tr2.train([(e[:32], l[:32])], nn.BCEWithLogitsLoss(reduction="sum"), epoch=1)
    lg = ref_step.adapter_logits(ip2, tp2, e[:32], bert_out)
    ref = nn.functional.binary_cross_entropy_with_logits(lg, l[:32], reduction="sum")
    assert abs(tr2.writer.scalars("train/Loss")[0][1] - float(ref)) / float(ref) < 1e-5
    ref.backward()
    opt2.step()
    for k, v in tr2.image_adapter.state_dict().items():
        assert float((v.cpu() - ip2[k].detach()).abs().max() / ip2[k].detach().abs().max()) < 1e-4, k
    # a criterion the fused kernel does not serve (pos_weight) is applied to the logits tensor as the caller's module defines it
    tr3, _, _ = _trainer(tmp_path / "c")
    ip3, tp3, _ = _oracle_state(tr3, classes, prompts)
    pw = torch.tensor([1.0, 2.0, 0.5, 3.0, 1.5])
    tr3.train([(e[:32], l[:32])], nn.BCEWithLogitsLoss(pos_weight=pw.to(DEV)), epoch=1)
    ref3 = nn.functional.binary_cross_entropy_with_logits(ref_step.adapter_logits(ip3, tp3, e[:32], bert_out), l[:32], pos_weight=pw)
    assert abs(tr3.writer.scalars("train/Loss")[0][1] - float(ref3)) / float(ref3) < 1e-5


def test_max_emb_training_step_matches_oracle(tmp_path, monkeypatch):
    """`MAX_EMB = True` (`Trainer.py:43,1691-1703`): scores are the maximum cosine over a class's prompts.  One training step of the
    fused path (`cxrk_pairwise_cosine_max_fwd/bwd` + BCE) and of the reference's per-class `myCosineSimilarity` path against the
    oracle: loss, both adapters after the Adam step, and the logged max-vs-mean gaps."""
    monkeypatch.setattr(TR, "MAX_EMB", True)
    tr, classes, prompts = _trainer(tmp_path)
    ip, tp, bert_out = _oracle_state(tr, classes, prompts)
    opt = torch.optim.Adam(list(tp.values()) + list(ip.values()), lr=1e-3)
    train, _, _ = TR.Trainer.synthetic_loaders(96, 8, 8, batch_size=96, shuffle=False)
    e, l = next(iter(train))
    crit = nn.BCEWithLogitsLoss()
    tr.train([(e, l)], crit, epoch=1)
    opt.zero_grad()
    logits = ref_step.adapter_logits_max_emb(ip, tp, e, bert_out)
    ref = nn.functional.binary_cross_entropy_with_logits(logits, l)
    ref.backward()
    opt.step()
    got = tr.writer.scalars("train/Loss")[0][1]
    assert abs(got - float(ref)) < 1e-5, (got, float(ref))
    for k, v in tr.image_adapter.state_dict().items():
        assert float((v.cpu() - ip[k].detach()).abs().max() / ip[k].detach().abs().max()) < 1e-4, k
    for k, v in tr.text_adapter.state_dict().items():
        assert float((v.cpu() - tp[k].detach()).abs().max() / tp[k].detach().abs().max()) < 1e-4, k
    gaps = [v for _, v, _ in tr.writer.scalars("max-mean-comparison/pos")]
    assert len(gaps) == len(classes) and all(g >= 0 for g in gaps)      # max >= mean, one log entry per class as in the reference
    # the reference's own per-call form (myCosineSimilarity on one class's un-averaged prompt set) gives the same numbers
    with torch.no_grad():
        x = tr.image_adapter(e.to(DEV))
        pe, ne = tr.bert_forward_mean(prompts[classes[0]]["positive"], prompts[classes[0]]["negative"], use_grad=False)
        assert pe.shape[0] == len(prompts[classes[0]]["positive"])      # not averaged under MAX_EMB
        ps = tr.myCosineSimilarity(x, pe, use_grad=False)
        full = ref_loss_cos(x.cpu(), pe.cpu()).max(dim=1).values
        assert float((ps.cpu() - full).abs().max()) < 1e-5


def test_max_emb_positive_only_with_ten_prompts_per_class(tmp_path, monkeypatch):
    """The reference's NEW_PROMPTS operating point with `TRAIN_LOGIT_DIFF = False` (`Trainer.py:42-44`, `new_texts_prompts.py`):
    10 prompts per group over 5 classes, positive-only logits — 2 x 5 x 10 = 100 prompt rows go through ONE max-over-prompts cosine
    call.  Its backward keeps a [prompts][128] accumulator per wave in LDS and walks the prompts in chunks of 32: round 2 refused more
    than 64 rows (`CXRK_ERR_UNSUPPORTED` from inside `loss.backward()`).  One training step against the oracle."""
    monkeypatch.setattr(TR, "MAX_EMB", True)
    monkeypatch.setattr(TR, "TRAIN_LOGIT_DIFF", False)
    classes = list(CHEXPERT_COMPETITION_CLASSES)
    prompts = {c: {"positive": [f"finding number {i} suggesting {c}" for i in range(10)],
                   "negative": [f"statement {i} excludes {c}" for i in range(10)]} for c in classes}
    tr = TR.Trainer(False, prompts, classes, "standard", 1e-3, DEV, TR.ScalarWriter(str(tmp_path / "run")), bert_encoder=_engine())
    syn.fill_module_(tr.image_adapter, "image_adapter.")
    syn.fill_module_(tr.text_adapter, "text_adapter.")
    ip = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in tr.image_adapter.state_dict().items()}
    tp = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in tr.text_adapter.state_dict().items()}
    pos = [tr.bert_encoder.get_embeddings_from_prompt(prompts[c]["positive"], normalize=False, verbose=False).cpu() for c in classes]
    bert_out = torch.stack([p for p in pos for _ in (0, 1)])          # positive-only: both rows of a class hold its positive prompts
    assert bert_out.shape == (10, 10, 128)
    opt = torch.optim.Adam(list(tp.values()) + list(ip.values()), lr=1e-3)
    train, _, _ = TR.Trainer.synthetic_loaders(200, 8, 8, batch_size=200, shuffle=False)
    e, l = next(iter(train))
    tr.train([(e, l)], nn.BCEWithLogitsLoss(), epoch=1)
    opt.zero_grad()
    ref = nn.functional.binary_cross_entropy_with_logits(ref_step.adapter_logits_max_emb(ip, tp, e, bert_out, diff=False), l)
    ref.backward()
    opt.step()
    got = tr.writer.scalars("train/Loss")[0][1]
    assert abs(got - float(ref)) < 1e-5, (got, float(ref))
    for mod, ref_p in ((tr.image_adapter, ip), (tr.text_adapter, tp)):
        for k, v in mod.state_dict().items():
            assert float((v.cpu() - ref_p[k].detach()).abs().max() / ref_p[k].detach().abs().max()) < 1e-4, k


def ref_loss_cos(x, y):
    from oracle import ref_loss
    return ref_loss.pairwise_cosine_similarity(x, y)


def test_class_incremental_and_weight_reset(tmp_path):
    tr, classes, prompts = _trainer(tmp_path)
    ip, tp, bert_out = _oracle_state(tr, classes, prompts)
    opt = torch.optim.Adam(list(tp.values()) + list(ip.values()), lr=1e-3)
    embs, labels, _ = syn.synthetic_adapter_batch(64, seed=5)
    crit = nn.BCEWithLogitsLoss()
    # task 2 with MORE_LABELS: logits/labels [:, :3] (Trainer.py:701-714)
    it = tr.train_class_more_labels_incremental([(embs, labels)], crit, epoch=1, current_task=2, last_batch=7)
    assert it == 8
    ref, _ = ref_step.adapter_step(ip, tp, embs, labels, bert_out, opt, n_cols=3)
    assert abs(tr.writer.scalars("train/Loss")[-1][1] - float(ref)) < 1e-5
    # single-column variant (Trainer.py:626-659): column 3 only
    tr.train_class_incremental([(embs, labels)], crit, epoch=1, current_task=3)
    opt.zero_grad()
    x = ref_step.mlp_adapter(ip, embs)
    pv = ref_step.prompt_vectors(tp, bert_out)
    from oracle import ref_loss
    lg = ref_loss.posneg_logits(x, pv[6:7], pv[7:8]).flatten()
    l1 = nn.functional.binary_cross_entropy_with_logits(lg, labels[:, 3])
    assert abs(tr.writer.scalars("train/Loss")[-1][1] - float(l1)) < 1e-5
    # myCL weight reset (Trainer.py:1556-1587): snapshot, one step, restore small updates
    tr.model_copy()
    before = [p.detach().cpu().clone() for p in tr.image_adapter.parameters()]
    tr._train_step((embs, labels), classes, crit)
    after = [p.detach().cpu().clone() for p in tr.image_adapter.parameters()]
    tr.myIncremental(0.4, 1)
    n_ref = 0
    for p, new, old in zip(tr.image_adapter.parameters(), after, before):
        exp, n = ref_step.weight_reset(new, old, 0.4)
        assert torch.equal(p.detach().cpu(), exp)
        n_ref += n
    n_reset, n_upd = tr._reset_stats()
    tot = sum(p.numel() for p in tr.image_adapter.parameters()) + sum(p.numel() for p in tr.text_adapter.parameters())
    assert n_reset >= n_ref and n_reset + n_upd == tot
    tr.save()
    w = tr.image_adapter.layer[0].weight.detach().clone()
    with torch.no_grad():
        tr.image_adapter.layer[0].weight.add_(1.0)
    tr.load()
    assert torch.equal(tr.image_adapter.layer[0].weight.detach(), w)


def test_data_incremental_schedule_runs(tmp_path):
    """DATA_INCREMENTAL.py:75-90 shape: 5 contiguous shards, epochs per shard, val + test after each."""
    torch.manual_seed(27)   # the shard loaders draw their batches with RandomSampler, like the reference's (Trainer.py:1214-1231)
    tr, classes, prompts = _trainer(tmp_path)
    train, val, test = TR.Trainer.synthetic_loaders(320, 64, 64, batch_size=32, shuffle=False)
    parts = TR.Trainer.split_dataloader_data_incremental(train, 5)
    crit = nn.BCEWithLogitsLoss()
    for part, loader in enumerate(parts, start=1):
        tr.train(loader, crit, epoch=1, part=part, epochs=1)
    losses = [v for _, v, _ in tr.writer.scalars("train/Loss")]
    # every batch is new data (5 disjoint shards x 2 batches), so single losses fluctuate: compare halves, not end points
    assert len(losses) == 10 and sum(losses[5:]) < sum(losses[:5])
    steps = [s for _, _, s in tr.writer.scalars("train/Loss")]
    assert steps == list(range(1, 11))
