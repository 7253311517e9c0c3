"""Size-independent properties at (near) BASELINE.json sizes, where the CPU oracle would take too long:
batch independence and padding invariance of the text encoder, scale invariance and zero-sum structure of the InfoNCE
gradient at global batch 1024, linearity of the convolution in its filter at a full-size ResNet layer, optimiser fixed
points, and loss = ln(B) for identical embeddings."""
import math

import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("precision")]

from incremental_multimodal_medical_learning_ii_amd import functional as Fh  # noqa: E402
from incremental_multimodal_medical_learning_ii_amd import kernels as K  # noqa: E402
from incremental_multimodal_medical_learning_ii_amd import optim as cxr_optim  # noqa: E402
from incremental_multimodal_medical_learning_ii_amd import synthetic as syn  # noqa: E402
from incremental_multimodal_medical_learning_ii_amd.health_multimodal.text import CXRBertConfig, CXRBertModel  # noqa: E402

DEV = "cuda"


def rel(a, b):
    return float((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-30))


def test_infonce_structure_at_global_batch_1024():
    B, D, tau = 1024, 128, 0.07
    g = torch.Generator().manual_seed(1)
    I = torch.randn(B, D, generator=g).to(DEV).requires_grad_(True)
    T = torch.randn(B, D, generator=g).to(DEV).requires_grad_(True)
    loss = Fh.infonce_loss(I, T, tau)
    loss.backward()
    assert 0.5 * math.log(B) < loss.item() < 3 * math.log(B)
    # the loss only sees normalised embeddings: the gradient is orthogonal to each row (scale invariance) ...
    assert float(((I * I.grad).sum(1).abs().max()) / (I.grad.norm(dim=1).max() * I.norm(dim=1).max())) < 1e-5
    # ... scaling the inputs leaves the loss unchanged and scales the gradient by 1/s
    I2 = (3.0 * I.detach()).requires_grad_(True)
    loss2 = Fh.infonce_loss(I2, T.detach(), tau)
    loss2.backward()
    assert abs(loss2.item() - loss.item()) < 1e-5 and rel(I2.grad * 3.0, I.grad) < 1e-4
    # symmetric in its two arguments
    assert abs(Fh.infonce_loss(T.detach(), I.detach(), tau).item() - loss.item()) < 1e-5
    # identical, mutually orthogonal embeddings at tau -> 0+: every row is classified perfectly; at tau=1 with identical
    # rows everywhere the logits are constant and loss = ln(B)
    ones = torch.ones(B, D, device=DEV)
    assert abs(Fh.infonce_loss(ones, ones, 1.0).item() - math.log(B)) < 1e-4
    # a ragged last batch is refused in the forward, before any collective, not inside backward()
    with pytest.raises(ValueError, match="multiples of 4"):
        Fh.infonce_loss(I.detach()[:1022], T.detach()[:1022], tau)


def test_text_encoder_batch_independence_and_padding_invariance():
    model = CXRBertModel(CXRBertConfig(num_hidden_layers=2)).eval()
    syn.fill_module_(model)
    model.to(DEV)
    ids, mask = syn.synthetic_tokens(256, 32, ragged=True)
    ids, mask = ids.to(DEV), mask.to(DEV)
    with torch.no_grad():
        e = model.get_projected_text_embeddings(ids, mask, normalize_embeddings=False)
        perm = torch.randperm(256, generator=torch.Generator().manual_seed(0)).to(DEV)
        e_perm = model.get_projected_text_embeddings(ids[perm], mask[perm], normalize_embeddings=False)
        assert torch.equal(e_perm, e[perm])                      # bit-exact: rows never mix
        junk = torch.where(mask == 0, torch.randint_like(ids, 5, 30000), ids)
        e_junk = model.get_projected_text_embeddings(junk, mask, normalize_embeddings=False)
        assert rel(e_junk, e) < 1e-6                            # padded tokens cannot influence the CLS embedding
        n = model.get_projected_text_embeddings(ids, mask, normalize_embeddings=True)
        assert rel(n.norm(dim=1), torch.ones(256, device=DEV)) < 1e-5


def test_conv_is_linear_in_its_filter_at_full_layer_size():
    # layer2 3x3 (128->128 at 28x28) at batch 256: y(w1 + 2*w2) = y(w1) + 2*y(w2) with zero shift and no ReLU
    N, H, C, Ko = 256, 28, 128, 128
    g = torch.Generator().manual_seed(2)
    x = torch.randn(N, H, H, C, generator=g).to(DEV)
    w1 = (torch.randn(Ko, 3, 3, C, generator=g) * 0.03).to(DEV)
    w2 = (torch.randn(Ko, 3, 3, C, generator=g) * 0.03).to(DEV)
    z = torch.zeros(Ko, device=DEV)
    out = [torch.empty(N, H, H, Ko, device=DEV) for _ in range(3)]
    for w, y in zip((w1, w2, w1 + 2 * w2), out):
        K.conv_fwd(x, w.contiguous(), z, None, y, N, H, H, C, Ko, 3, 3, 1, 1, False)
    assert rel(out[2], out[0] + 2 * out[1]) < 1e-4
    # data gradient is the adjoint of the forward map: <conv(x,w), dy> = <x, conv^T(dy,w)>
    dy = torch.randn(N, H, H, Ko, generator=g).to(DEV)
    dx = torch.empty_like(x)
    K.conv_bwd_data(dy, w1.contiguous(), None, None, dx, N, H, H, C, Ko, 3, 3, 1, 1)
    lhs, rhs = (out[0].double() * dy.double()).sum(), (x.double() * dx.double()).sum()
    from incremental_multimodal_medical_learning_ii_amd import _lib
    assert abs(lhs - rhs) / abs(lhs) < (1e-6 if _lib.get_precision() == "fp32" else 1e-4)


def test_optimizer_fixed_points_on_flat_buffers():
    p = [torch.nn.Parameter(torch.randn(1000, 37, device=DEV)), torch.nn.Parameter(torch.randn(5, device=DEV))]
    before = [q.detach().clone() for q in p]
    opt = cxr_optim.Adam(p, lr=1e-2)
    opt.zero_grad()
    opt.step()                                   # zero gradient: Adam must not move anything
    assert all(torch.equal(a, b.detach()) for a, b in zip(before, p))
    (p[0] * 2.0).sum().backward()                # constant gradient 2 on the first tensor only
    opt.step()
    ref = [torch.nn.Parameter(b.cpu().clone()) for b in before]   # torch.optim.Adam through the same two steps
    ropt = torch.optim.Adam(ref, lr=1e-2)
    for r_ in ref:
        r_.grad = torch.zeros_like(r_)
    ropt.step()
    ref[0].grad = torch.full_like(ref[0], 2.0)
    ropt.step()
    assert rel(p[0].detach().cpu(), ref[0].detach()) < 1e-6
    assert torch.equal(p[1].detach(), before[1])


def _cfg2_trainer():
    from incremental_multimodal_medical_learning_ii_amd.contrastive import JointContrastiveTrainer
    from incremental_multimodal_medical_learning_ii_amd.health_multimodal.image.model import get_biovil_resnet
    im = get_biovil_resnet(None).eval()
    tm = CXRBertModel(CXRBertConfig()).eval()
    syn.fill_module_(im)
    syn.fill_module_(tm)
    return JointContrastiveTrainer(im.to(DEV), tm.to(DEV), lr=1e-4, temperature=0.07)


def test_cfg2_joint_step_batch_256_full_models():
    """BASELINE config 2 (`Trainer.py` joint training shape at batch 256) on the full ResNet-50 (224 px) + 12-layer CXR-BERT, where
    the CPU oracle would take minutes.  The split-bf16 product path against the exact-fp32 path of the same kernels, for a FIXED
    cotangent on the embeddings:
      * embeddings, loss, every probed BERT gradient and the image projector's output layer: within the north star's 1e-3;
      * gradients BEHIND ReLUs of the image encoder (projector.0, layer3, layer1) are NOT compared free-running: the two
        precisions' forward passes differ in the last bits, a few 1e-5 of the 2.8e9 ReLU decisions of this batch fall on the other
        side of zero and the gradient is discontinuous there.  They are compared at this size under the fp32 forward's decisions,
        every tensor within 1e-3, in tests/test_precision_gpu.py; here the backward kernels are additionally checked through what
        must hold exactly: linearity in the cotangent under one set of decisions (1e-4);
      * run-to-run bit equality, and a loss that falls.
    (The InfoNCE gradient itself is not compared across precisions: with synthetic weights the embeddings of a batch are nearly
    parallel, (softmax - onehot) @ T cancels to ~1e-3 of its terms, and a 1e-5 change of the embeddings moves it by 2e-3.)"""
    from incremental_multimodal_medical_learning_ii_amd import _lib
    B = 256
    images = syn.synthetic_images(B, 224, seed=31).to(DEV)
    ids, mask = syn.synthetic_tokens(B, 32, seed=32)
    ids, mask = ids.to(DEV), mask.to(DEV)
    g = torch.Generator().manual_seed(5)
    cot_i, cot_t = torch.randn(B, 128, generator=g).to(DEV), torch.randn(B, 128, generator=g).to(DEV)
    tr = _cfg2_trainer()
    inamed, tnamed = dict(tr.image_model.named_parameters()), dict(tr.text_model.named_parameters())
    names = ("encoder.encoder.layer3.2.conv2.weight", "encoder.encoder.layer1.0.conv1.weight", "projector.model.0.weight",
             "projector.model.3.weight")
    probes = [inamed[n] for n in names] + [tnamed["bert.encoder.layer.5.attention.output.dense.weight"],
                                           tnamed["bert.embeddings.word_embeddings.weight"]]
    old = _lib.get_precision()
    out = {}
    try:
        for mode in ("fp32", "split_bf16", "split_bf16"):
            _lib.set_precision(mode)
            tr.optimizer.zero_grad()
            ie = tr.image_model(images)
            te = tr.text_model.get_projected_text_embeddings(ids, mask, normalize_embeddings=False)
            ((ie * cot_i).sum() + (te * cot_t).sum()).backward()
            grads = [p.grad.detach().clone() for p in probes]
            tr.optimizer.zero_grad()
            loss = float(tr.forward_loss(images, ids, mask))
            rec = (loss, ie.detach().clone(), te.detach().clone(), grads)
            if mode in out:   # second split-bf16 run: same bits (no floating-point atomics, fixed reduction order everywhere)
                assert rec[0] == out[mode][0] and torch.equal(rec[1], out[mode][1]) and torch.equal(rec[2], out[mode][2])
                assert all(torch.equal(x, y) for x, y in zip(rec[3], out[mode][3]))
            out[mode] = rec
        a, b = out["fp32"], out["split_bf16"]
        errs = {"loss": abs(b[0] - a[0]) / abs(a[0]), "image_embedding": rel(b[1], a[1]), "text_embedding": rel(b[2], a[2])}
        for name, ga, gb in zip(names + ("bert.layer5.attention.output", "bert.word_embeddings"), a[3], b[3]):   # same order as probes
            errs["grad " + name] = float((gb - ga).norm() / ga.norm())
        print("cfg2 split_bf16 vs fp32:", errs)
        behind_relu = ("grad encoder.encoder.layer3.2.conv2.weight", "grad encoder.encoder.layer1.0.conv1.weight", "grad projector.model.0.weight")
        assert math.isfinite(b[0]) and all(v < 1e-3 for k, v in errs.items() if k not in behind_relu), errs
        # linearity of the image backward in the cotangent (same forward, same decisions): g(c1) + g(c2) == g(c1 + c2)
        _lib.set_precision("split_bf16")
        gsum = None
        cot2 = torch.randn(B, 128, generator=g).to(DEV)
        for c in (cot_i, cot2, cot_i + cot2):
            tr.optimizer.zero_grad()
            (tr.image_model(images) * c).sum().backward()
            gs = [p.grad.detach().clone() for p in probes[:3]]
            if c is cot_i:
                gsum = gs
            elif c is cot2:
                gsum = [x + y for x, y in zip(gsum, gs)]
            else:
                lin = [float((x - y).norm() / y.norm()) for x, y in zip(gsum, gs)]
                assert max(lin) < 1e-4, lin
        _lib.set_precision("split_bf16")
        l0 = float(tr.step(images, ids, mask))
        for _ in range(3):
            l1 = float(tr.step(images, ids, mask))
        assert math.isfinite(l1) and l1 < l0 and abs(l0 - b[0]) < 1e-6
    finally:
        _lib.set_precision(old)


def test_joint_step_overfits_a_small_batch():
    """The step TRAINS: 40 steps of the full step (ResNet-50 at 64 px with calibrated BatchNorm statistics + a 2-layer CXR-BERT, InfoNCE,
    hand-written backward, fused Adam at lr 1e-4) on ONE batch of 32 pairs take the symmetric InfoNCE loss from above ln 32 = 3.47 to
    below 0.1 — the encoders memorise the pairing (scripts/overfit_check.py: 3.82 -> 0.035 after 15 steps, 0.001 after 300).  A wrong
    sign, a missing gradient or a mis-scaled update anywhere in the 133-M-parameter path does not get there."""
    from incremental_multimodal_medical_learning_ii_amd.contrastive import JointContrastiveTrainer
    from incremental_multimodal_medical_learning_ii_amd.health_multimodal.image.model import get_biovil_resnet
    B = 32
    im = get_biovil_resnet(None).eval()
    tm = CXRBertModel(CXRBertConfig(vocab_size=2048, hidden_size=128, num_attention_heads=2, intermediate_size=256, num_hidden_layers=2,
                                    max_position_embeddings=32)).eval()
    syn.fill_module_(im)
    syn.fill_module_(tm)
    im.to(DEV)
    from bench import structured_images
    images = structured_images(B, 64, seed=7).to(DEV)            # a different low-frequency pattern per image (bench.py's inputs)
    im.calibrate_batchnorm_(images)
    ids, mask = syn.synthetic_tokens(B, 16, vocab=2048, seed=8)
    tr = JointContrastiveTrainer(im, tm.to(DEV), lr=1e-4, temperature=0.07)
    trace = [float(tr.step(images, ids.to(DEV), mask.to(DEV))) for _ in range(40)]
    assert trace[0] > math.log(B) and all(math.isfinite(x) for x in trace), trace[:3]
    assert trace[-1] < 0.1 and min(trace[-5:]) < 0.05 * trace[0], (trace[0], trace[-5:])
    # the same with the image encoder's BatchNorm in TRAIN mode (batch statistics, the reference constructor's default state): no
    # calibration needed, the normalisation follows the batch
    im2 = get_biovil_resnet(None)
    tm2 = CXRBertModel(CXRBertConfig(vocab_size=2048, hidden_size=128, num_attention_heads=2, intermediate_size=256, num_hidden_layers=2,
                                     max_position_embeddings=32)).eval()
    syn.fill_module_(im2)
    syn.fill_module_(tm2)
    tr2 = JointContrastiveTrainer(im2.to(DEV).train(), tm2.to(DEV), lr=1e-4, temperature=0.07)
    before = im2.encoder.encoder.layer3[0].bn1.running_mean.clone()
    trace2 = [float(tr2.step(images, ids.to(DEV), mask.to(DEV))) for _ in range(40)]
    assert all(math.isfinite(x) for x in trace2) and trace2[-1] < 0.1 and min(trace2[-5:]) < 0.05 * trace2[0], (trace2[0], trace2[-5:])
    assert int(im2.projector.model[1].num_batches_tracked) == 40 and not torch.equal(im2.encoder.encoder.layer3[0].bn1.running_mean, before)


def test_embedding_precompute_at_reference_image_size():
    """`chexpert-get-embedding.py:48-74` operating point: frozen encoder, 512 x 512 images, a large batch.  Checks the batch
    pipeline against the same images encoded one at a time (the reference's batch size 1) and the chunk files it writes."""
    import tempfile
    from incremental_multimodal_medical_learning_ii_amd.embedding_precompute import compute_embeddings, synthetic_image_batches
    from incremental_multimodal_medical_learning_ii_amd.health_multimodal.image.model import get_biovil_resnet
    im = get_biovil_resnet(None).eval()
    syn.fill_module_(im)
    im = im.to(DEV)
    with tempfile.TemporaryDirectory() as d:
        embs, labels = compute_embeddings(im, synthetic_image_batches(80, 64, size=512), out_dir=d, checkpoint_interval=64)
        assert embs.shape == (80, 128) and labels.shape == (80, 5) and bool(torch.isfinite(embs).all())
        first = torch.load(d + "/embeddings_dataset_64.pt", weights_only=True)
        final = torch.load(d + "/embeddings_dataset_final_old.pt", weights_only=True)
        assert torch.equal(first["embs"], embs[:64]) and torch.equal(final["embs"], embs) and torch.equal(final["labels"], labels)
    one, _ = next(iter(synthetic_image_batches(80, 64, size=512)))
    with torch.no_grad():
        single = torch.cat([im(one[i:i + 1].to(DEV)) for i in (0, 17, 63)]).cpu()
    assert rel(embs[[0, 17, 63]], single) < 1e-4   # batch independence of the eval-mode encoder at 512 px
