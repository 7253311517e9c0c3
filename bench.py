#!/usr/bin/env python3
"""Headline benchmark: joint image+text contrastive TRAIN step, images/sec (BASELINE.json `metric`).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = ResNet-50 image encoder (224x224) + 12-layer CXR-BERT (32 tokens) forward, L2-normalise, all-gather,
InfoNCE over the global batch, hand-written backward through both encoders, gradient all-reduce, fused Adam on all
~133 M parameters.  Per-GPU batch is 1024 (global batch 1024 at N=1 = the configuration the metric is quoted on;
weak scaling for N>1: 8192 global at N=8 = BASELINE config 5).  fp32 storage and arithmetic everywhere; the large
contractions run in split-bf16 (three bf16 MFMAs per product, fp32 accumulate, ~2^-16 relative; `--precision fp32` =
exact fp32 MFMA, measured beside it as `other_precision`).  Synthetic data resident in HBM before the timed region,
seeded random-init weights.
The two encoders run on two HIP streams (contrastive.JointContrastiveTrainer); the per-kernel profile behind `roofline` is
taken over two extra single-stream steps after the timed region (see the comment there), `precision_check` compares the
split-bf16 forward and backward with the exact-fp32 ones on the bench batch (free-running, and with the fp32 forward's ReLU /
max-pool decisions imposed on the split-bf16 backward).
The timed steps train a model with LIVE gradients: BatchNorm statistics are calibrated on a sample batch (synthetic weights then
map different images to different embeddings, like a trained checkpoint), the learning rate keeps the 133 M-parameter model in the
first, descending phase of contrastive training for the whole run, and the line carries the loss trace, the norm of the loss
gradient w.r.t. the embeddings before and after the timed region (`cotangent_norm`) and per-step times (`step_ms`); the run
refuses to print a number measured on collapsed embeddings (cotangents fallen to < 2 % of their initial norm, or all embeddings
parallel; round 2's collapsed run sat at 4e-4 of it).
Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same guide, "Peak BF16/FP16 MFMA ~2.5 PF dense"
PEAK_HBM_GBS = 8000.0           # same guide, "HBM3E 8 TB/s peak" (about 6.3 TB/s achievable)
LR_DEFAULT = 1e-6               # see scripts/exp_bench_regime.py / DESIGN.md section 6: keeps the loss off ln(B) with live gradients
FLOP_PER_PAIR_STEP = 41.1e9     # 3 x (8.2 GFLOP ResNet-50+projector + 5.5 GFLOP CXR-BERT, L=32, no MLM head); SURVEY.md §8d


def log(msg: str) -> None:
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def usable_cores() -> int:
    """Host threads this process may really use: min(affinity mask, cgroup CPU quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except Exception:
            continue
    return max(1, min(n, int(os.environ.get("CXRK_CPU_THREADS", "64"))))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch-per-gpu", type=int, default=1024)
    ap.add_argument("--seq-len", type=int, default=32)
    ap.add_argument("--image-size", type=int, default=224)
    ap.add_argument("--temperature", type=float, default=0.07)
    ap.add_argument("--lr", type=float, default=LR_DEFAULT, help="Adam learning rate of the timed steps")
    ap.add_argument("--no-bn-calibration", action="store_true", help="keep the name-keyed BatchNorm statistics of synthetic.fill_module_")
    ap.add_argument("--batchnorm", default="eval", choices=["eval", "train"],
                    help="image-encoder BatchNorm mode of the timed steps: eval = running statistics folded into the filters (the only mode "
                         "the reference runs the encoder in; the headline), train = batch statistics (a labelled variant)")
    ap.add_argument("--cpu-baseline-batch", type=int, default=64)
    ap.add_argument("--cpu-baseline-steps", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--precision", default=os.environ.get("CXRK_PRECISION", "split_bf16"), choices=["split_bf16", "fp32"],
                    help="contraction precision of the headline number (the other one is measured too, on fewer steps)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the measurement in the other precision")
    return ap.parse_args()


def cpu_baseline(batch: int, seq_len: int, image_size: int, tau: float, steps: int = 3):
    """The CPU oracle (PyTorch fp32, all host cores) timed on a bounded sample of the same workload: full-size models,
    a global batch of `batch` pairs, 1 warm-up + `steps` timed steps (>= 3 steps at batch 64 / 256)."""
    from incremental_multimodal_medical_learning_ii_amd import synthetic as syn
    from oracle import ref_image, ref_step, ref_text
    ncores = usable_cores()
    torch.set_num_threads(ncores)
    log(f"cpu baseline: oracle joint step on {ncores} host threads, batch {batch}, 1 + {steps} steps")
    prm, buf = ref_image.image_param_shapes()
    g = torch.Generator().manual_seed(27)
    ip = {k: (torch.randn(s, generator=g) * (0.05 if len(s) > 1 else 0.0) + (1.0 if len(s) == 1 else 0.0)) for k, s in prm.items()}
    for k, s in buf.items():
        ip[k] = torch.zeros(s, dtype=torch.int64) if k.endswith("num_batches_tracked") else (torch.ones(s) if k.endswith("var") else torch.zeros(s))
    tp = {k: torch.randn(s, generator=g) * 0.02 + (1.0 if k.endswith("LayerNorm.weight") else 0.0)
          for k, s in ref_text.cxrbert_param_shapes().items()}
    leaves = [v.requires_grad_(True) for k, v in list(ip.items()) + list(tp.items()) if v.is_floating_point() and "running" not in k]
    opt = torch.optim.Adam(leaves, lr=1e-4)
    images = syn.synthetic_images(batch, image_size)
    ids, mask = syn.synthetic_tokens(batch, seq_len)
    ref_step.joint_step(ip, tp, images, ids, mask, tau, opt)
    log("cpu baseline: warm-up step done")
    times = []
    for _ in range(steps):
        t0 = time.perf_counter()
        ref_step.joint_step(ip, tp, images, ids, mask, tau, opt)
        times.append(time.perf_counter() - t0)
        log(f"cpu baseline: step {len(times)}/{steps}: {times[-1]:.2f} s")
    dt = sum(times) / len(times)
    return {"value": batch / dt, "unit": "images/sec", "cores": ncores, "kind": "port", "threads": ncores,
            "s_per_step": dt, "s_per_step_each": [round(t, 3) for t in times], "batch": batch, "timed_steps": steps,
            "sample": f"oracle joint step (ResNet-50 {image_size}px + 12-layer CXR-BERT L={seq_len} + InfoNCE + Adam), global batch "
                      f"{batch}, 1 warm-up + {steps} timed steps, {dt:.2f} s/step on {ncores} threads; per-pair cost is "
                      f"batch-independent apart from the negligible B^2 logits"}


def _free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_cmd(n_ranks: int, argv, port: int):
    """Command line that runs this script as `n_ranks` ranks of one node (one process per GPU over RCCL)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__), *argv]


def spawn_ranks(n_ranks: int, argv) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes (this parent never touches the
    GPU, so nothing that initialised HIP is ever replaced or re-executed), pass rank 0's JSON line through on stdout and
    return the launcher's exit status (non-zero when any rank failed)."""
    import subprocess
    cmd = launch_cmd(n_ranks, argv, _free_port())
    print(f"[bench] --gpus {n_ranks} without WORLD_SIZE: launching {n_ranks} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // n_ranks)))
    proc = subprocess.run(cmd, env=env)
    return proc.returncode


def selftest_rank() -> None:
    """CXRK_BENCH_SELFTEST=1: exercise the launcher, rendezvous and collectives only (no GPU work), so the spawn path can be
    tested on a CPU box with gloo.  Rank 0 prints a JSON line like the real run does."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(os.environ.get("CXRK_DIST_BACKEND", "gloo"))
    t = torch.ones(1) * (dist.get_rank() + 1)
    dist.all_reduce(t)
    w = dist.get_world_size()
    assert float(t) == w * (w + 1) / 2
    dist.barrier()
    if dist.get_rank() == 0:
        print(json.dumps({"selftest": True, "rccl_ranks": w, "n_gpus": w}), flush=True)
    dist.destroy_process_group()


def structured_images(batch: int, size: int, seed: int) -> torch.Tensor:
    """Synthetic 1-channel images replicated to 3 channels (reference: DataRetrieval.py:175-180) with per-image low-frequency
    structure (a few random plane waves + noise, scaled into [0,1)): iid-noise images (synthetic.synthetic_images, used by the
    parity fixtures) have the same statistics everywhere and an average-pooling encoder maps them all to nearly the same
    embedding.  Structure in the input is necessary but not sufficient for distinct embeddings — the BatchNorm statistics have to
    match the activations too (`ImageModel.calibrate_batchnorm_`, called by main())."""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, size), torch.linspace(0, 1, size), indexing="ij")
    img = torch.zeros(batch, size, size)
    for _ in range(4):
        f = torch.rand(batch, 2, generator=g) * 6.0
        ph = torch.rand(batch, 1, 1, generator=g) * 6.2832
        amp = torch.rand(batch, 1, 1, generator=g)
        img += amp * torch.cos(6.2832 * (f[:, 0, None, None] * yy + f[:, 1, None, None] * xx) + ph)
    img += 0.25 * torch.randn(batch, size, size, generator=g)
    img -= img.amin(dim=(1, 2), keepdim=True)
    img /= img.amax(dim=(1, 2), keepdim=True).clamp_min(1e-6) * 1.0001
    return img.unsqueeze(1).repeat_interleave(3, dim=1).contiguous()


def cotangent_norms(trainer, images, ids, mask):
    """||dL/dI||, ||dL/dT|| of the InfoNCE loss w.r.t. this rank's (un-normalised) embeddings at the current weights, the loss and
    the mean off-diagonal cosine of the image / text embeddings: what the two encoders' backward passes are fed.  N = 1 only (None
    under a process group: the loss there is a collective over the global batch)."""
    from incremental_multimodal_medical_learning_ii_amd import functional as Fh
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return None
    with torch.no_grad():
        ie = trainer.image_model(images)
        te = trainer.text_model.get_projected_text_embeddings(ids, mask, normalize_embeddings=False)
    ie, te = ie.clone().requires_grad_(True), te.clone().requires_grad_(True)
    loss = Fh.infonce_loss(ie, te, trainer.temperature)
    gi, gt = torch.autograd.grad(loss, (ie, te))

    def offdiag(e):   # mean of <n_i, n_j> over i != j = (|sum_i n_i|^2 - B) / (B^2 - B): no B x B matrix, no BLAS call
        n = torch.nn.functional.normalize(e.detach().double(), dim=1)
        b = n.shape[0]
        return float((n.sum(0).pow(2).sum() - b) / (b * b - b))
    return {"loss": float(loss), "dL_dI": float(gi.norm()), "dL_dT": float(gt.norm()), "mean_offdiag_cos_image": offdiag(ie),
            "mean_offdiag_cos_text": offdiag(te)}


def secondary_metrics(args, dev, trainer, images, ids, mask, step_ms):
    """SURVEY.md §8(d) / BASELINE.md §3 side measurements (N = 1 only; none of them is `value`)."""
    from incremental_multimodal_medical_learning_ii_amd import functional as Fh
    from incremental_multimodal_medical_learning_ii_amd import optim as cxr_optim
    from incremental_multimodal_medical_learning_ii_amd import synthetic as syn
    from incremental_multimodal_medical_learning_ii_amd.models import myMLP
    out = {}

    def timed(fn, n):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    # (1) the reference's own step (T-ref): two adapters on pre-computed embeddings, 10 prompt vectors, pos-neg BCE, Adam
    for B in (1024, 6144):
        embs, labels, bert_out = (t.to(dev) for t in syn.synthetic_adapter_batch(B, seed=29))
        ia, ta = myMLP().to(dev), myMLP().to(dev)
        opt = cxr_optim.Adam(list(ta.parameters()) + list(ia.parameters()), lr=1e-4)

        def ref_step():
            opt.zero_grad()
            pv = Fh.group_mean(ta(bert_out.reshape(40, 128)), 10, 4)
            loss, _ = Fh.posneg_bce_loss(Fh.pairwise_cosine_similarity(ia(embs), pv), labels)
            loss.backward()
            opt.step()
        dt = timed(ref_step, 30)
        out[f"adapter_step_b{B}"] = {"ms_per_step": dt * 1e3, "embeddings_per_sec": B / dt,
                                     "cpu_reference_ms": {1024: 15.7, 6144: 41.7}[B], "note": "reference-faithful step (Trainer.py:537-601); CPU figure from BASELINE.md §2 (8 threads, survey container)"}
    # (2) embedding pre-compute at the reference's operating point (chexpert-get-embedding.py:48-74): 512x512, frozen encoder
    x512 = syn.synthetic_images(64, 512, seed=31).to(dev)
    with torch.no_grad():
        dt = timed(lambda: trainer.image_model(x512), 3)
    out["embedding_precompute_512px_b64"] = {"images_per_sec": 64 / dt, "ms_per_batch": dt * 1e3}
    del x512
    # (2b) the same joint step with the image encoder's BatchNorm in TRAIN mode (batch statistics + running-stat updates, the state the
    # reference's constructor leaves the model in): separate statistics / normalisation / backward kernels around the same GEMMs
    try:
        trainer.image_model.train()
        dt = timed(lambda: trainer.step(images, ids, mask), 3)
        out["train_mode_batchnorm"] = {"ms_per_step": dt * 1e3, "images_per_sec": ids.shape[0] / dt,
                                       "note": "ImageModel.train(): batch statistics (csrc/bn_train.hip); the headline runs the eval-mode "
                                               "fold, the only mode the reference uses the encoder in"}
    finally:
        trainer.image_model.eval()
    # (3) ragged prompts (lengths ~U{8..32}, right-padded): same step, attention masked
    rid, rmask = syn.synthetic_tokens(ids.shape[0], ids.shape[1], seed=33, ragged=True)
    rid, rmask = rid.to(dev), rmask.to(dev)
    dt = timed(lambda: trainer.step(images, rid, rmask), 3)
    out["ragged_tokens"] = {"ms_per_step": dt * 1e3, "images_per_sec": ids.shape[0] / dt}
    # (4) PCIe-inclusive rate: the same batch copied from pinned host memory before every step, not overlapped
    himg = images.cpu().pin_memory()
    hid, hmask = ids.cpu().pin_memory(), mask.cpu().pin_memory()

    def pcie_step():
        trainer.step(himg.to(dev, non_blocking=True), hid.to(dev, non_blocking=True), hmask.to(dev, non_blocking=True))
    dt = timed(pcie_step, 3)
    out["pcie_inclusive"] = {"ms_per_step": dt * 1e3, "images_per_sec": ids.shape[0] / dt,
                             "note": "host -> device copy of every batch inside the step, serialised with it (worst case; a copy stream hides it)"}
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    if os.environ.get("CXRK_BENCH_SELFTEST") == "1":
        return selftest_rank()
    import faulthandler
    faulthandler.enable()
    faulthandler.dump_traceback_later(240, repeat=True, file=sys.stderr)   # a stuck phase shows where it is stuck
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.set_num_threads(max(1, min(usable_cores() // max(1, world), 16)))   # ranks share the host cores
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    # Rehearsal knobs (never set by the driver): CXRK_DIST_BACKEND=gloo + CXRK_BENCH_DEVICE=0 run the N>1 code path with
    # several ranks on ONE GPU (RCCL refuses two ranks per device), e.g. on the single-GPU test box.
    backend = os.environ.get("CXRK_DIST_BACKEND", "nccl")
    dev_index = int(os.environ.get("CXRK_BENCH_DEVICE", local_rank))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    if args.gpus != world:
        raise SystemExit(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                         f"(or run `python bench.py --gpus {args.gpus}` without a launcher: it starts the ranks itself)")
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)

    from incremental_multimodal_medical_learning_ii_amd import kernels as K
    from incremental_multimodal_medical_learning_ii_amd import synthetic as syn
    from incremental_multimodal_medical_learning_ii_amd.contrastive import JointContrastiveTrainer
    from incremental_multimodal_medical_learning_ii_amd.health_multimodal.image.model import get_biovil_resnet
    from incremental_multimodal_medical_learning_ii_amd.health_multimodal.text import CXRBertConfig, CXRBertModel

    log(f"building models (world={world}, batch/gpu={args.batch_per_gpu})")
    from incremental_multimodal_medical_learning_ii_amd import _lib as cxr_lib
    cxr_lib.set_precision(args.precision)
    im = get_biovil_resnet(None).eval()         # BN on running statistics (the reference's only mode)
    tm = CXRBertModel(CXRBertConfig()).eval()   # dropout inactive
    syn.fill_module_(im)                        # name-keyed deterministic weights: identical replicas on every rank
    syn.fill_module_(tm)                        # (the parity tests use the same fill)
    im = im.to(dev)
    if not args.no_bn_calibration:
        # a trained checkpoint's BatchNorm statistics match its activations; the name-keyed fill's do not, and 53 mismatched
        # BatchNorms in a row leave a common-mode component that makes all image embeddings parallel (cosine 1 - 1e-6): one Adam
        # step at any usable learning rate then pins the loss at ln(B) with zero cotangents (round 2's bench).  One calibration
        # pass over a fixed sample (same on every rank) gives the synthetic weights that property; it is set-up, not timed.
        im.calibrate_batchnorm_(structured_images(64, args.image_size, seed=4242).to(dev))
    if args.batchnorm == "train":
        im.train()
        args.no_secondary = True              # the precision / secondary legs are defined for the eval-mode headline only
    trainer = JointContrastiveTrainer(im, tm.to(dev), lr=args.lr, temperature=args.temperature)
    B = args.batch_per_gpu
    NB = 4                                      # resident batches the steps rotate through
    batches = []
    for j in range(NB):
        img = structured_images(B, args.image_size, seed=27 + 101 * j + rank).to(dev)
        ids, mask = syn.synthetic_tokens(B, args.seq_len, seed=28 + 101 * j + rank)
        batches.append((img, ids.to(dev), mask.to(dev)))
    images, ids, mask = batches[0]

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Evidence that the split-bf16 contractions meet the fp32 bar at THIS size: the same weights and batch through forward AND
    # backward in both precisions (no optimiser step; all ranks take part in the loss collectives) — BEFORE the first step, at
    # the initial weights, where the logits are not yet uniform and the loss (6.947 at B = 1024) is not ln B.  Gradients are compared by
    # norm-relative error: the two modes take different sides of ~1e-5 of the ReLU kinks.
    precision_check = None
    if not args.no_secondary:
        vals = {}
        probe_names = ("encoder.encoder.layer4.2.conv3.weight", "encoder.encoder.layer1.0.conv1.weight")
        tprobe = "bert.encoder.layer.11.intermediate.dense.weight"
        inamed, tnamed = dict(trainer.image_model.named_parameters()), dict(trainer.text_model.named_parameters())
        # third leg: exact fp32 again, with the images scaled by (1 + 1e-6): how far the image-encoder gradients move when the
        # forward changes in the last bits only — the conditioning against which the split-bf16 gradient differences must be read
        jitter = images * (1.0 + 1e-6)
        for mode in ("fp32", "fp32_jitter", "split_bf16"):
            cxr_lib.set_precision("split_bf16" if mode == "split_bf16" else "fp32")
            trainer.optimizer.zero_grad()
            src = jitter if mode == "fp32_jitter" else images
            ls = trainer.forward_loss(src, ids, mask)
            ls.backward()
            with torch.no_grad():
                ie = trainer.image_model(src[:64]).clone()
                te = trainer.text_model.get_projected_text_embeddings(ids[:64], mask[:64], normalize_embeddings=False).clone()
            vals[mode] = (ie, te, ls.detach().clone(), [inamed[n].grad.detach().clone() for n in probe_names], tnamed[tprobe].grad.detach().clone())
        del jitter
        trainer.optimizer.zero_grad()
        cxr_lib.set_precision(args.precision)
        a, b, j = vals["fp32"], vals["split_bf16"], vals["fp32_jitter"]
        rel = lambda x, y: float(((x - y).abs().max() / y.abs().max().clamp_min(1e-30)).item())
        nrel = lambda x, y: float(((x - y).norm() / y.norm().clamp_min(1e-30)).item())
        precision_check = {"what": "bench batch and weights through forward + backward in split_bf16 vs exact fp32 contractions "
                                   "(embeddings / loss: max abs diff / max abs; gradients: ||diff|| / ||fp32||)",
                           "image_embedding": rel(b[0], a[0]), "text_embedding": rel(b[1], a[1]),
                           "loss": abs(float(b[2]) - float(a[2])) / abs(float(a[2])),
                           "grad_image_layer4_conv3": nrel(b[3][0], a[3][0]), "grad_image_layer1_conv1": nrel(b[3][1], a[3][1]),
                           "grad_text_layer11_ffn": nrel(b[4], a[4]), "bar": 1e-3,
                           "fp32_vs_fp32_with_inputs_scaled_by_1p000001": {
                               "image_embedding": rel(j[0], a[0]), "grad_image_layer4_conv3": nrel(j[3][0], a[3][0]),
                               "grad_image_layer1_conv1": nrel(j[3][1], a[3][1]),
                               "note": "exact-fp32 against exact-fp32 on images * (1 + 1e-6): the image-encoder gradients move by this "
                                       "much when the forward changes in the last bits (ReLU / max-pool decisions on the other side of "
                                       "zero, DESIGN.md section 2); the split_bf16 rows above are to be read against it, not against `bar`"}}
        del vals
        # the same question with the discontinuity taken out: the fp32 forward's ReLU / max-pool decisions imposed on the split-bf16
        # backward (diagnostics.imposed_decision_gradient_errors), every image-encoder gradient tensor, cotangent = a fixed random one
        try:
            from incremental_multimodal_medical_learning_ii_amd.diagnostics import imposed_decision_gradient_errors
            cot = torch.randn(images.shape[0], 128, generator=torch.Generator().manual_seed(5)).to(dev)
            imp, free, flips, emb_err = imposed_decision_gradient_errors(trainer.image_model, images, cot)
            worst = max(((max(v), k) for k, v in imp.items()), key=lambda t: t[0])
            worst_free = max(((max(v), k) for k, v in free.items()), key=lambda t: t[0])
            precision_check["under_fp32_decisions"] = {
                "what": "split_bf16 backward with the exact-fp32 forward's ReLU masks / max-pool winners imposed, against the exact-fp32 "
                        "backward: [max |diff| / max |ref|, ||diff|| / ||ref||] per tensor; worst = the largest of either over all "
                        f"{len(imp)} image-encoder gradient tensors",
                "decisions_overridden": flips, "image_embedding": emb_err,
                "grad_image_layer4_conv3": imp[probe_names[0]], "grad_image_layer1_conv1": imp[probe_names[1]],
                "grad_image_stem_conv1": imp["encoder.encoder.conv1.weight"], "worst": [worst[0], worst[1]],
                "free_running_worst": [worst_free[0], worst_free[1]], "bar": 1e-3}
            del imp, free, cot
        except Exception as e:   # evidence, not the measurement
            precision_check["under_fp32_decisions"] = {"error": repr(e)}
        trainer.optimizer.zero_grad()
        log(f"precision check: {precision_check}")

    log(f"models + {NB} synthetic batches resident on the GPU; warm-up")
    cot_before = cotangent_norms(trainer, images, ids, mask)
    log(f"before the first step: {cot_before}")
    loss = None
    losses = []
    warm_ms = []
    for i in range(args.warmup):
        t_w = time.perf_counter()
        loss = trainer.step(*batches[i % NB])
        torch.cuda.synchronize()
        warm_ms.append((time.perf_counter() - t_w) * 1e3)
        losses.append(float(loss))
        log(f"warm-up step {i + 1}/{args.warmup} done in {warm_ms[-1]:.1f} ms, loss {float(loss):.4f}, "
            f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    sync()
    prof = (not args.no_roofline) and rank == 0
    single_stream = not trainer.two_streams
    if prof and single_stream:
        K.profiler.start()
    step_losses = []
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]   # per-step device times, no host sync inside
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        step_losses.append(trainer.step(*batches[(args.warmup + i) % NB]))
        marks[i + 1].record()
    sync()
    dt = time.perf_counter() - t0
    K.profiler.stop()
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    losses += [float(x) for x in step_losses]
    log(f"timed region: {args.steps} steps in {dt:.2f} s; loss {losses[0]:.4f} -> {losses[-1]:.4f}; "
        f"per step {' '.join(f'{x:.1f}' for x in step_ms)} ms")
    cot_after = cotangent_norms(trainer, images, ids, mask)
    log(f"after the timed region: {cot_after}")
    import math
    ln_b = math.log(world * B)
    # collapsed embeddings (all parallel: uniform logits, loss = ln(global batch) EXACTLY and staying there, cotangents -> 0) must
    # not be timed.  The loss may well pass through ln(B) on its way down, so the test is on what the backward is fed.
    collapsed = not all(math.isfinite(x) for x in losses)
    if cot_before is not None and cot_after is not None:
        collapsed = collapsed or cot_after["dL_dI"] < 0.02 * cot_before["dL_dI"] or cot_after["dL_dT"] < 0.02 * cot_before["dL_dT"] \
            or cot_after["mean_offdiag_cos_image"] > 0.99999 or cot_after["mean_offdiag_cos_text"] > 0.99999
    else:
        collapsed = collapsed or all(abs(x - ln_b) < 2e-5 for x in losses[-3:])
    if collapsed:
        raise SystemExit(f"[bench] the model collapsed during the run (loss trace {losses}, ln(global batch) = {ln_b:.5f}, cotangents "
                         f"{cot_before} -> {cot_after}): with uniform logits the InfoNCE cotangents vanish and both encoders' backward "
                         f"passes would be timed on zeros; refusing to report that number (lower --lr, or check the BatchNorm calibration)")
    prof_steps = args.steps
    if prof and not single_stream:
        # Per-kernel durations for the roofline: with the two encoders on two streams their kernels co-run, and an event
        # bracket around one launch then also covers the other stream's work.  The kernel profile is therefore taken over
        # two extra steps of the SAME step with both encoders on one stream (all ranks run them; not part of `value`).
        prof_steps = 2
        trainer.two_streams = False
        K.profiler.start()
    if not single_stream and not args.no_roofline:
        trainer.two_streams = False
        for j in range(2):
            trainer.step(*batches[j % NB])
        sync()
        K.profiler.stop()
        trainer.two_streams = True
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    final_loss = losses[-1]

    secondary = None
    if not args.no_secondary:   # same step in the other contraction precision, the full --steps (2 warm-up)
        other = "fp32" if args.precision == "split_bf16" else "split_bf16"
        cxr_lib.set_precision(other)
        for j in range(2):
            trainer.step(*batches[j % NB])
        sync()
        t1 = time.perf_counter()
        for j in range(args.steps):
            trainer.step(*batches[j % NB])
        sync()
        d2 = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(d2, op=dist.ReduceOp.MAX)
        secondary = {"precision": other, "value": world * B * args.steps / float(d2.item()), "unit": "images/sec",
                     "ms_per_step": float(d2.item()) / args.steps * 1e3, "steps": args.steps}
        cxr_lib.set_precision(args.precision)
        log(f"secondary measurement ({other}): {secondary['ms_per_step']:.1f} ms/step")

    side = None
    if world == 1 and not args.no_secondary:
        try:
            side = secondary_metrics(args, dev, trainer, images, ids, mask, dt / args.steps * 1e3)
            log(f"side measurements: {json.dumps(side)}")
        except Exception as e:  # never lose the headline over a side measurement
            side = {"error": repr(e)}

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * B * args.steps / dt
        out = {
            "metric": "contrastive train-step images/sec at global batch 1024; 1/2/4/8-GPU scaling",
            "value": value, "unit": "images/sec", "n_gpus": world, "rccl_ranks": dist.get_world_size() if world > 1 else 1,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "fp32" if args.precision == "fp32" else "bf16x3",
            "data": "synthetic",
            "config": {"workload": "joint image+text contrastive train step: ResNet-50 (224x224) + CXR-BERT (12 layers, 32 tokens) "
                                   "-> InfoNCE over the global batch -> backward through both encoders -> fused Adam "
                                   "(BASELINE config 3 at N=1: batch 1024 on one MI355X; config 5 at N=8: global 8192)",
                       "global_batch": world * B, "batch_per_gpu": B, "seq_len": args.seq_len, "image_size": args.image_size,
                       "temperature": args.temperature, "parallelism": f"dp{world}",
                       "weights": "name-keyed deterministic fill (synthetic.fill_module_)" + ("" if args.no_bn_calibration else
                                  "; BatchNorm running statistics calibrated on a 64-image sample (ImageModel.calibrate_batchnorm_)"),
                       "optimizer": f"Adam, lr {args.lr:g}",
                       "batches": f"{NB} resident synthetic batches, rotated",
                       "batchnorm": "running statistics (eval mode), gamma/beta trained" if args.batchnorm == "eval" else
                                    "VARIANT: batch statistics (train mode, csrc/bn_train.hip), running statistics updated",
                       "precision": args.precision,
                       "precision_note": "split_bf16: activations / gradients / weights that feed a contraction are stored as bf16 hi+lo "
                                         "planes (4 B per element) and every product is hi*hi + hi*lo + lo*hi on the bf16 MFMA with fp32 "
                                         "accumulation (~2^-16 relative); the parity suite (1e-3 relative on embeddings, loss, gradients) "
                                         "runs in both modes; fp32 = exact fp32 MFMA"},
            "final_loss": final_loss, "loss_trace": [round(x, 5) for x in losses], "ln_global_batch": round(ln_b, 5),
            "lr": args.lr, "step_ms": [round(x, 2) for x in step_ms], "warmup_step_ms": [round(x, 1) for x in warm_ms],
            "first_timed_step_ms": round(step_ms[0], 2), "steady_state_ms": round(sorted(step_ms)[len(step_ms) // 2], 2),
            "cotangent_norm": {"before_first_step": cot_before, "after_timed_region": cot_after,
                               "note": "norm of dL/d(embeddings) = what both encoders' backward passes are fed; 0 for a collapsed model"},
            "loss_note": "BatchNorm statistics calibrated on a sample batch, Adam at --lr: the run stays in the first, descending phase of "
                         "contrastive training (loss above ln(global batch) and falling, embeddings not parallel, cotangents O(1e-2..1)); "
                         "bench.py exits with an error instead of a number when the embeddings collapse (cotangent norm < 2 % of the initial one)",
            "model_tflops_per_s": FLOP_PER_PAIR_STEP * world * B * args.steps / dt / 1e12,
        }
        if prof:
            summ = K.profiler.summary()
            if summ:
                for k_, v_ in sorted(summ.items(), key=lambda kv: -kv[1]["ms"]):
                    log(f"  {k_:58s} {v_['launches'] / prof_steps:6.1f} launches/step {v_['ms'] / prof_steps:8.2f} ms/step "
                        f"{v_['flops'] / (v_['ms'] * 1e-3) / 1e12:6.1f} TFLOP/s {v_['bytes'] / (v_['ms'] * 1e-3) / 1e9:7.0f} GB/s (algorithmic)")
                key, d = max(summ.items(), key=lambda kv: kv[1]["ms"])
                secs = d["ms"] * 1e-3
                tflops = d["flops"] / secs / 1e12
                gbps = d["bytes"] / secs / 1e9
                bf16 = key.startswith("gemm_x3") or key.startswith("gemm_pw")
                peak_fl = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_FP32_MFMA_TFLOPS
                # price the dominant kernel against BOTH roofs and report the one it sits closer to (the binding one).  The
                # split-bf16 mainloops execute 3 bf16 MFMAs per algorithmic product: their MFMA roof for algorithmic FLOPs is peak / 3.
                frac_mfma, frac_hbm = tflops / (peak_fl / 3.0 if bf16 else peak_fl), gbps / PEAK_HBM_GBS
                tot_ms = sum(v["ms"] for v in summ.values())
                tot_fl = sum(v["flops"] for v in summ.values())
                tot_by = sum(v["bytes"] for v in summ.values())
                traffic = None
                mfma_busy = None
                try:  # HBM bytes per launch of this kernel from the committed PMC passes (scripts/pmc_traffic.sh)
                    pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
                    traffic = pmc["kernels"].get(key, {}).get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
                try:  # SQ counters of the same kernel as it runs inside this step (scripts/pmc_sq.sh pointed at bench.py)
                    sq = json.load(open(os.path.join(ROOT, "profiles", "pmc_sq.json")))
                    mfma_busy = sq["kernels"].get(key)
                except Exception:
                    mfma_busy = None
                mfma_note = ("algorithmic 2*M*N*K FLOPs against the dense bf16 MFMA peak / 3 (the split-bf16 mainloop executes 3 bf16 "
                             "MFMAs per product)") if bf16 else "exact fp32 MFMA, priced against the fp32 matrix peak"
                hbm = frac_hbm >= frac_mfma
                out["roofline"] = {"bound": "hbm" if hbm else "mfma",
                                   "achieved": gbps if hbm else tflops, "peak": PEAK_HBM_GBS if hbm else (peak_fl / 3.0 if bf16 else peak_fl),
                                   "unit": "GB/s" if hbm else "TFLOP/s", "frac": frac_hbm if hbm else frac_mfma,
                                   "traffic": traffic, "mfma_busy": mfma_busy, "kernel": key,
                                   "note": ("algorithmic bytes: every operand and fused side input (residual, ReLU bit mask) read once + "
                                            "the output written once, 4 B per element (1/8 B for masks); summed over the launches of this "
                                            "instantiation / their HIP-event time") if hbm else mfma_note,
                                   "launches_per_step": d["launches"] / prof_steps,
                                   "profiled": ("the timed region" if single_stream else
                                                "2 extra steps with both encoders on ONE stream (in the timed region their kernels "
                                                "co-run on two streams, so a per-launch event bracket would cover both)"),
                                   "avg_launch_ms": d["ms"] / d["launches"],
                                   "binding_roof_frac": d["floor_ms"] / d["ms"],
                                   "binding_roof_note": "sum over this instantiation's launches of max(FLOPs / MFMA peak, algorithmic bytes / HBM "
                                                        "peak) / their time: each launch priced against ITS binding roof (the instantiation mixes "
                                                        "HBM-bound K = 64 layers with MFMA-bound K = 2304 ones; `frac` prices them all against one roof)",
                                   "gflop_per_launch": d["flops"] / d["launches"] / 1e9,
                                   "algorithmic_mb_per_launch": d["bytes"] / d["launches"] / 1e6,
                                   "other_bound": {"bound": "mfma" if hbm else "hbm", "achieved": tflops if hbm else gbps,
                                                   "peak": (peak_fl / 3.0 if bf16 else peak_fl) if hbm else PEAK_HBM_GBS,
                                                   "unit": "TFLOP/s" if hbm else "GB/s",
                                                   "frac": frac_mfma if hbm else frac_hbm, "note": mfma_note if hbm else ""},
                                   "family": {"kernel": "gemm_*_kernel<*> (all MFMA mainloop instantiations)",
                                              "binding_roof_frac": sum(v["floor_ms"] for v in summ.values()) / tot_ms,
                                              "tflops": tot_fl / (tot_ms * 1e-3) / 1e12,
                                              "algorithmic_gbps": tot_by / (tot_ms * 1e-3) / 1e9,
                                              "ms_per_step": tot_ms / prof_steps}}
        if secondary is not None:
            out["other_precision"] = secondary
        if precision_check is not None:
            out["precision_check"] = precision_check
        if side is not None:
            out["secondary"] = side
        if world == 1 and not args.no_cpu_baseline:   # the CPU leg runs at N=1 only
            try:
                out["cpu_baseline"] = cpu_baseline(args.cpu_baseline_batch, args.seq_len, args.image_size, args.temperature,
                                                   args.cpu_baseline_steps)
            except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
                out["cpu_baseline"] = {"value": None, "error": repr(e)}
        faulthandler.cancel_dump_traceback_later()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
