"""N>1 protocol on CPU: 2 processes, gloo backend.  The HIP kernels cannot run here, so the orchestration in
`functional._InfoNCE` / `optim._FlatOptimizer.all_reduce_grads` is driven with a TEST-ONLY torch emulation of the
kernel wrappers (tests/cpu_kernels.py) — what is checked is the sharding / all-gather / all-reduce logic: a 2-rank
run over row shards reproduces the single-process global-batch loss and gradients of the CPU oracle."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    import cpu_kernels
    from incremental_multimodal_medical_learning_ii_amd import functional as Fh
    from incremental_multimodal_medical_learning_ii_amd import optim as cxr_optim
    from incremental_multimodal_medical_learning_ii_amd import synthetic as syn
    Fh.K = cpu_kernels  # test-only emulation of the kernel wrappers
    Bg, D, tau = 12, 128, 0.07
    B = Bg // world
    I = torch.from_numpy(syn._normal("dist.I", (Bg, D)))
    T = torch.from_numpy(syn._normal("dist.T", (Bg, D)))
    w = torch.nn.Parameter(torch.from_numpy(syn._normal("dist.W", (D, D))) * 0.1)   # a shared "encoder" weight
    sl = slice(rank * B, (rank + 1) * B)
    opt = cxr_optim.SGD([w], lr=0.1)
    opt.zero_grad()
    img = (I[sl] @ w).requires_grad_(True)
    img.retain_grad()
    txt = T[sl].clone().requires_grad_(True)
    loss = Fh.infonce_loss(img, txt, tau)
    loss.backward()
    opt.all_reduce_grads()
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), loss=loss.item(), dimg=img.grad.numpy(), dtxt=txt.grad.numpy(),
             dw=opt.flat_g[: D * D].reshape(D, D).numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_infonce_matches_single_process_oracle(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    from incremental_multimodal_medical_learning_ii_amd import synthetic as syn
    from oracle import ref_loss
    Bg, D, tau = 12, 128, 0.07
    I = torch.from_numpy(syn._normal("dist.I", (Bg, D)))
    T = torch.from_numpy(syn._normal("dist.T", (Bg, D))).requires_grad_(True)
    w = (torch.from_numpy(syn._normal("dist.W", (D, D))) * 0.1).requires_grad_(True)
    img = I @ w
    img.retain_grad()
    loss, _ = ref_loss.infonce(img, T, tau)
    loss.backward()
    r = [np.load(tmp_path / f"r{k}.npz") for k in range(world)]
    B = Bg // world
    for k in range(world):
        assert abs(float(r[k]["loss"]) - loss.item()) < 1e-5                      # every rank reports the global loss
        np.testing.assert_allclose(r[k]["dimg"], img.grad[k * B:(k + 1) * B].numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(r[k]["dtxt"], T.grad[k * B:(k + 1) * B].numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(r[k]["dw"], w.grad.numpy(), rtol=1e-4, atol=1e-6)  # summed over ranks = global grad
