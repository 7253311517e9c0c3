"""Diagnostic: 2 gloo ranks on one GPU run one DP step; per-parameter comparison rank0 vs rank1 vs single process."""
import os, sys, socket, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_dist_gpu as T

def named_flat(tr):
    out = {}
    for side, m in (("i", tr.image_model), ("t", tr.text_model)):
        for n, p in m.named_parameters():
            out[f"{side}.{n}"] = p.detach().float().cpu().numpy().copy()
    return out

def worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    tr, images, ids, mask = T._build()
    B = T.B_GLOBAL // world; sl = slice(rank * B, (rank + 1) * B)
    tr.optimizer.zero_grad()
    loss = tr.forward_loss(images[sl].cuda(), ids[sl].cuda(), mask[sl].cuda())
    loss.backward()
    torch.cuda.synchronize()
    g_local = {k: v for k, v in zip([f"{s}.{n}" for s, m in (("i", tr.image_model), ("t", tr.text_model)) for n, _ in m.named_parameters()],
                                    [p.grad.detach().cpu().numpy().copy() if p.grad is not None else None for m in (tr.image_model, tr.text_model) for p in m.parameters()])}
    tr.optimizer.all_reduce_grads(None)
    torch.cuda.synchronize()
    g_sum = {k: (p.grad.detach().cpu().numpy().copy() if p.grad is not None else None) for k, p in
             [(f"{s}.{n}", p) for s, m in (("i", tr.image_model), ("t", tr.text_model)) for n, p in m.named_parameters()]}
    offs = {k: p.data_ptr() - tr.optimizer.flat_p.data_ptr() for k, p in
            [(f"{s}.{n}", p) for s, m in (("i", tr.image_model), ("t", tr.text_model)) for n, p in m.named_parameters()]}
    np.save(os.path.join(out_dir, f"r{rank}.npy"), {"local": g_local, "sum": g_sum, "offs": offs}, allow_pickle=True)
    dist.barrier(); dist.destroy_process_group()

if __name__ == "__main__":
    import torch.multiprocessing as mp, tempfile
    d = tempfile.mkdtemp()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(worker, args=(2, port, d), nprocs=2, join=True)
    r = [np.load(os.path.join(d, f"r{k}.npy"), allow_pickle=True).item() for k in range(2)]
    bad_off = [k for k in r[0]["offs"] if r[0]["offs"][k] != r[1]["offs"][k]]
    print("parameters whose flat-buffer offset differs between ranks:", len(bad_off), bad_off[:5])
    nbad = 0
    for k in r[0]["sum"]:
        a, b = r[0]["sum"][k], r[1]["sum"][k]
        if a is None or b is None: continue
        ref = r[0]["local"][k] + r[1]["local"][k]
        e01 = np.abs(a - b).max(); e0 = np.abs(a - ref).max(); sc = np.abs(ref).max() + 1e-30
        if e01 > 0 or e0 > 1e-5 * sc:
            nbad += 1
            if nbad <= 12: print(f"{k:60s} |r0-r1| {e01:.3g}  |r0-(l0+l1)| {e0:.3g}  scale {sc:.3g}")
    print("tensors with a wrong / unequal summed gradient:", nbad, "of", len(r[0]["sum"]))
