import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from incremental_multimodal_medical_learning_ii_amd import kernels as K, _lib
dev = "cuda"
def run(N, H, C, Ko):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(N, H, H, C, generator=g).to(dev); dy = torch.randn(N, H, H, Ko, generator=g).to(dev)
    w = torch.randn(Ko, 1, 1, C, generator=g).to(dev)
    one = torch.ones(Ko, device=dev); z = torch.zeros(Ko, device=dev); s2 = torch.zeros(2, Ko, device=dev)
    outs = []
    for mode in ("fp32", "split_bf16"):
        _lib.set_precision(mode)
        dw = torch.empty(Ko, 1, 1, C, device=dev); dg = torch.empty(Ko, device=dev); db = torch.empty(Ko, device=dev)
        K.conv_bwd_params(x, dy, w, one, one, z, s2[0], one, s2[1], dw, dg, db, False, N, H, H, C, C, Ko, 1, 1, 1, 0)
        outs.append(dw.view(Ko, C).clone())
    ref = dy.view(-1, Ko).double().T @ x.view(-1, C).double()
    e32 = (outs[0].double() - ref).abs(); e3 = (outs[1].double() - ref).abs()
    sc = ref.abs().max()
    print(f"N={N} H={H} C={C} Ko={Ko}: fp32 err {e32.max()/sc:.2e}  x3 err {e3.max()/sc:.2e}")
    bad = (e3 > 1e-3 * sc).nonzero()
    if len(bad):
        print("  bad count", len(bad), "rows(ko)", sorted(set(bad[:,0].tolist()))[:16], "... cols(c)", sorted(set(bad[:,1].tolist()))[:16])
        i = tuple(bad[0].tolist()); print("  e.g.", i, outs[1][i].item(), ref[i].item())
for cfg in [(2,14,64,256),(2,7,512,2048),(2,28,512,128),(2,14,256,1024),(2,14,1024,256)]:
    run(*cfg)
