"""Micro-benchmark of the MFMA core on the shapes that dominate the step (GPU).  usage: gemm_bench.py [reps]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from incremental_multimodal_medical_learning_ii_amd import kernels as K
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = "cuda"
def timeit(fn, flops, name):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    print(f"{name:58s} {ms:8.3f} ms {flops / ms / 1e9:7.1f} TFLOP/s", flush=True)
T = 32768
x768, x3072 = torch.randn(T, 768, device=dev), torch.randn(T, 3072, device=dev)
w1, w2, wq = torch.randn(3072, 768, device=dev), torch.randn(768, 3072, device=dev), torch.randn(2304, 768, device=dev)
b1 = torch.randn(3072, device=dev)
y3072, y768, y2304 = torch.empty(T, 3072, device=dev), torch.empty(T, 768, device=dev), torch.empty(T, 2304, device=dev)
timeit(lambda: K.linear_fwd(x768, w1, b1, act=K.ACT_GELU, out=y3072), 2 * T * 768 * 3072, "dense NT  32768x3072x768 (+bias+gelu)")
timeit(lambda: K.linear_fwd(x768, w1, out=y3072), 2 * T * 768 * 3072, "dense NT  32768x3072x768 (plain)")
timeit(lambda: K.linear_fwd(x3072, w2, out=y768), 2 * T * 768 * 3072, "dense NT  32768x768x3072")
timeit(lambda: K.linear_fwd(x768, wq, out=y2304), 2 * T * 768 * 2304, "dense NT  32768x2304x768")
timeit(lambda: K.linear_bwd_data(y3072, w1, out=y768), 2 * T * 768 * 3072, "dense NN  32768x768x3072")
dw = torch.empty(3072, 768, device=dev)
timeit(lambda: K.linear_bwd_weight(y3072, x768, dw), 2 * T * 768 * 3072, "dense TN  3072x768x32768 (split-K)")
xa = torch.randn(200704, 1152, device=dev); wa = torch.randn(128, 1152, device=dev); ya = torch.empty(200704, 128, device=dev)
timeit(lambda: K.linear_fwd(xa, wa, out=ya), 2 * 200704 * 128 * 1152, "dense NT  200704x128x1152 (= 28x28 128->128 3x3 as a GEMM)")
xb = torch.randn(802816, 64, device=dev); wb = torch.randn(256, 64, device=dev); yb = torch.empty(802816, 256, device=dev)
timeit(lambda: K.linear_fwd(xb, wb, out=yb), 2 * 802816 * 256 * 64, "dense NT  802816x256x64   (= 56x56 64->256 1x1 as a GEMM)")
# conv shapes at batch 256 (same per-CU behaviour as 1024, 4x shorter)
N = 256
def conv_case(H, C, Ko, R, stride, tag):
    pad = R // 2
    Ho = (H + 2 * pad - R) // stride + 1
    x = torch.randn(N, H, H, C, device=dev); w = torch.randn(Ko, R, R, C, device=dev) * 0.05
    sh = torch.zeros(Ko, device=dev); y = torch.empty(N, Ho, Ho, Ko, device=dev)
    fl = 2.0 * N * Ho * Ho * Ko * R * R * C
    timeit(lambda: K.conv_fwd(x, w, sh, None, y, N, H, H, C, Ko, R, R, stride, pad, True), fl, f"conv fwd   {tag}")
    dy = torch.randn_like(y); dx = torch.empty_like(x)
    timeit(lambda: K.conv_bwd_data(dy, w, None, x, dx, N, H, H, C, Ko, R, R, stride, pad) if not (R == 1 and stride == 2) else
           K.conv_bwd_data(dy, w, None, None, dx, N, H, H, C, Ko, R, R, stride, pad), fl, f"conv dgrad {tag}")
    sc = torch.ones(Ko, device=dev); s2 = torch.zeros(2, Ko, device=dev)
    dwt = torch.empty_like(w); dg = torch.empty(Ko, device=dev); db = torch.empty(Ko, device=dev)
    timeit(lambda: K.conv_bwd_params(x, dy, w, sc, sc, sh, s2[0], sc, s2[1], dwt, dg, db, False, N, H, H, C, C, Ko, R, R, stride, pad),
           fl, f"conv wgrad {tag}")
conv_case(56, 64, 256, 1, 1, "56x56 64->256 1x1")
conv_case(56, 256, 64, 1, 1, "56x56 256->64 1x1")
conv_case(56, 64, 64, 3, 1, "56x56 64->64 3x3")
conv_case(28, 128, 128, 3, 1, "28x28 128->128 3x3")
conv_case(28, 512, 128, 1, 1, "28x28 512->128 1x1")
conv_case(14, 256, 256, 3, 1, "14x14 256->256 3x3")
conv_case(14, 1024, 256, 1, 1, "14x14 1024->256 1x1")
conv_case(7, 512, 512, 3, 1, "7x7 512->512 3x3")
conv_case(7, 512, 2048, 1, 1, "7x7 512->2048 1x1")
conv_case(56, 128, 128, 3, 2, "56->28 128->128 3x3 s2")
