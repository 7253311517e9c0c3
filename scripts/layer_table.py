"""Per-layer table of the ResNet-50 convolutions of the step (GPU): every distinct shape x {fprop, dgrad, wgrad} at batch N,
with its count per step, the time against a floor = max(FLOPs / F_REF, algorithmic bytes / B_REF), and the excess
count * (t - floor) the step pays for it.  usage: layer_table.py [batch] [reps]   (CXRK_PRECISION selects the mainloop)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from incremental_multimodal_medical_learning_ii_amd import kernels as K, _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
F_REF, B_REF = 265e12, 5.5e12   # what the dense BERT GEMMs reach in split-bf16 / a streaming kernel reaches from HBM
dev = "cuda"
print(f"precision {_lib.get_precision()}  batch {N}  reps {reps}  wide {os.environ.get('CXRK_WIDE', '0')}")

def timeit(fn):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

# (H, C, Ko, R, stride, count, has_dgrad)
L = [(56, 64, 64, 1, 1, 1, True), (56, 64, 64, 3, 1, 3, True), (56, 64, 256, 1, 1, 4, True), (56, 256, 64, 1, 1, 2, True),
     (56, 256, 128, 1, 1, 1, True), (56, 128, 128, 3, 2, 1, True), (28, 128, 512, 1, 1, 4, True), (56, 256, 512, 1, 2, 1, True),
     (28, 512, 128, 1, 1, 3, True), (28, 128, 128, 3, 1, 3, True),
     (28, 512, 256, 1, 1, 1, True), (28, 256, 256, 3, 2, 1, True), (14, 256, 1024, 1, 1, 6, True), (28, 512, 1024, 1, 2, 1, True),
     (14, 1024, 256, 1, 1, 5, True), (14, 256, 256, 3, 1, 5, True),
     (14, 1024, 512, 1, 1, 1, True), (14, 512, 512, 3, 2, 1, True), (7, 512, 2048, 1, 1, 3, True), (14, 1024, 2048, 1, 2, 1, True),
     (7, 2048, 512, 1, 1, 2, True), (7, 512, 512, 3, 1, 2, True), (7, 2048, 128, 1, 1, 1, True), (7, 128, 128, 1, 1, 1, True)]
rows = []
for (H, C, Ko, R, st, cnt, dg) in L:
    pad = R // 2
    Ho = (H + 2 * pad - R) // st + 1
    x = torch.randn(N, H, H, C, device=dev); w = torch.randn(Ko, R, R, C, device=dev) * 0.05
    sh = torch.zeros(Ko, device=dev); y = torch.empty(N, Ho, Ho, Ko, device=dev)
    fl = 2.0 * N * Ho * Ho * Ko * R * R * C
    bx, by, bw = x.numel() * 4, y.numel() * 4, w.numel() * 4
    tag = f"{H:3d}^2 {C:4d}->{Ko:4d} {R}x{R} s{st}"
    t = timeit(lambda: K.conv_fwd(x, w, sh, None, y, N, H, H, C, Ko, R, R, st, pad, True))
    rows.append((tag, "fprop", cnt, t, fl, bx + by + bw))
    dy = torch.randn_like(y); dx = torch.empty_like(x)
    relu_src = None if (R == 1 and st == 2) else x
    t = timeit(lambda: K.conv_bwd_data(dy, w, None, relu_src, dx, N, H, H, C, Ko, R, R, st, pad))
    rows.append((tag, "dgrad", cnt, t, fl, bx * (2 if relu_src is not None else 1) + by + bw))
    sc = torch.ones(Ko, device=dev); s2 = torch.zeros(2, Ko, device=dev)
    dwt = torch.empty_like(w); dgm = torch.empty(Ko, device=dev); db = torch.empty(Ko, device=dev)
    t = timeit(lambda: K.conv_bwd_params(x, dy, w, sc, sc, sh, s2[0], sc, s2[1], dwt, dgm, db, False, N, H, H, C, C, Ko, R, R, st, pad))
    rows.append((tag, "wgrad", cnt, t, fl, bx + by + bw))
    del x, y, dy, dx
tot = exc = 0.0
print(f"{'layer':26s} {'pass':6s} cnt {'ms':>8s} {'TFLOP/s':>8s} {'TB/s':>6s} {'floor':>7s} {'bound':>5s} {'excess*cnt':>10s}")
for (tag, ps, cnt, t, fl, by) in sorted(rows, key=lambda r: -r[2] * (r[3] - max(r[4] / F_REF, r[5] / B_REF) * 1e3)):
    ff, fb = fl / F_REF * 1e3, by / B_REF * 1e3
    floor = max(ff, fb)
    tot += cnt * t; exc += cnt * (t - floor)
    print(f"{tag:26s} {ps:6s} {cnt:3d} {t:8.3f} {fl / t / 1e9:8.1f} {by / t / 1e9:6.2f} {floor:7.3f} {'mfma' if ff > fb else 'hbm':>5s} {cnt * (t - floor):10.2f}", flush=True)
print(f"total {tot:.1f} ms/step in these launches, {exc:.1f} ms above the floors")
