"""Per-layer table of the ResNet-50 convolutions of the step (GPU, planes storage = the split_bf16 product path): every
distinct shape x {fprop, dgrad, wgrad} at batch N with its count per step, its time, and the hardware floor

    floor = max(FLOPs / F_PEAK, algorithmic bytes / B_PEAK)

with the PEAKS of /opt/skills/guides/MI355X_MICROARCH.md, not what this code reaches elsewhere: F_PEAK = 2500 / 3 = 833 TFLOP/s
(dense bf16 MFMA peak, three bf16 MFMAs per product in split-bf16), B_PEAK = 8 TB/s.  `excess` = count * (t - floor) is what the
step pays above the floor for that row; the table is sorted by it, so the top rows are the next kernels to fix.
usage: layer_table.py [batch] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from incremental_multimodal_medical_learning_ii_amd import _lib, kernels as K  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
F_PEAK, B_PEAK = 2500e12 / 3.0, 8.0e12
dev = "cuda"
_lib.set_precision("split_bf16")
print(f"precision split_bf16 (planes)  batch {N}  reps {reps}  floors: {F_PEAK / 1e12:.0f} TFLOP/s, {B_PEAK / 1e12:.0f} TB/s")


def timeit(fn):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def planes(*shape, scale=1.0):
    return K.split_planes(torch.randn(*shape, device=dev) * scale)


# (H, C, Ko, R, stride, count, identity-add in the forward epilogue)   — torchvision ResNet-50 bottlenecks, stride on the 3x3
L = [(56, 64, 64, 1, 1, 1, 0), (56, 64, 64, 3, 1, 3, 0), (56, 64, 256, 1, 1, 4, 3), (56, 256, 64, 1, 1, 2, 0),
     (56, 256, 128, 1, 1, 1, 0), (56, 128, 128, 3, 2, 1, 0), (28, 128, 512, 1, 1, 4, 4), (56, 256, 512, 1, 2, 1, 0),
     (28, 512, 128, 1, 1, 3, 0), (28, 128, 128, 3, 1, 3, 0),
     (28, 512, 256, 1, 1, 1, 0), (28, 256, 256, 3, 2, 1, 0), (14, 256, 1024, 1, 1, 6, 6), (28, 512, 1024, 1, 2, 1, 0),
     (14, 1024, 256, 1, 1, 5, 0), (14, 256, 256, 3, 1, 5, 0),
     (14, 1024, 512, 1, 1, 1, 0), (14, 512, 512, 3, 2, 1, 0), (7, 512, 2048, 1, 1, 3, 3), (14, 1024, 2048, 1, 2, 1, 0),
     (7, 2048, 512, 1, 1, 2, 0), (7, 512, 512, 3, 1, 2, 0)]
if os.environ.get("CXRK_LAYER_ROWS"):      # e.g. CXRK_LAYER_ROWS=1,9: only these entries of L (A/B runs of one kernel)
    L = [L[int(i)] for i in os.environ["CXRK_LAYER_ROWS"].split(",")]
rows = []
for (H, C, Ko, R, st, cnt, nres) in L:
    pad = R // 2
    Ho = (H + 2 * pad - R) // st + 1
    x, w = planes(N, H, H, C), planes(Ko, R * R * C, scale=0.05)
    wf = torch.randn(Ko, R, R, C, device=dev) * 0.05
    sh = torch.zeros(Ko, device=dev)
    y = K.Planes.empty(N * Ho * Ho, Ko, device=dev)
    res = planes(N * Ho * Ho, Ko) if nres else None
    mask = torch.empty(N * Ho * Ho, Ko // 8, dtype=torch.uint8, device=dev)
    xmask = torch.randint(0, 256, (N * H * H, C // 8), dtype=torch.uint8, device=dev)
    fl = 2.0 * N * Ho * Ho * Ko * R * R * C
    bx, by, bw = N * H * H * C * 4, N * Ho * Ho * Ko * 4, Ko * R * R * C * 4
    tag = f"{H:3d}^2 {C:4d}->{Ko:4d} {R}x{R} s{st}"
    t = timeit(lambda: K.conv_fwd_pl(x, w, sh, res, y, mask, N, H, H, C, Ko, R, R, st, pad, True))
    rows.append((tag, "fprop", cnt, t, fl, bx + by * (2 if nres else 1) + bw + by / 32))
    dy, dx = planes(N * Ho * Ho, Ko), K.Planes.empty(N * H * H, C, device=dev)
    sums = torch.empty(C, device=dev)
    proj = R == 1 and st == 2   # projection shortcut: its input is the block input, whose ReLU mask / column sums belong to conv1's dgrad
    t = timeit(lambda: K.conv_bwd_data_pl(dy, w, None, None if proj else xmask, dx, N, H, H, C, Ko, R, R, st, pad, sums=None if proj else sums))
    rows.append((tag, "dgrad", cnt, t, fl, bx + by + bw + (0 if proj else bx / 32)))
    sc, zero = torch.ones(Ko, device=dev), torch.zeros(Ko, device=dev)
    dwt, dgm, db = torch.empty_like(wf), torch.empty(Ko, device=dev), torch.empty(Ko, device=dev)
    t = timeit(lambda: K.conv_bwd_params_pl(x, dy, wf, sc, sc, zero, zero, dwt, dgm, db, False, N, H, H, C, Ko, R, R, st, pad))
    rows.append((tag, "wgrad", cnt, t, fl, bx + by + bw))
    print("measured", tag, " ".join(f"{r[1]} {r[3]:.3f} ms" for r in rows[-3:]), flush=True)
    del x, y, dy, dx, res
tot = exc = 0.0
print(f"{'layer':26s} {'pass':6s} cnt {'ms':>8s} {'TFLOP/s':>8s} {'TB/s':>6s} {'floor':>7s} {'bound':>5s} {'t/floor':>7s} {'excess*cnt':>10s}")
for (tag, ps, cnt, t, fl, by) in sorted(rows, key=lambda r: -r[2] * (r[3] - max(r[4] / F_PEAK, r[5] / B_PEAK) * 1e3)):
    ff, fb = fl / F_PEAK * 1e3, by / B_PEAK * 1e3
    floor = max(ff, fb)
    tot += cnt * t; exc += cnt * (t - floor)
    print(f"{tag:26s} {ps:6s} {cnt:3d} {t:8.3f} {fl / t / 1e9:8.1f} {by / t / 1e9:6.2f} {floor:7.3f} {'mfma' if ff > fb else 'hbm':>5s} "
          f"{t / floor:7.2f} {cnt * (t - floor):10.2f}", flush=True)
print(f"total {tot:.1f} ms/step in these launches; {tot - exc:.1f} ms at the floors, {exc:.1f} ms above them")
