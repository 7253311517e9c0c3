"""Conditioning study of the BatchNorm gamma gradient taken from the weight gradient (csrc/conv_wgrad.hip):

    dgamma = rstd * (<w, dW_raw> - mean * sum(dy))          (what the kernels compute: no read of the activation)
    dgamma = sum(dy * (z - mean) * rstd)                    (direct form; fp64 on the host is the reference here)

on one conv + BN unit whose input is post-ReLU (all positive, like every unit of the trunk) so that the pre-BN output z has a
mean well away from zero, swept over |mean| / sigma of the running statistics and over both contraction precisions.  The two
forms are algebraically equal; the question is how much the subtraction <w, dW> - mean * sum(dy) cancels.  Prints the worst
relative error over the channels, against max |dgamma|.   usage: exp_dgamma.py"""
import math
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from incremental_multimodal_medical_learning_ii_amd import _lib, kernels as K  # noqa: E402

dev = "cuda"
N, H, C, Ko, R = 16, 28, 128, 128, 3
g = torch.Generator().manual_seed(3)
x = torch.randn(N, C, H, H, generator=g).abs()                       # post-ReLU input
w = torch.randn(Ko, C, R, R, generator=g) / math.sqrt(C * R * R)
dy = torch.randn(N, Ko, H, H, generator=g)
dy = dy * (torch.rand(N, Ko, H, H, generator=g) > 0.5)                  # gradient behind the unit's own ReLU
z = F.conv2d(x.double(), w.double(), padding=1)
sig = z.std(dim=(0, 2, 3))
for ratio in (0.0, 1.0, 5.0, 20.0, 50.0):
    rm = (z.mean(dim=(0, 2, 3)) + ratio * sig).float()                # running mean off the batch mean by `ratio` sigma
    rv = (sig * sig).float()
    rstd64 = 1.0 / torch.sqrt(rv.double() + 1e-5)
    ref = (dy.double() * (z - rm.double()[None, :, None, None]) * rstd64[None, :, None, None]).sum(dim=(0, 2, 3))
    for mode in ("fp32", "split_bf16"):
        _lib.set_precision(mode)
        gamma, beta = torch.ones(Ko, device=dev), torch.zeros(Ko, device=dev)
        xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
        dyd = dy.permute(0, 2, 3, 1).contiguous().to(dev)
        w_cl = w.permute(0, 2, 3, 1).contiguous().to(dev)
        sc, sh, rstd = (torch.empty(Ko, device=dev) for _ in range(3))
        dw, dg, db = torch.empty(Ko, R, R, C, device=dev), torch.empty(Ko, device=dev), torch.empty(Ko, device=dev)
        if mode == "fp32":
            ws = torch.empty(Ko, R, R, C, device=dev)
            K.bn_fold(w_cl, gamma, beta, rm.to(dev), rv.to(dev), 1e-5, Ko, R * R, C, C, ws, sc, sh, rstd)
            sumdy = K.colsum(dyd.view(-1, Ko), torch.empty(Ko, device=dev))
            K.conv_bwd_params(xd, dyd, w_cl, sc, rstd, rm.to(dev), sumdy, dw, dg, db, False, N, H, H, C, C, Ko, R, R, 1, 1)
        else:
            wsp = K.Planes.empty(Ko, R * R * C, device=dev)
            K.bn_fold_pl(w_cl, gamma, beta, rm.to(dev), rv.to(dev), 1e-5, Ko, R * R, C, C, wsp, sc, sh, rstd)
            xp, dyp = K.split_planes(xd.view(-1, C)), K.split_planes(dyd.view(-1, Ko))
            sumdy = K.colsum(dyp, torch.empty(Ko, device=dev))
            K.conv_bwd_params_pl(xp, dyp, w_cl, sc, rstd, rm.to(dev), sumdy, dw, dg, db, False, N, H, H, C, Ko, R, R, 1, 1)
        err = float((dg.double().cpu() - ref).abs().max() / ref.abs().max())
        canc = float(((rm.double() * dy.double().sum(dim=(0, 2, 3))).abs() * rstd64).max() / ref.abs().max())
        print(f"|running mean - batch mean| = {ratio:4.1f} sigma  {mode:10s}  max |dgamma - ref| / max |ref| = {err:.2e}   "
              f"(cancelled term / result = {canc:.1f})", flush=True)
