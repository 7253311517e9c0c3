"""TEST-ONLY torch emulation of the handful of cxrk kernel wrappers that the data-parallel orchestration calls.
It exists so the N>1 protocol (all-gather of embeddings / log-sum-exps, flat-gradient all-reduce) can be exercised
with the gloo backend on CPU, where no HIP kernel can run.  It is never imported by the package."""
import torch


def l2norm_fwd(x, eps=1e-12, out=None):
    n = x.norm(dim=1).clamp_min(eps)
    if out is None:
        return x / n[:, None], n
    out.copy_(x / n[:, None])
    return out, n


def l2norm_bwd(dxhat, xhat, norm):
    return (dxhat - xhat * (xhat * dxhat).sum(1, keepdim=True)) / norm[:, None]


def gemm(a, b, out, M, N, K, trans_a, trans_b, alpha=1.0, **kw):
    A = a.T if trans_a else a
    B = b.T if trans_b else b
    out.copy_(alpha * (A @ B))
    return out


def infonce_row_lse(S, diag_off, loss_out=None, loss_scale=0.0, loss_accumulate=False):
    lse = torch.logsumexp(S, dim=1)
    idx = torch.arange(S.shape[0]) + diag_off
    diag = S[torch.arange(S.shape[0]), idx]
    if loss_out is not None:
        v = (lse - diag).sum() * loss_scale
        loss_out.copy_(loss_out + v if loss_accumulate else v)
    return lse, diag


def infonce_grad_inplace(S, diag_off, lse_row, lse_col):
    g = torch.exp(S - lse_row[:, None]) + torch.exp(S - lse_col[None, :])
    idx = torch.arange(S.shape[0]) + diag_off
    g[torch.arange(S.shape[0]), idx] -= 2.0
    S.copy_(g)
    return S


def scale_mask(x, mask_src=None, alpha_dev=None, alpha=1.0, out=None):
    v = x * alpha
    if alpha_dev is not None:
        v = v * alpha_dev
    if mask_src is not None:
        v = v * (mask_src > 0)
    if out is None:
        return v
    out.copy_(v)
    return out
