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


# ---- the adapter (T-ref) step: what Trainer._train_step calls, for the 2-rank CPU rehearsal of its data-parallel form ----------
ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2
AUX_NONE, AUX_RELU_MASK, AUX_GELU_GRAD, AUX_MASK_BITS = 0, 1, 2, 3


def linear_fwd(x, w, bias=None, act=0, residual=None, preact_out=None, out=None):
    y = x @ w.T
    if bias is not None:
        y = y + bias
    if act == ACT_RELU:
        y = torch.relu(y)
    return y


def linear_bwd_data(dy, w, aux=None, auxmode=0, residual=None, out=None, accumulate=False):
    dx = dy @ w
    if auxmode == AUX_RELU_MASK:
        dx = dx * (aux > 0)
    return dx


def linear_bwd_weight(dy, x, dw, accumulate=False):
    g = dy.T @ x
    dw.copy_(dw + g if accumulate else g)
    return dw


def colsum(x, out, alpha=1.0, accumulate=False):
    s = alpha * x.sum(0)
    out.copy_(out + s if accumulate else s)
    return out


def group_mean_fwd(x, G, n):
    return x.reshape(G, n, -1).mean(1)


def group_mean_bwd(dout, G, n):
    return (dout / n).unsqueeze(1).expand(G, n, dout.shape[-1]).reshape(G * n, -1).contiguous()


def pairwise_cosine_fwd(x, y):
    xn, yn = x.norm(dim=1), y.norm(dim=1)
    return (x / xn[:, None]) @ (y / yn[:, None]).T, xn, yn


def pairwise_cosine_bwd(x, y, cosv, dcos, xn, yn, need_dx=True):
    xh, yh = x / xn[:, None], y / yn[:, None]
    dx = (dcos @ yh - (dcos * cosv).sum(1, keepdim=True) * xh) / xn[:, None]
    dy = (dcos.T @ xh - (dcos * cosv).sum(0)[:, None] * yh) / yn[:, None]
    return (dx if need_dx else None), dy


def bce_posneg_fwd_bwd(cosv, labels, diff=True, need_grad=True):
    B, C2 = cosv.shape
    cp, cn = cosv[:, 0::2], cosv[:, 1::2]
    z = cp - cn if diff else cp
    loss = torch.nn.functional.binary_cross_entropy_with_logits(z, labels)
    dz = (torch.sigmoid(z) - labels) / z.numel()
    dcos = torch.zeros_like(cosv)
    dcos[:, 0::2] = dz
    if diff:
        dcos[:, 1::2] = -dz
    return z, dcos, loss


def adam_fused(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    gg = g * grad_scale
    m.mul_(beta1).add_(gg, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(gg, gg, value=1 - beta2)
    mh, vh = m / (1 - beta1 ** step), v / (1 - beta2 ** step)
    p.sub_(lr * mh / (vh.sqrt() + eps))
