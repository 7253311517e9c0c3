"""Tensor-level wrappers over the cxrk C-ABI (one Python function per entry point family).

PyTorch is plumbing here: it owns device memory (caching allocator) and the current HIP stream.  Every function
checks that its operands are CUDA(HIP) fp32 tensors laid out as the kernel expects and raises otherwise; nothing
falls back to torch arithmetic.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import check

ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2
AUX_NONE, AUX_RELU_MASK, AUX_GELU_GRAD, AUX_MASK_BITS = 0, 1, 2, 3


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _chk(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise ValueError(f"{name}: expected a tensor on the GPU (the cxrk path has no CPU fallback), got "
                         f"{type(t).__name__} on {getattr(t, 'device', None)}")
    if t.dtype != dtype:
        raise ValueError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    return t


def _rowmajor2d(t: torch.Tensor, name: str) -> Tuple[torch.Tensor, int]:
    _chk(t, name)
    if t.dim() != 2:
        raise ValueError(f"{name}: expected 2-D, got shape {tuple(t.shape)}")
    if t.shape[1] != 1 and t.stride(1) != 1:
        t = t.contiguous()
    return t, (t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0)))


class Planes:
    """A tensor in split-bf16 storage: `t` is a bf16 tensor [2, *shape] — plane 0 = bf16(x), plane 1 = bf16(x - plane 0) — i.e. the
    same 4 bytes per element as fp32, x ~ hi + lo to ~2^-17 relative.  It is the activation / weight format of the encoders in
    split-bf16 mode: the MFMA mainloops load the planes as they are (no conversion work), see csrc/gemm_loaders.h."""
    __slots__ = ("t",)

    def __init__(self, t: torch.Tensor):
        if t.dtype != torch.bfloat16 or t.dim() < 2 or t.shape[0] != 2 or not t.is_cuda:
            raise ValueError(f"Planes: expected a bf16 GPU tensor [2, ...], got {t.dtype} {tuple(t.shape)} on {t.device}")
        self.t = t

    @staticmethod
    def empty(*shape, device) -> "Planes":
        return Planes(torch.empty((2,) + tuple(shape), dtype=torch.bfloat16, device=device))

    @staticmethod
    def zeros(*shape, device) -> "Planes":
        return Planes(torch.zeros((2,) + tuple(shape), dtype=torch.bfloat16, device=device))

    @property
    def shape(self):
        return self.t.shape[1:]

    @property
    def device(self):
        return self.t.device

    @property
    def plane(self) -> int:
        """distance between the hi and the lo plane, in elements"""
        return self.t.stride(0)

    def ptr(self) -> int:
        return self.t.data_ptr()

    def numel(self) -> int:
        return self.t.numel() // 2

    def view(self, *shape) -> "Planes":
        return Planes(self.t.view((2,) + tuple(shape)))

    def rows(self, sl) -> "Planes":
        """row slice / strided row view of a 2-D planes tensor"""
        return Planes(self.t[:, sl])

    def is_contiguous(self) -> bool:
        return self.t[0].is_contiguous()

    def float(self) -> torch.Tensor:
        """fp32 value hi + lo (host-side consumers, tests)"""
        if self.is_contiguous() and self.numel() % 8 == 0:
            out = torch.empty(tuple(self.shape), dtype=torch.float32, device=self.t.device)
            check(_lib.load().cxrk_merge_planes(self.ptr(), self.plane, self.numel(), _p(out), _stream()), "cxrk_merge_planes")
            return out
        return self.t[0].float() + self.t[1].float()


def split_planes(x: torch.Tensor, out: Optional[Planes] = None) -> Planes:
    """fp32 tensor -> its split-bf16 planes (numel % 8 == 0, contiguous)."""
    _chk(x, "split_planes.x")
    x = x.contiguous()
    if out is None:
        out = Planes.empty(*x.shape, device=x.device)
    check(_lib.load().cxrk_split_planes(_p(x), x.numel(), out.ptr(), out.plane, _stream()), "cxrk_split_planes")
    return out


def _pl2d(t: Planes, name: str):
    """(pointer, row stride, plane stride) of a 2-D planes operand with contiguous columns"""
    if not isinstance(t, Planes) or len(t.shape) != 2:
        raise ValueError(f"{name}: expected a 2-D Planes tensor")
    v = t.t
    if v.shape[2] != 1 and v.stride(2) != 1:
        raise ValueError(f"{name}: planes columns must be contiguous")
    return v.data_ptr(), (v.stride(1) if v.shape[1] > 1 else max(v.shape[2], v.stride(1))), v.stride(0)


class LaunchProfiler:
    """Optional per-launch timing of the MFMA GEMM family with HIP events recorded on the launch stream
    (bench.py's `roofline` object).  Off by default; when on, every GEMM / implicit-GEMM launch is bracketed by two
    events and tagged with its kernel instantiation (the name rocprofv3 reports, scripts/pmc_summary_key.py), its algorithmic
    FLOPs (2*M*N*K) and its algorithmic HBM bytes (every operand and side input read once, every output written once)."""

    PEAK_BF16X3, PEAK_F32, PEAK_HBM = 2500e12 / 3.0, 157.3e12, 8.0e12   # /opt/skills/guides/MI355X_MICROARCH.md (bench.py's constants)

    def __init__(self):
        self.on = False
        self.rows = []  # (key, flops, bytes, start_event, end_event, kernel launches inside the bracket)

    def start(self):
        self.on, self.rows = True, []

    def stop(self):
        self.on = False

    def bracket(self, key: str, flops: float, nbytes: float = 0.0, sub: int = 1):
        """`sub`: kernel launches inside the bracket (a stride-2 data gradient is one launch per output-parity class)"""
        if not self.on:
            return None
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        self.rows.append((key, flops, nbytes, a, b, sub))
        return b

    def summary(self):
        """{kernel: {"launches", "flops", "bytes", "ms"}} — call after torch.cuda.synchronize()."""
        out = {}
        for key, flops, nbytes, a, b, sub in self.rows:
            d = out.setdefault(key, {"launches": 0, "flops": 0.0, "bytes": 0.0, "ms": 0.0, "floor_ms": 0.0})
            d["launches"] += sub
            d["flops"] += flops
            d["bytes"] += nbytes
            d["ms"] += a.elapsed_time(b)
            # the launch's own binding roof: max(FLOPs / MFMA peak of its mainloop, algorithmic bytes / HBM peak)
            peak = self.PEAK_BF16X3 if (key.startswith("gemm_pw") or key.startswith("gemm_x3")) else self.PEAK_F32
            d["floor_ms"] += 1e3 * max(flops / peak, nbytes / self.PEAK_HBM)
        return out


profiler = LaunchProfiler()


def _tile(M: int, N: int) -> str:
    return "4,1" if N <= 64 else ("1,4" if M <= 64 else "2,2")


def _kern(flops: float, exact: bool = False) -> str:
    """Name of the mainloop a launch runs (mirrors launch_gemm's policy in csrc/gemm_core.h)."""
    split = _lib.get_precision() == "split_bf16" and not exact and flops >= 1073741824.0
    return "gemm_x3_kernel" if split else "gemm_f32_kernel"


def _label(la: str, lb: str, tile: str, M: int, N: int, K: int, splitk: int, kind: int, exact: bool = False,
           planes: bool = False) -> str:
    """Profiler key of a launch: mainloop<A loader,B loader,tile>.  Planes operands take the LDS-DMA pipelined kernel: 256x256
    tiles when the library's policy picks them, 256x64 / 64x256 (two blocks per CU) where the output has <= 64 columns / rows,
    128x128 (two blocks per CU) otherwise."""
    if planes:
        if _lib.load().cxrk_gemm_wide_tile(M, N, K, splitk, kind):
            return f"gemm_pw_kernel<Pw256,Dma{la},Dma{lb}>"
        cfg = {"2,2": "Pw128", "4,1": "Pw256x64", "1,4": "Pw64x256"}[tile]
        return f"gemm_pw_kernel<{cfg},Dma{la},Dma{lb}>"
    return f"{_kern(2.0 * M * N * K, exact)}<{la},{lb},{tile}>"


class _Workspace:
    """Grow-on-demand scratch buffer per device and stream; reuse is safe because kernels of one stream run in order."""

    def __init__(self):
        self.bufs = {}

    def get(self, nbytes: int, device) -> torch.Tensor:
        # one buffer per (device, stream): the two encoders of the joint step run on two streams
        key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)
        buf = self.bufs.get(key)
        if buf is None or buf.numel() * 4 < nbytes:
            n = max(int(nbytes * 1.25) // 4 + 64, 1 << 20)
            buf = torch.empty(n, dtype=torch.float32, device=device)
            self.bufs[key] = buf
        return buf


_ws = _Workspace()


def workspace(nbytes: int, device) -> torch.Tensor:
    return _ws.get(nbytes, device)


# ----------------------------------------------------------------------------------------------------------------
# GEMM family
# ----------------------------------------------------------------------------------------------------------------

def gemm(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor, M: int, N: int, K: int, trans_a: bool, trans_b: bool,
         bias=None, residual=None, aux=None, auxmode: int = 0, preact_out=None, act: int = 0, alpha: float = 1.0,
         accumulate: bool = False, splitk: int = 1) -> torch.Tensor:
    lib = _lib.load()
    a, lda = _rowmajor2d(a, "gemm.a")
    b, ldb = _rowmajor2d(b, "gemm.b")
    _chk(out, "gemm.out")
    if out.dim() != 2 or out.stride(1) != 1:
        raise ValueError("gemm.out must be a row-major 2-D tensor")
    ldc = out.stride(0)
    ldr = ldaux = ldc2 = 0
    if residual is not None:
        residual, ldr = _rowmajor2d(residual, "gemm.residual")
    if aux is not None:
        aux, ldaux = _rowmajor2d(aux, "gemm.aux")
    if preact_out is not None:
        _chk(preact_out, "gemm.preact_out")
        ldc2 = preact_out.stride(0)
    if bias is not None:
        _chk(bias, "gemm.bias")
        if bias.numel() != N or not bias.is_contiguous():
            raise ValueError("gemm.bias must be contiguous with N elements")
    ws = None
    wsb = 0
    if splitk > 1:
        wsb = lib.cxrk_gemm_splitk_ws_bytes(M, N, splitk)
        ws = workspace(wsb, out.device)
        wsb = ws.numel() * 4
    plain = bias is None and residual is None and aux is None and preact_out is None and act == 0
    ev = profiler.bracket(_label(f"Dense{'MC' if trans_a else 'KC'}", f"Dense{'KC' if trans_b else 'MC'}", _tile(M, N), M, N, K, splitk,
                                 0 if (plain or splitk > 1) else 3), 2.0 * M * N * K,
                          4.0 * (M * K + N * K + M * N * (1 + (residual is not None) + (aux is not None) + (preact_out is not None)
                                                       + int(bool(accumulate))))) if profiler.on else None
    rc = lib.cxrk_gemm_f32(int(trans_a), int(trans_b), M, N, K, _p(a), lda, _p(b), ldb, _p(out), ldc, _p(bias),
                           _p(residual), ldr, _p(aux), ldaux, auxmode, _p(preact_out), ldc2, act, float(alpha),
                           int(accumulate), int(splitk), _p(ws), wsb, _stream())
    if ev is not None:
        ev.record()
    check(rc, f"cxrk_gemm_f32(M={M},N={N},K={K},tA={trans_a},tB={trans_b})")
    return out


def linear_fwd(x: torch.Tensor, w: torch.Tensor, bias=None, act: int = 0, residual=None, preact_out=None,
               out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y[M,N] = act(x[M,K] @ w[N,K]^T + bias + residual)."""
    M, K = x.shape
    N = w.shape[0]
    if w.shape[1] != K:
        raise ValueError(f"linear_fwd: x is [{M},{K}] but w is {tuple(w.shape)}")
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=x.device)
    return gemm(x, w, out, M, N, K, False, True, bias=bias, residual=residual, preact_out=preact_out, act=act)


def linear_bwd_data(dy: torch.Tensor, w: torch.Tensor, aux=None, auxmode: int = 0, residual=None,
                    out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """dx[M,K] = (dy[M,N] @ w[N,K] + residual) (* mask/gelu'(aux))."""
    M, N = dy.shape
    K = w.shape[1]
    if out is None:
        out = torch.empty(M, K, dtype=torch.float32, device=dy.device)
    return gemm(dy, w, out, M, K, N, False, False, aux=aux, auxmode=auxmode, residual=residual, accumulate=accumulate)


def _wgrad_splitk(n_out: int, n_in: int, rows: int, planes: bool = False) -> int:
    return int(_lib.load().cxrk_gemm_wgrad_splitk(n_out, n_in, rows, int(planes)))


def linear_bwd_weight(dy: torch.Tensor, x: torch.Tensor, dw: torch.Tensor, accumulate: bool = False) -> torch.Tensor:
    """dw[N,K] (+)= dy[M,N]^T @ x[M,K]  (deterministic split-K over M)."""
    M, N = dy.shape
    K = x.shape[1]
    sk = _wgrad_splitk(N, K, M)
    if sk > 1:
        return gemm(dy, x, dw, N, K, M, True, False, accumulate=accumulate, splitk=sk)
    return gemm(dy, x, dw, N, K, M, True, False, accumulate=accumulate)


def gemm_pl(a: Planes, b: Planes, M: int, N: int, K: int, trans_a: bool, trans_b: bool, out=None, bias=None, residual=None,
            aux=None, auxmode: int = 0, maskin=None, maskout=None, preact_out=None, act: int = 0, alpha: float = 1.0,
            accumulate: bool = False, splitk: int = 1, colsum=None, colsum_accumulate: bool = False):
    """The GEMM family on planes operands.  `out`: fp32 tensor or Planes (row-major [M, N]); `residual`: fp32 tensor or Planes;
    `aux` (auxmode 2): fp32 pre-activation for gelu'; `maskin` (auxmode 3) / `maskout` (with ReLU): uint8 bit masks [M, N/8];
    `colsum` ([N], optional, splitk == 1): receives (accumulates) the column sums of the stored output, reduced in the epilogue."""
    lib = _lib.load()
    ap, lda, apl = _pl2d(a, "gemm_pl.a")
    bp, ldb, bpl = _pl2d(b, "gemm_pl.b")
    C = Cp = None
    cpl = 0
    if isinstance(out, Planes):
        Cp, ldc, cpl = _pl2d(out, "gemm_pl.out")
    else:
        _chk(out, "gemm_pl.out")
        if out.dim() != 2 or out.stride(1) != 1:
            raise ValueError("gemm_pl.out must be a row-major 2-D tensor")
        C, ldc = out.data_ptr(), out.stride(0)
    R = Rp = None
    ldr = rpl = 0
    if isinstance(residual, Planes):
        Rp, ldr, rpl = _pl2d(residual, "gemm_pl.residual")
    elif residual is not None:
        residual, ldr = _rowmajor2d(residual, "gemm_pl.residual")
        R = residual.data_ptr()
    ldaux = ldc2 = ldmi = ldmo = 0
    if aux is not None:
        aux, ldaux = _rowmajor2d(aux, "gemm_pl.aux")
    if preact_out is not None:
        _chk(preact_out, "gemm_pl.preact_out")
        ldc2 = preact_out.stride(0)
    if maskin is not None:
        _chk(maskin, "gemm_pl.maskin", torch.uint8)
        ldmi = maskin.stride(0)
    if maskout is not None:
        _chk(maskout, "gemm_pl.maskout", torch.uint8)
        ldmo = maskout.stride(0)
    if bias is not None:
        _chk(bias, "gemm_pl.bias")
        if bias.numel() != N or not bias.is_contiguous():
            raise ValueError("gemm_pl.bias must be contiguous with N elements")
    ws, wsb = None, 0
    if splitk > 1:
        wsb = lib.cxrk_gemm_splitk_ws_bytes(M, N, splitk)
        ws = workspace(wsb, a.device)
        wsb = ws.numel() * 4
    elif colsum is not None:
        _chk(colsum, "gemm_pl.colsum")
        if colsum.numel() != N or not colsum.is_contiguous():
            raise ValueError("gemm_pl.colsum must be contiguous with N elements")
        ws = workspace(lib.cxrk_gemm_pl_colsum_ws_bytes(M, N), a.device)
        wsb = ws.numel() * 4
    if colsum is not None and splitk > 1:
        raise ValueError("gemm_pl: fused column sums need splitk == 1")
    plain = bias is None and residual is None and aux is None and maskin is None and preact_out is None and act == 0
    ev = profiler.bracket(_label(f"Dense{'MC' if trans_a else 'KC'}", f"Dense{'KC' if trans_b else 'MC'}", _tile(M, N), M, N, K, splitk,
                                 0 if (plain or splitk > 1) else 3, planes=True), 2.0 * M * N * K,
                          4.0 * (M * K + N * K + M * N * (1 + (residual is not None) + (aux is not None) + (preact_out is not None)
                                                       + int(bool(accumulate)))) + M * N / 8.0 * ((maskin is not None) + (maskout is not None))) if profiler.on else None
    rc = lib.cxrk_gemm_pl(int(trans_a), int(trans_b), M, N, K, ap, lda, apl, bp, ldb, bpl, C, Cp, ldc, cpl, _p(bias), R, Rp, ldr, rpl,
                          _p(aux), ldaux, auxmode, _p(maskin), ldmi, _p(maskout), ldmo, _p(preact_out), ldc2, act, float(alpha),
                          int(accumulate), int(splitk), _p(colsum), int(colsum_accumulate), _p(ws), wsb, _stream())
    if ev is not None:
        ev.record()
    check(rc, f"cxrk_gemm_pl(M={M},N={N},K={K},tA={trans_a},tB={trans_b})")
    return out


def linear_fwd_pl(x: Planes, w: Planes, bias=None, act: int = 0, residual=None, preact_out=None, out=None, out_planes: bool = False,
                  maskout=None):
    """y[M,N] = act(x[M,K] @ w[N,K]^T + bias + residual); y as fp32 (default) or Planes."""
    M, K = x.shape
    N = w.shape[0]
    if w.shape[1] != K:
        raise ValueError(f"linear_fwd_pl: x is [{M},{K}] but w is {tuple(w.shape)}")
    if out is None:
        out = Planes.empty(M, N, device=x.device) if out_planes else torch.empty(M, N, dtype=torch.float32, device=x.device)
    return gemm_pl(x, w, M, N, K, False, True, out=out, bias=bias, residual=residual, preact_out=preact_out, act=act, maskout=maskout)


def linear_bwd_data_pl(dy: Planes, w: Planes, aux=None, auxmode: int = 0, maskin=None, residual=None, out=None,
                       out_planes: bool = False, accumulate: bool = False, colsum=None, colsum_accumulate: bool = False):
    """dx[M,K] = (dy[M,N] @ w[N,K] + residual) (* gelu'(aux) | * mask bits); `colsum` ([K], optional) receives the column sums of dx
    (the bias gradient of the layer that produced this layer's input) from the same epilogue."""
    M, N = dy.shape
    K = w.shape[1]
    if out is None:
        out = Planes.empty(M, K, device=dy.device) if out_planes else torch.empty(M, K, dtype=torch.float32, device=dy.device)
    if maskin is not None:
        auxmode = AUX_MASK_BITS
    return gemm_pl(dy, w, M, K, N, False, False, out=out, aux=aux, auxmode=auxmode, maskin=maskin, residual=residual, accumulate=accumulate,
                   colsum=colsum, colsum_accumulate=colsum_accumulate)


def linear_bwd_weight_pl(dy: Planes, x: Planes, dw: torch.Tensor, accumulate: bool = False) -> torch.Tensor:
    """dw[N,K] (+)= dy[M,N]^T @ x[M,K]  (deterministic split-K over M); dw fp32."""
    M, N = dy.shape
    K = x.shape[1]
    sk = _wgrad_splitk(N, K, M, planes=True)
    return gemm_pl(dy, x, N, K, M, True, False, out=dw, accumulate=accumulate, splitk=sk if sk > 1 else 1)


def colsum_pl(x: Planes, out: torch.Tensor, alpha: float = 1.0, accumulate: bool = False) -> torch.Tensor:
    lib = _lib.load()
    xp, ldx, xpl = _pl2d(x, "colsum_pl.x")
    _chk(out, "colsum_pl.out")
    rows, cols = x.shape
    ws = workspace(lib.cxrk_colsum_ws_bytes(rows, cols), x.device)
    check(lib.cxrk_colsum_pl(xp, ldx, xpl, rows, cols, _p(out), float(alpha), int(accumulate), _p(ws), ws.numel() * 4, _stream()),
          "cxrk_colsum_pl")
    return out


def colsum(x: torch.Tensor, out: torch.Tensor, alpha: float = 1.0, accumulate: bool = False) -> torch.Tensor:
    if isinstance(x, Planes):
        return colsum_pl(x, out, alpha, accumulate)
    lib = _lib.load()
    x, ldx = _rowmajor2d(x, "colsum.x")
    _chk(out, "colsum.out")
    rows, cols = x.shape
    wsb = lib.cxrk_colsum_ws_bytes(rows, cols)
    ws = workspace(wsb, x.device)
    check(lib.cxrk_colsum(_p(x), ldx, rows, cols, _p(out), float(alpha), int(accumulate), _p(ws), ws.numel() * 4,
                          _stream()), "cxrk_colsum")
    return out


def colvar(x, mean: torch.Tensor, out: torch.Tensor, alpha: float = 1.0) -> torch.Tensor:
    """out[c] = alpha * sum_r (x[r, c] - mean[c])^2 for a 2-D fp32 tensor or Planes (second pass of a two-pass batch variance)."""
    lib = _lib.load()
    if isinstance(x, Planes):
        xp, ldx, xpl = _pl2d(x, "colvar.x")
        rows, cols = x.shape
    else:
        x, ldx = _rowmajor2d(x, "colvar.x")
        xp, xpl = x.data_ptr(), 0
        rows, cols = x.shape
    _chk(mean, "colvar.mean")
    _chk(out, "colvar.out")
    if mean.numel() != cols or out.numel() != cols or not mean.is_contiguous() or not out.is_contiguous():
        raise ValueError(f"colvar: mean / out must be contiguous with {cols} elements")
    ws = workspace(lib.cxrk_colsum_ws_bytes(rows, cols), out.device)
    check(lib.cxrk_colvar(xp, ldx, xpl, rows, cols, _p(mean), _p(out), float(alpha), _p(ws), ws.numel() * 4, _stream()), "cxrk_colvar")
    return out


# ---- train-mode BatchNorm2d around the convolutions (csrc/bn_train.hip) ------------------------------------------------------------

def _fmt(t, name: str):
    """(pointer, plane stride) of a contiguous fp32 tensor (plane 0) or Planes tensor"""
    if isinstance(t, Planes):
        if not t.is_contiguous():
            raise ValueError(f"{name}: planes tensor must be contiguous")
        return t.ptr(), t.plane
    _chk(t, name)
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    return t.data_ptr(), 0


def _vec(t: torch.Tensor, C: int, name: str) -> torch.Tensor:
    _chk(t, name)
    if t.numel() != C or not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous vector of {C} elements, got {tuple(t.shape)}")
    return t


def colstats(x, unbiased: bool = False):
    """(mean[C], var[C]) over the rows of x ([..., C] fp32 or Planes, C % 8 == 0) in one pass; var biased (1 / n) or unbiased (1 / (n - 1))."""
    lib = _lib.load()
    C = x.shape[-1]
    rows = x.numel() // C
    xp, xpl = _fmt(x, "colstats.x")
    mean = torch.empty(C, dtype=torch.float32, device=x.device)
    var = torch.empty(C, dtype=torch.float32, device=x.device)
    ws = workspace(2 * lib.cxrk_coldot_ws_bytes(rows, C), x.device)
    check(lib.cxrk_colstats(xp, xpl, rows, C, _p(mean), _p(var), 1.0 / max(1, rows - 1) if unbiased else 1.0 / rows, _p(ws), ws.numel() * 4,
                            _stream()), f"cxrk_colstats(rows={rows},C={C})")
    return mean, var


def bn_train_fwd_coeffs(mean, var, gamma, beta, eps: float, n: int, momentum: float = 0.0, rmean=None, rvar=None):
    """(scale, shift, rstd) of a train-mode BatchNorm from its batch statistics (var: biased); updates the running statistics in place
    (r = (1 - momentum) r + momentum stat, unbiased variance) when given."""
    lib = _lib.load()
    C = mean.numel()
    for nme, t in (("mean", mean), ("var", var), ("gamma", gamma), ("beta", beta)):
        _vec(t, C, "bn_train." + nme)
    scale, shift, rstd = (torch.empty(C, dtype=torch.float32, device=mean.device) for _ in range(3))
    if rmean is not None:
        _vec(rmean, C, "bn_train.running_mean")
        _vec(rvar, C, "bn_train.running_var")
    check(lib.cxrk_bn_train_fwd_coeffs(_p(mean), _p(var), _p(gamma), _p(beta), float(eps), int(n), float(momentum), _p(scale), _p(shift),
                                       _p(rstd), _p(rmean), _p(rvar), C, _stream()), "cxrk_bn_train_fwd_coeffs")
    return scale, shift, rstd


def bn_apply(z, scale, shift, residual=None, relu: bool = True, want_mask: bool = False):
    """y = relu?(z * scale + shift + residual?) in z's storage format ([..., C], C % 8 == 0); mask: ReLU decision bits [rows, C / 8]."""
    lib = _lib.load()
    C = z.shape[-1]
    rows = z.numel() // C
    zp, zpl = _fmt(z, "bn_apply.z")
    if isinstance(z, Planes):
        y = Planes.empty(*z.shape, device=z.device)
    else:
        y = torch.empty_like(z)
    yp, ypl = _fmt(y, "bn_apply.y")
    rp, rpl = _fmt(residual, "bn_apply.residual") if residual is not None else (None, 0)
    if residual is not None and tuple(residual.shape) != tuple(z.shape):
        raise ValueError(f"bn_apply: residual {tuple(residual.shape)} vs z {tuple(z.shape)}")
    mask = torch.empty(rows, C // 8, dtype=torch.uint8, device=z.device) if (want_mask and relu) else None
    check(lib.cxrk_bn_apply(zp, zpl, _p(_vec(scale, C, "bn_apply.scale")), _p(_vec(shift, C, "bn_apply.shift")), rp, rpl, yp, ypl, _p(mask),
                            rows, C, int(relu), _stream()), f"cxrk_bn_apply(rows={rows},C={C})")
    return y, mask


def coldot(a, b, bshift=None) -> torch.Tensor:
    """out[c] = sum_rows a[r, c] * (b[r, c] - bshift[c]); a, b: [..., C] fp32 or Planes (formats may differ); bshift: fp32 [C] or None."""
    lib = _lib.load()
    C = a.shape[-1]
    rows = a.numel() // C
    if b.shape[-1] != C or b.numel() != a.numel():
        raise ValueError(f"coldot: {tuple(a.shape)} vs {tuple(b.shape)}")
    ap, apl = _fmt(a, "coldot.a")
    bp, bpl = _fmt(b, "coldot.b")
    out = torch.empty(C, dtype=torch.float32, device=a.device)
    ws = workspace(lib.cxrk_coldot_ws_bytes(rows, C), a.device)
    check(lib.cxrk_coldot(ap, apl, bp, bpl, _p(_vec(bshift, C, "coldot.bshift")) if bshift is not None else None, rows, C, _p(out), _p(ws), ws.numel() * 4, _stream()), "cxrk_coldot")
    return out


def bn_train_bwd_coeffs(gamma, mean, rstd, sumdy, dot, n: int, dgamma, dbeta, accumulate: bool):
    """dot = sum dy (z - mean) (coldot(dy, z, mean)); dgamma (+)= rstd dot, dbeta (+)= sumdy; returns (A, B, Cc), dz = A dy + B + Cc z."""
    lib = _lib.load()
    C = gamma.numel()
    for nme, t in (("gamma", gamma), ("mean", mean), ("rstd", rstd), ("sumdy", sumdy), ("dot", dot), ("dgamma", dgamma), ("dbeta", dbeta)):
        _vec(t, C, "bn_train_bwd." + nme)
    A, B, Cc = (torch.empty(C, dtype=torch.float32, device=gamma.device) for _ in range(3))
    check(lib.cxrk_bn_train_bwd_coeffs(_p(gamma), _p(mean), _p(rstd), _p(sumdy), _p(dot), int(n), _p(A), _p(B), _p(Cc), _p(dgamma), _p(dbeta),
                                       int(accumulate), C, _stream()), "cxrk_bn_train_bwd_coeffs")
    return A, B, Cc


def bn_train_dz(dy, z, A, B, Cc):
    """dz = A dy + B + Cc z (per channel), in dy's storage format."""
    lib = _lib.load()
    C = dy.shape[-1]
    rows = dy.numel() // C
    dp, dpl = _fmt(dy, "bn_train_dz.dy")
    zp, zpl = _fmt(z, "bn_train_dz.z")
    if z.numel() != dy.numel():
        raise ValueError(f"bn_train_dz: {tuple(dy.shape)} vs {tuple(z.shape)}")
    dz = Planes.empty(*dy.shape, device=dy.device) if isinstance(dy, Planes) else torch.empty_like(dy)
    op, opl = _fmt(dz, "bn_train_dz.dz")
    check(lib.cxrk_bn_train_dz(dp, dpl, zp, zpl, _p(_vec(A, C, "A")), _p(_vec(B, C, "B")), _p(_vec(Cc, C, "Cc")), op, opl, rows, C, _stream()),
          "cxrk_bn_train_dz")
    return dz


# ----------------------------------------------------------------------------------------------------------------
# image encoder pieces (NHWC)
# ----------------------------------------------------------------------------------------------------------------

def bn_fold(w, gamma, beta, rmean, rvar, eps, Ko, taps, C, Cpad, w_scaled, scale, shift, rstd):
    lib = _lib.load()
    for n, t in (("w", w), ("gamma", gamma), ("beta", beta), ("rmean", rmean), ("rvar", rvar)):
        _chk(t, "bn_fold." + n)
    check(lib.cxrk_bn_fold(_p(w), _p(gamma), _p(beta), _p(rmean), _p(rvar), float(eps), Ko, taps, C, Cpad,
                           _p(w_scaled), _p(scale), _p(shift), _p(rstd), _stream()), "cxrk_bn_fold")


def conv_fwd(x, w_scaled, shift, residual, y, N, H, W, C, Ko, R, S, stride, pad, relu):
    lib = _lib.load()
    ev = None
    if profiler.on:
        Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1
        fl = 2.0 * N * Ho * Wo * Ko * R * S * C
        ev = profiler.bracket(_label("ConvIm2colKC", "DenseKC", "4,1" if Ko <= 64 else "2,2", N * Ho * Wo, Ko, R * S * C, 1,
                                     1 if C % 32 == 0 else 3), fl,
                              4.0 * (N * H * W * C + Ko * R * S * C + N * Ho * Wo * Ko * (1 + (residual is not None))))
    rc = lib.cxrk_conv_bn_act_fwd(_p(_chk(x, "conv.x")), _p(w_scaled), _p(shift), _p(residual), _p(y), N, H, W, C, Ko,
                                  R, S, stride, pad, int(relu), _stream())
    if ev is not None:
        ev.record()
    check(rc, f"cxrk_conv_bn_act_fwd(N={N},H={H},W={W},C={C},Ko={Ko},R={R},s={stride})")
    return y


def _conv_out(H, W, R, S, stride, pad):
    return (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1


def _dgrad_label(N, H, W, Ko, R, S, stride):
    """(A loader, B loader, GEMM rows, GEMM K, kernel launches) of a data gradient as the library runs it: stride 1 = one launch over
    all pixels; stride 2 = one launch per output-parity class that a tap reaches (4 for a 3x3, 1 for a 1x1), each over a quarter of the
    pixels with that class's taps (csrc/conv_dgrad.hip) — a different kernel instantiation, labelled as such."""
    if stride == 1:
        return "ConvDgradKC", "ConvFilterMC", N * H * W, R * S * Ko, 1
    classes = 4 if R > 1 else 1
    return "ConvDgradS2KC", "ConvFilterS2MC", N * ((H + 1) // 2) * ((W + 1) // 2), Ko * max(1, (R * S) // 4), classes


def conv_bwd_data(dy, w_scaled, residual, relu_src, dx, N, H, W, C, Ko, R, S, stride, pad, sums=None):
    """dx = (relu_src > 0) * (conv^T(dy, w_scaled) + residual); `sums` ([C], optional) receives the column sums of dx (the BN beta
    gradient of the unit that produced relu_src), reduced in the same epilogue."""
    lib = _lib.load()
    ws = workspace(lib.cxrk_conv_bwd_data_colsum_ws_bytes(N, H, W, C, stride), dy.device) if sums is not None else None
    ev = None
    if profiler.on:  # algorithmic FLOPs of a data gradient = those of the forward conv (stride-2 zero taps are waste)
        Ho, Wo = _conv_out(H, W, R, S, stride, pad)
        fl = 2.0 * N * Ho * Wo * Ko * R * S * C
        la, lb, Ml, Kl, sub = _dgrad_label(N, H, W, Ko, R, S, stride)
        ev = profiler.bracket(_label(la, lb, "4,1" if C <= 64 else "2,2", Ml, C, Kl, 1, 2), fl,
                              4.0 * (N * Ho * Wo * Ko + Ko * R * S * C + N * H * W * C * (1 + (residual is not None) + (relu_src is not None))), sub)
    rc = lib.cxrk_conv_bn_act_bwd_data(_p(_chk(dy, "conv.dy")), _p(w_scaled), _p(residual), _p(relu_src), _p(dx), N, H,
                                       W, C, Ko, R, S, stride, pad, _p(sums), _p(ws), ws.numel() * 4 if ws is not None else 0, _stream())
    if ev is not None:
        ev.record()
    check(rc, f"cxrk_conv_bn_act_bwd_data(N={N},H={H},W={W},C={C},Ko={Ko},R={R},s={stride})")
    return dx


def conv_bwd_params(x, dy, w, scale, rstd, rmean, sumdy, dw, dgamma, dbeta, accumulate, N, H, W, C, Cpad, Ko, R, S, stride, pad):
    """dW = scale * wgrad(x, dy); dbeta = sumdy; dgamma = rstd * (<w, wgrad> - mean * sumdy)  (csrc/conv.hip)."""
    lib = _lib.load()
    wsb = lib.cxrk_conv_wgrad_ws_bytes(N, H, W, Cpad, Ko, R, S, stride, pad)
    ws = workspace(wsb, x.device)
    ev = None
    if profiler.on:
        Ho, Wo = _conv_out(H, W, R, S, stride, pad)
        fl = 2.0 * N * Ho * Wo * Ko * R * S * Cpad
        sk = lib.cxrk_gemm_wgrad_splitk(Ko, R * S * Cpad, N * Ho * Wo, 0)
        ev = profiler.bracket(_label("DenseMC", "ConvIm2colMC", "1,4" if Ko <= 64 else "2,2", Ko, R * S * Cpad, N * Ho * Wo, sk, 0,
                                     False), fl, 4.0 * (N * H * W * Cpad + N * Ho * Wo * Ko + Ko * R * S * Cpad))
    check(lib.cxrk_conv_bn_act_bwd_params(_p(_chk(x, "conv.x")), _p(_chk(dy, "conv.dy")), _p(w), _p(scale), _p(rstd),
                                          _p(rmean), _p(sumdy), _p(dw), _p(dgamma), _p(dbeta), int(accumulate), N, H, W,
                                          C, Cpad, Ko, R, S, stride, pad, _p(ws), ws.numel() * 4, _stream()),
          f"cxrk_conv_bn_act_bwd_params(N={N},H={H},W={W},C={C},Ko={Ko},R={R},s={stride})")
    if ev is not None:
        ev.record()


# ---- planes storage (split-bf16 mode of the image encoder) -------------------------------------------------------------------

def bn_fold_pl(w, gamma, beta, rmean, rvar, eps, Ko, taps, C, Cpad, w_scaled: Planes, scale, shift, rstd):
    lib = _lib.load()
    for n, t in (("w", w), ("gamma", gamma), ("beta", beta), ("rmean", rmean), ("rvar", rvar)):
        _chk(t, "bn_fold." + n)
    check(lib.cxrk_bn_fold_pl(_p(w), _p(gamma), _p(beta), _p(rmean), _p(rvar), float(eps), Ko, taps, C, Cpad, w_scaled.ptr(),
                              w_scaled.plane, _p(scale), _p(shift), _p(rstd), _stream()), "cxrk_bn_fold_pl")


def conv_fwd_pl(x, w_scaled, shift, residual: Optional[Planes], y: Planes, maskout, N, H, W, C, Ko, R, S, stride, pad, relu):
    """x / w_scaled: both Planes, or both fp32 tensors (the stem); y (and residual) Planes; maskout: uint8 [N*Ho*Wo, Ko/8] or None."""
    lib = _lib.load()
    in_planes = isinstance(x, Planes)
    if in_planes != isinstance(w_scaled, Planes):
        raise ValueError("conv_fwd_pl: x and w_scaled must share the storage format")
    ev = None
    if profiler.on:
        Ho, Wo = _conv_out(H, W, R, S, stride, pad)
        fl = 2.0 * N * Ho * Wo * Ko * R * S * C
        nb = 4.0 * (N * H * W * C + Ko * R * S * C + N * Ho * Wo * Ko * (1 + (residual is not None))) + (N * Ho * Wo * Ko / 8.0 if maskout is not None else 0.0)
        if in_planes:
            key = _label("ConvIm2colKC", "DenseKC", "4,1" if Ko <= 64 else "2,2", N * Ho * Wo, Ko, R * S * C, 1, 1, planes=True)
        else:
            key = _label("ConvIm2colKC", "DenseKC", "4,1" if Ko <= 64 else "2,2", N * Ho * Wo, Ko, R * S * C, 1, 3)
        ev = profiler.bracket(key, fl, nb)
    if maskout is not None:
        _chk(maskout, "conv.maskout", torch.uint8)
    xp, xpl = (x.ptr(), x.plane) if in_planes else (_p(_chk(x, "conv.x")), 0)
    wp, wpl = (w_scaled.ptr(), w_scaled.plane) if in_planes else (_p(_chk(w_scaled, "conv.w")), 0)
    rc = lib.cxrk_conv_bn_act_fwd_pl(xp, xpl, wp, wpl, int(in_planes), _p(shift), residual.ptr() if residual is not None else None,
                                     residual.plane if residual is not None else 0, y.ptr(), y.plane, _p(maskout), N, H, W, C, Ko, R, S,
                                     stride, pad, int(relu), _stream())
    if ev is not None:
        ev.record()
    check(rc, f"cxrk_conv_bn_act_fwd_pl(N={N},H={H},W={W},C={C},Ko={Ko},R={R},s={stride})")
    return y


def conv_bwd_data_pl(dy: Planes, w_scaled: Planes, residual: Optional[Planes], maskin, dx: Planes, N, H, W, C, Ko, R, S, stride, pad,
                     sums=None, residual_s2: bool = False):
    """dx = mask(conv^T(dy, w_scaled) + residual) on planes operands (+ column sums of dx).  residual_s2: `residual` is COMPACT,
    [N, (H+1)//2, (W+1)//2, C] = the identity-branch gradient at the even (h, w) pixels only (what a 1x1 / stride-2 projection
    shortcut's data gradient is: see `conv1x1_s2_bwd_data_compact_pl`); the other pixels receive nothing."""
    lib = _lib.load()
    if residual_s2:
        if residual is None or tuple(residual.shape) != (N, (H + 1) // 2, (W + 1) // 2, C):
            raise ValueError(f"conv_bwd_data_pl: compact residual must be [N, (H+1)//2, (W+1)//2, C], got {None if residual is None else tuple(residual.shape)}")
    ws = workspace(lib.cxrk_conv_bwd_data_colsum_ws_bytes(N, H, W, C, stride), dy.device) if sums is not None else None
    ev = None
    if profiler.on:
        Ho, Wo = _conv_out(H, W, R, S, stride, pad)
        fl = 2.0 * N * Ho * Wo * Ko * R * S * C
        la, lb, Ml, Kl, sub = _dgrad_label(N, H, W, Ko, R, S, stride)
        ev = profiler.bracket(_label(la, lb, "4,1" if C <= 64 else "2,2", Ml, C, Kl, 1, 2, planes=True), fl,
                              4.0 * (N * Ho * Wo * Ko + Ko * R * S * C + N * H * W * C * (1 + (0.0 if residual is None else 0.25 if residual_s2 else 1.0)))
                              + (N * H * W * C / 8.0 if maskin is not None else 0.0), sub)
    fn = lib.cxrk_conv_bn_act_bwd_data_pl_s2res if residual_s2 else lib.cxrk_conv_bn_act_bwd_data_pl
    rc = fn(dy.ptr(), dy.plane, w_scaled.ptr(), w_scaled.plane, residual.ptr() if residual is not None else None,
                                          residual.plane if residual is not None else 0, _p(maskin), dx.ptr(), dx.plane, N, H, W, C, Ko, R, S,
                                          stride, pad, _p(sums), _p(ws), ws.numel() * 4 if ws is not None else 0, _stream())
    if ev is not None:
        ev.record()
    check(rc, f"cxrk_conv_bn_act_bwd_data_pl{'_s2res' if residual_s2 else ''}(N={N},H={H},W={W},C={C},Ko={Ko},R={R},s={stride})")
    return dx


def conv1x1_s2_bwd_data_compact_pl(dy: Planes, w_scaled: Planes, N, H, W, C, Ko) -> Optional[Planes]:
    """Data gradient of a 1x1 / stride-2 / pad-0 convolution in COMPACT form: only the input pixels with even (h, w) receive
    anything, dx[n, 2a, 2b, :] = dy[n, a, b, :] @ w — a dense [N*Ho*Wo, Ko] x [Ko, C] product, returned as [N, Ho, Wo, C] for
    `conv_bwd_data_pl(..., residual_s2=True)`.  None when the compact tensor is too large for that epilogue's 32-bit offsets
    (the caller then takes the dense form)."""
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    if N * Ho * Wo * C * 2 >= (1 << 31):
        return None
    if tuple(dy.shape) not in ((N, Ho, Wo, Ko), (N * Ho * Wo, Ko)) or tuple(w_scaled.shape) != (Ko, C):
        raise ValueError(f"conv1x1_s2_bwd_data_compact_pl: dy {tuple(dy.shape)}, w {tuple(w_scaled.shape)} for N={N},H={H},W={W},C={C},Ko={Ko}")
    out = Planes.empty(N * Ho * Wo, C, device=dy.device)
    linear_bwd_data_pl(dy.view(N * Ho * Wo, Ko), w_scaled, out=out)
    return out.view(N, Ho, Wo, C)


def conv_bwd_params_pl(x: Planes, dy: Planes, w, scale, rstd, rmean, sumdy, dw, dgamma, dbeta, accumulate, N, H, W, C, Ko, R, S, stride, pad):
    lib = _lib.load()
    ws = workspace(lib.cxrk_conv_wgrad_ws_bytes(N, H, W, C, Ko, R, S, stride, pad), x.device)
    ev = None
    if profiler.on:
        Ho, Wo = _conv_out(H, W, R, S, stride, pad)
        fl = 2.0 * N * Ho * Wo * Ko * R * S * C
        sk = lib.cxrk_gemm_wgrad_splitk(Ko, R * S * C, N * Ho * Wo, 1)
        ev = profiler.bracket(_label("DenseMC", "ConvIm2colMC", "1,4" if Ko <= 64 else "2,2", Ko, R * S * C, N * Ho * Wo, sk, 0, planes=True),
                              fl, 4.0 * (N * H * W * C + N * Ho * Wo * Ko + Ko * R * S * C))
    check(lib.cxrk_conv_bn_act_bwd_params_pl(x.ptr(), x.plane, dy.ptr(), dy.plane, _p(w), _p(scale), _p(rstd), _p(rmean), _p(sumdy),
                                             _p(dw), _p(dgamma), _p(dbeta), int(accumulate), N, H, W, C, Ko, R, S, stride, pad, _p(ws),
                                             ws.numel() * 4, _stream()),
          f"cxrk_conv_bn_act_bwd_params_pl(N={N},H={H},W={W},C={C},Ko={Ko},R={R},s={stride})")
    if ev is not None:
        ev.record()


def maxpool_fwd_pl(x: Planes):
    lib = _lib.load()
    N, H, W, C = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = Planes.empty(N, Ho, Wo, C, device=x.device)
    idx = torch.empty(N, Ho, Wo, C, dtype=torch.uint8, device=x.device)
    check(lib.cxrk_maxpool_fwd_pl(x.ptr(), x.plane, y.ptr(), y.plane, _p(idx), N, H, W, C, _stream()), "cxrk_maxpool_fwd_pl")
    return y, idx


def maxpool_bwd_pl(dy: Planes, idx, pooled: Planes, H: int, W: int) -> torch.Tensor:
    """fp32 gradient w.r.t. the (post-ReLU) stem output, masked by the stem ReLU through the sign of `pooled`."""
    lib = _lib.load()
    N, _, _, C = pooled.shape
    dx = torch.empty(N, H, W, C, dtype=torch.float32, device=dy.device)
    check(lib.cxrk_maxpool_bwd_pl(dy.ptr(), dy.plane, _p(idx), pooled.ptr(), _p(dx), N, H, W, C, _stream()), "cxrk_maxpool_bwd_pl")
    return dx


def spatial_mean_bwd_pl(dy: torch.Tensor, Pn: int, add: Optional[torch.Tensor] = None) -> Planes:
    lib = _lib.load()
    N, C = dy.shape
    dx = Planes.empty(N, Pn, C, device=dy.device)
    if add is not None:
        add = _chk(add, "spatial_mean.add").contiguous()
    check(lib.cxrk_spatial_mean_bwd_pl(_p(_chk(dy.contiguous(), "spatial_mean.dy")), _p(add), dx.ptr(), dx.plane, N, Pn, C, _stream()),
          "cxrk_spatial_mean_bwd_pl")
    return dx


def unpack_mask(mask: torch.Tensor, C: int) -> torch.Tensor:
    """ReLU decision bits [rows, C/8] (bit c % 8 of byte c / 8) -> bool [rows, C] on the host (tests only)."""
    m = mask.cpu().reshape(-1, C // 8).to(torch.int32)
    bits = (m.unsqueeze(-1) >> torch.arange(8, dtype=torch.int32)) & 1
    return bits.reshape(-1, C).bool()


def nchw_to_nhwc(x: torch.Tensor, cpad: int) -> torch.Tensor:
    lib = _lib.load()
    _chk(x, "nchw_to_nhwc.x")
    x = x.contiguous()
    N, C, H, W = x.shape
    y = torch.empty(N, H, W, cpad, dtype=torch.float32, device=x.device)
    check(lib.cxrk_nchw_to_nhwc(_p(x), _p(y), N, C, H, W, cpad, _stream()), "cxrk_nchw_to_nhwc")
    return y


def nhwc_to_nchw(x: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    _chk(x, "nhwc_to_nchw.x")
    N, H, W, C = x.shape
    y = torch.empty(N, C, H, W, dtype=torch.float32, device=x.device)
    check(lib.cxrk_nhwc_to_nchw(_p(x), _p(y), N, C, H, W, _stream()), "cxrk_nhwc_to_nchw")
    return y


def maxpool_fwd(x: torch.Tensor):
    lib = _lib.load()
    N, H, W, C = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty(N, Ho, Wo, C, dtype=torch.float32, device=x.device)
    idx = torch.empty(N, Ho, Wo, C, dtype=torch.uint8, device=x.device)
    check(lib.cxrk_maxpool_fwd(_p(_chk(x, "maxpool.x")), _p(y), _p(idx), N, H, W, C, _stream()), "cxrk_maxpool_fwd")
    return y, idx


def maxpool_bwd(dy, idx, x, relu_mask: bool):
    lib = _lib.load()
    N, H, W, C = x.shape
    dx = torch.empty_like(x)
    check(lib.cxrk_maxpool_bwd(_p(_chk(dy, "maxpool.dy")), _p(idx), _p(x), _p(dx), N, H, W, C, int(relu_mask), _stream()),
          "cxrk_maxpool_bwd")
    return dx


def spatial_mean_fwd(x: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    N, Pn, C = x.shape
    y = torch.empty(N, C, dtype=torch.float32, device=x.device)
    check(lib.cxrk_spatial_mean_fwd(_p(_chk(x, "spatial_mean.x")), _p(y), N, Pn, C, _stream()), "cxrk_spatial_mean_fwd")
    return y


def spatial_mean_bwd(dy: torch.Tensor, Pn: int) -> torch.Tensor:
    lib = _lib.load()
    N, C = dy.shape
    dx = torch.empty(N, Pn, C, dtype=torch.float32, device=dy.device)
    check(lib.cxrk_spatial_mean_bwd(_p(_chk(dy.contiguous(), "spatial_mean.dy")), _p(dx), N, Pn, C, _stream()),
          "cxrk_spatial_mean_bwd")
    return dx


# ----------------------------------------------------------------------------------------------------------------
# text encoder pieces
# ----------------------------------------------------------------------------------------------------------------

def _new_out(rows, cols, device, planes: bool):
    """(tensor-or-Planes, pointer, plane stride) of a fresh [rows, cols] output in the requested storage format"""
    if planes:
        o = Planes.empty(rows, cols, device=device)
        return o, o.ptr(), o.plane
    o = torch.empty(rows, cols, dtype=torch.float32, device=device)
    return o, o.data_ptr(), 0


def embed_ln_fwd(ids, word, pos, type_row, gamma, beta, eps, L, out_planes: bool = False):
    lib = _lib.load()
    _chk(ids, "embed.ids", torch.int64)
    ids = ids.contiguous()
    T = ids.numel()
    H = word.shape[1]
    y, yp, ypl = _new_out(T, H, word.device, out_planes)
    xhat = torch.empty(T, H, dtype=torch.float32, device=word.device)
    rstd = torch.empty(T, dtype=torch.float32, device=word.device)
    check(lib.cxrk_embed_ln_fwd(_p(ids), _p(_chk(word, "embed.word")), _p(pos), _p(type_row), _p(gamma), _p(beta),
                                float(eps), T, L, H, yp, ypl, _p(xhat), _p(rstd), _stream()), "cxrk_embed_ln_fwd")
    return y, xhat, rstd


def residual_ln_fwd(x, res, gamma, beta, eps, save: bool = True, out_planes: bool = False):
    lib = _lib.load()
    rows, H = x.shape
    y, yp, ypl = _new_out(rows, H, x.device, out_planes)
    xhat = torch.empty_like(x) if save else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if save else None
    check(lib.cxrk_residual_ln_fwd(_p(_chk(x, "ln.x")), _p(res), _p(gamma), _p(beta), float(eps), rows, H, yp, ypl,
                                   _p(xhat), _p(rstd), _stream()), "cxrk_residual_ln_fwd")
    return y, xhat, rstd


def residual_ln_bwd(dy, xhat, rstd, gamma, dgamma, dbeta, dx_add=None, accumulate: bool = False, out=None, out_planes: bool = False,
                    dxsum=None, dxsum_accumulate: bool = False):
    """LayerNorm backward (+ dx_add); `dxsum` ([H], optional) receives the column sums of the returned gradient in the same pass."""
    lib = _lib.load()
    rows, H = dy.shape
    if out is None:
        dx, dxp, dxpl = _new_out(rows, H, dy.device, out_planes)
    elif isinstance(out, Planes):
        dx, dxp, dxpl = out, out.ptr(), out.plane
    else:
        dx, dxp, dxpl = out, out.data_ptr(), 0
    wsb = lib.cxrk_residual_ln_bwd_ws_bytes(rows, H)
    ws = workspace(wsb, dy.device)
    check(lib.cxrk_residual_ln_bwd(_p(_chk(dy, "ln.dy")), _p(xhat), _p(rstd), _p(gamma), rows, H, _p(dx_add), dxp, dxpl,
                                   _p(dgamma), _p(dbeta), int(accumulate), _p(dxsum), int(dxsum_accumulate), _p(ws), ws.numel() * 4,
                                   _stream()),
          "cxrk_residual_ln_bwd")
    return dx


def attn_fwd(qkv, mask, B, L, nH, dH, save_probs: bool = True, out_planes: bool = False):
    lib = _lib.load()
    ctx, cp, cpl = _new_out(B * L, nH * dH, qkv.device, out_planes)
    probs = torch.empty(B, nH, L, L, dtype=torch.float32, device=qkv.device) if save_probs else None
    if mask is not None:
        _chk(mask, "attn.mask", torch.int64)
    check(lib.cxrk_attn_fwd(_p(_chk(qkv, "attn.qkv")), _p(mask), B, L, nH, dH, cp, cpl, _p(probs), _stream()),
          f"cxrk_attn_fwd(B={B},L={L},nH={nH},dH={dH})")
    return ctx, probs


def attn_bwd(qkv, probs, dctx, B, L, nH, dH, out_planes: bool = False):
    lib = _lib.load()
    dqkv, dp, dpl = _new_out(qkv.shape[0], qkv.shape[1], qkv.device, out_planes)
    wsb = lib.cxrk_attn_bwd_ws_bytes(B, L, nH, dH)        # the dS matrix of the tiled form (L > 64); 0 otherwise
    ws = workspace(wsb, qkv.device) if wsb else None
    check(lib.cxrk_attn_bwd(_p(qkv), _p(probs), _p(_chk(dctx, "attn.dctx")), B, L, nH, dH, dp, dpl, _p(ws),
                            ws.numel() * 4 if ws is not None else 0, _stream()), f"cxrk_attn_bwd(B={B},L={L},nH={nH},dH={dH})")
    return dqkv


def planes_add_rows(src: Planes, dst: torch.Tensor) -> torch.Tensor:
    """dst[r, :] += src[r, :] for a row-strided fp32 `dst` view (e.g. the CLS rows of a [N, L*H] gradient)."""
    lib = _lib.load()
    rows, cols = src.shape
    if tuple(dst.shape) != (rows, cols) or dst.stride(1) != 1 or not src.is_contiguous():
        raise ValueError("planes_add_rows: shape / layout mismatch")
    check(lib.cxrk_planes_add_rows(src.ptr(), src.plane, rows, cols, _p(_chk(dst, "add_rows.dst")), dst.stride(0), _stream()),
          "cxrk_planes_add_rows")
    return dst


def embed_bwd(ids, dx, dword):
    """dword[ids[t]] += dx[t]; every row is summed in ascending token position (deterministic: csrc/embed.hip)."""
    lib = _lib.load()
    T, H = dx.shape
    ws = workspace(lib.cxrk_embed_bwd_ws_bytes(T, H), dx.device)
    check(lib.cxrk_embed_bwd(_p(ids), _p(_chk(dx.contiguous(), "embed.dx")), T, H, _p(dword), _p(ws), ws.numel() * 4, _stream()),
          "cxrk_embed_bwd")


def gelu_bwd(dy, pre):
    lib = _lib.load()
    dx = torch.empty_like(pre)
    check(lib.cxrk_gelu_bwd(_p(_chk(dy.contiguous(), "gelu.dy")), _p(pre), pre.numel(), _p(dx), _stream()), "cxrk_gelu_bwd")
    return dx


# ----------------------------------------------------------------------------------------------------------------
# heads
# ----------------------------------------------------------------------------------------------------------------

def l2norm_fwd(x: torch.Tensor, eps: float = 1e-12, out: Optional[torch.Tensor] = None):
    """xhat = x / max(|x|, eps) row-wise; `out` may be a column slice of a wider buffer (rows out.stride(0) apart)."""
    lib = _lib.load()
    x = _chk(x, "l2norm.x").contiguous()
    rows, D = x.shape
    xhat = torch.empty_like(x) if out is None else _chk(out, "l2norm.out")
    if tuple(xhat.shape) != (rows, D) or xhat.stride(1) != 1:
        raise ValueError(f"l2norm.out must be [{rows},{D}] with contiguous columns, got {tuple(xhat.shape)} strides {xhat.stride()}")
    norm = torch.empty(rows, dtype=torch.float32, device=x.device)
    check(lib.cxrk_l2norm_fwd(_p(x), rows, D, float(eps), _p(xhat), xhat.stride(0), _p(norm), _stream()), "cxrk_l2norm_fwd")
    return xhat, norm


def l2norm_bwd(dxhat, xhat, norm):
    lib = _lib.load()
    rows, D = xhat.shape
    if xhat.stride(1) != 1:
        raise ValueError("l2norm_bwd: xhat columns must be contiguous")
    dx = torch.empty(rows, D, dtype=torch.float32, device=xhat.device)
    check(lib.cxrk_l2norm_bwd(_p(_chk(dxhat.contiguous(), "l2norm.dxhat")), _p(xhat), xhat.stride(0), _p(norm), rows, D, _p(dx),
                              _stream()), "cxrk_l2norm_bwd")
    return dx


def infonce_row_lse(S, diag_off, loss_out=None, loss_scale: float = 0.0, loss_accumulate: bool = False):
    lib = _lib.load()
    rows, cols = S.shape
    lse = torch.empty(rows, dtype=torch.float32, device=S.device)
    diag = torch.empty(rows, dtype=torch.float32, device=S.device)
    check(lib.cxrk_infonce_row_lse(_p(_chk(S, "infonce.S")), S.stride(0), rows, cols, diag_off, _p(lse), _p(diag),
                                   _p(loss_out), float(loss_scale), int(loss_accumulate), _stream()),
          "cxrk_infonce_row_lse")
    return lse, diag


def infonce_grad_inplace(S, diag_off, lse_row, lse_col):
    lib = _lib.load()
    rows, cols = S.shape
    check(lib.cxrk_infonce_grad_inplace(_p(S), S.stride(0), rows, cols, diag_off, _p(lse_row), _p(lse_col), _stream()),
          "cxrk_infonce_grad_inplace")
    return S


def pairwise_cosine_fwd(x, y):
    lib = _lib.load()
    x = _chk(x, "cosine.x").contiguous()
    y = _chk(y, "cosine.y").contiguous()
    B, D = x.shape
    Pn = y.shape[0]
    if y.shape[1] != D:
        raise ValueError(f"pairwise_cosine: x is [{B},{D}] but y is {tuple(y.shape)}")
    cosv = torch.empty(B, Pn, dtype=torch.float32, device=x.device)
    xn = torch.empty(B, dtype=torch.float32, device=x.device)
    yn = torch.empty(Pn, dtype=torch.float32, device=x.device)
    check(lib.cxrk_pairwise_cosine_fwd(_p(x), _p(y), B, Pn, D, _p(cosv), _p(xn), _p(yn), _stream()),
          "cxrk_pairwise_cosine_fwd")
    return cosv, xn, yn


def pairwise_cosine_bwd(x, y, cosv, dcos, xn, yn, need_dx: bool = True):
    lib = _lib.load()
    B, D = x.shape
    Pn = y.shape[0]
    dx = torch.empty_like(x) if need_dx else None
    dy = torch.empty_like(y)
    wsb = lib.cxrk_pairwise_cosine_bwd_ws_bytes(B, Pn, D)
    ws = workspace(wsb, x.device)
    check(lib.cxrk_pairwise_cosine_bwd(_p(x), _p(y), _p(cosv), _p(_chk(dcos.contiguous(), "cosine.dcos")), _p(xn), _p(yn),
                                       B, Pn, D, _p(dx), _p(dy), 0, _p(ws), ws.numel() * 4, _stream()),
          "cxrk_pairwise_cosine_bwd")
    return dx, dy


def pairwise_cosine_max_fwd(x, y, groups: int):
    """x [B,D], y [groups*Pg, D] -> cos [B,groups*Pg], max / mean over each group's prompts [B,groups], winner index (int32)."""
    lib = _lib.load()
    x = _chk(x, "cosine.x").contiguous()
    y = _chk(y, "cosine.y").contiguous()
    B, D = x.shape
    Pn = y.shape[0]
    if y.shape[1] != D or groups <= 0 or Pn % groups:
        raise ValueError(f"pairwise_cosine_max: x is [{B},{D}], y is {tuple(y.shape)}, groups = {groups}")
    f = dict(dtype=torch.float32, device=x.device)
    cosv, xn, yn = torch.empty(B, Pn, **f), torch.empty(B, **f), torch.empty(Pn, **f)
    mx, mean = torch.empty(B, groups, **f), torch.empty(B, groups, **f)
    arg = torch.empty(B, groups, dtype=torch.int32, device=x.device)
    check(lib.cxrk_pairwise_cosine_max_fwd(_p(x), _p(y), B, groups, Pn // groups, D, _p(cosv), _p(xn), _p(yn), _p(mx), _p(mean),
                                           _p(arg), _stream()), "cxrk_pairwise_cosine_max_fwd")
    return cosv, xn, yn, mx, mean, arg


def pairwise_cosine_max_bwd(x, y, cosv, dmax, arg, xn, yn, need_dx: bool = True):
    lib = _lib.load()
    B, D = x.shape
    Pn = y.shape[0]
    G = arg.shape[1]
    dx = torch.empty_like(x) if need_dx else None
    dy = torch.empty_like(y)
    ws = workspace(lib.cxrk_pairwise_cosine_bwd_ws_bytes(B, Pn, D), x.device)
    check(lib.cxrk_pairwise_cosine_max_bwd(_p(x), _p(y), _p(cosv), _p(_chk(dmax.contiguous(), "cosine.dmax")), _p(arg), _p(xn), _p(yn),
                                           B, G, Pn // G, D, _p(dx), _p(dy), 0, _p(ws), ws.numel() * 4, _stream()),
          "cxrk_pairwise_cosine_max_bwd")
    return dx, dy


def patch_similarity(patches, text):
    """patches [R,D] fp32, text [D] -> [R]: <patch_r, text> (vlp/inference_engine.py:104)."""
    lib = _lib.load()
    patches = _chk(patches, "patch_similarity.patches").contiguous()
    text = _chk(text, "patch_similarity.text").contiguous().reshape(-1)
    R, D = patches.shape
    if text.numel() != D:
        raise ValueError(f"patch_similarity: patches are [{R},{D}] but the text embedding has {text.numel()} features")
    out = torch.empty(R, dtype=torch.float32, device=patches.device)
    check(lib.cxrk_patch_similarity(_p(patches), _p(text), R, D, _p(out), _stream()), "cxrk_patch_similarity")
    return out


def bce_posneg_fwd_bwd(cosv, labels, diff: bool = True, need_grad: bool = True):
    """cos [B,2C] (2c = pos, 2c+1 = neg), labels [B,C] view (stride(1)==1) -> logits [B,C], dcos [B,2C], loss []"""
    lib = _lib.load()
    B, C2 = cosv.shape
    C = C2 // 2
    _chk(labels, "bce.labels")
    if labels.dim() == 1:
        labels = labels.unsqueeze(1)
    if labels.shape[1] > 1 and labels.stride(1) != 1:
        labels = labels.contiguous()
    logits = torch.empty(B, C, dtype=torch.float32, device=cosv.device)
    dcos = torch.empty_like(cosv) if need_grad else None
    loss = torch.empty((), dtype=torch.float32, device=cosv.device)
    ws = workspace(lib.cxrk_bce_posneg_ws_bytes(), cosv.device)
    check(lib.cxrk_bce_posneg_fwd_bwd(_p(_chk(cosv, "bce.cos")), _p(labels), B, C, labels.stride(0), int(diff), _p(logits),
                                      _p(dcos), _p(loss), _p(ws), ws.numel() * 4, _stream()), "cxrk_bce_posneg_fwd_bwd")
    return logits, dcos, loss


def eval_score(cosv, pred_diff: bool = False):
    lib = _lib.load()
    B, C2 = cosv.shape
    C = C2 // 2
    score = torch.empty(B, C, dtype=torch.float32, device=cosv.device)
    pred = torch.empty(B, C, dtype=torch.float32, device=cosv.device)
    check(lib.cxrk_eval_score(_p(_chk(cosv, "eval.cos")), B, C, int(pred_diff), _p(score), _p(pred), _stream()),
          "cxrk_eval_score")
    return score, pred


def scale_mask(x, mask_src=None, alpha_dev=None, alpha: float = 1.0, out=None):
    """out = alpha * (*alpha_dev) * x * (mask_src > 0); alpha_dev is a 0-d device tensor (no host sync)."""
    lib = _lib.load()
    x = _chk(x, "scale_mask.x").contiguous()
    if out is None:
        out = torch.empty_like(x)
    check(lib.cxrk_scale_mask(_p(x), _p(mask_src), _p(alpha_dev), float(alpha), x.numel(), _p(out), _stream()),
          "cxrk_scale_mask")
    return out


def group_mean_fwd(x, G, n):
    lib = _lib.load()
    D = x.shape[-1]
    out = torch.empty(G, D, dtype=torch.float32, device=x.device)
    check(lib.cxrk_group_mean_fwd(_p(_chk(x.contiguous(), "group_mean.x")), G, n, D, _p(out), _stream()), "cxrk_group_mean_fwd")
    return out


def group_mean_bwd(dout, G, n):
    lib = _lib.load()
    D = dout.shape[-1]
    din = torch.empty(G * n, D, dtype=torch.float32, device=dout.device)
    check(lib.cxrk_group_mean_bwd(_p(_chk(dout.contiguous(), "group_mean.dout")), G, n, D, _p(din), _stream()),
          "cxrk_group_mean_bwd")
    return din


# ----------------------------------------------------------------------------------------------------------------
# optimiser / continual learning
# ----------------------------------------------------------------------------------------------------------------

def adam_fused(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale: float = 1.0):
    lib = _lib.load()
    n = p.numel()
    for nme, t in (("p", p), ("g", g), ("m", m), ("v", v)):
        _chk(t, "adam." + nme)
        if not t.is_contiguous() or t.numel() != n:
            raise ValueError(f"adam.{nme}: must be contiguous with {n} elements")
    check(lib.cxrk_adam_fused(_p(p), _p(g), _p(m), _p(v), n, float(lr), float(beta1), float(beta2), float(eps),
                              float(weight_decay), int(step), float(grad_scale), _stream()), "cxrk_adam_fused")


def sgd(p, g, lr, weight_decay: float = 0.0, grad_scale: float = 1.0):
    lib = _lib.load()
    check(lib.cxrk_sgd(_p(_chk(p, "sgd.p")), _p(_chk(g, "sgd.g")), p.numel(), float(lr), float(weight_decay),
                       float(grad_scale), _stream()), "cxrk_sgd")


def weight_reset(pnew, pold, threshold, counters):
    """In place: restore entries of pnew whose |pnew-pold| < min + thr*(max-min); counters[0] += #restored (uint64)."""
    lib = _lib.load()
    ws = workspace(lib.cxrk_weight_reset_ws_bytes(), pnew.device)
    check(lib.cxrk_weight_reset(_p(_chk(pnew, "reset.new")), _p(_chk(pold, "reset.old")), pnew.numel(), float(threshold),
                                _p(counters), _p(ws), ws.numel() * 4, _stream()), "cxrk_weight_reset")
