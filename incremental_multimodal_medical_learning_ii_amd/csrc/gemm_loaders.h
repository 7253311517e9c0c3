// Operand loaders of the MFMA mainloops (included by gemm_core.h).
//
// A loader turns an (idx, k) tile coordinate into global loads of 16 B per lane and stages the values into LDS.
// TILE = extent of the idx dimension in the block tile.  Every loader exposes
//   P (host-filled parameters), V (register type of one 16-byte load), NV (loads per K-tile), init(P, idx0, tid),
//   load(k0, v[NV]), store2(hi, lo, v[NV])  [split-bf16 planes in LDS]  and, fp32 sources only, store(S, v[NV]).
// load() must be called with k0 advancing by BK from one call to the next (any start): the convolution loaders carry
// the filter tap / pixel position of k0 as wave-uniform state instead of re-deriving it with integer divisions.
//
// Source formats (template parameter FMT of every loader):
//   F32  fp32 tensor.  The split-bf16 mainloops split every value into bf16 hi/lo halves while staging it (5 vector
//        instructions per pair); the exact-fp32 mainloop stages it as is.
//   PL   pre-split tensor ("planes"): the producer already wrote x as two bf16 planes hi = bf16(x), lo = bf16(x - hi), the lo
//        plane `plane` elements behind the hi plane (same 4 bytes per element as fp32).  A lane loads 8 consecutive bf16 of
//        one plane and stores them to LDS with ONE ds_write_b128: no conversion work is left in the mainloop.
//
// Addressing.  The mainloop is bound by vector-ALU issue, not by the matrix pipe (rocprofv3 SQ_INSTS_VALU / SQ_INSTS_MFMA
// was 10 for the dense and 16-27 for the convolution loaders when every load carried its own 64-bit address, bounds
// compare and branch), so the loaders spend as few vector instructions per load as possible:
//  * every load is a buffer_load_dwordx4 through a per-K-tile descriptor whose base holds everything that is uniform over
//    the block (tile origin, k0, the filter tap): scalar ALU only;
//  * what depends on the lane is ONE 32-bit byte offset per load, computed once in init();
//  * a lane that must read zero (row outside the tensor, padding halo, K tail) gets bit 31 set in its offset: the
//    descriptor declares 2^31 bytes, so the hardware range check returns 0 and no branch or select guards the load.
//    Convolution halos are a per-row bit mask over the (<= 32) filter taps, tested with one v_bfe_u32 per load.
// A block's loads stay within 2 GiB of its tile origin (host-side checks in launch_gemm's callers).
#pragma once

namespace cxrk {

constexpr unsigned VOFF_OOB = 0x80000000u;

struct F32 {
  static constexpr bool PLANES = false;
  static constexpr int ESZ = 4;   // bytes per element
  static constexpr int EPL = 4;   // elements per 16-byte lane load
  static constexpr int NPL = 1;   // planes
  typedef float T;
  typedef float4 V;
};
struct PL {
  static constexpr bool PLANES = true;
  static constexpr int ESZ = 2;
  static constexpr int EPL = 8;
  static constexpr int NPL = 2;
  typedef unsigned short T;
  typedef uint4 V;
};

// `live` false = a descriptor of zero bytes: every load through it returns 0 and moves no data (the pipelined mainloop
// issues the loads of tiles past the end of its K range that way instead of branching around them)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const void* p, bool live = true) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, live ? (int)0x80000000 : 0, 0x00020000);
}
__device__ __forceinline__ float4 bload4(__amdgpu_buffer_rsrc_t r, unsigned voff) {
  // bit_cast, not a conversion: the builtin returns a 128-bit scalar-like value, and converting THAT to a vector splats
  // its low dword (the compiler then narrows the load to buffer_load_dword)
  const auto q = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0);
  static_assert(sizeof(q) == 16, "raw_buffer_load_b128 must return 16 bytes");
  const f32x4 f = __builtin_bit_cast(f32x4, q);
  return make_float4(f[0], f[1], f[2], f[3]);
}
__device__ __forceinline__ uint4 bloadu4(__amdgpu_buffer_rsrc_t r, unsigned voff) {
  const auto q = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0);
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 u = __builtin_bit_cast(u32x4, q);
  return make_uint4(u[0], u[1], u[2], u[3]);
}
template <class FMT> __device__ __forceinline__ typename FMT::V bloadv(__amdgpu_buffer_rsrc_t r, unsigned voff);
template <> __device__ __forceinline__ float4 bloadv<F32>(__amdgpu_buffer_rsrc_t r, unsigned voff) { return bload4(r, voff); }
template <> __device__ __forceinline__ uint4 bloadv<PL>(__amdgpu_buffer_rsrc_t r, unsigned voff) { return bloadu4(r, voff); }

// offset | (bit t of inv ? OOB : 0): two vector instructions (v_bfe_u32, v_lshl_or_b32)
__device__ __forceinline__ unsigned masked_off(unsigned off, unsigned inv, int t) {
  return (__builtin_amdgcn_ubfe(inv, (unsigned)t, 1u) << 31) | off;
}

// Halo masks without data-dependent loops (a runtime-bounded loop inside the unrolled per-row loop makes the row index
// dynamic and sends the loader's off[] / inv[] arrays to scratch memory).
__device__ __forceinline__ unsigned low_bits(int k) { return k >= 32 ? 0xffffffffu : ((1u << k) - 1u); }
// bits i in [0, n) that fall outside [lo, hi)   (0 <= lo, hi <= n after clamping)
__device__ __forceinline__ unsigned outside_bits(int lo, int hi, int n) {
  lo = min(max(lo, 0), n); hi = min(max(hi, 0), n);
  return low_bits(lo) | (low_bits(n) & ~low_bits(hi));
}
// bit (r*S + s) = badr bit r | bads bit s, for r < R <= 8, R*S <= 32
__device__ __forceinline__ unsigned tap_mask(unsigned badr, unsigned bads, int R, int S) {
  const unsigned full = low_bits(S);
  unsigned m = 0;
#pragma unroll
  for (int r = 0; r < 8; ++r)
    if (r < R) m |= (((badr >> r) & 1u) ? full : bads) << (r * S);
  return m;
}

// ---- LDS staging shared by the loaders -------------------------------------------------------------------------------
// K-contiguous operand: kq = tid % KQ -> EPL consecutive k, r0 = tid / KQ -> row, +RP rows per j   (KQ = BK / EPL)
template <int NV, int LD, int RP>
__device__ __forceinline__ void stage_kc(float* S, int r0, int k4, const float4 (&v)[NV]) {
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    float* d = S + (k4 * 4) * LD + r0 + j * RP;
    d[0] = v[j].x; d[LD] = v[j].y; d[2 * LD] = v[j].z; d[3 * LD] = v[j].w;
  }
}
template <int NV, int RP>
__device__ __forceinline__ void stage2_kc(unsigned short* hi, unsigned short* lo, int r0, int kq, const float4 (&v)[NV]) {
#pragma unroll
  for (int j = 0; j < NV; ++j) store2_kc(hi, lo, r0 + j * RP, kq * 4, v[j]);
}
template <int NV, int RP>   // planes: v[0 .. NV/2) = hi loads, v[NV/2 .. NV) = lo loads
__device__ __forceinline__ void stage2_kc(unsigned short* hi, unsigned short* lo, int r0, int kq, const uint4 (&v)[NV]) {
#pragma unroll
  for (int j = 0; j < NV / 2; ++j) {
    *reinterpret_cast<uint4*>(hi + (r0 + j * RP) * LDH + kq * 8) = v[j];
    *reinterpret_cast<uint4*>(lo + (r0 + j * RP) * LDH + kq * 8) = v[NV / 2 + j];
  }
}
// idx-contiguous operand: cq = tid % VPR -> EPL consecutive idx, kr0 = tid / VPR -> k row, +RPP rows per j   (VPR = TILE / EPL)
template <int NV, int LD, int RPP>
__device__ __forceinline__ void stage_mc(float* S, int kr0, int c4, const float4 (&v)[NV]) {
#pragma unroll
  for (int j = 0; j < NV; ++j) *reinterpret_cast<float4*>(S + (kr0 + j * RPP) * LD + c4 * 4) = v[j];
}
template <int NV, int LDT, int RPP>
__device__ __forceinline__ void stage2_mc(unsigned short* hi, unsigned short* lo, int kr0, int cq, const float4 (&v)[NV]) {
#pragma unroll
  for (int j = 0; j < NV; ++j) store2_mc<LDT>(hi, lo, cq * 4, kr0 + j * RPP, v[j]);
}
template <int NV, int LDT, int RPP>
__device__ __forceinline__ void stage2_mc(unsigned short* hi, unsigned short* lo, int kr0, int cq, const uint4 (&v)[NV]) {
#pragma unroll
  for (int j = 0; j < NV / 2; ++j) {
    *reinterpret_cast<uint4*>(hi + (kr0 + j * RPP) * LDT + cq * 8) = v[j];
    *reinterpret_cast<uint4*>(lo + (kr0 + j * RPP) * LDT + cq * 8) = v[NV / 2 + j];
  }
}

// Geometry shared by the K-contiguous loaders
template <int TILE, class FMT, int NT>
struct KCGeom {
  static constexpr bool KC = true;
  static constexpr bool FMT_PLANES = FMT::PLANES;
  static constexpr int PLANE = TILE * LDH;      // halfwords of one bf16 plane in LDS
  static constexpr int KQ = BK / FMT::EPL;      // lanes per row
  static constexpr int RP = NT / KQ;            // rows per pass
  static constexpr int NVR = TILE / RP;         // loads per plane
  static constexpr int NV = NVR * FMT::NPL;
  static constexpr int LD = TILE + 1;           // LDS row stride of the exact-fp32 mainloop (floats)
  static_assert(NVR >= 1 && NVR * RP == TILE, "tile / thread-count mismatch");
};
// Geometry shared by the idx-contiguous loaders
template <int TILE, class FMT, int NT>
struct MCGeom {
  static constexpr bool KC = false;
  static constexpr bool FMT_PLANES = FMT::PLANES;
  static constexpr int LDT = TILE + 32;
  static constexpr int PLANE = BK * LDT;        // halfwords of one bf16 plane in LDS
  static constexpr int LD = TILE + 4;           // LDS row stride of the exact-fp32 mainloop (floats)
  static constexpr int VPR = TILE / FMT::EPL;   // lane loads per k-row
  static constexpr int RPP = NT / VPR;          // k-rows per pass
  static constexpr int NVR = BK / RPP;
  static constexpr int NV = NVR * FMT::NPL;
  static_assert(NVR >= 1 && NVR * RPP == BK, "tile / thread-count mismatch");
};

// X(idx, k) = ptr[idx*ld + k]   (k contiguous)
template <int TILE, class FMT = F32, int NT = NTHREADS>
struct DenseKC : KCGeom<TILE, FMT, NT> {
  typedef KCGeom<TILE, FMT, NT> G;
  typedef typename FMT::T T; typedef typename FMT::V V;
  using G::NV; using G::NVR; using G::RP; using G::KQ; using G::LD;
  struct P { const T* ptr; long ld; int rows; int K; long plane; };
  const T* bp; long plane; unsigned voff[NVR]; int kq, r0, K;
  __device__ __forceinline__ void init(const P& p, int idx0, int tid) {
    kq = tid % KQ; r0 = tid / KQ; K = p.K; plane = p.plane;
    bp = p.ptr + (long)idx0 * p.ld;
    const int ld = (int)p.ld;
#pragma unroll
    for (int j = 0; j < NVR; ++j)
      voff[j] = idx0 + r0 + j * RP < p.rows ? (unsigned)(((r0 + j * RP) * ld + kq * FMT::EPL) * FMT::ESZ) : VOFF_OOB;
  }
  __device__ __forceinline__ void load(int k0, V (&v)[NV], bool live = true) {
    const unsigned t = (k0 + BK <= K || k0 + kq * FMT::EPL < K) ? 0u : VOFF_OOB;   // K tail (the last K-tile only)
#pragma unroll
    for (int pl = 0; pl < FMT::NPL; ++pl) {
      const __amdgpu_buffer_rsrc_t rs = tile_rsrc(bp + pl * plane + k0, live);
#pragma unroll
      for (int j = 0; j < NVR; ++j) v[pl * NVR + j] = bloadv<FMT>(rs, voff[j] | t);
    }
  }
  __device__ __forceinline__ void store2(unsigned short* hi, unsigned short* lo, const V (&v)[NV]) const { stage2_kc<NV, RP>(hi, lo, r0, kq, v); }
  __device__ __forceinline__ void store(float* S, const float4 (&v)[NV]) const { stage_kc<NV, LD, RP>(S, r0, kq, v); }
};

// X(idx, k) = ptr[k*ld + idx]   (idx contiguous)
template <int TILE, class FMT = F32, int NT = NTHREADS>
struct DenseMC : MCGeom<TILE, FMT, NT> {
  typedef MCGeom<TILE, FMT, NT> G;
  typedef typename FMT::T T; typedef typename FMT::V V;
  using G::NV; using G::NVR; using G::RPP; using G::VPR; using G::LD; using G::LDT;
  struct P { const T* ptr; long ld; int cols; int K; long plane; };
  const T* bp; long ld_, plane; unsigned voff[NVR];
  int cq, kr0, K;
  __device__ __forceinline__ void init(const P& p, int idx0, int tid) {
    cq = tid % VPR; kr0 = tid / VPR; K = p.K; ld_ = p.ld; plane = p.plane;
    bp = p.ptr + idx0;
    const bool ok = idx0 + cq * FMT::EPL < p.cols;
    const int ld = (int)p.ld;
#pragma unroll
    for (int j = 0; j < NVR; ++j) voff[j] = ok ? (unsigned)(((kr0 + j * RPP) * ld + cq * FMT::EPL) * FMT::ESZ) : VOFF_OOB;
  }
  __device__ __forceinline__ void load(int k0, V (&v)[NV], bool live = true) {
    const bool full = k0 + BK <= K;
#pragma unroll
    for (int pl = 0; pl < FMT::NPL; ++pl) {
      const __amdgpu_buffer_rsrc_t rs = tile_rsrc(bp + pl * plane + (long)k0 * ld_, live);
#pragma unroll
      for (int j = 0; j < NVR; ++j) v[pl * NVR + j] = bloadv<FMT>(rs, (full || k0 + kr0 + j * RPP < K) ? voff[j] : VOFF_OOB);
    }
  }
  __device__ __forceinline__ void store2(unsigned short* hi, unsigned short* lo, const V (&v)[NV]) const { stage2_mc<NV, LDT, RPP>(hi, lo, kr0, cq, v); }
  __device__ __forceinline__ void store(float* S, const float4 (&v)[NV]) const { stage_mc<NV, LD, RPP>(S, kr0, cq, v); }
};

// NHWC geometry shared by the convolution gathers.
struct ConvGeom {
  int N, H, W, C;        // input  x[N][H][W][C]
  int Ho, Wo, Ko;        // output y[N][Ho][Wo][Ko]
  int R, S, stride, pad; // filter w[Ko][R][S][C]
};

// fprop A operand: idx = (n,ho,wo), k = (r,s,c) -> x[n][ho*st-pad+r][wo*st-pad+s][c]   (c contiguous)
// TAPWISE: C % BK == 0, so a K-tile lies inside ONE filter tap (r,s): the tap is wave-uniform state, the halo a bit mask
// over the R*S <= 32 taps.  Otherwise (the stem: C = 4, 49 taps; fp32 source only) every lane derives its own tap.
template <int TILE, class FMT = F32, bool TAPWISE = true, int NT = NTHREADS>
struct ConvIm2colKC : KCGeom<TILE, FMT, NT> {
  typedef KCGeom<TILE, FMT, NT> G;
  typedef typename FMT::T T; typedef typename FMT::V V;
  using G::NV; using G::NVR; using G::RP; using G::KQ; using G::LD;
  static_assert(TAPWISE || !FMT::PLANES, "the per-lane-tap gather exists for the fp32 stem only");
  struct P { const T* x; ConvGeom g; int rows; int K; long plane; };
  const T* bp; long plane;
  unsigned off[NVR];                // byte offset of (row, tap 0, c = EPL*kq) from bp   [TAPWISE: >= 0 by construction]
  unsigned inv[NVR];                // TAPWISE: bit (r*S+s) set = that tap falls outside the image for this row
  int hi0[NVR], wi0[NVR];           // !TAPWISE
  int kq, r0, K, H, W, C, S;
  int tr, ts, tc, knext;            // TAPWISE: tap (tr,ts) and channel offset tc of K-tile `knext` (wave-uniform)
  __device__ __forceinline__ void init(const P& p, int idx0, int tid) {
    kq = tid % KQ; r0 = tid / KQ; K = p.K; H = p.g.H; W = p.g.W; C = p.g.C; S = p.g.S; plane = p.plane;
    const int HoWo = p.g.Ho * p.g.Wo;
    const int n_first = idx0 / HoWo;
    // bias so that off >= 0 for every row of the tile: the halo can reach `pad` rows / columns before the image
    const int bias = TAPWISE ? (p.g.pad * W + p.g.pad) * C : 0;
    bp = p.x + (long)n_first * H * W * C - bias;
    knext = -1; tr = ts = tc = 0;
#pragma unroll
    for (int j = 0; j < NVR; ++j) {
      const int row = idx0 + r0 + j * RP;
      const bool in = row < p.rows;
      const int rowc = in ? row : 0;   // every value below is computed for a valid row and then selected (no branches)
      const int wo = rowc % p.g.Wo; const int t = rowc / p.g.Wo; const int ho = t % p.g.Ho; const int n = t / p.g.Ho;
      const int h0 = ho * p.g.stride - p.g.pad, w0 = wo * p.g.stride - p.g.pad;
      const unsigned o = (unsigned)(((((n - n_first) * H + h0) * W + w0) * C + bias + (TAPWISE ? kq * FMT::EPL : 0)) * FMT::ESZ);
      off[j] = in ? o : 0u;
      if constexpr (TAPWISE) {   // tap (r,s) reads (h0 + r, w0 + s): inside the image for r in [-h0, H - h0), s in [-w0, W - w0)
        const unsigned m = tap_mask(outside_bits(-h0, H - h0, p.g.R), outside_bits(-w0, W - w0, S), p.g.R, S);
        inv[j] = in ? m : 0xffffffffu;
      } else {
        hi0[j] = in ? h0 : -(1 << 28); wi0[j] = in ? w0 : 0;
      }
    }
  }
  __device__ __forceinline__ void load(int k0, V (&v)[NV], bool live = true) {
    if constexpr (TAPWISE) {
      if (k0 != knext) { const int tap = k0 / C; tc = k0 - tap * C; tr = tap / S; ts = tap - tr * S; }
      const int t = tr * S + ts;
#pragma unroll
      for (int pl = 0; pl < FMT::NPL; ++pl) {
        const __amdgpu_buffer_rsrc_t rs = tile_rsrc(bp + pl * plane + ((long)(tr * W + ts) * C + tc), live);
#pragma unroll
        for (int j = 0; j < NVR; ++j) v[pl * NVR + j] = bloadv<FMT>(rs, masked_off(off[j], inv[j], t));
      }
      tc += BK;
      if (tc >= C) { tc = 0; if (++ts == S) { ts = 0; ++tr; } }
      knext = k0 + BK;
    } else {
      const __amdgpu_buffer_rsrc_t rs = tile_rsrc(bp, live);
      const int k = k0 + kq * 4;
      const int tap = k / C, c = k - tap * C;
      const int r = tap / S, s = tap - r * S;
      const unsigned toff = (unsigned)(((r * W + s) * C + c) * 4);
#pragma unroll
      for (int j = 0; j < NVR; ++j) {
        const bool ok = (k < K) && ((unsigned)(hi0[j] + r) < (unsigned)H) && ((unsigned)(wi0[j] + s) < (unsigned)W);
        v[j] = bloadv<FMT>(rs, ok ? off[j] + toff : VOFF_OOB);
      }
    }
  }
  __device__ __forceinline__ void store2(unsigned short* hi, unsigned short* lo, const V (&v)[NV]) const { stage2_kc<NV, RP>(hi, lo, r0, kq, v); }
  __device__ __forceinline__ void store(float* Sm, const float4 (&v)[NV]) const { stage_kc<NV, LD, RP>(Sm, r0, kq, v); }
};

// dgrad A operand (stride 1): idx = (n,hi,wi), k = (r,s,ko) -> dy[n][hi+pad-r][wi+pad-s][ko].  Ko % BK == 0: a K-tile
// lies inside one tap.  Stride-2 layers go through ConvDgradS2KC.
template <int TILE, class FMT = F32, int NT = NTHREADS>
struct ConvDgradKC : KCGeom<TILE, FMT, NT> {
  typedef KCGeom<TILE, FMT, NT> G;
  typedef typename FMT::T T; typedef typename FMT::V V;
  using G::NV; using G::NVR; using G::RP; using G::KQ; using G::LD;
  struct P { const T* dy; ConvGeom g; int rows; int K; long plane; };
  const T* bp; long plane; unsigned off[NVR], inv[NVR];
  int kq, r0, Wo, Ko, R, S;
  int tr, ts, tk, knext;  // tap (tr,ts), channel offset tk of K-tile `knext`
  __device__ __forceinline__ void init(const P& p, int idx0, int tid) {
    kq = tid % KQ; r0 = tid / KQ; Wo = p.g.Wo; Ko = p.g.Ko; R = p.g.R; S = p.g.S; plane = p.plane;
    const int Ho = p.g.Ho, H = p.g.H, W = p.g.W, pad = p.g.pad;
    const int n_first = idx0 / (H * W);
    // address of tap (r,s) = bp + [((R-1-r)*Wo + (S-1-s))*Ko + ko0]  (uniform, >= 0)  +  off[row]  (>= 0)
    bp = p.dy + ((long)n_first * Ho * Wo - ((R - 1) * Wo + (S - 1))) * Ko;
    knext = -1; tr = ts = tk = 0;
#pragma unroll
    for (int j = 0; j < NVR; ++j) {
      const int row = idx0 + r0 + j * RP;
      if (row < p.rows) {
        const int wi = row % W; const int t = row / W; const int hi = t % H; const int n = t / H;
        off[j] = (unsigned)(((((n - n_first) * Ho + hi + pad) * Wo + wi + pad) * Ko + kq * FMT::EPL) * FMT::ESZ);
        // tap (r,s) reads (hi + pad - r, wi + pad - s): inside for r in (hi + pad - Ho, hi + pad], s likewise
        inv[j] = tap_mask(outside_bits(hi + pad - Ho + 1, hi + pad + 1, R), outside_bits(wi + pad - Wo + 1, wi + pad + 1, S), R, S);
      } else { off[j] = 0; inv[j] = 0xffffffffu; }
    }
  }
  __device__ __forceinline__ void load(int k0, V (&v)[NV], bool live = true) {
    if (k0 != knext) { const int tap = k0 / Ko; tk = k0 - tap * Ko; tr = tap / S; ts = tap - tr * S; }
    const int t = tr * S + ts;
#pragma unroll
    for (int pl = 0; pl < FMT::NPL; ++pl) {
      const __amdgpu_buffer_rsrc_t rs = tile_rsrc(bp + pl * plane + ((long)((R - 1 - tr) * Wo + (S - 1 - ts)) * Ko + tk), live);
#pragma unroll
      for (int j = 0; j < NVR; ++j) v[pl * NVR + j] = bloadv<FMT>(rs, masked_off(off[j], inv[j], t));
    }
    tk += BK;
    if (tk >= Ko) { tk = 0; if (++ts == S) { ts = 0; ++tr; } }
    knext = k0 + BK;
  }
  __device__ __forceinline__ void store2(unsigned short* hi, unsigned short* lo, const V (&v)[NV]) const { stage2_kc<NV, RP>(hi, lo, r0, kq, v); }
  __device__ __forceinline__ void store(float* Sm, const float4 (&v)[NV]) const { stage_kc<NV, LD, RP>(Sm, r0, kq, v); }
};

// dgrad B operand: k = (r,s,ko), idx = c -> w[ko][r][s][c]   (c contiguous)
template <int TILE, class FMT = F32, int NT = NTHREADS>
struct ConvFilterMC : MCGeom<TILE, FMT, NT> {
  typedef MCGeom<TILE, FMT, NT> G;
  typedef typename FMT::T T; typedef typename FMT::V V;
  using G::NV; using G::NVR; using G::RPP; using G::VPR; using G::LD; using G::LDT;
  struct P { const T* w; ConvGeom g; int cols; int K; long plane; };
  const T* bp; long plane; unsigned voff[NVR]; int cq, kr0, Ko, C; long RSC;
  int tap, tk, knext;
  __device__ __forceinline__ void init(const P& p, int idx0, int tid) {
    cq = tid % VPR; kr0 = tid / VPR; Ko = p.g.Ko; C = p.g.C; RSC = (long)p.g.R * p.g.S * p.g.C; plane = p.plane;
    bp = p.w + idx0;
    const bool ok = idx0 + cq * FMT::EPL < p.cols;
    knext = -1; tap = tk = 0;
#pragma unroll
    for (int j = 0; j < NVR; ++j) voff[j] = ok ? (unsigned)(((kr0 + j * RPP) * (int)RSC + cq * FMT::EPL) * FMT::ESZ) : VOFF_OOB;
  }
  __device__ __forceinline__ void load(int k0, V (&v)[NV], bool live = true) {
    if (k0 != knext) { tap = k0 / Ko; tk = k0 - tap * Ko; }
#pragma unroll
    for (int pl = 0; pl < FMT::NPL; ++pl) {
      const __amdgpu_buffer_rsrc_t rs = tile_rsrc(bp + pl * plane + ((long)tk * RSC + (long)tap * C), live);
#pragma unroll
      for (int j = 0; j < NVR; ++j) v[pl * NVR + j] = bloadv<FMT>(rs, voff[j]);
    }
    tk += BK;
    if (tk >= Ko) { tk = 0; ++tap; }
    knext = k0 + BK;
  }
  __device__ __forceinline__ void store2(unsigned short* hi, unsigned short* lo, const V (&v)[NV]) const { stage2_mc<NV, LDT, RPP>(hi, lo, kr0, cq, v); }
  __device__ __forceinline__ void store(float* Sm, const float4 (&v)[NV]) const { stage_mc<NV, LD, RPP>(Sm, kr0, cq, v); }
};

// Stride-2 dgrad, one output-parity class (ph,pw) per launch: only the taps r = (ph+pad) mod 2 (+2) reach pixels
// (2a+ph, 2b+pw), so the K loop runs over those taps only (no multiply-by-zero work: 9 taps -> 1+2+2+4 over the
// four classes).  idx = (n,a,b) on the half-resolution grid; k = (ti,ko) with ti indexing the class's tap list.
struct S2Taps { int nr, ns; int r[2], s[2]; int dr[2], ds[2]; };  // ho = a + dr[tr], wo = b + ds[ts]   (dr, ds >= 0)

template <int TILE, class FMT = F32, int NT = NTHREADS>
struct ConvDgradS2KC : KCGeom<TILE, FMT, NT> {
  typedef KCGeom<TILE, FMT, NT> G;
  typedef typename FMT::T T; typedef typename FMT::V V;
  using G::NV; using G::NVR; using G::RP; using G::KQ; using G::LD;
  struct P { const T* dy; ConvGeom g; S2Taps t; int Hs, Ws; int rows; int K; long plane; };
  const T* bp; long plane; unsigned off[NVR], inv[NVR];
  int kq, r0, Wo, Ko; S2Taps t;
  int ti, tk, knext;
  __device__ __forceinline__ void init(const P& p, int idx0, int tid) {
    kq = tid % KQ; r0 = tid / KQ; Wo = p.g.Wo; Ko = p.g.Ko; t = p.t; plane = p.plane;
    const int Ho = p.g.Ho;
    const int n_first = idx0 / (p.Hs * p.Ws);
    bp = p.dy + (long)n_first * Ho * Wo * Ko;
    knext = -1; ti = tk = 0;
#pragma unroll
    for (int j = 0; j < NVR; ++j) {
      const int row = idx0 + r0 + j * RP;
      if (row < p.rows) {
        const int b = row % p.Ws; const int q = row / p.Ws; const int a = q % p.Hs; const int n = q / p.Hs;
        off[j] = (unsigned)(((((n - n_first) * Ho + a) * Wo + b) * Ko + kq * FMT::EPL) * FMT::ESZ);
        unsigned m = 0;
#pragma unroll
        for (int ir = 0; ir < 2; ++ir)
#pragma unroll
          for (int is = 0; is < 2; ++is)
            if (ir < t.nr && is < t.ns && (a + t.dr[ir] >= Ho || b + t.ds[is] >= Wo)) m |= 1u << (ir * t.ns + is);
        inv[j] = m;
      } else { off[j] = 0; inv[j] = 0xffffffffu; }
    }
  }
  __device__ __forceinline__ void load(int k0, V (&v)[NV], bool live = true) {
    if (k0 != knext) { ti = k0 / Ko; tk = k0 - ti * Ko; }
    const int ir = ti / t.ns, is = ti - ir * t.ns;
    const int dr = ir == 0 ? t.dr[0] : t.dr[1], ds = is == 0 ? t.ds[0] : t.ds[1];
#pragma unroll
    for (int pl = 0; pl < FMT::NPL; ++pl) {
      const __amdgpu_buffer_rsrc_t rs = tile_rsrc(bp + pl * plane + ((long)(dr * Wo + ds) * Ko + tk), live);
#pragma unroll
      for (int j = 0; j < NVR; ++j) v[pl * NVR + j] = bloadv<FMT>(rs, masked_off(off[j], inv[j], ti));
    }
    tk += BK;
    if (tk >= Ko) { tk = 0; ++ti; }
    knext = k0 + BK;
  }
  __device__ __forceinline__ void store2(unsigned short* hi, unsigned short* lo, const V (&v)[NV]) const { stage2_kc<NV, RP>(hi, lo, r0, kq, v); }
  __device__ __forceinline__ void store(float* Sm, const float4 (&v)[NV]) const { stage_kc<NV, LD, RP>(Sm, r0, kq, v); }
};

template <int TILE, class FMT = F32, int NT = NTHREADS>
struct ConvFilterS2MC : MCGeom<TILE, FMT, NT> {
  typedef MCGeom<TILE, FMT, NT> G;
  typedef typename FMT::T T; typedef typename FMT::V V;
  using G::NV; using G::NVR; using G::RPP; using G::VPR; using G::LD; using G::LDT;
  struct P { const T* w; ConvGeom g; S2Taps t; int cols; int K; long plane; };
  const T* bp; long plane; unsigned voff[NVR]; int cq, kr0, Ko, C, S; long RSC; S2Taps t;
  int ti, tk, knext;
  __device__ __forceinline__ void init(const P& p, int idx0, int tid) {
    cq = tid % VPR; kr0 = tid / VPR; Ko = p.g.Ko; C = p.g.C; S = p.g.S; RSC = (long)p.g.R * p.g.S * p.g.C; t = p.t; plane = p.plane;
    bp = p.w + idx0;
    const bool ok = idx0 + cq * FMT::EPL < p.cols;
    knext = -1; ti = tk = 0;
#pragma unroll
    for (int j = 0; j < NVR; ++j) voff[j] = ok ? (unsigned)(((kr0 + j * RPP) * (int)RSC + cq * FMT::EPL) * FMT::ESZ) : VOFF_OOB;
  }
  __device__ __forceinline__ void load(int k0, V (&v)[NV], bool live = true) {
    if (k0 != knext) { ti = k0 / Ko; tk = k0 - ti * Ko; }
    const int ir = ti / t.ns, is = ti - ir * t.ns;
    const int tap = (ir == 0 ? t.r[0] : t.r[1]) * S + (is == 0 ? t.s[0] : t.s[1]);
#pragma unroll
    for (int pl = 0; pl < FMT::NPL; ++pl) {
      const __amdgpu_buffer_rsrc_t rs = tile_rsrc(bp + pl * plane + ((long)tk * RSC + (long)tap * C), live);
#pragma unroll
      for (int j = 0; j < NVR; ++j) v[pl * NVR + j] = bloadv<FMT>(rs, voff[j]);
    }
    tk += BK;
    if (tk >= Ko) { tk = 0; ++ti; }
    knext = k0 + BK;
  }
  __device__ __forceinline__ void store2(unsigned short* hi, unsigned short* lo, const V (&v)[NV]) const { stage2_mc<NV, LDT, RPP>(hi, lo, kr0, cq, v); }
  __device__ __forceinline__ void store(float* Sm, const float4 (&v)[NV]) const { stage_mc<NV, LD, RPP>(Sm, kr0, cq, v); }
};

// wgrad B operand: k = (n,ho,wo), idx = (r,s,c) -> x[n][ho*st-pad+r][wo*st-pad+s][c]   (c contiguous)
// A lane keeps its (r,s,c) for the whole K loop; its NVR pixels advance by BK per K-tile.
//  * "same" stride-1 convolutions (Ho == H, Wo == W): the pixel index IS k, so the address is linear in k: uniform base
//    k0*C + a fixed lane offset.  1x1: nothing else to do.  3x3: the halo test needs (ho,wo) of each pixel, carried per
//    lane and stepped by (BK / Wo, BK % Wo) with a carry (no division).
//  * otherwise (stride 2): image / row / column of each pixel are carried the same way and the address is rebuilt from
//    them, relative to the image of pixel k0 (wave-uniform, stepped alongside).
template <int TILE, class FMT = F32, int NT = NTHREADS>
struct ConvIm2colMC : MCGeom<TILE, FMT, NT> {
  typedef MCGeom<TILE, FMT, NT> G;
  typedef typename FMT::T T; typedef typename FMT::V V;
  using G::NV; using G::NVR; using G::RPP; using G::VPR; using G::LD; using G::LDT;
  struct P { const T* x; ConvGeom g; int cols; int K; long plane; };
  const T* x; long plane; int cq, kr0, K, H, W, C, Ho, Wo, st, dh, dw, cc, bias; bool ok;
  bool linear, halo;
  unsigned voff[NVR];                      // linear mode: fixed lane offsets from x - bias + k0*C
  int pn[NVR], pho[NVR], pwo[NVR];         // pixel (image, row, column) of this lane's j-th k-row in K-tile `knext`
  int stepq, stepr;                        // BK = stepq * Wo + stepr
  int un, urem, knext;                     // wave-uniform: image of pixel `knext` and its index inside that image
  __device__ __forceinline__ void init(const P& p, int idx0, int tid) {
    cq = tid % VPR; kr0 = tid / VPR; K = p.K; x = p.x; H = p.g.H; W = p.g.W; C = p.g.C; Ho = p.g.Ho; Wo = p.g.Wo; st = p.g.stride;
    plane = p.plane;
    const int col = idx0 + cq * FMT::EPL;
    ok = col < p.cols;
    const int tap = col / C; cc = col - tap * C;
    const int r = tap / p.g.S, s = tap - r * p.g.S;
    dh = r - p.g.pad; dw = s - p.g.pad;
    linear = (st == 1 && Ho == H && Wo == W);
    halo = (p.g.R * p.g.S > 1) || p.g.pad != 0;
    stepq = BK / Wo; stepr = BK - stepq * Wo;
    knext = -1; un = urem = 0;
    bias = (p.g.pad * W + p.g.pad) * C;   // keeps the linear-mode lane offset >= 0
#pragma unroll
    for (int j = 0; j < NVR; ++j) {
      voff[j] = ok ? (unsigned)(((kr0 + j * RPP + dh * W + dw) * C + cc + bias) * FMT::ESZ) : VOFF_OOB;
      pn[j] = pho[j] = pwo[j] = 0;
    }
  }
  __device__ __forceinline__ void seek(int k0) {
    const int HoWo = Ho * Wo;
    un = k0 / HoWo; urem = k0 - un * HoWo;
#pragma unroll
    for (int j = 0; j < NVR; ++j) {
      const int k = k0 + kr0 + j * RPP;
      pwo[j] = k % Wo; const int t = k / Wo; pho[j] = t % Ho; pn[j] = t / Ho;
    }
  }
  __device__ __forceinline__ void load(int k0, V (&v)[NV], bool live = true) {
    if (k0 != knext) seek(k0);
    const bool tail = k0 + BK > K;
    if (linear) {
      unsigned o[NVR];
#pragma unroll
      for (int j = 0; j < NVR; ++j) {
        o[j] = voff[j];
        if (halo || tail) {
          bool valid = ((unsigned)(pho[j] + dh) < (unsigned)H) && ((unsigned)(pwo[j] + dw) < (unsigned)W);
          if (!halo) valid = true;
          if (tail) valid = valid && (k0 + kr0 + j * RPP < K);
          if (!valid) o[j] = VOFF_OOB;
        }
      }
#pragma unroll
      for (int pl = 0; pl < FMT::NPL; ++pl) {
        const __amdgpu_buffer_rsrc_t rs = tile_rsrc(x + pl * plane + ((long)k0 * C - bias), live);
#pragma unroll
        for (int j = 0; j < NVR; ++j) v[pl * NVR + j] = bloadv<FMT>(rs, o[j]);
      }
    } else {
      unsigned o[NVR];
#pragma unroll
      for (int j = 0; j < NVR; ++j) {
        const int hi = pho[j] * st + dh, wi = pwo[j] * st + dw;
        bool valid = ok && ((unsigned)hi < (unsigned)H) && ((unsigned)wi < (unsigned)W);
        if (tail) valid = valid && (k0 + kr0 + j * RPP < K);
        o[j] = valid ? (unsigned)(((((pn[j] - un) * H + hi) * W + wi) * C + cc) * FMT::ESZ) : VOFF_OOB;
      }
#pragma unroll
      for (int pl = 0; pl < FMT::NPL; ++pl) {
        const __amdgpu_buffer_rsrc_t rs = tile_rsrc(x + pl * plane + (long)un * H * W * C, live);
#pragma unroll
        for (int j = 0; j < NVR; ++j) v[pl * NVR + j] = bloadv<FMT>(rs, o[j]);
      }
    }
    if (halo || !linear) {  // step every pixel by BK
#pragma unroll
      for (int j = 0; j < NVR; ++j) {
        int wo = pwo[j] + stepr, ho = pho[j] + stepq;
        if (wo >= Wo) { wo -= Wo; ++ho; }
        int n = pn[j];
        while (ho >= Ho) { ho -= Ho; ++n; }   // one iteration at most unless the image is narrower than BK / Ho
        pwo[j] = wo; pho[j] = ho; pn[j] = n;
      }
      urem += BK;
      const int HoWo = Ho * Wo;
      while (urem >= HoWo) { urem -= HoWo; ++un; }
    }
    knext = k0 + BK;
  }
  __device__ __forceinline__ void store2(unsigned short* hi, unsigned short* lo, const V (&v)[NV]) const { stage2_mc<NV, LDT, RPP>(hi, lo, kr0, cq, v); }
  __device__ __forceinline__ void store(float* Sm, const float4 (&v)[NV]) const { stage_mc<NV, LD, RPP>(Sm, kr0, cq, v); }
};

}  // namespace cxrk
