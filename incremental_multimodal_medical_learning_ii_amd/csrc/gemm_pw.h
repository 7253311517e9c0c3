// 256x256 split-bf16 mainloop for pre-split ("planes") operands: LDS-DMA staging + software-pipelined fragment reads.
// (Included at the end of gemm_core.h.)
//
// Why a second wide kernel.  With planes operands the register-staged 256x256 kernel (gemm_x3w_kernel) has no conversion work
// left, yet it stays at ~35 % of the matrix pipe: its waves read ALL fragments of a K-step, then issue its 24 MFMAs, then
// store the next tile to LDS, and the two waves that share a SIMD do that in lockstep, so the matrix pipe idles during every
// read / store phase (round-1 ablation: MFMAs alone 666 TFLOP/s, + fragment reads 485-520, + stores ~400, + global loads ~310).
// Here
//  * operand tiles go global -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`): no staging registers, no ds_write pass, and the
//    range-checked zero fill of out-of-range lanes (bit 31 of the offset; verified on gfx950 by scripts/probe_ldsdma.hip: a lane
//    outside the descriptor writes ZERO to LDS) still implements tile edges, convolution halos and K tails;
//  * an LDS-DMA piece is lane-linear (wave-uniform base + 16 B x lane), so the bank-conflict-free layouts are obtained by
//    permuting the per-lane SOURCE addresses (cdna_hip_programming.md rule 21): K-contiguous planes are [row][64 B] with the four
//    16-byte k-chunks of row r XOR-ed by (r >> 2) & 3 (conflict-free ds_read_b128 for the 32x32x16 operand map);
//    idx-contiguous planes are [k][2*TILE B] with the 64-byte column groups of k-row k XOR-ed by (k & 3) (conflict-free
//    ds_read_b64_tr_b16 blocks);
//  * the 48 MFMAs of a K-tile are issued as four groups of 12; the fragments of group g+1 are read while group g computes
//    (two register sets), the DMA of the next tile is issued during groups 0 and 1, and the ONE barrier of the K-tile sits
//    before the last group, whose MFMAs then cover the first fragment reads of the next tile: no phase of the loop is without
//    matrix work in flight.
// Hazards (cdna_hip_programming.md, "Read a staged buffer one phase AFTER the wait that retires it"):
//    RAW  tile t+1 is DMA-ed during groups 0-1 of tile t; every wave waits vmcnt(0) and then passes the barrier before any
//         wave reads it;
//    WAR  the DMA of tile t+2 overwrites the buffer of tile t; it is issued after that barrier, which every wave reaches only
//         after lgkmcnt(0) has retired its last fragment reads of tile t (those of group 3, issued in group 2).
#pragma once

namespace cxrk {

// Tile configurations (TILE = rows of an operand tile, NW = waves of the block):
//   256 x 256, 8 waves (2 x 4, 128 x 64 outputs each), one block per CU, 2 x 64 KiB of LDS: the big compute-bound shapes;
//   128 x 128, 4 waves (2 x 2,  64 x 64 outputs each), two blocks per CU, 2 x 32 KiB each: outputs narrower than 256 columns
//              or rows, and every shape whose 256 x 256 tiling would leave most of the chip idle; the two co-resident blocks
//              run out of phase, so one block's epilogue and DMA waits overlap the other's matrix work.
template <int TILE> constexpr int pw_plane_bytes() { return TILE * BK * 2; }   // one bf16 plane of a TILE x BK operand tile

typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, unsigned char* lds_wave_uniform, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)lds_wave_uniform, 16, (int)voff, 0, 0, 0);
}

// ---- fragment reads of the swizzled LDS images ---------------------------------------------------------------------
// K-contiguous plane [row][4 chunks of 16 B]: MFMA 32x32x16 operand (8 bf16 = k 8h .. 8h+7 of k-step kc) of row row0 + (lane & 31)
__device__ __forceinline__ bf16x8 pw_frag_kc(const unsigned char* plane, int row0, int kc, int lane) {
  const int r = row0 + (lane & 31);
  const int q = (2 * kc + (lane >> 5)) ^ ((lane >> 2) & 3);
  return *reinterpret_cast<const bf16x8*>(plane + r * 64 + q * 16);
}
// idx-contiguous plane [k][2*TILE B]: two transposing reads (see frag_mc in gemm_core.h for the lane map).  The 64-byte column
// groups of k-row k are XOR-ed by pw_mc_swz(k): (k & 3) for rows of >= 256 B; rows of 128 B (TILE 64) have two groups only and
// use bit 1 of k, which also separates the four k-rows of a transposing read into four bank windows.
template <int TILE> __device__ __forceinline__ int pw_mc_swz(int k) { return TILE >= 128 ? ((k & 3) << 6) : (((k >> 1) & 1) << 6); }
template <int TILE>
__device__ __forceinline__ bf16x8 pw_frag_mc(const unsigned char* plane, int row0, int kc, int lane) {
  constexpr int RB = 2 * TILE;
  const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
  const int k = kc * 16 + 8 * (g >> 1) + q;                             // (k & 3) == q, also for the second read (k + 4)
  const int colb = ((row0 + 16 * (g & 1) + 4 * pp) * 2) ^ pw_mc_swz<TILE>(q);
  const unsigned char* a0 = plane + k * RB + colb;
  typedef s16x4 __attribute__((address_space(3))) * lds_v4;
  const s16x4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(a0));
  const s16x4 x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(a0 + 4 * RB));
  const s16x8 x = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
  return __builtin_bit_cast(bf16x8, x);
}

// ---- DMA loaders (planes sources only) -----------------------------------------------------------------------------------
// Every loader<TILE, NW>: init(P, idx0, wave, lane); issue(k0, lds_hi, live): this wave's share of the TILE x BK tile.
// K-contiguous images: wave w fills rows [RW*w, RW*(w+1)), RW = TILE / NW, of each plane in NP = RW / 16 pieces of 16 rows;
// lane -> (row 16j + lane/4, slot lane%4), and the slot holds k-chunk q = slot ^ ((row >> 2) & 3) = (lane & 3) ^ ((lane >> 4) & 3)
// (the same for every piece).

// X(idx, k) = ptr[idx*ld + k]
template <int TILE, int NW>
struct DmaDenseKC {
  static constexpr int PLANEB = pw_plane_bytes<TILE>();
  static constexpr bool KC = true;
  struct P { const unsigned short* ptr; long ld; int rows; int K; long plane; };
  const unsigned short* bp; long plane; static constexpr int RW = TILE / NW, NP = RW / 16;
  unsigned voff[NP]; int kq8, K, wave;
  __device__ __forceinline__ void init(const P& p, int idx0, int wave_, int lane) {
    wave = wave_; K = p.K; plane = p.plane;
    bp = p.ptr + (long)idx0 * p.ld;
    kq8 = 8 * ((lane & 3) ^ ((lane >> 4) & 3));
    const int ld = (int)p.ld;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int r = RW * wave + 16 * j + (lane >> 2);
      voff[j] = idx0 + r < p.rows ? (unsigned)((r * ld + kq8) * 2) : VOFF_OOB;
    }
  }
  __device__ __forceinline__ void issue(int k0, unsigned char* lds_hi, bool live) const {
    const unsigned t = (k0 + BK <= K || k0 + kq8 < K) ? 0u : VOFF_OOB;   // K tail (K % 8 == 0)
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      const __amdgpu_buffer_rsrc_t rs = tile_rsrc(bp + pl * plane + k0, live);
#pragma unroll
      for (int j = 0; j < NP; ++j) dma16(rs, lds_hi + pl * PLANEB + (RW * wave + 16 * j) * 64, voff[j] | t);
    }
  }
  static __device__ __forceinline__ bf16x8 frag(const unsigned char* plane_, int row0, int kc, int lane) { return pw_frag_kc(plane_, row0, kc, lane); }
};

// X(idx, k) = ptr[k*ld + idx]: the plane is [BK k-rows][RB = 2*TILE bytes]; a 1-KiB piece covers KPP = 1024 / RB k-rows and the
// plane has 32 / KPP pieces, NP = 32 / KPP / NW per wave; lane -> (k-row lane / LPR, LDS bytes 16*(lane % LPR) ..), LPR = RB / 16,
// which hold the columns at byte (16*(lane % LPR)) ^ pw_mc_swz(k) of the global row.
template <int TILE, int NW>
struct DmaDenseMC {
  static constexpr int PLANEB = pw_plane_bytes<TILE>();
  static constexpr bool KC = false;
  struct P { const unsigned short* ptr; long ld; int cols; int K; long plane; };
  static constexpr int RB = 2 * TILE, KPP = 1024 / RB, LPR = RB / 16, NP = 32 / KPP / NW;
  static_assert(NP >= 1, "tile too small for this wave count");
  const unsigned short* bp; long plane, ld_; unsigned voff[NP]; int kl[NP], K, wave;
  __device__ __forceinline__ void init(const P& p, int idx0, int wave_, int lane) {
    wave = wave_; K = p.K; plane = p.plane; ld_ = p.ld;
    bp = p.ptr + idx0;
    const int ld = (int)p.ld;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int k = (NP * wave + j) * KPP + lane / LPR;
      const int colb = (16 * (lane % LPR)) ^ pw_mc_swz<TILE>(k);
      kl[j] = k;
      voff[j] = idx0 + colb / 2 < p.cols ? (unsigned)(k * ld * 2 + colb) : VOFF_OOB;   // cols % 8 == 0
    }
  }
  __device__ __forceinline__ void issue(int k0, unsigned char* lds_hi, bool live) const {
    const bool full = k0 + BK <= K;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      const __amdgpu_buffer_rsrc_t rs = tile_rsrc(bp + pl * plane + (long)k0 * ld_, live);
#pragma unroll
      for (int j = 0; j < NP; ++j) dma16(rs, lds_hi + pl * PLANEB + (NP * wave + j) * 1024, (full || k0 + kl[j] < K) ? voff[j] : VOFF_OOB);
    }
  }
  static __device__ __forceinline__ bf16x8 frag(const unsigned char* plane_, int row0, int kc, int lane) { return pw_frag_mc<TILE>(plane_, row0, kc, lane); }
};

// ---- convolution gathers (same index algebra as the register loaders in gemm_loaders.h, DMA lane map) -------------------------
// K-contiguous images: a lane serves LDS row r_j = 32*wave + 16*j + lane/4 (j = 0, 1) and the 8 channels kq8.. of the K-tile.

// fprop A operand: idx = (n,ho,wo), k = (r,s,c) -> x[n][ho*st-pad+r][wo*st-pad+s][c]; C % BK == 0 (a K-tile inside one tap)
template <int TILE, int NW>
struct DmaConvIm2colKC {
  static constexpr int PLANEB = pw_plane_bytes<TILE>();
  static constexpr bool KC = true;
  struct P { const unsigned short* x; ConvGeom g; int rows; int K; long plane; };
  const unsigned short* bp; long plane; static constexpr int RW = TILE / NW, NP = RW / 16;
  unsigned off[NP], inv[NP]; int kq8, W, C, S, wave;
  int tr, ts, tc, knext;
  __device__ __forceinline__ void init(const P& p, int idx0, int wave_, int lane) {
    wave = wave_; plane = p.plane; W = p.g.W; C = p.g.C; S = p.g.S;
    const int H = p.g.H;
    kq8 = 8 * ((lane & 3) ^ ((lane >> 4) & 3));
    const int HoWo = p.g.Ho * p.g.Wo;
    const int n_first = idx0 / HoWo;
    const int bias = (p.g.pad * W + p.g.pad) * C;   // keeps off >= 0: the halo reaches `pad` rows / columns before the image
    bp = p.x + (long)n_first * H * W * C - bias;
    knext = -1; tr = ts = tc = 0;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int row = idx0 + RW * wave + 16 * j + (lane >> 2);
      const bool in = row < p.rows;
      const int rowc = in ? row : 0;
      const int wo = rowc % p.g.Wo; const int t = rowc / p.g.Wo; const int ho = t % p.g.Ho; const int n = t / p.g.Ho;
      const int h0 = ho * p.g.stride - p.g.pad, w0 = wo * p.g.stride - p.g.pad;
      off[j] = in ? (unsigned)(((((n - n_first) * H + h0) * W + w0) * C + bias + kq8) * 2) : 0u;
      inv[j] = in ? tap_mask(outside_bits(-h0, H - h0, p.g.R), outside_bits(-w0, W - w0, S), p.g.R, S) : 0xffffffffu;
    }
  }
  __device__ __forceinline__ void issue(int k0, unsigned char* lds_hi, bool live) {
    if (k0 != knext) { const int tap = k0 / C; tc = k0 - tap * C; tr = tap / S; ts = tap - tr * S; }
    const int t = tr * S + ts;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      const __amdgpu_buffer_rsrc_t rs = tile_rsrc(bp + pl * plane + ((long)(tr * W + ts) * C + tc), live);
#pragma unroll
      for (int j = 0; j < NP; ++j) dma16(rs, lds_hi + pl * PLANEB + (RW * wave + 16 * j) * 64, masked_off(off[j], inv[j], t));
    }
    tc += BK;
    if (tc >= C) { tc = 0; if (++ts == S) { ts = 0; ++tr; } }
    knext = k0 + BK;
  }
  static __device__ __forceinline__ bf16x8 frag(const unsigned char* plane_, int row0, int kc, int lane) { return pw_frag_kc(plane_, row0, kc, lane); }
};

// dgrad A operand (stride 1): idx = (n,hi,wi), k = (r,s,ko) -> dy[n][hi+pad-r][wi+pad-s][ko]; Ko % BK == 0
template <int TILE, int NW>
struct DmaConvDgradKC {
  static constexpr int PLANEB = pw_plane_bytes<TILE>();
  static constexpr bool KC = true;
  struct P { const unsigned short* dy; ConvGeom g; int rows; int K; long plane; };
  const unsigned short* bp; long plane; static constexpr int RW = TILE / NW, NP = RW / 16;
  unsigned off[NP], inv[NP]; int kq8, Wo, Ko, R, S, wave;
  int tr, ts, tk, knext;
  __device__ __forceinline__ void init(const P& p, int idx0, int wave_, int lane) {
    wave = wave_; plane = p.plane; Wo = p.g.Wo; Ko = p.g.Ko; R = p.g.R; S = p.g.S;
    const int Ho = p.g.Ho, H = p.g.H, W = p.g.W, pad = p.g.pad;
    kq8 = 8 * ((lane & 3) ^ ((lane >> 4) & 3));
    const int n_first = idx0 / (H * W);
    bp = p.dy + ((long)n_first * Ho * Wo - ((R - 1) * Wo + (S - 1))) * Ko;
    knext = -1; tr = ts = tk = 0;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int row = idx0 + RW * wave + 16 * j + (lane >> 2);
      if (row < p.rows) {
        const int wi = row % W; const int t = row / W; const int hi = t % H; const int n = t / H;
        off[j] = (unsigned)(((((n - n_first) * Ho + hi + pad) * Wo + wi + pad) * Ko + kq8) * 2);
        inv[j] = tap_mask(outside_bits(hi + pad - Ho + 1, hi + pad + 1, R), outside_bits(wi + pad - Wo + 1, wi + pad + 1, S), R, S);
      } else { off[j] = 0; inv[j] = 0xffffffffu; }
    }
  }
  __device__ __forceinline__ void issue(int k0, unsigned char* lds_hi, bool live) {
    if (k0 != knext) { const int tap = k0 / Ko; tk = k0 - tap * Ko; tr = tap / S; ts = tap - tr * S; }
    const int t = tr * S + ts;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      const __amdgpu_buffer_rsrc_t rs = tile_rsrc(bp + pl * plane + ((long)((R - 1 - tr) * Wo + (S - 1 - ts)) * Ko + tk), live);
#pragma unroll
      for (int j = 0; j < NP; ++j) dma16(rs, lds_hi + pl * PLANEB + (RW * wave + 16 * j) * 64, masked_off(off[j], inv[j], t));
    }
    tk += BK;
    if (tk >= Ko) { tk = 0; if (++ts == S) { ts = 0; ++tr; } }
    knext = k0 + BK;
  }
  static __device__ __forceinline__ bf16x8 frag(const unsigned char* plane_, int row0, int kc, int lane) { return pw_frag_kc(plane_, row0, kc, lane); }
};

// stride-2 dgrad A operand, one output-parity class: idx = (n,a,b) on the half-resolution grid, k = (ti,ko)
template <int TILE, int NW>
struct DmaConvDgradS2KC {
  static constexpr int PLANEB = pw_plane_bytes<TILE>();
  static constexpr bool KC = true;
  struct P { const unsigned short* dy; ConvGeom g; S2Taps t; int Hs, Ws; int rows; int K; long plane; };
  const unsigned short* bp; long plane; static constexpr int RW = TILE / NW, NP = RW / 16;
  unsigned off[NP], inv[NP]; int kq8, Wo, Ko, wave; S2Taps t;
  int ti, tk, knext;
  __device__ __forceinline__ void init(const P& p, int idx0, int wave_, int lane) {
    wave = wave_; plane = p.plane; Wo = p.g.Wo; Ko = p.g.Ko; t = p.t;
    const int Ho = p.g.Ho;
    kq8 = 8 * ((lane & 3) ^ ((lane >> 4) & 3));
    const int n_first = idx0 / (p.Hs * p.Ws);
    bp = p.dy + (long)n_first * Ho * Wo * Ko;
    knext = -1; ti = tk = 0;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int row = idx0 + RW * wave + 16 * j + (lane >> 2);
      if (row < p.rows) {
        const int b = row % p.Ws; const int q = row / p.Ws; const int a = q % p.Hs; const int n = q / p.Hs;
        off[j] = (unsigned)(((((n - n_first) * Ho + a) * Wo + b) * Ko + kq8) * 2);
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
  __device__ __forceinline__ void issue(int k0, unsigned char* lds_hi, bool live) {
    if (k0 != knext) { ti = k0 / Ko; tk = k0 - ti * Ko; }
    const int ir = ti / t.ns, is = ti - ir * t.ns;
    const int dr = ir == 0 ? t.dr[0] : t.dr[1], ds = is == 0 ? t.ds[0] : t.ds[1];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      const __amdgpu_buffer_rsrc_t rs = tile_rsrc(bp + pl * plane + ((long)(dr * Wo + ds) * Ko + tk), live);
#pragma unroll
      for (int j = 0; j < NP; ++j) dma16(rs, lds_hi + pl * PLANEB + (RW * wave + 16 * j) * 64, masked_off(off[j], inv[j], ti));
    }
    tk += BK;
    if (tk >= Ko) { tk = 0; ++ti; }
    knext = k0 + BK;
  }
  static __device__ __forceinline__ bf16x8 frag(const unsigned char* plane_, int row0, int kc, int lane) { return pw_frag_kc(plane_, row0, kc, lane); }
};

// idx-contiguous images: a lane serves k-row kl_j = 4*wave + 2*j + lane/32 and the 8 idx at LDS bytes 16*(lane%32), i.e. the
// logical columns (16*(lane%32) ^ ((kl & 3) << 6)) / 2 .. +7 of the tile.

// dgrad B operand: k = (r,s,ko), idx = c -> w[ko][r][s][c]
template <int TILE, int NW>
struct DmaConvFilterMC {
  static constexpr int PLANEB = pw_plane_bytes<TILE>();
  static constexpr bool KC = false;
  struct P { const unsigned short* w; ConvGeom g; int cols; int K; long plane; };
  static constexpr int RB = 2 * TILE, KPP = 1024 / RB, LPR = RB / 16, NP = 32 / KPP / NW;
  const unsigned short* bp; long plane, RSC; unsigned voff[NP]; int Ko, C, wave;
  int tap, tk, knext;
  __device__ __forceinline__ void init(const P& p, int idx0, int wave_, int lane) {
    wave = wave_; plane = p.plane; Ko = p.g.Ko; C = p.g.C; RSC = (long)p.g.R * p.g.S * p.g.C;
    bp = p.w + idx0;
    knext = -1; tap = tk = 0;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int kl = (NP * wave + j) * KPP + lane / LPR;
      const int colb = (16 * (lane % LPR)) ^ pw_mc_swz<TILE>(kl);
      voff[j] = idx0 + colb / 2 < p.cols ? (unsigned)(kl * (int)RSC * 2 + colb) : VOFF_OOB;
    }
  }
  __device__ __forceinline__ void issue(int k0, unsigned char* lds_hi, bool live) {
    if (k0 != knext) { tap = k0 / Ko; tk = k0 - tap * Ko; }
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      const __amdgpu_buffer_rsrc_t rs = tile_rsrc(bp + pl * plane + ((long)tk * RSC + (long)tap * C), live);
#pragma unroll
      for (int j = 0; j < NP; ++j) dma16(rs, lds_hi + pl * PLANEB + (NP * wave + j) * 1024, voff[j]);
    }
    tk += BK;
    if (tk >= Ko) { tk = 0; ++tap; }
    knext = k0 + BK;
  }
  static __device__ __forceinline__ bf16x8 frag(const unsigned char* plane_, int row0, int kc, int lane) { return pw_frag_mc<TILE>(plane_, row0, kc, lane); }
};

template <int TILE, int NW>
struct DmaConvFilterS2MC {
  static constexpr int PLANEB = pw_plane_bytes<TILE>();
  static constexpr bool KC = false;
  struct P { const unsigned short* w; ConvGeom g; S2Taps t; int cols; int K; long plane; };
  static constexpr int RB = 2 * TILE, KPP = 1024 / RB, LPR = RB / 16, NP = 32 / KPP / NW;
  const unsigned short* bp; long plane, RSC; unsigned voff[NP]; int Ko, C, S, wave; S2Taps t;
  int ti, tk, knext;
  __device__ __forceinline__ void init(const P& p, int idx0, int wave_, int lane) {
    wave = wave_; plane = p.plane; Ko = p.g.Ko; C = p.g.C; S = p.g.S; RSC = (long)p.g.R * p.g.S * p.g.C; t = p.t;
    bp = p.w + idx0;
    knext = -1; ti = tk = 0;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int kl = (NP * wave + j) * KPP + lane / LPR;
      const int colb = (16 * (lane % LPR)) ^ pw_mc_swz<TILE>(kl);
      voff[j] = idx0 + colb / 2 < p.cols ? (unsigned)(kl * (int)RSC * 2 + colb) : VOFF_OOB;
    }
  }
  __device__ __forceinline__ void issue(int k0, unsigned char* lds_hi, bool live) {
    if (k0 != knext) { ti = k0 / Ko; tk = k0 - ti * Ko; }
    const int ir = ti / t.ns, is = ti - ir * t.ns;
    const int tap = (ir == 0 ? t.r[0] : t.r[1]) * S + (is == 0 ? t.s[0] : t.s[1]);
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      const __amdgpu_buffer_rsrc_t rs = tile_rsrc(bp + pl * plane + ((long)tk * RSC + (long)tap * C), live);
#pragma unroll
      for (int j = 0; j < NP; ++j) dma16(rs, lds_hi + pl * PLANEB + (NP * wave + j) * 1024, voff[j]);
    }
    tk += BK;
    if (tk >= Ko) { tk = 0; ++ti; }
    knext = k0 + BK;
  }
  static __device__ __forceinline__ bf16x8 frag(const unsigned char* plane_, int row0, int kc, int lane) { return pw_frag_mc<TILE>(plane_, row0, kc, lane); }
};

// wgrad B operand: k = (n,ho,wo), idx = (r,s,c) -> x[n][ho*st-pad+r][wo*st-pad+s][c]   (see ConvIm2colMC in gemm_loaders.h)
template <int TILE, int NW>
struct DmaConvIm2colMC {
  static constexpr int PLANEB = pw_plane_bytes<TILE>();
  static constexpr bool KC = false;
  struct P { const unsigned short* x; ConvGeom g; int cols; int K; long plane; };
  static constexpr int RB = 2 * TILE, KPP = 1024 / RB, LPR = RB / 16, NP = 32 / KPP / NW;
  const unsigned short* x; long plane; int K, H, W, C, Ho, Wo, st, bias, wave; bool linear, halo;
  unsigned voff[NP]; int kl[NP];
  int dhj[NP], dwj[NP], ccv[NP]; bool okv[NP];     // (r - pad, s - pad, c) and column validity of each piece (the pieces differ)
  int pn[NP], pho[NP], pwo[NP];                    // pixel (image, row, column) of each piece's k-row in K-tile `knext`
  int stepq, stepr, un, urem, knext;
  __device__ __forceinline__ void init(const P& p, int idx0, int wave_, int lane) {
    wave = wave_; plane = p.plane; K = p.K; x = p.x; H = p.g.H; W = p.g.W; C = p.g.C; Ho = p.g.Ho; Wo = p.g.Wo; st = p.g.stride;
    // the 8 idx of a lane share (r,s) and are 8 consecutive channels (C % 8 == 0); the XOR moves whole 64-byte groups
    linear = (st == 1 && Ho == H && Wo == W);
    halo = (p.g.R * p.g.S > 1) || p.g.pad != 0;
    stepq = BK / Wo; stepr = BK - stepq * Wo;
    knext = -1; un = urem = 0;
    bias = (p.g.pad * W + p.g.pad) * C;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      kl[j] = (NP * wave + j) * KPP + lane / LPR;
      const int colb = (16 * (lane % LPR)) ^ pw_mc_swz<TILE>(kl[j]);
      const int col = idx0 + colb / 2;
      const bool okj = col < p.cols;
      const int tap = col / C, ccj = col - tap * C;
      const int r = tap / p.g.S, s = tap - r * p.g.S;
      dhj[j] = r - p.g.pad; dwj[j] = s - p.g.pad; ccv[j] = ccj; okv[j] = okj;
      voff[j] = okj ? (unsigned)(((kl[j] + dhj[j] * W + dwj[j]) * C + ccj + bias) * 2) : VOFF_OOB;
      pn[j] = pho[j] = pwo[j] = 0;
    }
  }
  __device__ __forceinline__ void seek(int k0) {
    const int HoWo = Ho * Wo;
    un = k0 / HoWo; urem = k0 - un * HoWo;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int k = k0 + kl[j];
      pwo[j] = k % Wo; const int t = k / Wo; pho[j] = t % Ho; pn[j] = t / Ho;
    }
  }
  __device__ __forceinline__ void issue(int k0, unsigned char* lds_hi, bool live) {
    if (k0 != knext) seek(k0);
    const bool tail = k0 + BK > K;
    unsigned o[NP];
    if (linear) {
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        o[j] = voff[j];
        if (halo || tail) {
          bool valid = !halo || (((unsigned)(pho[j] + dhj[j]) < (unsigned)H) && ((unsigned)(pwo[j] + dwj[j]) < (unsigned)W));
          if (tail) valid = valid && (k0 + kl[j] < K);
          if (!valid) o[j] = VOFF_OOB;
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        const int hi = pho[j] * st + dhj[j], wi = pwo[j] * st + dwj[j];
        bool valid = okv[j] && ((unsigned)hi < (unsigned)H) && ((unsigned)wi < (unsigned)W);
        if (tail) valid = valid && (k0 + kl[j] < K);
        o[j] = valid ? (unsigned)(((((pn[j] - un) * H + hi) * W + wi) * C + ccv[j]) * 2) : VOFF_OOB;
      }
    }
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      const __amdgpu_buffer_rsrc_t rs = linear ? tile_rsrc(x + pl * plane + ((long)k0 * C - bias), live) : tile_rsrc(x + pl * plane + (long)un * H * W * C, live);
#pragma unroll
      for (int j = 0; j < NP; ++j) dma16(rs, lds_hi + pl * PLANEB + (NP * wave + j) * 1024, o[j]);
    }
    if (halo || !linear) {
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        int wo = pwo[j] + stepr, ho = pho[j] + stepq;
        if (wo >= Wo) { wo -= Wo; ++ho; }
        int n = pn[j];
        while (ho >= Ho) { ho -= Ho; ++n; }
        pwo[j] = wo; pho[j] = ho; pn[j] = n;
      }
      urem += BK;
      const int HoWo = Ho * Wo;
      while (urem >= HoWo) { urem -= HoWo; ++un; }
    }
    knext = k0 + BK;
  }
  static __device__ __forceinline__ bf16x8 frag(const unsigned char* plane_, int row0, int kc, int lane) { return pw_frag_mc<TILE>(plane_, row0, kc, lane); }
};

// ---- kernel -----------------------------------------------------------------------------------------------------------
// WGM x WGN waves, each WTM x 2 blocks of 32 x 32 outputs (WTM = 4: 128 x 64 per wave, WTM = 2: 64 x 64).
#ifndef CXRK_PW_DMA_EARLY
#define CXRK_PW_DMA_EARLY (-1)   // experiment override of PwCfg::DMA_EARLY (0 / 1), scripts/tune_pw.hip
#endif
template <int WGM_, int WGN_, int WTM_>
struct PwCfg {
  static constexpr int WGM = WGM_, WGN = WGN_, WTM = WTM_, WTN = 2;
  static constexpr int NW = WGM * WGN, NT = 64 * NW;
  static constexpr int TM = WGM * WTM * 32, TN = WGN * WTN * 32;
  static constexpr int PLANE_A = pw_plane_bytes<TM>(), PLANE_B = pw_plane_bytes<TN>();
  static constexpr int STAGE = 2 * (PLANE_A + PLANE_B);        // A hi | A lo | B hi | B lo
  static constexpr int BLOCKS_PER_CU = (2 * STAGE <= 80 * 1024 && NW <= 4) ? 2 : 1;
  static constexpr int WAVES_PER_SIMD = NW * BLOCKS_PER_CU / 4;
  // When the LDS-DMA of a K-tile is issued.  false: tile t+1, spread over the first MFMA groups of tile t (in flight for the rest of
  // that tile, the issue instructions hidden behind MFMAs of the other wave on the SIMD).  true: tile t+2, right behind tile t's
  // barrier into the buffer it vacated (in flight for a whole tile, but issued where all waves are in step, so nothing hides it).
  // Measured (scripts/tune_pw.hip, r2g): four-group tiles 3604 vs 4152 cycles per K-tile -> late; two-group tiles, whose K-tile
  // is too short to cover the load latency, 193 vs 231 TFLOP/s on 12544x256x2304 -> early.
  static constexpr bool DMA_EARLY = CXRK_PW_DMA_EARLY >= 0 ? CXRK_PW_DMA_EARLY != 0 : WTM < 4;
  static_assert(NW * 32 * 64 * 4 <= 2 * STAGE, "operand LDS too small to stage the epilogue");
};
typedef PwCfg<2, 4, 4> Pw256;   // 256 x 256, 8 waves, 128 KiB LDS, one block per CU
typedef PwCfg<2, 2, 2> Pw128;   // 128 x 128, 4 waves,  64 KiB LDS, two blocks per CU
typedef PwCfg<4, 1, 2> Pw256x64;   // 256 x 64 (outputs of <= 64 columns), 4 waves, 80 KiB LDS, two blocks per CU
typedef PwCfg<1, 4, 2> Pw64x256;   // 64 x 256 (<= 64 rows: weight gradients of <= 64 filters), 4 waves, 80 KiB LDS, two blocks per CU

struct PwFrag { bf16x8 h[2], l[2]; };   // two 32-row (or 32-column) blocks of an operand, hi / lo

template <class L>
__device__ __forceinline__ void pw_read(PwFrag& f, const unsigned char* plane_hi, int plane_bytes, int row0, int kc, int lane) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    f.h[i] = L::frag(plane_hi, row0 + i * 32, kc, lane);
    f.l[i] = L::frag(plane_hi + plane_bytes, row0 + i * 32, kc, lane);
  }
}
__device__ __forceinline__ void pw_mfma12(f32x16 (&acc)[2][2], const PwFrag& a, const PwFrag& b) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l[i], b.h[j], acc[i][j], 0, 0, 0);
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h[i], b.l[j], acc[i][j], 0, 0, 0);
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h[i], b.h[j], acc[i][j], 0, 0, 0);
    }   // (term-major order, dependent MFMAs four apart, measured the same: 3584 cycles per K-tile either way)
}

#ifndef CXRK_PW_SCHED
#define CXRK_PW_SCHED 1   // 1: pin the group boundaries (reads of group g+1 stay in front of the MFMAs of group g)
#endif
#ifdef CXRK_PW_EXP_NODMA      // experiment (scripts/tune_pw.hip only): what the in-loop DMA costs — results are then wrong
struct PwNoIssue { __device__ __forceinline__ void operator()(int, unsigned char*, bool) const {} };
#define PW_LOOP_ISSUE(l) PwNoIssue()
#else
#define PW_LOOP_ISSUE(l) (l).issue
#endif
#define PW_FENCE() do { if (CXRK_PW_SCHED) __builtin_amdgcn_sched_barrier(0); } while (0)

// Schedule of one K-tile t (buffer `st`, the other buffer `nx`), NG = WTM groups of 12 MFMAs (a 64 x 64 output block, one k-step):
//   groups 0 .. NG-2   read the fragments of the next group, then issue this group's MFMAs (CFG::DMA_EARLY == false: the LDS-DMA
//                      of tile t+1 into `nx` is issued here too, A in group 0 and B in group 1);
//   group  NG-1        wait for this wave's outstanding LDS reads (the last of tile t) and LDS-DMA pieces (tile t+1), barrier,
//                      (DMA_EARLY: issue the DMA of tile t+2 into `st`, which tile t has just vacated,) read the first fragments
//                      of tile t+1 from `nx`, then the last MFMAs of tile t.
// One barrier per K-tile: it orders "every wave has read all of tile t" before "anyone overwrites `st`" and "every wave's pieces
// of tile t+1 have landed" before "anyone reads `nx`".
template <class CFG, class LA, class LB>
__global__ __launch_bounds__(CFG::NT, CFG::WAVES_PER_SIMD) void gemm_pw_kernel(typename LA::P pa, typename LB::P pb, EpiParams ep, int M, int N,
                                                                               int K, int nMt, int nNt, int nZ, int kchunk) {
  constexpr int STAGE = CFG::STAGE, PA = CFG::PLANE_A, PB = CFG::PLANE_B, WTM = CFG::WTM;
  constexpr bool EARLY = CFG::DMA_EARLY;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE];
  int mt, nt, z;
  tile_coords(nMt, nNt, nZ, mt, nt, z);
  const int m0 = mt * CFG::TM, n0 = nt * CFG::TN;
  const int kbeg = z * kchunk;
  const int kend = min(K, kbeg + kchunk);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / CFG::WGN, wn = wave % CFG::WGN;
  const int arow = wm * (WTM * 32), bcol = wn * 64;

#define PW_STAMP(i) do { if (ep.stamps && tid == 0) { ep.stamps[(long)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
                            if ((i) == 0 || (i) == 4) ep.stamps[(long)blockIdx.x * 8 + 5 + (i) / 4] = __builtin_amdgcn_s_memrealtime(); } } while (0)
  PW_STAMP(0);
  LA la; LB lb;
  la.init(pa, m0, wave, lane);
  lb.init(pb, n0, wave, lane);

  f32x16 acc[2][2][2];             // [slab][i][j]: rows 0-63 / 64-127 of the wave's outputs (slab 1 is dead when WTM == 2)
  f32x16 (&acc0)[2][2] = acc[0];
  f32x16 (&acc1)[2][2] = acc[1];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) { acc0[i][j][e] = 0.f; acc1[i][j][e] = 0.f; }

  la.issue(kbeg, smem, kbeg < kend);
  lb.issue(kbeg, smem + 2 * PA, kbeg < kend);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (EARLY) {
    la.issue(kbeg + BK, smem + STAGE, kbeg + BK < kend);
    lb.issue(kbeg + BK, smem + STAGE + 2 * PA, kbeg + BK < kend);
  }
  PW_STAMP(1);

  PwFrag A0, A1, B0, B1;
  pw_read<LA>(A0, smem, PA, arow, 0, lane);
  pw_read<LB>(B0, smem + 2 * PA, PB, bcol, 0, lane);

  int cur = 0;
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    unsigned char* st = smem + cur * STAGE;
    unsigned char* nx = smem + (cur ^ 1) * STAGE;
    const bool more2 = k0 + 2 * BK < kend, more = k0 + BK < kend;
    if constexpr (WTM == 4) {
      // group 0: rows 0-63, k-step 0
      pw_read<LA>(A1, st, PA, arow + 64, 0, lane);
      if (!EARLY) PW_LOOP_ISSUE(la)(k0 + BK, nx, more);
      PW_FENCE();
      pw_mfma12(acc0, A0, B0);
      PW_FENCE();
      // group 1: rows 64-127, k-step 0
      pw_read<LA>(A0, st, PA, arow, 1, lane);
      pw_read<LB>(B1, st + 2 * PA, PB, bcol, 1, lane);
      if (!EARLY) PW_LOOP_ISSUE(lb)(k0 + BK, nx + 2 * PA, more);   // (both in group 0: 3683 vs 3605 cycles per K-tile, r2n)
      PW_FENCE();
      pw_mfma12(acc1, A1, B0);
      PW_FENCE();
      // group 2: rows 0-63, k-step 1
      pw_read<LA>(A1, st, PA, arow + 64, 1, lane);
      PW_FENCE();
      pw_mfma12(acc0, A0, B1);
      PW_FENCE();
      // group 3: rows 64-127, k-step 1 — behind the K-tile's one barrier (see the hazard notes in the header comment)
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (EARLY) { PW_LOOP_ISSUE(la)(k0 + 2 * BK, st, more2); PW_LOOP_ISSUE(lb)(k0 + 2 * BK, st + 2 * PA, more2); }
      pw_read<LA>(A0, nx, PA, arow, 0, lane);
      pw_read<LB>(B0, nx + 2 * PA, PB, bcol, 0, lane);
      PW_FENCE();
      pw_mfma12(acc1, A1, B1);
      PW_FENCE();
    } else {
      // group 0: k-step 0
      pw_read<LA>(A1, st, PA, arow, 1, lane);
      pw_read<LB>(B1, st + 2 * PA, PB, bcol, 1, lane);
      if (!EARLY) { PW_LOOP_ISSUE(la)(k0 + BK, nx, more); PW_LOOP_ISSUE(lb)(k0 + BK, nx + 2 * PA, more); }
      PW_FENCE();
      pw_mfma12(acc0, A0, B0);
      PW_FENCE();
      // group 1: k-step 1 — behind the barrier
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (EARLY) { PW_LOOP_ISSUE(la)(k0 + 2 * BK, st, more2); PW_LOOP_ISSUE(lb)(k0 + 2 * BK, st + 2 * PA, more2); }
      pw_read<LA>(A0, nx, PA, arow, 0, lane);
      pw_read<LB>(B0, nx + 2 * PA, PB, bcol, 0, lane);
      PW_FENCE();
      pw_mfma12(acc0, A1, B1);
      PW_FENCE();
    }
    cur ^= 1;
  }
  PW_STAMP(2);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();   // every wave is done with the operand buffers: they become the epilogue's transpose staging
  float* stg = reinterpret_cast<float*>(smem) + wave * (32 * 64);
  // planes operands are always 16-byte aligned (launch_gemm_pw refuses anything else): the fast epilogue only.  The wave's WTM / 2
  // slabs of 64 rows go through ONE pipelined pass (side inputs of the next 32-row half requested while this one is stored).
  epi_pw_dispatch<0, WTM / 2>(ep.kind, acc, ep, stg, M, N, m0 + arow, n0 + bcol, mt * (CFG::TM / 64) + wm * (WTM / 2), z, lane);
  PW_STAMP(3);
  if (ep.stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); PW_STAMP(4); }
#undef PW_STAMP
}

template <class CFG, class LA, class LB>
static int launch_gemm_pw(const typename LA::P& pa, const typename LB::P& pb, const EpiParams& ep, int M, int N, int K, int splitk,
                          hipStream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0) return CXRK_ERR_ARG;
  const int nMt = ceil_div(M, CFG::TM), nNt = ceil_div(N, CFG::TN);
  int kchunk = K;
  if (splitk > 1) { kchunk = ceil_div(ceil_div(K, splitk), BK) * BK; splitk = ceil_div(K, kchunk); }
  else splitk = 1;
  dim3 grid((unsigned)(nMt * nNt * splitk), 1, 1);
  EpiParams e = ep;
  if (!prep_epilogue(e, M, N, splitk) || !e.fast) return CXRK_ERR_ARG;
  hipLaunchKernelGGL((gemm_pw_kernel<CFG, LA, LB>), grid, dim3(CFG::NT), 0, stream, pa, pb, e, M, N, K, nMt, nNt, splitk, kchunk);
  CXRK_LAUNCH_CHECK();
  return splitk;
}

}  // namespace cxrk
