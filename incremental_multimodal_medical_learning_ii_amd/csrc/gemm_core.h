// fp32 MFMA GEMM mainloop shared by every dense / implicit-GEMM op of the path.
//
//   C[m][n] = epilogue( sum_k A(m,k) * B(k,n) )
//
// * v_mfma_f32_32x32x2_f32: exact fp32 (a k-ordered fmaf chain), 64 FLOP/clk/SIMD -> 157 TFLOP/s chip peak
//   (MI355X_MICROARCH.md "Peak FP32 (matrix)").  fp32 keeps the 1e-3 parity bar of the north star with room.
// * 256 threads = 4 waves, each wave owns a 64x64 output sub-tile = 2x2 MFMA tiles (64 accumulator VGPRs).
//   Tile shapes: WMxWN waves -> (64*WM) x (64*WN); 2x2 = 128x128 (default), 4x1 = 256x64, 1x4 = 64x256.
// * Operand tiles are staged global -> registers -> LDS as [k][idx] (idx contiguous, +4 pad) so the MFMA
//   operand read is a conflict-free ds_read_b32 (32 consecutive dwords per half-wave); the next K-tile's global
//   loads are issued before the MFMA block of the current one (register prefetch, one LDS buffer).
// * "Loaders" turn an (idx, k) tile coordinate into global addresses: dense K-contiguous, dense idx-contiguous,
//   and the three NHWC convolution gathers (im2col for fprop, strided-tap gather for dgrad, pixel gather for
//   wgrad).  All of them load 16 B per lane.
// * blockIdx -> tile mapping is XCD-aware (bijective T1 remap): each XCD owns a contiguous chunk of the M-panel-major
//   tile list, so the A panel is re-read from that XCD's L2 rather than from HBM.
#pragma once
#include "cxrk_common.h"
#include <cstdlib>

#ifndef CXRK_ABL
#define CXRK_ABL 0  // ablation switch for scripts/tune_gemm.hip only; the library is always built with 0
#endif

namespace cxrk {

constexpr int BK = 32;
constexpr int NTHREADS = 256;
constexpr int LPAD = 4;

struct EpiParams {
  float* C; long ldc;
  const float* bias;             // per column n, or null
  const float* R; long ldr;      // residual added before the activation, or null
  const float* aux; long ldaux;  // auxmode 1: v *= (aux > 0); auxmode 2: v *= gelu'(aux)
  int auxmode;                   // auxmode 3 (planes epilogue): v *= bit of maskin
  float* C2; long ldc2;          // optional copy of the pre-activation value
  int act;                       // 0 none, 1 relu, 2 gelu(erf)
  float alpha;                   // scales the accumulator
  long slab_stride;              // split-K: slab z is written at C + z*slab_stride (epilogue must be plain)
  // optional output-row remap (stride-2 dgrad parity classes): GEMM row (n,a,b) -> pixel (n, 2a+ph, 2b+pw) of [N,H,W]
  int rm_on, rm_Hs, rm_Ws, rm_H, rm_W, rm_ph, rm_pw;
  int vec;  // set by launch_gemm: every pointer / leading dimension allows 16-byte row accesses
  int nt;   // set by launch_gemm: store C non-temporally (large outputs)
  // optional fused column sums of the stored values over the rows of this launch (the BatchNorm beta gradient of the unit
  // whose output gradient this launch produces): per-wave partials go to colsum_part[(mt*WM + wm)][N]; a second kernel adds
  // them up (deterministic).  Needs vec.
  float* colsum_part;
  // ---- "planes" operands (split-bf16 storage, see gemm_loaders.h): any of these selects the 8-columns-per-lane epilogue
  unsigned short* Cp; long cplane;          // output as bf16 hi/lo planes (hi at Cp, lo at Cp + cplane; row stride ldc) instead of C
  const unsigned short* Rp; long rplane;    // residual given as planes (row stride ldr) instead of R
  const unsigned char* maskin; long ldmaskin;   // auxmode 3: ReLU decision bits, byte [row][col / 8], bit col % 8
  unsigned char* maskout; long ldmaskout;       // act 1: also store the ReLU decision bits of this launch's output
  int pl;   // set by launch_gemm: a planes / bit-mask operand is present
  int fast; // set by launch_gemm: every operand allows the 8-columns-per-lane epilogue (gemm_epilogue.h)
  int kind; // set by launch_gemm: index of the compiled-in feature set (EPI_KINDS), or -1 = generic
  // diagnostics only (scripts/tune_pw.hip): per-block s_memtime stamps [block][8]; null in the library
  unsigned long long* stamps;
  // optional COMPACT planes residual (the data gradient of a 1x1 / stride-2 projection shortcut; LDS-DMA kernels only): the output
  // rows are the pixels of [N, rs2_H, rs2_W], Rp holds rows for the pixels with even (h, w) only, as [N, rs2_Ho, rs2_Wo] — the
  // other pixels get nothing.  (At the end of the block: the register-staged kernels' code does not move, see gemm_epilogue.h epi64.)
  int rs2_on, rs2_H, rs2_W, rs2_Ho, rs2_Wo;
};

__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// ---- split-bf16 ("bf16x3") helpers --------------------------------------------------------------------------------
// x = hi + lo + O(2^-17 |x|) with hi = bf16(x), lo = bf16(x - hi): the product a*b is taken as
// a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on the bf16 MFMA (fp32 accumulate), dropping only a_lo*b_lo (2^-16 relative).
constexpr int LDH = BK + 8;  // LDS row stride of a bf16 operand tile, in halfwords (80 B: conflict-free ds_read_b128)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split4(const float4& v, uint2& hi, uint2& lo) {
  split2(v.x, v.y, hi.x, lo.x);
  split2(v.z, v.w, hi.y, lo.y);
}
// K-contiguous operand: 4 consecutive k of one row -> one 8-byte store per plane
__device__ __forceinline__ void store2_kc(unsigned short* hi, unsigned short* lo, int row, int k, const float4& v) {
  uint2 h, l; split4(v, h, l);
  *reinterpret_cast<uint2*>(hi + row * LDH + k) = h;
  *reinterpret_cast<uint2*>(lo + row * LDH + k) = l;
}
// idx-contiguous operand: the plane keeps the global orientation [k][idx] (row stride TILE+32 halfwords), so the 4
// consecutive idx of a float4 are ONE 8-byte store per plane; the k-major -> MFMA-operand transpose is done by the
// reads (ds_read_b64_tr_b16, see frag_mc).
template <int LDT>
__device__ __forceinline__ void store2_mc(unsigned short* hi, unsigned short* lo, int idx, int k, const float4& v) {
  uint2 h, l; split4(v, h, l);
  *reinterpret_cast<uint2*>(hi + k * LDT + idx) = h;
  *reinterpret_cast<uint2*>(lo + k * LDT + idx) = l;
}

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// MFMA 32x32x16 operand fragment (8 bf16: k = 8h..8h+7 of row/col r) for the 32-wide tile starting at `row0`.
// K-contiguous plane [idx][LDH]: one ds_read_b128.
__device__ __forceinline__ bf16x8 frag_kc(const unsigned short* plane, int row0, int kc, int lane) {
  return *reinterpret_cast<const bf16x8*>(plane + (row0 + (lane & 31)) * LDH + kc * 16 + 8 * (lane >> 5));
}
// idx-contiguous plane [k][LDT]: two transposing reads.  ds_read_b64_tr_b16 works per 16-lane group on a 4(k) x 16(idx)
// block: lane 4q+p supplies the address of row q, columns 4p..4p+3, and lane i receives column i of the 4 rows
// (cdna_hip_programming.md T10).  Group g = lane>>4 covers idx 16*(g&1).. and k 8*(g>>1).. (+4 for the second read).
// LDT = TILE+32 puts the four rows of a block in disjoint 16-dword bank windows (conflict-free).  EXEC is all ones here.
template <int LDT>
__device__ __forceinline__ bf16x8 frag_mc(const unsigned short* plane, int row0, int kc, int lane) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
  const unsigned short* a0 = plane + (kc * 16 + 8 * (g >> 1) + q) * LDT + row0 + 16 * (g & 1) + 4 * pp;
  typedef s16x4 __attribute__((address_space(3))) * lds_v4;
  const s16x4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(a0));
  const s16x4 x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(a0 + 4 * LDT));
  const s16x8 x = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
  return __builtin_bit_cast(bf16x8, x);
}

}  // namespace cxrk
#include "gemm_loaders.h"
namespace cxrk {

// ---------------------------------------------------------------------------------------------------------------
// Epilogues shared by the fp32 and the split-bf16 mainloops (both MFMA shapes have the same 32x32 C/D map).
// C/D map of the 32x32 MFMAs: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5), i.e. a lane owns ONE column — stored
// straight from the accumulators that is 64 dword stores per lane in 128-B pieces.  Instead each wave transposes its 64x64
// sub-tile through the (now free) operand LDS in two 32-row passes and leaves with whole 16-byte pieces of a row per lane;
// the residual / mask / GELU' side inputs are read the same way.  This is what the HBM-bound shapes (1x1 convolutions with
// K = 64) pay for.
// `st` = this wave's 32x64-float staging area in LDS (no other wave touches it); `part` = index of this 64-row slab among the
// fused column-sum partials (row0 / 64 over the padded row range).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ long epi_row(const EpiParams& ep, int grow) {
  if (!ep.rm_on) return grow;
  const int b_ = grow % ep.rm_Ws; const int q_ = grow / ep.rm_Ws; const int a_ = q_ % ep.rm_Hs; const int n_ = q_ / ep.rm_Hs;
  return ((long)n_ * ep.rm_H + 2 * a_ + ep.rm_ph) * ep.rm_W + 2 * b_ + ep.rm_pw;
}

// fp32 operands everywhere: a lane owns 4 consecutive columns (one float4), 16 lanes a 64-column row, 4 rows per pass.
__device__ __forceinline__ void gemm_epilogue64_f32(f32x16 (&acc)[2][2], const EpiParams& ep, float* st, int M, int N, int row0,
                                                    int col0, int part, int z, int lane) {
  const int r = lane & 31, h = lane >> 5;
  float* C = ep.C + (long)z * ep.slab_stride;
  const int c4 = lane & 15, rq = lane >> 4;
  const int col = col0 + c4 * 4;
  float4 bv = zero4();
  if (ep.bias && col < N) {
    if (ep.vec) bv = *reinterpret_cast<const float4*>(ep.bias + col);
    else { bv.x = ep.bias[col]; if (col + 1 < N) bv.y = ep.bias[col + 1]; if (col + 2 < N) bv.z = ep.bias[col + 2]; if (col + 3 < N) bv.w = ep.bias[col + 3]; }
  }
  float bs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) st[((e & 3) + 8 * (e >> 2) + 4 * h) * 64 + j * 32 + r] = acc[i][j][e];
    __builtin_amdgcn_wave_barrier();  // LDS serves one wave's accesses in issue order; keep the compiler from reordering
    // Side inputs first, four rows at a time: the stores to C may alias them as far as the compiler knows, so left in
    // program order every load would wait behind the previous row's store (one HBM round trip per row).
#pragma unroll
    for (int tg = 0; tg < 2; ++tg) {
      long rows[4]; bool live[4];
      float4 pr[4], pa[4];   // residual and ReLU / GELU' source
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int rl = (tg * 4 + u) * 4 + rq;
        const int grow = row0 + i * 32 + rl;
        live[u] = grow < M && col < N;
        const long row = epi_row(ep, grow);
        rows[u] = row;
        pr[u] = pa[u] = zero4();
        if (ep.vec && live[u]) {
          if (ep.R) pr[u] = *reinterpret_cast<const float4*>(ep.R + row * ep.ldr + col);
          if (ep.auxmode) pa[u] = *reinterpret_cast<const float4*>(ep.aux + row * ep.ldaux + col);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int rl = (tg * 4 + u) * 4 + rq;
        const float4 v4 = *reinterpret_cast<const float4*>(st + rl * 64 + c4 * 4);
        if (!live[u]) continue;
        const long row = rows[u];
        float v[4] = {ep.alpha * v4.x + bv.x, ep.alpha * v4.y + bv.y, ep.alpha * v4.z + bv.z, ep.alpha * v4.w + bv.w};
        if ((CXRK_ABL == 5 || CXRK_ABL == 6) && v[0] != 12345.678f) continue;  // ablation: drop the epilogue traffic
        if (ep.vec) {
          if (ep.R) { v[0] += pr[u].x; v[1] += pr[u].y; v[2] += pr[u].z; v[3] += pr[u].w; }
          if (ep.C2) *reinterpret_cast<float4*>(ep.C2 + row * ep.ldc2 + col) = make_float4(v[0], v[1], v[2], v[3]);
          if (ep.act == 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
          } else if (ep.act == 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = gelu_erf(v[q]);
          }
          if (ep.auxmode) {
            const float ax[4] = {pa[u].x, pa[u].y, pa[u].z, pa[u].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = ep.auxmode == 1 ? (ax[q] > 0.f ? v[q] : 0.f) : v[q] * gelu_erf_grad(ax[q]);
          }
          if (ep.colsum_part) {
#pragma unroll
            for (int q = 0; q < 4; ++q) bs[q] += v[q];
          }
          if (ep.nt) { const f32x4 o = {v[0], v[1], v[2], v[3]}; __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(C + row * ep.ldc + col)); }
          else *reinterpret_cast<float4*>(C + row * ep.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (col + q >= N) break;
            float x = v[q];
            if (ep.R) x += ep.R[row * ep.ldr + col + q];
            if (ep.C2) ep.C2[row * ep.ldc2 + col + q] = x;
            if (ep.act == 1) x = fmaxf(x, 0.f);
            else if (ep.act == 2) x = gelu_erf(x);
            if (ep.auxmode == 1) x = ep.aux[row * ep.ldaux + col + q] > 0.f ? x : 0.f;
            else if (ep.auxmode == 2) x *= gelu_erf_grad(ep.aux[row * ep.ldaux + col + q]);
            C[row * ep.ldc + col + q] = x;
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (ep.colsum_part) {  // add the four row-quads of the wave (lanes c4, c4+16, c4+32, c4+48), lanes 0..15 write
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float t = bs[q];
      t += __shfl_xor(t, 16, 64);
      t += __shfl_xor(t, 32, 64);
      bs[q] = t;
    }
    if (rq == 0 && col < N) *reinterpret_cast<float4*>(ep.colsum_part + (long)part * N + col) = make_float4(bs[0], bs[1], bs[2], bs[3]);
  }
}

}  // namespace cxrk
#include "gemm_epilogue.h"
namespace cxrk {

__device__ __forceinline__ void gemm_epilogue64(f32x16 (&acc)[2][2], const EpiParams& ep, float* st, int M, int N, int row0,
                                                int col0, int part, int z, int lane) {
  if (ep.fast) epi64_dispatch<0>(ep.kind, acc, ep, st, M, N, row0, col0, part, z, lane);
  else gemm_epilogue64_f32(acc, ep, st, M, N, row0, col0, part, z, lane);   // odd widths / unaligned fp32 operands
}

template <int WM, int WN>
__device__ __forceinline__ void gemm_epilogue(f32x16 (&acc)[2][2], const EpiParams& ep, float* smem, int M, int N, int m0,
                                              int n0, int mt, int z, int wave, int lane) {
  const int wm = wave / WN, wn = wave % WN;
  gemm_epilogue64(acc, ep, smem + wave * (32 * 64), M, N, m0 + wm * 64, n0 + wn * 64, mt * WM + wm, z, lane);
}

// XCD-aware tile mapping (speed only, never correctness).  Workgroups are dealt round-robin over the 8 XCDs, so blocks b
// and b+8 share an L2.  Bijective remap (cdna_hip_programming.md T1): XCD x owns a contiguous chunk of the tile list, so
// every XCD gets work even when there are fewer than 8 M panels.  Inside the list the tiles are ordered in column chunks
// of CXRK_GN N-tiles, M-panel-major inside a chunk: the blocks an XCD runs at the same time then share a few B panels
// (they stay in its 4 MiB L2) while the A panels stream through, each re-read CXRK_GN times back to back.
#ifndef CXRK_GN
#define CXRK_GN 8   // 0 = one chunk = plain M-panel-major order; 8: +3..7 % on the BERT GEMMs (N = 2304 / 3072) over 0
#endif
// Split-K launches (nz > 1 slabs): the grid is ONE list of nz * nMt * nNt workgroups, slab-major, and the XCD remap runs over the
// whole list, so an XCD owns ~1/8 of it = whole slabs (or a contiguous run of tiles of one slab).  All tiles of a slab read the
// same K-range of both operands: kept on one XCD, that range is fetched into ONE L2 instead of all eight.  (Round 2 remapped
// inside each slab only — every XCD then held tiles of every slab, and the weight-gradient launches moved 2-3x their algorithmic
// bytes through the fabric: profiles/r02_z_pmc_hbm_traffic.txt.)
#ifndef CXRK_SLAB_MAJOR
#define CXRK_SLAB_MAJOR 1   // 0 = round 2's order (tile list remapped per slab, slabs interleaved over the XCDs), for A/B measurements
#endif
__device__ __forceinline__ void tile_coords(int nMt, int nNt, int nz, int& mt, int& nt, int& z) {
  const int per = nMt * nNt;
  int wgid;
  if (CXRK_SLAB_MAJOR || nz == 1) {
    const int nwg = per * nz;
    const int b = blockIdx.x;
    const int xcd = b & 7, idx = b >> 3;
    const int qq = nwg >> 3, rr = nwg & 7;
    const int w = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
    z = w / per;
    wgid = w - z * per;
  } else {
    z = blockIdx.x / per;
    const int b = blockIdx.x - z * per;
    const int xcd = b & 7, idx = b >> 3;
    const int qq = per >> 3, rr = per & 7;
    wgid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
  }
  if (CXRK_GN <= 0 || nNt <= CXRK_GN) { mt = wgid / nNt; nt = wgid - mt * nNt; return; }
  const int per_c = nMt * CXRK_GN;          // tiles of a full chunk
  const int c = wgid / per_c, rem = wgid - c * per_c;
  const int width = min(CXRK_GN, nNt - c * CXRK_GN);
  mt = rem / width;                         // the last (narrower) chunk still holds nMt * width tiles
  nt = c * CXRK_GN + (rem - mt * width);
}

// Waves per SIMD the register-staged kernels are compiled for.  Rounds 1-3 ran the 128x128 tile at 3 (168 registers: a third block
// per CU hides more of the staging / epilogue phases); with the 21-feature-set epilogue inline those instantiations spill 270-300
// registers (212 B of scratch per lane), and at the end of round 3 exactly these kernels turned out to be FRAGILE: two unrelated
// additions to gemm_epilogue.h — code the exact-fp32 mode never executes — made fp32-mode model tests fail (a silently aborted
// queue in one build, a wrong BatchNorm beta gradient in another; deterministic per build, dependent on test order), and the same
// failing build passes every test when compiled for 2 waves per SIMD (no spills).  Whether the spill code or the scratch set-up at
// that occupancy is at fault is not established; the kernels are therefore built without spills.  -DCXRK_OCC=3 restores the old
// configuration for measurements.
#ifndef CXRK_OCC
#define CXRK_OCC 2
#endif
template <class LA, class LB, int WM, int WN>
__global__ __launch_bounds__(NTHREADS, CXRK_OCC) void gemm_f32_kernel(typename LA::P pa, typename LB::P pb, EpiParams ep,
                                                            int M, int N, int K, int nMt, int nNt, int nZ, int kchunk) {
  constexpr int BM = WM * 64, BN = WN * 64;
  constexpr int LDA = LA::LD, LDB = LB::LD;
  constexpr int ASZ = BK * LDA, BSZ = BK * LDB;
  // ONE LDS buffer per operand (33-41 KB) and two barriers per K-tile.  Measured on 32768x3072x768 (scripts/tune_gemm.hip) with
  // the round-1 epilogue: 113 TFLOP/s at 3 blocks per CU, against 99 for two buffers at 2 blocks/CU (66 KB) and 107 for one buffer
  // at 4 waves/SIMD (register pressure).  The kernels are now compiled for 2 waves per SIMD (CXRK_OCC above: no spills); the LDS
  // footprint still lets a third block in where the register file allows it.  NBUF = 2 is kept as a tuning option.
  // All LDS lives in ONE array (cdna_hip_programming.md: a second __shared__ object can de-pipeline the loop).
#ifdef CXRK_NBUF
  constexpr int NBUF = CXRK_NBUF;  // tuning override (scripts/tune_gemm.hip)
#else
  constexpr int NBUF = 1;
#endif
  __shared__ __attribute__((aligned(16))) float smem[NBUF * (ASZ + BSZ)];
  float* const As0 = smem;
  float* const Bs0 = smem + NBUF * ASZ;

  int mt, nt, z;
  tile_coords(nMt, nNt, nZ, mt, nt, z);
  const int m0 = mt * BM, n0 = nt * BN;
  const int kbeg = z * kchunk;
  const int kend = min(K, kbeg + kchunk);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;

  LA la; LB lb;
  la.init(pa, m0, tid);
  lb.init(pb, n0, tid);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  static_assert(!LA::FMT_PLANES && !LB::FMT_PLANES, "the exact-fp32 mainloop reads fp32 operands");
  float4 ra[LA::NV], rb[LB::NV];
  if (kbeg < kend) {
    la.load(kbeg, ra); lb.load(kbeg, rb);
    la.store(As0, ra); lb.store(Bs0, rb);
  }
  __syncthreads();
  if (kbeg + BK < kend) { la.load(kbeg + BK, ra); lb.load(kbeg + BK, rb); }

  // Main loop.  MFMAs are asynchronous to the issuing wave (64 cycles each), so the staging of the NEXT tiles is
  // interleaved between them instead of being appended: with two LDS buffers the registers holding tile t+1 are
  // written to the other buffer a quarter of the way through tile t, and the global loads of tile t+2 are issued at
  // the half-way point; only one barrier per K-tile remains.  (Single-buffer tiles stage after the MFMA block.)
  int cur = 0;
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    const float* Ap = As0 + cur * ASZ + wm * 64 + r + h * LDA;
    const float* Bp = Bs0 + cur * BSZ + wn * 64 + r + h * LDB;
    const bool has1 = k0 + BK < kend, has2 = k0 + 2 * BK < kend;
    float a0 = Ap[0], a1 = Ap[32], b0 = Bp[0], b1 = Bp[32];
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      float a0n = 0.f, a1n = 0.f, b0n = 0.f, b1n = 0.f;
      if (CXRK_ABL == 4 || CXRK_ABL == 6) { a0n = a0 + 1.f; a1n = a1; b0n = b0; b1n = b1; } else
      if (kk + 1 < BK / 2) {
        a0n = Ap[(2 * kk + 2) * LDA]; a1n = Ap[(2 * kk + 2) * LDA + 32];
        b0n = Bp[(2 * kk + 2) * LDB]; b1n = Bp[(2 * kk + 2) * LDB + 32];
      }
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      a0 = a0n; a1 = a1n; b0 = b0n; b1 = b1n;
      if (NBUF == 2 && CXRK_ABL != 4 && CXRK_ABL != 6) {
        if (kk == 3 && has1 && CXRK_ABL != 2) { la.store(As0 + (cur ^ 1) * ASZ, ra); lb.store(Bs0 + (cur ^ 1) * BSZ, rb); }
        if (kk == 7 && has2 && CXRK_ABL != 1) { la.load(k0 + 2 * BK, ra); lb.load(k0 + 2 * BK, rb); }
      }
    }
    if (NBUF == 1 && CXRK_ABL != 4 && CXRK_ABL != 6) {
      if (CXRK_ABL != 3) __syncthreads();  // single buffer: everyone must be done reading before it is overwritten
      if (has1 && CXRK_ABL != 2) { la.store(As0, ra); lb.store(Bs0, rb); }
    }
    if (CXRK_ABL != 3 && CXRK_ABL != 4 && CXRK_ABL != 6) __syncthreads();
    if (NBUF == 1 && has2 && CXRK_ABL != 1 && CXRK_ABL != 4 && CXRK_ABL != 6) { la.load(k0 + 2 * BK, ra); lb.load(k0 + 2 * BK, rb); }
    if (NBUF == 2) cur ^= 1;
  }

  static_assert(BK * (LDA + LDB) >= 4 * 32 * 64, "operand LDS too small to stage the epilogue");
  __syncthreads();  // every wave is done reading operand tiles
  gemm_epilogue<WM, WN>(acc, ep, smem, M, N, m0, n0, mt, z, wave, lane);
}

// ---------------------------------------------------------------------------------------------------------------
// Split-bf16 mainloop: same tiling, loaders, remap and epilogue as gemm_f32_kernel; the operand tiles are split into
// bf16 hi/lo planes while they are staged into LDS ([idx][k], 80-byte rows) and each 32x32x16 product costs three
// v_mfma_f32_32x32x16_bf16 (96 cycles) instead of eight v_mfma_f32_32x32x2_f32 (512 cycles).
// ---------------------------------------------------------------------------------------------------------------
template <class LA, class LB, int WM, int WN>
__global__ __launch_bounds__(NTHREADS, (WM == 2 && WN == 2) ? 3 : 2) void gemm_x3_kernel(typename LA::P pa, typename LB::P pb, EpiParams ep,
                                                              int M, int N, int K, int nMt, int nNt, int nZ, int kchunk) {
  constexpr int BM = WM * 64, BN = WN * 64;
  constexpr int PLANE_A = LA::PLANE, PLANE_B = LB::PLANE;  // halfwords
  static_assert(2 * (PLANE_A + PLANE_B) * 2 >= 4 * 32 * 64 * 4, "operand LDS too small to stage the epilogue");
  __shared__ __attribute__((aligned(16))) unsigned short smem16[2 * (PLANE_A + PLANE_B)];
  unsigned short* const Ahi = smem16;
  unsigned short* const Alo = Ahi + PLANE_A;
  unsigned short* const Bhi = Alo + PLANE_A;
  unsigned short* const Blo = Bhi + PLANE_B;

  int mt, nt, z;
  tile_coords(nMt, nNt, nZ, mt, nt, z);
  const int m0 = mt * BM, n0 = nt * BN;
  const int kbeg = z * kchunk;
  const int kend = min(K, kbeg + kchunk);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  LA la; LB lb;
  la.init(pa, m0, tid);
  lb.init(pb, n0, tid);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  typename LA::V ra[LA::NV]; typename LB::V rb[LB::NV];
  if (kbeg < kend) {
    la.load(kbeg, ra); lb.load(kbeg, rb);
    la.store2(Ahi, Alo, ra); lb.store2(Bhi, Blo, rb);
  }
  __syncthreads();
  if (kbeg + BK < kend) { la.load(kbeg + BK, ra); lb.load(kbeg + BK, rb); }

  // lane (r, h) of v_mfma_f32_32x32x16_bf16 holds A[row r][k = 8h..8h+7] and B[k = 8h..8h+7][col r]
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
#pragma unroll
    for (int kc = 0; kc < BK / 16; ++kc) {
      bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if constexpr (LA::KC) {
          ah[i] = frag_kc(Ahi, wm * 64 + i * 32, kc, lane); al[i] = frag_kc(Alo, wm * 64 + i * 32, kc, lane);
        } else {
          ah[i] = frag_mc<LA::LDT>(Ahi, wm * 64 + i * 32, kc, lane); al[i] = frag_mc<LA::LDT>(Alo, wm * 64 + i * 32, kc, lane);
        }
        if constexpr (LB::KC) {
          bh[i] = frag_kc(Bhi, wn * 64 + i * 32, kc, lane); bl[i] = frag_kc(Blo, wn * 64 + i * 32, kc, lane);
        } else {
          bh[i] = frag_mc<LB::LDT>(Bhi, wn * 64 + i * 32, kc, lane); bl[i] = frag_mc<LB::LDT>(Blo, wn * 64 + i * 32, kc, lane);
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();  // everyone is done reading before the single buffer is overwritten
    if (k0 + BK < kend && !(CXRK_ABL & 2)) { la.store2(Ahi, Alo, ra); lb.store2(Bhi, Blo, rb); }
    __syncthreads();
    if (k0 + 2 * BK < kend) { la.load(k0 + 2 * BK, ra, !(CXRK_ABL & 1)); lb.load(k0 + 2 * BK, rb, !(CXRK_ABL & 1)); }
  }
  __syncthreads();
  if ((CXRK_ABL & 4) && acc[0][0][0] != 12345.678f) return;
  gemm_epilogue<WM, WN>(acc, ep, reinterpret_cast<float*>(smem16), M, N, m0, n0, mt, z, wave, lane);
}

// 0 = exact fp32 MFMA (default), 1 = split-bf16.  Process-wide, set through cxrk_set_precision().
inline int& gemm_precision_mode() { static int mode = 0; return mode; }

// Outputs that cannot stay in the 32 MiB of L2 anyway are stored non-temporally, so that they do not evict the operand
// panels the co-running tiles are re-reading (dense 32768x3072x768: +10 % with the 256x256 tile).
static inline int stream_output(int M, int N, int splitk) {
  static const double thr = [] { const char* e = getenv("CXRK_NT_MB"); return (e ? atof(e) : 64.0) * 1048576.0; }();  // tuning override
  return (double)M * N * 4.0 * (splitk > 1 ? splitk : 1) >= thr;
}

int wgrad_splitk_policy(int M, int N, int K, bool planes);   // gemm.hip

// 256x256-tile policy.  CXRK_WIDE (environment, read once; cxrk_set_wide_mode() at run time): 0 = never, 1 (default) = where it
// pays, 2 = every launch on planes operands (test coverage on small / ragged shapes).  "Pays": split-bf16 launch, both tile dimensions filled, a K
// loop long enough to amortise the exposed prologue / epilogue of a one-block-per-CU kernel, and a tile count that fills
// the 256 CUs in whole rounds to at least 60 % (the other encoder's stream fills a partial round: 80 -> 70 % was worth 3 % of
// the step, below 60 % nothing more).
inline int& wide_mode_ref() { static int m = [] { const char* e = getenv("CXRK_WIDE"); return e ? atoi(e) : 1; }(); return m; }
inline int wide_mode() { return wide_mode_ref(); }
// min_k: shortest K loop (per split-K slab) for which the caller's kind of launch gains (measured per kind on the step's
// shapes, scripts/layer_table.py under CXRK_MINK_* overrides; r2k: data gradients gain on the 256x256 tile at every K — their
// N = C_in >= 256 outputs dominate and the larger tile halves the dy / filter re-reads — forward convolutions only from K = 512).
static inline long env_long(const char* name, long dflt) { const char* e = getenv(name); return e ? atol(e) : dflt; }
#define WIDE_MINK_PLAIN (wide_mink(0))   // dense layers, weight gradients
#define WIDE_MINK_FPROP (wide_mink(1))   // convolution forward (shift + residual + ReLU)
#define WIDE_MINK_DGRAD (wide_mink(2))   // convolution data gradient (ReLU mask + fused BatchNorm sums)
inline long wide_mink(int kind) {
  static const long v[3] = {env_long("CXRK_MINK_PLAIN", 512), env_long("CXRK_MINK_FPROP", 512), env_long("CXRK_MINK_DGRAD", 64)};
  return v[kind];
}
static inline bool use_wide256(int M, int N, long K, int splitk, bool planes, long min_k = WIDE_MINK_PLAIN) {
  if (!planes) return false;                     // the 256x256 kernel (gemm_pw.h) reads planes operands
  const int mode = wide_mode();
  if (mode == 2) return true;                    // test coverage: every planes launch, ragged and tiny ones included
  if (2.0 * M * N * (double)K < 1073741824.0) return false;
  if (mode == 0 || M < 256 || N < 256) return false;
  const long kper = splitk > 1 ? K / splitk : K;
  if (kper < min_k) return false;
  const long tiles = (long)ceil_div(M, 256) * ceil_div(N, 256) * (splitk > 1 ? splitk : 1);
  const long rounds = (tiles + 255) / 256;
  static const long eff_pct = env_long("CXRK_WIDE_EFF", 60);   // tuning override: minimum round efficiency in percent
  return tiles * 100 >= rounds * 256 * eff_pct;
}

// Fills the launch-derived epilogue switches; false = the operand formats / alignments cannot be served.
static inline bool prep_epilogue(EpiParams& e, int M, int N, int splitk) {
  auto ok16 = [](const void* p_, long ld) { return p_ == nullptr || (aligned16(p_) && (ld % 4) == 0); };
  auto ok8 = [](const void* p_, long ld) { return p_ == nullptr || (aligned16(p_) && (ld % 8) == 0); };
  e.vec = (N % 4 == 0) && ok16(e.C, e.ldc) && ((e.slab_stride % 4) == 0) && ok16(e.R, e.ldr) && ok16(e.aux, e.ldaux) &&
          ok16(e.C2, e.ldc2) && (e.bias == nullptr || aligned16(e.bias));
  e.pl = (e.Cp || e.Rp || e.maskin || e.maskout || e.auxmode == 3) ? 1 : 0;
  e.fast = e.vec && (N % 8) == 0 && ok8(e.Cp, e.ldc) && ok8(e.Rp, e.ldr) && (e.cplane % 8) == 0 && (e.rplane % 8) == 0 &&
           (!e.maskout || ((N % 64) == 0 && (e.ldmaskout % 8) == 0 && (((uintptr_t)e.maskout) & 7) == 0));
  e.kind = epi_kind(epi_flags(e));
  // fp32-output feature sets use the 4 + 4 column lane map, whose two column groups must be inside or outside the tensor together
  if (e.kind >= 0 && (EPI_KINDS[e.kind] & (EF_OUTPL | EF_AUX_MASK | EF_MASKOUT)) == 0u && (N % 64) != 0) e.kind = -1;
  if (e.rs2_on && !(e.Rp && e.Cp && e.rs2_H > 0 && e.rs2_W > 0 && (long)e.rs2_Ho * e.rs2_Wo > 0)) return false;
  if (e.pl) {   // planes / mask operands exist in the fast epilogue only
    if (!e.fast) return false;
    if ((e.C != nullptr) == (e.Cp != nullptr)) return false;           // exactly one output format
    if (e.R && e.Rp) return false;
    if (e.auxmode == 3 && !e.maskin) return false;
    if (e.maskout && e.act != 1) return false;
    if (e.Cp && splitk > 1) return false;                               // split-K slabs are fp32
  }
  e.nt = stream_output(M, N, splitk);
  return true;
}

// Host launcher.  splitk > 1 writes plain partial slabs (caller reduces them).
template <class LA, class LB, int WM, int WN>
static int launch_gemm(const typename LA::P& pa, const typename LB::P& pb, const EpiParams& ep, int M, int N, int K,
                       int splitk, hipStream_t stream, bool force_fp32 = false) {
  constexpr int BM = WM * 64, BN = WN * 64;
  if (M <= 0 || N <= 0 || K <= 0) return CXRK_ERR_ARG;
  const int nMt = ceil_div(M, BM), nNt = ceil_div(N, BN);
  int kchunk = K;
  if (splitk > 1) { kchunk = ceil_div(ceil_div(K, splitk), BK) * BK; splitk = ceil_div(K, kchunk); }
  else splitk = 1;
  dim3 grid((unsigned)(nMt * nNt * splitk), 1, 1);
  EpiParams e = ep;
  if (!prep_epilogue(e, M, N, splitk)) return CXRK_ERR_ARG;
  if (e.rs2_on) return CXRK_ERR_UNSUPPORTED;   // the compact stride-2 residual exists in the LDS-DMA kernels' epilogue only
  // split-bf16 only where it pays and is well conditioned: small problems (adapters, heads: < 1 GFLOP) and launches the
  // caller marks exact (the stem convolution: an all-positive input makes its weight gradient a cancelling sum) stay fp32.
  // Pre-split ("planes") operands exist in the split-bf16 format only: they always take the split mainloop.
  constexpr bool planes_in = LA::FMT_PLANES || LB::FMT_PLANES;
  const bool split = planes_in || (gemm_precision_mode() == 1 && !force_fp32 && 2.0 * M * N * (double)K >= 1073741824.0);
  if constexpr (planes_in) {
    static_assert(LA::FMT_PLANES && LB::FMT_PLANES, "both operands must share the storage format");
    hipLaunchKernelGGL((gemm_x3_kernel<LA, LB, WM, WN>), grid, dim3(NTHREADS), 0, stream, pa, pb, e, M, N, K, nMt, nNt, splitk, kchunk);
  } else if (split)
    hipLaunchKernelGGL((gemm_x3_kernel<LA, LB, WM, WN>), grid, dim3(NTHREADS), 0, stream, pa, pb, e, M, N, K, nMt, nNt, splitk, kchunk);
  else
    hipLaunchKernelGGL((gemm_f32_kernel<LA, LB, WM, WN>), grid, dim3(NTHREADS), 0, stream, pa, pb, e, M, N, K, nMt, nNt, splitk, kchunk);
  CXRK_LAUNCH_CHECK();
  return splitk;  // >= 1: number of slabs actually written
}

}  // namespace cxrk
#include "gemm_pw.h"
