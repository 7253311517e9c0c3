// 3x3 / stride 1 / pad 1 convolution with 64 input channels and 64 filters (the 56 x 56 layers of ResNet-50's layer1) on planes
// operands, forward and data gradient: the activation operand stays RESIDENT in LDS for all nine taps.
//
// Why.  As an implicit GEMM on the 256 x 64 tile (gemm_pw_kernel<Pw256x64, DmaConvIm2colKC, ...>) these layers run at 3.5-4.2x
// their MFMA floor (profiles/r03_z_layer_table.log): with only 64 output columns a K-tile carries 26 FLOP per staged byte (the
// 256 x 256 tile: 64), and the im2col expansion stages every activation nine times — 9.2 GB of L2 -> LDS traffic per launch at
// batch 1024, which at the ~9 TB/s the LDS-DMA path sustains IS the measured 0.99 ms.  Here a block owns 256 CONSECUTIVE pixels
// of the flattened [N*H*W] index m0 .. m0+255 and stages, once, the pixels m0-(W+1) .. m0+255+(W+1) (all 64 channels, both
// planes: 94 KiB).  Tap (r, s) of output pixel m reads pixel m + (r-1) W + (s-1) of that window (data gradient: m + (1-r) W +
// (1-s) of dy): the MFMA operand of a tap is the same LDS image read at a shifted row.  Neighbours that do not exist (image
// borders; the flattened index then points into the previous / next row or image) are per-lane tap masks that redirect the read
// to an all-zero pixel.  Only the filter streams: a four-tap LDS-DMA ring (16 KiB per tap), tap t+4 issued behind the barrier of
// tap t, i.e. three taps (~2.5 us) ahead of its use.  Staged bytes per launch: 3.0 GB instead of 9.2.  Because the tile is a run
// of consecutive output rows, the epilogue is the shared one (gemm_epilogue.h).
//
// Waves.  The window and the ring fill the LDS, so one block per CU.  With four waves (one per SIMD) nothing covers a wave's own
// fragment reads, address arithmetic, DMA issue and barrier: measured 1 495 cycles per K-tile against 768 of MFMA work
// (scripts/tune_halo.hip stamps; the same with the data L2-resident, so not memory).  The block therefore has EIGHT waves: the two
// waves of a SIMD own the same 64 output rows and split the 64 output COLUMNS (wave w: rows 64 (w & 3).., columns 32 (w >> 2)..);
// they run out of phase and cover each other.  (Splitting the contraction instead — each wave one channel half — measured the
// same mainloop, 950 cycles per K-tile, but needs the accumulators added through LDS and leaves the epilogue to four waves:
// 6 500 cycles for a plain store, more with masks and sums, nothing overlapping it.)  The epilogue is the shared one in its pair
// form (gemm_epilogue.h, PAIR): the two waves stage their column halves side by side and each finishes 32 of the 64 rows.
//
// LDS image of the window: [pixel][8 chunks of 16 B] per plane, chunk c of pixel p at slot c ^ ((p >> 1) & 7): each of the four
// 16-lane groups a ds_read_b128 is served in ({0-3,12-15,20-27}, ...: MI355X_MICROARCH.md, LDS) reads 16 pixels that land in 16
// different 16-byte bank groups, for every shift.  The LDS-DMA writes lane-linear, so the swizzle is applied to the per-lane
// SOURCE address (gemm_pw.h).
// Hazards: the window is written once (its low 264 pixels before the first barrier, the rest before the second) and only read
// afterwards.  Filter ring: tap t+4 goes into the
// slot tap t used, behind the barrier of tap t that every wave passes only after its last reads of that slot retired
// (lgkmcnt(0)); before that barrier every wave waits for its own pieces of tap t+1 (vmcnt(4): the four younger loads are those
// of taps t+2 and t+3).  Waves 0-3 load the K-tile of channels 0-31 of a tap, waves 4-7 that of channels 32-63; all read both.
#pragma once
#include "gemm_core.h"

namespace cxrk {

constexpr int HALO_CH = 64;                                  // input channels (= K per tap)
constexpr int HALO_TM = 256;                                 // output pixels per block, 4 waves x 64
constexpr int HALO_HP = 376;                                 // pixel slots of the window; the last one is the all-zero pixel
constexpr int HALO_PLANE = HALO_HP * 2 * HALO_CH;            // 47 KiB per plane
constexpr int HALO_NSTB = 4;                                 // filter taps in the ring: window 94 KiB + ring 64 KiB = 158 of 160 KiB
constexpr int HALO_BPLANE = pw_plane_bytes<64>();            // 4 KiB: one plane of a 64 x 32 filter tile
constexpr int HALO_BSTAGE = 2 * HALO_BPLANE;                 // a K-tile: hi | lo
constexpr int HALO_TAPB = 2 * HALO_BSTAGE;                   // a tap: the K-tiles of channels 0-31 and 32-63
constexpr int HALO_MAXW = (HALO_HP - 1 - HALO_TM) / 2 - 1;   // 58: the window 256 + 2 (W + 1) must leave the zero pixel free
constexpr int HALO_PIECES = HALO_HP / 8;                     // 1-KiB LDS-DMA pieces (8 pixels) per plane
constexpr int HALO_PART1 = 33;                               // pieces holding the pixels 0 .. 263 >= 258 the first three taps read

__device__ __forceinline__ bf16x8 halo_frag(const unsigned char* plane, int px, int c) {
  return *reinterpret_cast<const bf16x8*>(plane + px * 128 + ((c ^ ((px >> 1) & 7)) << 4));
}

// x: planes [M = N*H*W][64] (forward: the input; data gradient: dy); LB: the filter operand of the matching implicit GEMM for FOUR
// waves (forward DmaDenseKC<64, 4> over w[Ko][9 * 64]; data gradient DmaConvFilterMC<64, 4>), whose K-tiles run tap-major.
// Column sums (ep.colsum_part): EIGHT partial rows per block, row (4 mt + w & 3) * 2 + (w >> 2).
struct HaloFrag { bf16x8 ah[2], al[2], bh, bl; };   // A: the wave's two 32-row blocks, B: its 32 columns; hi / lo

template <class LB, bool DGRAD>
__global__ __launch_bounds__(512, 2) void conv3x3_halo_kernel(const unsigned short* __restrict__ x, long xplane, typename LB::P pb, EpiParams ep,
                                                             int M, int N, int H, int W, int nMt) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * HALO_PLANE + HALO_NSTB * HALO_TAPB];
  int mt, nt, z;
  tile_coords(nMt, 1, 1, mt, nt, z);
  const int m0 = mt * HALO_TM;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int jsel = wave >> 2, wq = wave & 3;      // column half, row block
  const int W1 = W + 1;
  const int arow = wq * 64;
  unsigned char* sA = smem;
  unsigned char* sB = smem + 2 * HALO_PLANE;

  auto tap_of = [](int t) { return DGRAD ? 8 - t : t; };   // filter tap of step t: the window shift grows with t either way
#define HALO_STAMP(i) do { if (ep.stamps && tid == 0) { ep.stamps[(long)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
                              if ((i) == 0 || (i) == 4) ep.stamps[(long)blockIdx.x * 8 + 5 + (i) / 4] = __builtin_amdgcn_s_memrealtime(); } } while (0)
  HALO_STAMP(0);   // diagnostics (scripts/tune_halo.hip): s_memtime at start / window landed / mainloop done / epilogue issued / stores drained
  LB lb;
  lb.init(pb, 0, wq, lane);

  // ---- the window: 47 pieces of 8 pixels per plane; lane -> (pixel 8 piece + lane / 8, slot lane % 8).  The taps run from the low
  // end of the window upwards (the data gradient walks the filter backwards for that), so the first three taps need the pixels
  // 0 .. 257 and the zero pixel only: pieces 0-32 and 46 of both planes (part 1, 68 piece loads) and the first tap of the ring are
  // waited for here; the other 13 pieces per plane (part 2: exactly four loads per wave, the last six repeating earlier ones) and
  // ring taps 1-3 land behind the first MFMAs — the wait in front of tap 0's barrier (vmcnt(4): only ring taps 2 and 3 may be
  // outstanding) covers them.
  const unsigned short* wbase = x + ((long)m0 - W1) * HALO_CH;
  auto window_piece = [&](int pl, int piece) {
    const int span = HALO_TM + 2 * W1;
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc(wbase + pl * xplane);
    const int pp = piece * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((pp >> 1) & 7);
    const long gp = (long)m0 - W1 + pp;
    const bool ok = pp < span && gp >= 0 && gp < (long)M;
    dma16(rs, sA + pl * HALO_PLANE + piece * 1024, ok ? (unsigned)((pp * HALO_CH + c * 8) * 2) : VOFF_OOB);
  };
  constexpr int P1 = HALO_PART1 + 1, P2 = HALO_PIECES - P1;   // 34 and 13 pieces per plane
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    const int f = wave * 9 + j;                    // 68 loads over 8 waves: 9 each for waves 0-6, 5 for wave 7
    if (f >= 2 * P1) continue;                     // wave-uniform
    const int pl = f >= P1 ? 1 : 0, q = f - pl * P1;
    window_piece(pl, q < HALO_PART1 ? q : HALO_PIECES - 1);
  }
  // a wave loads the K-tile `jsel` (channels 32 jsel .. +31) of every tap: 2 loads per tap and wave
  auto ring_issue = [&](int step, bool live) { lb.issue((2 * tap_of(live ? step : 0) + jsel) * BK, sB + (step & (HALO_NSTB - 1)) * HALO_TAPB + jsel * HALO_BSTAGE, live); };
  ring_issue(0, true);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int f = wave * 4 + j;                          // 26 loads + 6 repeats = 4 per wave
    if (f >= 2 * P2) f -= 2 * P2;
    const int pl = f >= P2 ? 1 : 0;
    window_piece(pl, HALO_PART1 + f - pl * P2);
  }
#pragma unroll
  for (int t = 1; t < HALO_NSTB; ++t) ring_issue(t, true);

  // ---- per-lane rows: window slot of the wave's two 32-row blocks and the 9-bit mask of the taps that exist
  int pxb[2]; unsigned vm[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int rl = arow + i * 32 + (lane & 31);
    const int row = m0 + rl;
    pxb[i] = W1 + rl;
    unsigned m = 0;
    if (row < M) {
      const int w_ = row % W, h_ = (row / W) % H;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          const int hh = DGRAD ? h_ + 1 - r : h_ + r - 1, ww = DGRAD ? w_ + 1 - s : w_ + s - 1;
          if ((unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W) m |= 1u << (r * 3 + s);
        }
    }
    vm[i] = m;
  }

  f32x16 acc[2][2][2];             // only acc[0][i][0] is live (64 rows x 32 columns); the shape is the shared epilogue's signature
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[0][i][0][e] = 0.f;

  // fragments of step `step` (filter tap tap_of(step)), K-tile kt2 (channel half) and k-step kc: window rows shifted by the tap, and
  // the wave's 32 filter columns
  auto read_f = [&](HaloFrag& f, int step, int kt2, int kc) {
    const int tap = step < 9 ? tap_of(step) : 9;      // step 9: no such tap -> the zero pixel (never used)
    const int r = tap / 3, s = tap - 3 * r;
    const int shift = DGRAD ? (1 - r) * W + (1 - s) : (r - 1) * W + (s - 1);
    const int c = kt2 * 4 + 2 * kc + (lane >> 5);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int px = ((vm[i] >> tap) & 1u) ? pxb[i] + shift : HALO_HP - 1;
      f.ah[i] = halo_frag(sA, px, c);
      f.al[i] = halo_frag(sA + HALO_PLANE, px, c);
    }
    const unsigned char* bt = sB + (step & (HALO_NSTB - 1)) * HALO_TAPB + kt2 * HALO_BSTAGE;
    f.bh = LB::frag(bt, jsel * 32, kc, lane);
    f.bl = LB::frag(bt + HALO_BPLANE, jsel * 32, kc, lane);
  };
  auto mfma6 = [&](const HaloFrag& f) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      acc[0][i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[i], f.bh, acc[0][i][0], 0, 0, 0);
      acc[0][i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], f.bl, acc[0][i][0], 0, 0, 0);
      acc[0][i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], f.bh, acc[0][i][0], 0, 0, 0);
    }
  };

  asm volatile("s_waitcnt vmcnt(10)" ::: "memory");   // part 1 and ring tap 0 have landed; younger: 4 (part 2) + 6 (ring taps 1-3)
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  HALO_STAMP(1);
  HaloFrag F0, F1;
  read_f(F0, 0, 0, 0);
  for (int step = 0; step < 9; ++step) {
    // groups 0-2: the next group's fragments are read in front of this group's six MFMAs
    read_f(F1, step, 0, 1);
    PW_FENCE();
    mfma6(F0);
    PW_FENCE();
    read_f(F0, step, 1, 0);
    PW_FENCE();
    mfma6(F1);
    PW_FENCE();
    read_f(F1, step, 1, 1);
    PW_FENCE();
    mfma6(F0);
    PW_FENCE();
    // group 3, behind the tap's barrier
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");   // 2 loads per tap and wave: taps step+2, step+3 may be in flight
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    ring_issue(step + HALO_NSTB, step + HALO_NSTB < 9);          // into the slot this tap has just vacated
    read_f(F0, step + 1, 0, 0);
    PW_FENCE();
    mfma6(F1);
    PW_FENCE();
  }
  HALO_STAMP(2);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();   // every wave is done with the window and the ring: the window becomes the epilogue's staging space
  float* stg = reinterpret_cast<float*>(smem) + wq * (2 * 32 * 64);   // 16 KiB per wave pair
  epi_pair_dispatch<0>(ep.kind, acc, ep, stg, M, N, m0 + arow, 0, (mt * (HALO_TM / 64) + wq) * 2 + jsel, lane, jsel);
  HALO_STAMP(3);
  if (ep.stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); HALO_STAMP(4); }
#undef HALO_STAMP
}

static inline bool halo_enabled() { static const bool on = [] { const char* e = getenv("CXRK_HALO"); return e ? atoi(e) != 0 : true; }(); return on; }
// the layers this kernel takes: planes operands, 3 x 3, stride 1, pad 1, 64 -> 64 channels, rows of <= 62 pixels
static inline bool halo_applies(int H, int W, int C, int Ko, int R, int S, int stride, int pad) {
  return halo_enabled() && R == 3 && S == 3 && stride == 1 && pad == 1 && C == HALO_CH && Ko == 64 && W <= HALO_MAXW && H >= 1;
}

template <class LB, bool DGRAD>
static int launch_conv3x3_halo(const unsigned short* x, long xplane, const typename LB::P& pb, const EpiParams& ep, int M, int N, int H, int W,
                               hipStream_t stream) {
  if (M <= 0 || N != 64 || W > HALO_MAXW) return CXRK_ERR_ARG;
  const int nMt = ceil_div(M, HALO_TM);
  EpiParams e = ep;
  if (!prep_epilogue(e, M, N, 1) || !e.fast) return CXRK_ERR_ARG;
  hipLaunchKernelGGL((conv3x3_halo_kernel<LB, DGRAD>), dim3((unsigned)nMt), dim3(512), 0, stream, x, xplane, pb, e, M, N, H, W, nMt);
  CXRK_LAUNCH_CHECK();
  return 1;
}

}  // namespace cxrk
