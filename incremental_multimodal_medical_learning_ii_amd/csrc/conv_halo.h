// 3x3 / stride 1 / pad 1 convolution with 64 input channels and 64 filters (the 56 x 56 layers of ResNet-50's layer1) on planes
// operands, forward and data gradient: the activation operand stays RESIDENT in LDS for all nine taps.
//
// Why.  As an implicit GEMM on the 256 x 64 tile (gemm_pw_kernel<Pw256x64, DmaConvIm2colKC, ...>) these layers run at 3.5-4.2x
// their MFMA floor (profiles/r03_z_layer_table.log): with only 64 output columns a K-tile carries 26 FLOP per staged byte (the
// 256 x 256 tile: 64), and the im2col expansion stages every activation nine times — 9.2 GB of L2 -> LDS traffic per launch at
// batch 1024.  Here a block owns 256 CONSECUTIVE pixels of the flattened [N*H*W] index, m0 .. m0+255, and stages, once, the
// pixels m0-(W+1) .. m0+255+(W+1) (all 64 channels, both planes: 94 KiB).  Tap (r, s) of output pixel m reads pixel
// m + (r-1) W + (s-1) of that window (data gradient: m + (1-r) W + (1-s) of dy): the MFMA operand of a tap is the same LDS image
// read at a shifted row.  Neighbours that do not exist (image borders; the flattened index then points into the previous / next
// row or image) are per-lane tap masks that redirect the read to an all-zero pixel.  Only the filter streams: a four-tap LDS-DMA
// ring (16 KiB per tap), tap t+4 issued behind the barrier of tap t, three taps ahead of its use.  Staged bytes per launch:
// 3.0 GB instead of 9.2.  Because a tile is a run of consecutive output rows, the epilogue is the shared one (gemm_epilogue.h).
//
// Waves (each step measured with the s_memtime stamps of scripts/tune_halo.hip, profiles/r03_halo_tune_stamps.log).  The window
// and the ring fill the LDS, so one block per CU, and nothing of another block covers this one's prologue, waits or epilogue.
//   * four waves (one per SIMD), each issuing its share of the DMA: 1 495 cycles per K-tile against 768 of MFMA work (the same with
//     the data L2-resident, so not memory: one wave per SIMD cannot cover its own reads, address arithmetic, DMA issue, barrier);
//   * eight waves, the two of a SIMD splitting the CONTRACTION (one channel half each): 950 per K-tile, but the accumulators have
//     to be added through LDS and four waves run the epilogue alone — 6 500 cycles for a plain store, more with masks and sums;
//   * eight waves splitting the 64 output COLUMNS (wave w: rows 64 (w & 3).., columns 32 (w >> 2)..), epilogue in the pair form
//     of gemm_epilogue.h (the two waves stage their column halves side by side and each finishes 32 of the 64 rows): 3 100;
//   * persistent blocks with four LOADER waves (below): the 10-12 000-cycle prologue (window latency + issue) disappears under
//     the previous tile's epilogue; no scheduling fences in the mainloop: 924 per K-tile.
// A tile now costs ~1 200 (start) + 16 600 (mainloop) + 7 000-11 000 (epilogue incl. waiting for the loaders at PAIR) + ~2 500
// (waiting for the next window at DONE) cycles against 13 800 of MFMA work: 0.98 / 0.96 ms forward / data gradient with the
// step's epilogues against 1.14 / 1.17 ms as implicit GEMM on the same box (profiles/r03_halo_layer_ab.log), t / floor 3.4.
// What is left is the epilogue: VALU-bound (split to planes, ReLU bits, column sums) and, with the LDS full, not overlappable
// with the next tile's MFMAs.
//
// LDS image of the window: [pixel][8 chunks of 16 B] per plane, chunk c of pixel p at slot c ^ ((p >> 1) & 7): each of the four
// 16-lane groups a ds_read_b128 is served in ({0-3,12-15,20-27}, ...: MI355X_MICROARCH.md, LDS) reads 16 pixels that land in 16
// different 16-byte bank groups, for every shift.  The LDS-DMA writes lane-linear, so the swizzle is applied to the per-lane
// SOURCE address (gemm_pw.h).
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

__device__ __forceinline__ bf16x8 halo_frag(const unsigned char* plane, int px, int c) {
  return *reinterpret_cast<const bf16x8*>(plane + px * 128 + ((c ^ ((px >> 1) & 7)) << 4));
}

// x: planes [M = N*H*W][64] (forward: the input; data gradient: dy); LB: the filter operand of the matching implicit GEMM for FOUR
// waves (forward DmaDenseKC<64, 4> over w[Ko][9 * 64]; data gradient DmaConvFilterMC<64, 4>), whose K-tiles run tap-major.
// Column sums (ep.colsum_part): EIGHT partial rows per tile, row (4 mt + w & 3) * 2 + (w >> 2).
//
// PERSISTENT, with four loader waves.  A block is 8 compute waves + 4 loader waves and walks a list of tiles (the XCD's contiguous
// run of tiles, strided by the blocks of that XCD, so the blocks of an XCD sweep neighbouring windows through its L2 together).
// Only the loader waves issue LDS-DMA: loader (L, G) stages half G of plane L of the window and half G of K-tile L of every tap.
// That takes the DMA issue (60-185 cycles a piece) out of the MFMA waves' instruction streams, and — vmcnt being a per-wave, in-order counter —
// lets the NEXT tile's window load run under this tile's epilogue without the epilogue's own side-input loads having to wait
// for it.  The epilogue stages through the ring (free after the last tap), so the window buffer is free for that load.
// Every wave executes the same twelve barriers per tile:
//   START    loaders: window and ring tap 0 have landed                   compute: row masks of the tile are ready
//   TAP s    (s = 0..8) loaders: ring tap s+1 has landed                    compute: my reads of ring slot s & 3 have retired
//            -> behind it the loaders refill that slot with tap s+4; behind TAP 8 they issue the next tile's window
//   PAIR     the pair-form epilogue's internal barrier (gemm_epilogue.h)
//   DONE     loaders: the next window has landed                            compute: done with the staging space (= the ring)
//            -> behind it the loaders issue ring taps 0-3 of the next tile
struct HaloFrag { bf16x8 ah[2], al[2], bh, bl; };   // A: the wave's two 32-row blocks, B: its 32 columns; hi / lo

template <class LB, bool DGRAD>
__global__ __launch_bounds__(768, 3) void conv3x3_halo_kernel(const unsigned short* __restrict__ x, long xplane, typename LB::P pb, EpiParams ep,
                                                             int M, int N, int H, int W, int nMt) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * HALO_PLANE + HALO_NSTB * HALO_TAPB];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int W1 = W + 1;
  unsigned char* sA = smem;
  unsigned char* sB = smem + 2 * HALO_PLANE;
  // this block's tiles: XCD x = blockIdx & 7 owns tiles [t0, t1); its blocks take t0 + slot, t0 + slot + nslot, ...
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
  const int qq = nMt >> 3, rr = nMt & 7;
  const int t0 = xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq;
  const int t1 = t0 + qq + (xcd < rr ? 1 : 0);
  auto tap_of = [](int t) { return DGRAD ? 8 - t : t; };   // filter tap of step t: the window shift grows with t either way
#define HALO_BARRIER() do { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
#define HALO_STAMP(i) do { if (ep.stamps && (tid & 511) == 0) { ep.stamps[(long)mt * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
                              if ((i) == 0 || (i) == 4) ep.stamps[(long)mt * 8 + 5 + (i) / 4] = __builtin_amdgcn_s_memrealtime(); } } while (0)
  // diagnostics (scripts/tune_halo.hip): s_memtime per tile at [0] loop top (loader) / [1] START passed / [2] last tap done /
  // [3] epilogue issued / [4] stores drained (compute wave 0)

  if (wave >= 8) {
    // ================================================================================================ loader waves
    const int L = (wave - 8) & 1, G = (wave - 8) >> 1;    // loader (L, G): plane / K-tile L, piece half / filter-row half G
    LB lbv[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) lbv[v].init(pb, 0, 2 * G + v, lane);
    auto window = [&](int mt) {           // half G of plane L of the window of tile mt: pieces of 8 pixels, 24 + 23
      const int m0 = mt * HALO_TM;
      const unsigned short* wbase = x + ((long)m0 - W1) * HALO_CH + L * xplane;
      const __amdgpu_buffer_rsrc_t rs = tile_rsrc(wbase);
      const int span = HALO_TM + 2 * W1;
      const int p0 = G * 24, p1 = G ? HALO_PIECES : 24;
#pragma unroll 1
      for (int piece = p0; piece < p1; ++piece) {
        const int pp = piece * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((pp >> 1) & 7);
        const long gp = (long)m0 - W1 + pp;
        const bool ok = pp < span && gp >= 0 && gp < (long)M;
        dma16(rs, sA + L * HALO_PLANE + piece * 1024, ok ? (unsigned)((pp * HALO_CH + c * 8) * 2) : VOFF_OOB);
      }
    };
    auto ring = [&](int step, bool live) {   // filter rows 32 G .. +31 of K-tile L of the tap of `step` into its slot: 4 loads
      unsigned char* dst = sB + (step & (HALO_NSTB - 1)) * HALO_TAPB + L * HALO_BSTAGE;
      const int k0 = (2 * tap_of(step) + L) * BK;
#pragma unroll
      for (int v = 0; v < 2; ++v) lbv[v].issue(k0, dst, live);
    };
    bool first = true;
    for (int mt = t0 + slot; mt < t1; mt += nslot) {
      if (wave == 8) HALO_STAMP(0);
      if (first) { window(mt); first = false; }
#pragma unroll
      for (int t = 0; t < HALO_NSTB; ++t) ring(t, true);
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // the window and ring tap 0 have landed; younger: taps 1-3, 4 loads each
      HALO_BARRIER();                                      // START
#pragma unroll 1
      for (int step = 0; step < 9; ++step) {
        // tap step+1 has landed; the younger loads are those of taps step+2 and step+3 as far as they exist.  (No dummy issues to
        // keep the count uniform: an LDS-DMA through a dead descriptor still WRITES zeros, and behind TAP 8 the ring is the
        // epilogue's staging space.)
        if (step <= 5) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (step == 6) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        HALO_BARRIER();                                    // TAP step
        if (step + HALO_NSTB < 9) ring(step + HALO_NSTB, true);
      }
      HALO_BARRIER();                                      // PAIR (first: the compute waves stand at it with their accumulators staged)
      if (mt + nslot < t1) window(mt + nslot);             // under this tile's epilogue
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      HALO_BARRIER();                                      // DONE
    }
    return;
  }

  // ================================================================================================== compute waves
  const int jsel = wave >> 2, wq = wave & 3;      // column half, row block
  const int arow = wq * 64;
  for (int mt = t0 + slot; mt < t1; mt += nslot) {
    const int m0 = mt * HALO_TM;
    // per-lane rows: window slot of the wave's two 32-row blocks and the 9-bit mask of the taps that exist
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

    // per step: byte offset of the lane's window pixel (shifted by the tap, or the zero pixel) and its swizzle term, per 32-row block
    struct TapAddr { int base[2], sw[2]; };
    auto tap_addr = [&](TapAddr& t, int step) {
      const int tap = tap_of(step);
      const int r = tap / 3, s = tap - 3 * r;
      const int shift = DGRAD ? (1 - r) * W + (1 - s) : (r - 1) * W + (s - 1);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int px = ((vm[i] >> tap) & 1u) ? pxb[i] + shift : HALO_HP - 1;
        t.base[i] = px * 128; t.sw[i] = (px >> 1) & 7;
      }
    };
    // fragments of a step (its TapAddr and ring slot), K-tile kt2 (channel half) and k-step kc: window rows shifted by the tap, and
    // the wave's 32 filter columns
    auto read_f = [&](HaloFrag& f, const TapAddr& t, int step, int kt2, int kc) {
      const int c = kt2 * 4 + 2 * kc + (lane >> 5);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const unsigned char* a = sA + t.base[i] + ((c ^ t.sw[i]) << 4);
        f.ah[i] = *reinterpret_cast<const bf16x8*>(a);
        f.al[i] = *reinterpret_cast<const bf16x8*>(a + HALO_PLANE);
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

    HALO_BARRIER();                                        // START
    HALO_STAMP(1);
    HaloFrag F0, F1;
    TapAddr ta;
    tap_addr(ta, 0);
    read_f(F0, ta, 0, 0, 0);
#pragma unroll 1
    for (int step = 0; step < 9; ++step) {
      // groups 0-2: the next group's fragments are read in front of this group's six MFMAs.  (No scheduling fences here, unlike
      // gemm_pw.h: left to the compiler the mainloop takes 924 instead of 996 cycles per K-tile, scripts/tune_halo.hip.)
      read_f(F1, ta, step, 0, 1);
      mfma6(F0);
      read_f(F0, ta, step, 1, 0);
      mfma6(F1);
      read_f(F1, ta, step, 1, 1);
      if (step < 8) tap_addr(ta, step + 1);
      mfma6(F0);
      // group 3, behind the tap's barrier
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      HALO_BARRIER();                                      // TAP step
      if (step < 8) read_f(F0, ta, step + 1, 0, 0);
      mfma6(F1);
    }
    HALO_STAMP(2);
    float* stg = reinterpret_cast<float*>(sB) + wq * (2 * 32 * 64);   // 16 KiB per wave pair, in the ring (free behind TAP 8)
    epi_pair_dispatch<0>(ep.kind, acc, ep, stg, M, N, m0 + arow, 0, (mt * (HALO_TM / 64) + wq) * 2 + jsel, lane, jsel);   // PAIR inside
    HALO_STAMP(3);
    if (ep.stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); HALO_STAMP(4); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // my reads of the staging space have retired
    HALO_BARRIER();                                        // DONE
  }
#undef HALO_STAMP
#undef HALO_BARRIER
}

static inline bool halo_enabled() { static const bool on = [] { const char* e = getenv("CXRK_HALO"); return e ? atoi(e) != 0 : true; }(); return on; }
// the layers this kernel takes: planes operands, 3 x 3, stride 1, pad 1, 64 -> 64 channels, rows of <= 62 pixels
static inline bool halo_applies(int H, int W, int C, int Ko, int R, int S, int stride, int pad) {
  return halo_enabled() && R == 3 && S == 3 && stride == 1 && pad == 1 && C == HALO_CH && Ko == 64 && W <= HALO_MAXW && H >= 1;
}

static inline int halo_cus() {
  static const int n = [] { int dev = 0, v = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256; return v; }();
  return n;
}
template <class LB, bool DGRAD>
static int launch_conv3x3_halo(const unsigned short* x, long xplane, const typename LB::P& pb, const EpiParams& ep, int M, int N, int H, int W,
                               hipStream_t stream) {
  if (M <= 0 || N != 64 || W > HALO_MAXW) return CXRK_ERR_ARG;
  const int nMt = ceil_div(M, HALO_TM);
  EpiParams e = ep;
  if (!prep_epilogue(e, M, N, 1) || !e.fast) return CXRK_ERR_ARG;
  // one block per CU, in groups of 8 (block b runs on XCD b % 8): 8 x min(CUs / 8, tiles of the longest XCD run)
  const int per_xcd = ceil_div(nMt, 8);
  int nslot = halo_cus() / 8; if (nslot < 1) nslot = 1; if (nslot > per_xcd) nslot = per_xcd;
  hipLaunchKernelGGL((conv3x3_halo_kernel<LB, DGRAD>), dim3((unsigned)(8 * nslot)), dim3(768), 0, stream, x, xplane, pb, e, M, N, H, W, nMt);
  CXRK_LAUNCH_CHECK();
  return 1;
}

}  // namespace cxrk
