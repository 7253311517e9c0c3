// PROTOTYPE (not part of the library): a 256 x 128 tile, 4 waves (128 x 64 outputs each, as in the 256 x 256 kernel), BK = 16,
// THREE LDS stages of 24 KiB = 72 KiB per workgroup -> TWO workgroups per CU, dense NT on planes operands only.
// Question it answers: does a second workgroup over the prologue + epilogue of a tile pay for the 1.5x
// L2 -> LDS bytes per FLOP and the barrier every 24 MFMAs?  Timed against the library's 256 x 256 (one workgroup per CU) and
// 128 x 128 (two per CU) LDS-DMA kernels on the step's GEMM shapes; results checked against the 256 x 256 kernel.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iincremental_multimodal_medical_learning_ii_amd/csrc scripts/proto_pw3.hip -o proto_pw3
//   ./proto_pw3 M N K [planes_out]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "gemm_core.h"
using namespace cxrk;

namespace proto {
constexpr int BK3 = 16, TM = 256, TN = 128, NW = 4, NST = 3;
constexpr int PLANE_A = TM * BK3 * 2, PLANE_B = TN * BK3 * 2;      // bytes of one bf16 plane of an operand tile
constexpr int STAGE = 2 * (PLANE_A + PLANE_B);                     // A hi | A lo | B hi | B lo = 24 KiB

// K-contiguous plane [row][32 B]: the two 16-byte k-chunks of row r sit at slot h ^ ((r >> 3) & 1) (conflict-free ds_read_b128 for the
// 32x32x16 operand map: a quarter-wave reads 16 rows = 8 even + 8 odd slots = all 64 banks once)
__device__ __forceinline__ bf16x8 frag(const unsigned char* plane, int row0, int lane) {
  const int r = row0 + (lane & 31);
  const int slot = (lane >> 5) ^ ((r >> 3) & 1);
  return *reinterpret_cast<const bf16x8*>(plane + r * 32 + slot * 16);
}
typedef PwFrag Frag;
__device__ __forceinline__ void rd(Frag& f, const unsigned char* hi, int plane_bytes, int row0, int lane) {
#pragma unroll
  for (int i = 0; i < 2; ++i) { f.h[i] = frag(hi, row0 + 32 * i, lane); f.l[i] = frag(hi + plane_bytes, row0 + 32 * i, lane); }
}

// X(idx, k) = ptr[idx * ld + k]; a wave fills RW = TILE / NW rows of each plane in RW / 32 pieces of 32 rows (1 KiB = 64 lanes x 16 B:
// lane -> row lane / 2, slot lane % 2, which holds k-chunk (lane % 2) ^ ((row >> 3) & 1))
template <int TILE>
struct Loader {
  static constexpr int RW = TILE / NW, NP = RW / 32, PLANEB = TILE * BK3 * 2;
  const unsigned short* bp; long plane; unsigned voff[NP]; int kq8, K, wave;
  __device__ __forceinline__ void init(const unsigned short* ptr, long ld, long plane_, int rows, int K_, int idx0, int wave_, int lane) {
    wave = wave_; K = K_; plane = plane_; bp = ptr + (long)idx0 * ld;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int r = RW * wave + 32 * j + (lane >> 1);
      const int q = (lane & 1) ^ ((r >> 3) & 1);
      if (j == 0) kq8 = 8 * q;                                         // (r >> 3) & 1 is the same for every piece: pieces are 32 rows apart
      voff[j] = idx0 + r < rows ? (unsigned)((r * (int)ld + 8 * q) * 2) : VOFF_OOB;
    }
  }
  __device__ __forceinline__ void issue(int k0, unsigned char* lds_hi, bool live) const {
    const unsigned t = (k0 + kq8 < K) ? 0u : VOFF_OOB;                 // K tail (K % 8 == 0)
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      const __amdgpu_buffer_rsrc_t rs = tile_rsrc(bp + pl * plane + k0, live);
#pragma unroll
      for (int j = 0; j < NP; ++j) dma16(rs, lds_hi + pl * PLANEB + (RW * wave + 32 * j) * 32, voff[j] | t);
    }
  }
};

__global__ __launch_bounds__(256, 2) void kernel(const unsigned short* A, long lda, long aplane, const unsigned short* B, long ldb, long bplane,
                                                  EpiParams ep, int M, int N, int K, int nMt, int nNt) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * STAGE];
  int mt, nt, z;
  tile_coords(nMt, nNt, 1, mt, nt, z);
  const int m0 = mt * TM, n0 = nt * TN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int arow = wm * 128, bcol = wn * 64;
  Loader<TM> la; Loader<TN> lb;
  la.init(A, lda, aplane, M, K, m0, wave, lane);
  lb.init(B, ldb, bplane, N, K, n0, wave, lane);
  f32x16 acc[2][2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[s][i][j][e] = 0.f;
  auto stage = [&](int t) { return smem + (t % NST) * STAGE; };
  la.issue(0, stage(0), true);            lb.issue(0, stage(0) + 2 * PLANE_A, true);
  la.issue(BK3, stage(1), BK3 < K);       lb.issue(BK3, stage(1) + 2 * PLANE_A, BK3 < K);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // this wave's 6 pieces of tile 0 have landed (tile 1's 6 may still fly)
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  Frag A0, A1, Bc, Bn;
  rd(A0, stage(0), PLANE_A, arow, lane);
  rd(Bc, stage(0) + 2 * PLANE_A, PLANE_B, bcol, lane);
  const int NK = (K + BK3 - 1) / BK3;
  // one K-tile: group 0 (rows 0-63) with the DMA of tile t + 2 into the stage tile t - 1 vacated (every wave passed tile t - 1's
  // barrier), group 1 (rows 64-127) behind the K-tile's one barrier (tile t + 1 has landed everywhere, nobody reads tile t - 1's stage)
  auto ktile = [&](int t, Frag& bcur, Frag& bnext) {
    unsigned char* st = stage(t);
    unsigned char* nx = stage(t + 1);
    rd(A1, st, PLANE_A, arow + 64, lane);
    { const int k2 = (t + 2) * BK3; la.issue(k2, stage(t + 2), k2 < K); lb.issue(k2, stage(t + 2) + 2 * PLANE_A, k2 < K); }
    __builtin_amdgcn_sched_barrier(0);
    pw_mfma12(acc[0], A0, bcur);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    rd(A0, nx, PLANE_A, arow, lane);
    rd(bnext, nx + 2 * PLANE_A, PLANE_B, bcol, lane);
    __builtin_amdgcn_sched_barrier(0);
    pw_mfma12(acc[1], A1, bcur);
    __builtin_amdgcn_sched_barrier(0);
  };
  int t = 0;
  for (; t + 1 < NK; t += 2) { ktile(t, Bc, Bn); ktile(t + 1, Bn, Bc); }
  if (t < NK) ktile(t, Bc, Bn);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  float* stg = reinterpret_cast<float*>(smem) + wave * (32 * 64);
  epi_pw_dispatch<0, 2>(ep.kind, acc, ep, stg, M, N, m0 + arow, n0 + bcol, mt * (TM / 64) + wm * 2, 0, lane);
}

static int launch(const unsigned short* A, long aplane, const unsigned short* B, long bplane, const EpiParams& ep, int M, int N, int K) {
  const int nMt = ceil_div(M, TM), nNt = ceil_div(N, TN);
  EpiParams e = ep;
  if (!prep_epilogue(e, M, N, 1) || !e.fast) return -1;
  hipLaunchKernelGGL(kernel, dim3(nMt * nNt), dim3(256), 0, 0, A, (long)K, aplane, B, (long)K, bplane, e, M, N, K, nMt, nNt);
  return 0;
}
}  // namespace proto

__global__ void split_k(const float* x, long n8, unsigned short* out, long plane) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const float4 a = *reinterpret_cast<const float4*>(x + i * 8), b = *reinterpret_cast<const float4*>(x + i * 8 + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    planes_store8(out, plane, i * 8, v);
  }
}
template <class KERN>
static float time_kernel(KERN launch, int reps) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch();
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main(int argc, char** argv) {
  const int M = argc > 3 ? atoi(argv[1]) : 32768, N = argc > 3 ? atoi(argv[2]) : 3072, K = argc > 3 ? atoi(argv[3]) : 768;
  const bool planes_out = argc > 4 && atoi(argv[4]) != 0;
  const size_t nA = (size_t)M * K, nB = (size_t)N * K, nC = (size_t)M * N;
  float *A, *B, *C0, *C1, *C2; unsigned short *Ap, *Bp;
  (void)hipMalloc(&A, nA * 4); (void)hipMalloc(&B, nB * 4); (void)hipMalloc(&C0, nC * 4); (void)hipMalloc(&C1, nC * 4); (void)hipMalloc(&C2, nC * 4);
  (void)hipMalloc(&Ap, nA * 4); (void)hipMalloc(&Bp, nB * 4);
  std::vector<float> h(nA > nB ? nA : nB);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 100003) / 50001.f - 1.f;
  (void)hipMemcpy(A, h.data(), nA * 4, hipMemcpyHostToDevice);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 40503u + 17) % 100019) / 50009.f - 1.f;
  (void)hipMemcpy(B, h.data(), nB * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(split_k, dim3(2048), dim3(256), 0, 0, A, (long)(nA / 8), Ap, (long)nA);
  hipLaunchKernelGGL(split_k, dim3(2048), dim3(256), 0, 0, B, (long)(nB / 8), Bp, (long)nB);
  const double fl = 2.0 * M * N * K;
  EpiParams ep{}; ep.ldc = N; ep.alpha = 1.f;
  auto out = [&](float* c) { EpiParams e = ep; if (planes_out) { e.Cp = reinterpret_cast<unsigned short*>(c); e.cplane = (long)nC; } else e.C = c; return e; };
  EpiParams e0 = out(C0), e1 = out(C1), e2 = out(C2);
  DmaDenseKC<256, 8>::P da{Ap, K, M, K, (long)nA}; DmaDenseKC<256, 8>::P db{Bp, K, N, K, (long)nB};
  DmaDenseKC<128, 4>::P qa{Ap, K, M, K, (long)nA}; DmaDenseKC<128, 4>::P qb{Bp, K, N, K, (long)nB};
  (void)hipMemset(C0, 0xff, nC * 4); (void)hipMemset(C1, 0xff, nC * 4); (void)hipMemset(C2, 0xff, nC * 4);
  const float t0 = time_kernel([&] { launch_gemm_pw<Pw256, DmaDenseKC<256, 8>, DmaDenseKC<256, 8>>(da, db, e0, M, N, K, 1, 0); }, 10);
  const float t1 = time_kernel([&] { launch_gemm_pw<Pw128, DmaDenseKC<128, 4>, DmaDenseKC<128, 4>>(qa, qb, e1, M, N, K, 1, 0); }, 10);
  if (proto::launch(Ap, (long)nA, Bp, (long)nB, e2, M, N, K) != 0) { printf("proto launch refused\n"); return 2; }
  const float t2 = time_kernel([&] { proto::launch(Ap, (long)nA, Bp, (long)nB, e2, M, N, K); }, 10);
  std::vector<unsigned> c0(nC), c2(nC);
  (void)hipMemcpy(c0.data(), C0, nC * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(c2.data(), C2, nC * 4, hipMemcpyDeviceToHost);
  size_t bad = 0; double md = 0;
  if (planes_out) { for (size_t i = 0; i < nC; ++i) bad += c0[i] != c2[i]; }        // same products, same k order: bit-identical planes
  else for (size_t i = 0; i < nC; ++i) { const double d = fabs((double)__builtin_bit_cast(float, c0[i]) - __builtin_bit_cast(float, c2[i])); md = fmax(md, d); bad += !(d <= 1e-3); }
  printf("NT %dx%dx%d %s out | 256x256 1 blk/CU %.3f ms %.0f TF | 128x128 2 blk/CU %.3f ms %.0f TF | PROTO 256x128 BK16 3-stage 2 blk/CU %.3f ms %.0f TF (%+.1f %% vs best) | max diff %.3g bad %zu\n",
         M, N, K, planes_out ? "planes" : "fp32", t0, fl / t0 / 1e9, t1, fl / t1 / 1e9, t2, fl / t2 / 1e9, 100.0 * (fmin(t0, t1) / t2 - 1.0), md, bad);
  return bad != 0;
}
