// Epilogue of the MFMA mainloops for 16-byte-aligned operands (every shape of the training step): included by gemm_core.h.
//
// One wave owns a 64x64 output sub-tile in 2x2 accumulator tiles of the 32x32 MFMA C/D map (col = lane & 31,
// row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)): a lane holds ONE column.  Stored from there it would take 64 dword stores
// per lane in 128-byte pieces; instead the wave transposes 32 rows at a time through its private 8 KiB of (now free) operand
// LDS and leaves with 8 consecutive columns per lane: 16-byte pieces of every operand, fp32 (two per lane) or bf16 planes.
//
// Round 1's epilogue decided everything (operand formats, activation, residual, masks, row remap, ragged edges) with
// run-time branches per element group; s_memtime stamps put it at 85,000 cycles per 256x256 tile — as long as 24 K-tiles of
// the mainloop — for ~1,000 useful instructions per wave: it spilled scalar registers into vector lanes and re-derived the
// predicates per row.  This version
//  * is a template over the feature set F: the flag combinations the training step uses are compiled in (EPI_KINDS) and picked
//    by ONE switch; every other combination runs the same code with F = EF_GENERIC, where each feature test is a (wave-uniform)
//    run-time test of the parameter block;
//  * never branches per lane: every global access is a buffer load / store whose lane offset carries bit 31 when the lane's row
//    or column is outside the tensor (descriptors declare 2^31 bytes: the range check drops the access);
//  * issues the side-input loads of all four 8-row passes of a 32-row half before the first store of that half.
// The row-remapped form (stride-2 data-gradient parity classes) addresses rows individually and is predicated instead.
#pragma once

namespace cxrk {

enum : unsigned {
  EF_OUTPL = 1u << 0,      // output as bf16 hi/lo planes (else fp32)
  EF_BIAS = 1u << 1,
  EF_RES_F32 = 1u << 2,    // residual fp32
  EF_RES_PL = 1u << 3,     // residual planes
  EF_RELU = 1u << 4,
  EF_GELU = 1u << 5,
  EF_AUX_SIGN = 1u << 6,   // v *= (aux > 0), aux fp32          (auxmode 1)
  EF_AUX_GELU = 1u << 7,   // v *= gelu'(aux), aux fp32         (auxmode 2)
  EF_AUX_MASK = 1u << 8,   // v *= bit of maskin                (auxmode 3)
  EF_C2 = 1u << 9,         // fp32 copy of the pre-activation value
  EF_MASKOUT = 1u << 10,   // store the ReLU decision bits
  EF_COLSUM = 1u << 11,    // per-wave column sums of the stored values
  EF_REMAP = 1u << 12,     // output-row remap (stride-2 dgrad parity class)
  EF_RES_S2 = 1u << 13,    // the planes residual is compact: rows for the even (h, w) pixels of the output only
  EF_GENERIC = 1u << 31,   // every feature decided at run time
};

// Feature sets compiled in (index = EpiParams::kind); anything else runs as EF_GENERIC.
constexpr unsigned EPI_KINDS[] = {
    0u,                                                          //  0 plain fp32: split-K slabs, weight gradients, dctx
    EF_BIAS,                                                     //  1 qkv projection
    EF_BIAS | EF_RES_PL,                                         //  2 attention output / FFN down projection (+ residual)
    EF_OUTPL | EF_BIAS | EF_GELU | EF_C2,                        //  3 FFN up projection
    EF_OUTPL | EF_AUX_GELU,                                      //  4 gradient through GELU
    EF_RES_PL,                                                   //  5 data gradient + residual gradient
    EF_OUTPL | EF_BIAS | EF_RELU | EF_MASKOUT,                   //  6 conv + BN + ReLU
    EF_OUTPL | EF_BIAS | EF_RES_PL | EF_RELU | EF_MASKOUT,       //  7 conv + BN + identity + ReLU
    EF_OUTPL | EF_BIAS,                                          //  8 downsample conv + BN
    EF_OUTPL | EF_AUX_MASK | EF_COLSUM,                          //  9 conv data gradient, ReLU mask, beta-gradient sums
    EF_OUTPL | EF_RES_PL | EF_AUX_MASK | EF_COLSUM,              // 10 the same + gradient of the identity branch
    EF_OUTPL,                                                    // 11 plain planes output
    EF_BIAS | EF_RELU,                                           // 12 fp32 mode: conv + BN + ReLU
    EF_BIAS | EF_RES_F32 | EF_RELU,                              // 13 fp32 mode: conv + BN + identity + ReLU
    EF_AUX_SIGN | EF_COLSUM,                                     // 14 fp32 mode: conv data gradient
    EF_RES_F32 | EF_AUX_SIGN | EF_COLSUM,                        // 15 fp32 mode: conv data gradient + identity gradient
    EF_BIAS | EF_RES_F32,                                        // 16 fp32 mode: dense + residual
    EF_BIAS | EF_GELU | EF_C2,                                   // 17 fp32 mode: FFN up projection
    EF_RES_F32,                                                  // 18 fp32 mode: data gradient + residual gradient
    EF_OUTPL | EF_BIAS | EF_RELU,                                // 19 stem conv + BN + ReLU (its mask is the sign of the pooled value)
    EF_OUTPL | EF_AUX_GELU | EF_COLSUM,                          // 20 gradient through GELU + its column sums (FFN-up bias gradient)
    EF_OUTPL | EF_RES_PL | EF_RES_S2 | EF_AUX_MASK | EF_COLSUM,  // 21 kind 10 with the identity-branch gradient of a stride-2 projection (compact)
};
constexpr int EPI_NKINDS = (int)(sizeof(EPI_KINDS) / sizeof(EPI_KINDS[0]));

// host: feature word of a parameter block / its compiled-in index (-1: generic)
static inline unsigned epi_flags(const EpiParams& e) {
  unsigned f = 0;
  if (e.Cp) f |= EF_OUTPL;
  if (e.bias) f |= EF_BIAS;
  if (e.R) f |= EF_RES_F32;
  if (e.Rp) f |= EF_RES_PL;
  if (e.act == 1) f |= EF_RELU;
  if (e.act == 2) f |= EF_GELU;
  if (e.auxmode == 1) f |= EF_AUX_SIGN;
  if (e.auxmode == 2) f |= EF_AUX_GELU;
  if (e.auxmode == 3) f |= EF_AUX_MASK;
  if (e.C2) f |= EF_C2;
  if (e.maskout) f |= EF_MASKOUT;
  if (e.colsum_part) f |= EF_COLSUM;
  if (e.rm_on) f |= EF_REMAP;
  if (e.rs2_on) f |= EF_RES_S2;
  return f;
}
static inline int epi_kind(unsigned flags) {
  for (int i = 0; i < EPI_NKINDS; ++i) if (EPI_KINDS[i] == flags) return i;
  return -1;
}

__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned voff) { return bloadu4(r, voff); }
__device__ __forceinline__ void buf_store16(__amdgpu_buffer_rsrc_t r, unsigned voff, const uint4& v, bool nt) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 u = {v.x, v.y, v.z, v.w};
  if (nt) __builtin_amdgcn_raw_buffer_store_b128(u, r, (int)voff, 0, 2);   // aux 2 = nt
  else __builtin_amdgcn_raw_buffer_store_b128(u, r, (int)voff, 0, 0);
}
__device__ __forceinline__ uint2 buf_load8(__amdgpu_buffer_rsrc_t r, unsigned voff) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const u32x2 u = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, 0, 0));
  return make_uint2(u[0], u[1]);
}
__device__ __forceinline__ uint4 f4_bits(float a, float b, float c, float d) {
  return make_uint4(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, c), __builtin_bit_cast(unsigned, d));
}

// OR over each aligned group of 8 lanes, result in all 8: three DPP steps on the vector ALU (lane ^ 1, lane ^ 2 inside a quad, then
// 7 - lane inside the half row, which joins the two quads) instead of three ds_bpermute round trips through the LDS crossbar: the
// forward epilogue of conv_halo.h went from 6 960 to 5 570 cycles (scripts/tune_halo.hip).  (First tried while the register-staged
// kernels still spilled at 3 waves per SIMD; that build aborted an fp32-mode test — see CXRK_OCC in gemm_core.h.)
#ifndef CXRK_OR8_DPP
#define CXRK_OR8_DPP 1   // 0: the ds_bpermute form (A/B measurements)
#endif
__device__ __forceinline__ unsigned or8_lanes(unsigned x) {
  x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
  x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
  x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xF, 0xF, true);   // row_half_mirror
  return x;
}

// Experiment switch (round 3, measured and NOT kept): 1 = pipeline the side inputs of the LDS-DMA kernels' epilogues (see epi_rows).
// With it the data-gradient launches of the step took 20.34 instead of 20.45 ms and the step 154.0 instead of 155.3 ms — inside
// the box-to-box spread — while the 256x256 instantiations went from 249 registers to 256 + 60 spilled (scratch): the epilogue is
// bound by its stores (256 KB per tile), not by the latency of its side inputs.
#ifndef CXRK_EPI_PIPE
#define CXRK_EPI_PIPE 0
#endif
constexpr bool epi_has_side(unsigned f) { return (f & (EF_RES_F32 | EF_RES_PL | EF_AUX_SIGN | EF_AUX_GELU | EF_AUX_MASK)) != 0u; }
constexpr bool epi_two_fp32_sides(unsigned f) { return (f & EF_RES_F32) != 0u && (f & (EF_AUX_SIGN | EF_AUX_GELU)) != 0u; }

#define EH(bit, rt) (GEN ? (rt) : ((F & (bit)) != 0u))

// NSUB 64-row slabs of a wave's outputs (acc[s] = rows 64 s .. 64 s + 63 of the block at row0; NSUB = 2: the 128 x 64 outputs of a
// wave of the 256 x 256 tile) go through 2 NSUB halves of 32 rows.
// PIPE (compiled-in feature sets on the LDS-DMA kernels): the side inputs (residual, mask bits, GELU' source) of an 8-row pass of
// half h + 1 are requested as soon as the same pass of half h has been stored, so only the first half's loads are waited for at
// full memory latency; the un-pipelined form (generic feature set, register-staged kernels) loads the side inputs of a half and
// consumes them at once — a round trip per half with nothing of this wave in between.
// PAIR (conv_halo.h): two waves share one 64 x 64 output block, wave `psel` holding its columns 32 psel .. +31 (acc[0][i][0], all
// 64 rows).  Both stage their column block of BOTH 32-row halves into the pair's shared staging area `st` ([2 halves][32][64]
// fp32), meet at the workgroup barrier (every wave of the block must call this), and wave psel then finishes half psel alone:
// the 64 x 64 block's epilogue runs on two waves.  Column sums: one partial row per (block, half) — `part` is that row.
// S2RES: the compact stride-2 residual (EF_RES_S2) is compiled in.  Only the LDS-DMA kernels (epi_pw_dispatch / epi_pair_dispatch)
// carry it: the register-staged fp32 / x3 kernels never see such a launch, and their code must not grow (see the note at epi64).
template <unsigned F, int NSUB, bool PIPE, bool PAIR = false, bool S2RES = true>
__device__ __forceinline__ void epi_rows(f32x16 (*acc)[2][2], const EpiParams& ep, float* st, int M, int N, int row0, int col0,
                                         int part, int z, int lane, int psel = 0) {
  constexpr bool GEN = (F & EF_GENERIC) != 0u;
  static_assert(!(GEN && PIPE), "the generic feature set is not pipelined");
  static_assert(!PAIR || (NSUB == 1 && !PIPE), "the pair form: one 64-row block, un-pipelined");
  const bool outpl = EH(EF_OUTPL, ep.Cp != nullptr), has_bias = EH(EF_BIAS, ep.bias != nullptr);
  const bool res_f32 = EH(EF_RES_F32, ep.R != nullptr), res_pl = EH(EF_RES_PL, ep.Rp != nullptr);
  const bool relu = EH(EF_RELU, ep.act == 1), gelu = EH(EF_GELU, ep.act == 2);
  const bool aux_sign = EH(EF_AUX_SIGN, ep.auxmode == 1), aux_gelu = EH(EF_AUX_GELU, ep.auxmode == 2), aux_mask = EH(EF_AUX_MASK, ep.auxmode == 3);
  const bool has_c2 = EH(EF_C2, ep.C2 != nullptr), maskout = EH(EF_MASKOUT, ep.maskout != nullptr);
  const bool colsum = EH(EF_COLSUM, ep.colsum_part != nullptr), remap = EH(EF_REMAP, ep.rm_on != 0);
  const bool res_s2 = S2RES && EH(EF_RES_S2, ep.rs2_on != 0);   // (planes output only: prep_epilogue)
  const bool nt = ep.nt != 0;

  const int r = lane & 31, h = lane >> 5;
  const int c8 = lane & 7, rq = lane >> 3;
  // Lane -> columns of the 64-wide sub-tile.  v[0..3] are columns cA..cA+3, v[4..7] columns cB..cB+3.
  //   planes output / bit masks: cA = 8*c8, cB = cA + 4 — 8 consecutive columns, one 16-byte piece of each bf16 plane (one mask byte);
  //   fp32 output (MAPF):        cA = 4*c8, cB = 32 + 4*c8 — each store instruction then writes 128 contiguous bytes per row
  //   (8 lanes x 16 B) instead of every other 16 bytes of 256 (two half-filled lines per row and instruction: measured 19-25 k
  //   cycles per 256x256 tile against 11.7 k for the planes form with the same number of stores).
  //   The second group is a CONSTANT byte distance behind the first (DB4 in an fp32 row, DB2 in a bf16 plane row), which the
  //   compiler folds into the instruction's immediate offset: one offset register per (pass, tensor), as before.  For that both
  //   groups of a lane must be inside or outside the tensor together: N % 8 == 0 gives it to the 8-column map, and the host
  //   selects a MAPF feature set only when N % 64 == 0 (prep_epilogue; other widths run the generic form).
  constexpr bool MAPF = !GEN && (F & (EF_OUTPL | EF_AUX_MASK | EF_MASKOUT)) == 0u;
  constexpr unsigned DB4 = MAPF ? 128u : 16u, DB2 = 64u;
  const int cA = MAPF ? c8 * 4 : c8 * 8, cB = MAPF ? 32 + c8 * 4 : c8 * 8 + 4;
  const unsigned deadA = col0 + cA < N ? 0u : VOFF_OOB;

  float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (has_bias) {
    const __amdgpu_buffer_rsrc_t rb = tile_rsrc(ep.bias + col0);
    const unsigned ob = (unsigned)(cA * 4) | deadA;
    const uint4 b0 = buf_load16(rb, ob), b1 = buf_load16(rb, ob + DB4);
    bv[0] = __builtin_bit_cast(float, b0.x); bv[1] = __builtin_bit_cast(float, b0.y); bv[2] = __builtin_bit_cast(float, b0.z); bv[3] = __builtin_bit_cast(float, b0.w);
    bv[4] = __builtin_bit_cast(float, b1.x); bv[5] = __builtin_bit_cast(float, b1.y); bv[6] = __builtin_bit_cast(float, b1.z); bv[7] = __builtin_bit_cast(float, b1.w);
  }
  float bs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // Wave-uniform descriptors at the block origin (row0, col0); lane offsets below are relative to it.  In the remapped form
  // (GEMM row (n,a,b) -> pixel (n, 2a+ph, 2b+pw)) the origin is the pixel of the block's first row and a lane adds the
  // distance of its own pixel from it (the map is monotonic, and 128 consecutive GEMM rows stay within a few image rows).
  const long rbase = remap ? epi_row(ep, row0 < M ? row0 : M - 1) : (long)row0;   // remap is monotonic: lane offsets stay >= 0
  float* Cf = ep.C ? ep.C + (long)z * ep.slab_stride : nullptr;
  const __amdgpu_buffer_rsrc_t d_out = tile_rsrc(outpl ? (const void*)(ep.Cp + rbase * ep.ldc + col0) : (const void*)(Cf + rbase * ep.ldc + col0));
  const __amdgpu_buffer_rsrc_t d_out2 = tile_rsrc(outpl ? (const void*)(ep.Cp + ep.cplane + rbase * ep.ldc + col0) : (const void*)nullptr, outpl);
  const long rres = res_s2 ? 0 : rbase;   // the compact residual is addressed from its first row
  const __amdgpu_buffer_rsrc_t d_res = tile_rsrc(res_pl ? (const void*)(ep.Rp + rres * ep.ldr + col0) : (const void*)(ep.R + rbase * ep.ldr + col0), res_pl || res_f32);
  const __amdgpu_buffer_rsrc_t d_res2 = tile_rsrc(res_pl ? (const void*)(ep.Rp + ep.rplane + rres * ep.ldr + col0) : (const void*)nullptr, res_pl);
  const __amdgpu_buffer_rsrc_t d_aux = tile_rsrc(ep.aux + rbase * ep.ldaux + col0, aux_sign || aux_gelu);
  const __amdgpu_buffer_rsrc_t d_c2 = tile_rsrc(ep.C2 + rbase * ep.ldc2 + col0, has_c2);
  const __amdgpu_buffer_rsrc_t d_min = tile_rsrc(ep.maskin + rbase * ep.ldmaskin + (col0 >> 3), aux_mask);
  const __amdgpu_buffer_rsrc_t d_mout = tile_rsrc(ep.maskout + rbase * ep.ldmaskout + (col0 >> 3), maskout);

  // side inputs of PF 8-row passes (all four of a half in the compiled-in feature sets; the generic form, which may carry every
  // side input at once, goes pass by pass to stay inside the register budget; an fp32 residual AND an fp32 aux tensor — the
  // fp32-mode data gradient with identity branch — are four 16-byte loads per pass: two passes ahead is what the 168-register
  // budget of the three-blocks-per-CU kernels holds without spilling)
  constexpr int PF = GEN ? 1 : (((F & EF_RES_F32) != 0u && (F & (EF_AUX_SIGN | EF_AUX_GELU)) != 0u) ? 2 : 4);
  static_assert(!PIPE || PF == 4, "the pipelined form keeps the side inputs of a whole half");
  struct Side { unsigned dead[PF]; unsigned rowo[PF]; uint4 r0v[PF], r1v[PF], a0v[PF], a1v[PF]; unsigned mk[PF]; };

  // request the side inputs of pass ua (8 rows) of half hh (rows 32 hh .. 32 hh + 31 of the block) into slot u
  auto load_side = [&](Side& sd, int u, int hh, int ua) {
    {
      const int rl = hh * 32 + ua * 8 + rq;              // row inside the block
      const int grow = row0 + rl;
      sd.dead[u] = grow < M ? 0u : VOFF_OOB;   // row only; the column part (deadA) is OR-ed in per access
      sd.rowo[u] = (unsigned)(remap ? epi_row(ep, grow < M ? grow : M - 1) - rbase : (long)rl);
      const unsigned rowo = sd.rowo[u];
      if (res_pl) {
        if constexpr (MAPF) {   // 4 + 4 columns: two 8-byte pieces of each plane
          const unsigned oa = ((rowo * (unsigned)ep.ldr + cA) * 2u) | sd.dead[u] | deadA;
          const uint2 ha = buf_load8(d_res, oa), hb = buf_load8(d_res, oa + DB2), la = buf_load8(d_res2, oa), lb = buf_load8(d_res2, oa + DB2);
          sd.r0v[u] = make_uint4(ha.x, ha.y, hb.x, hb.y); sd.r1v[u] = make_uint4(la.x, la.y, lb.x, lb.y);
        } else {
          unsigned o = ((rowo * (unsigned)ep.ldr + cA) * 2u) | sd.dead[u] | deadA;
          if (res_s2) {   // output pixel (n, h, w) -> compact row (n, h / 2, w / 2) when h and w are even, else no residual
            const int gr = grow < M ? grow : M - 1;
            const int w_ = gr % ep.rs2_W; const int t_ = gr / ep.rs2_W; const int h_ = t_ % ep.rs2_H; const int n_ = t_ / ep.rs2_H;
            const unsigned crow = (unsigned)((n_ * ep.rs2_Ho + (h_ >> 1)) * ep.rs2_Wo + (w_ >> 1));
            o = ((crow * (unsigned)ep.ldr + cA) * 2u) | (((w_ | h_) & 1) ? VOFF_OOB : 0u) | sd.dead[u] | deadA;
          }
          sd.r0v[u] = buf_load16(d_res, o); sd.r1v[u] = buf_load16(d_res2, o);
        }
      } else if (res_f32) {
        const unsigned o = ((rowo * (unsigned)ep.ldr + cA) * 4u) | sd.dead[u] | deadA;
        sd.r0v[u] = buf_load16(d_res, o); sd.r1v[u] = buf_load16(d_res, o + DB4);
      }
      if (aux_sign || aux_gelu) {
        const unsigned o = ((rowo * (unsigned)ep.ldaux + cA) * 4u) | sd.dead[u] | deadA;
        sd.a0v[u] = buf_load16(d_aux, o); sd.a1v[u] = buf_load16(d_aux, o + DB4);
      }
      if (aux_mask) sd.mk[u] = __builtin_amdgcn_raw_buffer_load_b8(d_min, (int)((rowo * (unsigned)ep.ldmaskin + c8) | sd.dead[u] | deadA), 0, 0);
    }
  };

  // values and stores of pass ua of half hh, from the staged accumulators and the side inputs in slot u of `sd`
  auto finish_pass = [&](const Side& sd, int u, int hh, int ua) {
    {
      const float* sp = st + (PAIR ? psel * 2048 : 0) + (ua * 8 + rq) * 64;
      const float4 s0 = *reinterpret_cast<const float4*>(sp + cA), s1 = *reinterpret_cast<const float4*>(sp + cB);
      float v[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
      const unsigned rowo = sd.rowo[u];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = ep.alpha * v[q] + bv[q];
      if (res_pl) {
        float rv[8]; planes_unpack8(sd.r0v[u], sd.r1v[u], rv);
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] += rv[q];
      } else if (res_f32) {
        const unsigned rw[8] = {sd.r0v[u].x, sd.r0v[u].y, sd.r0v[u].z, sd.r0v[u].w, sd.r1v[u].x, sd.r1v[u].y, sd.r1v[u].z, sd.r1v[u].w};
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] += __builtin_bit_cast(float, rw[q]);
      }
      if (has_c2) {
        const unsigned o = ((rowo * (unsigned)ep.ldc2 + cA) * 4u) | sd.dead[u] | deadA;
        buf_store16(d_c2, o, f4_bits(v[0], v[1], v[2], v[3]), false);
        buf_store16(d_c2, o + DB4, f4_bits(v[4], v[5], v[6], v[7]), false);
      }
      unsigned obits = 0;
      if (relu) {
#pragma unroll
        for (int q = 0; q < 8; ++q) { obits |= (v[q] > 0.f ? 1u : 0u) << q; v[q] = fmaxf(v[q], 0.f); }
      } else if (gelu) {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = gelu_erf(v[q]);
      }
      if (aux_mask) {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = ((sd.mk[u] >> q) & 1u) ? v[q] : 0.f;
      } else if (aux_sign || aux_gelu) {
        const unsigned aw[8] = {sd.a0v[u].x, sd.a0v[u].y, sd.a0v[u].z, sd.a0v[u].w, sd.a1v[u].x, sd.a1v[u].y, sd.a1v[u].z, sd.a1v[u].w};
        if (aux_sign) {
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] = __builtin_bit_cast(float, aw[q]) > 0.f ? v[q] : 0.f;
        } else {
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] *= gelu_erf_grad(__builtin_bit_cast(float, aw[q]));
        }
      }
      if (colsum) {   // rows outside the tensor contribute nothing (dead columns are dropped at the final store)
        const bool livel = sd.dead[u] == 0u;
#pragma unroll
        for (int q = 0; q < 8; ++q) bs[q] += livel ? v[q] : 0.f;
      }
      if (outpl) {
        const unsigned oo = ((rowo * (unsigned)ep.ldc + cA) * 2u) | sd.dead[u] | deadA;
        uint4 hv, lv; planes_pack8(v, hv, lv);
        buf_store16(d_out, oo, hv, nt); buf_store16(d_out2, oo, lv, nt);
      } else {
        const unsigned oo = ((rowo * (unsigned)ep.ldc + cA) * 4u) | sd.dead[u] | deadA;
        buf_store16(d_out, oo, f4_bits(v[0], v[1], v[2], v[3]), nt);
        buf_store16(d_out, oo + DB4, f4_bits(v[4], v[5], v[6], v[7]), nt);
      }
      if (maskout) {   // the 8 lanes of a row hold its 8 mask bytes: OR them together, lane c8 == 0 stores the 8 bytes
        unsigned w0 = c8 < 4 ? obits << (8 * c8) : 0u, w1 = c8 >= 4 ? obits << (8 * (c8 - 4)) : 0u;
        if (CXRK_OR8_DPP) { w0 = or8_lanes(w0); w1 = or8_lanes(w1); }
        else {
#pragma unroll
          for (int o = 1; o < 8; o <<= 1) { w0 |= (unsigned)__shfl_xor((int)w0, o, 64); w1 |= (unsigned)__shfl_xor((int)w1, o, 64); }
        }
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 w = {w0, w1};
        const unsigned mo = (rowo * (unsigned)ep.ldmaskout) | ((c8 == 0 && sd.dead[u] == 0u && col0 < N) ? 0u : VOFF_OOB);
        __builtin_amdgcn_raw_buffer_store_b64(w, d_mout, (int)mo, 0, 0);
      }
    }
  };

  auto flush_colsum = [&](int prt) {  // add the eight row lanes of the wave (lanes c8 + 8*rq), lanes 0..7 write; restart the sums
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float t = bs[q];
      t += __shfl_xor(t, 8, 64); t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64);
      bs[q] = t;
    }
    const __amdgpu_buffer_rsrc_t d_cs = tile_rsrc(ep.colsum_part + (long)prt * N + col0);
    const unsigned keep = rq == 0 ? 0u : VOFF_OOB;
    const unsigned o = (unsigned)(cA * 4) | keep | deadA;
    buf_store16(d_cs, o, f4_bits(bs[0], bs[1], bs[2], bs[3]), false);
    buf_store16(d_cs, o + DB4, f4_bits(bs[4], bs[5], bs[6], bs[7]), false);
#pragma unroll
    for (int q = 0; q < 8; ++q) bs[q] = 0.f;
  };

  // accumulators of half hh -> the wave's private LDS staging area, rows of the half x 64 columns
  auto stage = [&](int hh) {
    const int s_ = hh >> 1, i = hh & 1;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) st[((e & 3) + 8 * (e >> 2) + 4 * h) * 64 + j * 32 + r] = acc[s_][i][j][e];
    __builtin_amdgcn_wave_barrier();  // LDS serves one wave's accesses in issue order; keep the compiler from reordering
  };

  constexpr int NHALF = 2 * NSUB;
  Side sd;
  if constexpr (PAIR) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) st[i * 2048 + ((e & 3) + 8 * (e >> 2) + 4 * h) * 64 + psel * 32 + r] = acc[0][i][0][e];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int ug = 0; ug < 4; ug += PF) {
#pragma unroll
      for (int up = 0; up < PF; ++up) load_side(sd, up, psel, ug + up);
#pragma unroll
      for (int up = 0; up < PF; ++up) finish_pass(sd, up, psel, ug + up);
    }
    if (colsum) flush_colsum(part);
  } else if constexpr (PIPE) {
    // slot u holds the side inputs of pass u of the current half; the moment pass u has been combined and stored, the slot is
    // re-requested for pass u of the NEXT half: that load is in flight for a whole half's worth of staging, arithmetic and stores
    // (one set of side-input registers, as in the un-pipelined form)
#pragma unroll
    for (int u = 0; u < 4; ++u) load_side(sd, u, 0, u);
#pragma unroll
    for (int hh = 0; hh < NHALF; ++hh) {
      stage(hh);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        finish_pass(sd, u, hh, u);
        if (hh + 1 < NHALF) load_side(sd, u, hh + 1, u);
      }
      __builtin_amdgcn_wave_barrier();
      if (colsum && (hh & 1)) flush_colsum(part + (hh >> 1));
    }
  } else {
#pragma unroll
    for (int hh = 0; hh < NHALF; ++hh) {
      stage(hh);
#pragma unroll
      for (int ug = 0; ug < 4; ug += PF) {
#pragma unroll
        for (int up = 0; up < PF; ++up) load_side(sd, up, hh, ug + up);
#pragma unroll
        for (int up = 0; up < PF; ++up) finish_pass(sd, up, hh, ug + up);
      }
      __builtin_amdgcn_wave_barrier();
      if (colsum && (hh & 1)) flush_colsum(part + (hh >> 1));
    }
  }
}
#undef EH

// The register-staged kernels (gemm_f32_kernel, gemm_x3_kernel) do not carry the compact stride-2 residual (S2RES = false; the
// kinds with EF_RES_S2 fall through, and launch_gemm refuses a launch with rs2_on).  History: the first build with that feature
// compiled into them made fp32-mode model tests fail although the exact-fp32 mode never executes it — one wrong BatchNorm beta
// gradient, deterministic per build, dependent on the order of the tests — as had a build with the DPP lane-OR above (a silently
// aborted queue).  Both times the 128x128 instantiations, then compiled for 3 waves per SIMD, spilled ~300 registers; the failing
// build (-DCXRK_EXP_S2_EVERYWHERE=1) passes every test when compiled for 2 waves per SIMD, which is how these kernels are built
// now (CXRK_OCC in gemm_core.h).  The feature stays out of them anyway: they never see such a launch.
#ifndef CXRK_EXP_S2_EVERYWHERE
#define CXRK_EXP_S2_EVERYWHERE 0   // experiment: 1 = compile the compact-residual code into the register-staged kernels too (the failing build)
#endif
template <unsigned F>
__device__ __forceinline__ void epi64(f32x16 (&acc)[2][2], const EpiParams& ep, float* st, int M, int N, int row0, int col0,
                                      int part, int z, int lane) {
  epi_rows<F, 1, false, false, CXRK_EXP_S2_EVERYWHERE != 0>(&acc, ep, st, M, N, row0, col0, part, z, lane);
}

template <int KIND>
__device__ __forceinline__ void epi64_dispatch(int kind, f32x16 (&acc)[2][2], const EpiParams& ep, float* st, int M, int N, int row0,
                                               int col0, int part, int z, int lane) {
  if constexpr (KIND >= EPI_NKINDS) epi64<EF_GENERIC>(acc, ep, st, M, N, row0, col0, part, z, lane);
  else if constexpr (!CXRK_EXP_S2_EVERYWHERE && (EPI_KINDS[KIND] & EF_RES_S2) != 0u) epi64_dispatch<KIND + 1>(kind, acc, ep, st, M, N, row0, col0, part, z, lane);
  else {
    if (kind == KIND) epi64<EPI_KINDS[KIND]>(acc, ep, st, M, N, row0, col0, part, z, lane);
    else epi64_dispatch<KIND + 1>(kind, acc, ep, st, M, N, row0, col0, part, z, lane);
  }
}

// The LDS-DMA kernels (gemm_pw.h): NSUB slabs of 64 rows per wave, compiled-in feature sets pipelined (see epi_rows).
template <int KIND, int NSUB>
__device__ __forceinline__ void epi_pw_dispatch(int kind, f32x16 (*acc)[2][2], const EpiParams& ep, float* st, int M, int N, int row0,
                                                int col0, int part, int z, int lane) {
  if constexpr (KIND >= EPI_NKINDS) epi_rows<EF_GENERIC, NSUB, false>(acc, ep, st, M, N, row0, col0, part, z, lane);
  else {
    if (kind == KIND) epi_rows<EPI_KINDS[KIND], NSUB, (CXRK_EPI_PIPE != 0) && epi_has_side(EPI_KINDS[KIND]) && !epi_two_fp32_sides(EPI_KINDS[KIND])>(acc, ep, st, M, N, row0, col0, part, z, lane);
    else epi_pw_dispatch<KIND + 1, NSUB>(kind, acc, ep, st, M, N, row0, col0, part, z, lane);
  }
}

// The pair form (conv_halo.h): see epi_rows<..., PAIR = true>.
template <int KIND>
__device__ __forceinline__ void epi_pair_dispatch(int kind, f32x16 (*acc)[2][2], const EpiParams& ep, float* st, int M, int N, int row0,
                                                  int col0, int part, int lane, int psel) {
  if constexpr (KIND >= EPI_NKINDS) epi_rows<EF_GENERIC, 1, false, true>(acc, ep, st, M, N, row0, col0, part, 0, lane, psel);
  else {
    if (kind == KIND) epi_rows<EPI_KINDS[KIND], 1, false, true>(acc, ep, st, M, N, row0, col0, part, 0, lane, psel);
    else epi_pair_dispatch<KIND + 1>(kind, acc, ep, st, M, N, row0, col0, part, lane, psel);
  }
}

}  // namespace cxrk
