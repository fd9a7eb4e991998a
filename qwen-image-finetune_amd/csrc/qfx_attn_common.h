// qfx_attn_common.h -- helpers shared by the attention translation units (qfx_attn.hip: 32-query waves on the 16x16x32 MFMA;
// qfx_attn64.hip: 64-query waves on the 32x32x16 MFMA): LDS-DMA staging, bank swizzles, the XCD-aware block map, the epilogue pieces
// (fused rank-r projection, wide fragment stores).  Everything here has internal linkage.
#pragma once
#include "qfx_common.h"
#include <cstdlib>

namespace {

constexpr float LOG2E = 1.4426950408889634f;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4v;   // operand type of the transpose-read builtin

// One LDS-DMA piece (1 KiB per wave instruction).  Issued from an ASM statement (cdna_hip_programming.md 5.7: M0 is written in the statement
// that reads it): hipcc does not know these writes to LDS exist.  With the builtin it does, cannot tell which ring stage a ds_read
// touches, and guards fragment reads with s_waitcnt vmcnt(...) / turns its counted lgkmcnt waits into lgkmcnt(0) -- the asynchronous
// tile ring then waits for its youngest pieces inside the tile loop (found in round 5 on the 64-query kernels).  Every kernel tracks
// completion by hand: s_waitcnt vmcnt(N) + barrier at the tile seams.  -DQFX_ATTN_BUILTIN_DMA = the builtin of rounds 1-4 (A/B lever).
__device__ __forceinline__ void glds16a(const bf16_t* g, char* lds) {
  const uint32_t l = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(l) : "memory");
}
__device__ __forceinline__ void glds4a(const float* g, char* lds) {
  const uint32_t l = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(l) : "memory");
}
__device__ __forceinline__ void glds16b(const bf16_t* g, char* lds) {   // the builtin: hipcc counts it (and guards every LDS read with it)
  __builtin_amdgcn_global_load_lds((const QFX_AS1 void*)g, (QFX_AS3 void*)lds, 16, 0, 0);
}
__device__ __forceinline__ void glds16(const bf16_t* g, char* lds) {
#if defined(QFX_ATTN_BUILTIN_DMA)
  glds16b(g, lds);
#else
  glds16a(g, lds);
#endif
}
__device__ __forceinline__ void glds4(const float* g, char* lds) {
#if defined(QFX_ATTN_BUILTIN_DMA)
  __builtin_amdgcn_global_load_lds((const QFX_AS1 void*)g, (QFX_AS3 void*)lds, 4, 0, 0);
#else
  glds4a(g, lds);
#endif
}

template <int DH> __device__ __forceinline__ int swz_row(int row) {
  // DH == 128 (256-byte rows, 16 chunks).  f(r) = (m << 1) | h with h = (r>>3)&1, m = (r&7) ^ (h<<2) satisfies both
  // access patterns on 64 banks:
  //  * ds_read_b128 is serviced in 16-lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31}: rows A={0-3,12-15} at chunk c
  //    together with rows B={4-11} at chunk c^1.  f(A) = {0..7}, f(B) = {8..15} are both closed under ^1, so the two
  //    halves never share a 16-byte slot;
  //  * the transpose read (8 aligned consecutive rows x 32 bytes per 32-lane group) needs f(r)>>1 distinct over r&7.
  if constexpr (DH == 128) { const int h = (row >> 3) & 1; return ((((row & 7) ^ (h << 2)) << 1) | h); }
  else return (row >> 1) & 7;
}

// row * ld of a staging source as a FULL-rate 24-bit multiply (v_mul_u32_u24; the 32 / 64-bit forms are quarter rate and sit in
// every tile iteration): rows and row strides < 2^24 and S * ld < 2^32 are checked by the launchers (check_common).
__device__ __forceinline__ unsigned row_off(int row, int64_t ld) { return __umul24((unsigned)row, (unsigned)ld); }

// 64 rows x DH tile of a token-major tensor (rows s0..s0+63 clamped to S-1) -> LDS [64][DH], swizzled.
// base points at element [b, 0, h, 0]; ld = row stride in elements.
template <int DH>
__device__ __forceinline__ void stage_rows(char* lds, const bf16_t* base, int64_t ld, int s0, int S, int w, int lane) {
  constexpr int CPR = DH / 8;        // 16-byte chunks per row
  constexpr int RPI = 64 / CPR;      // rows per wave-instruction
  constexpr int NI = 16 / RPI;       // instructions per wave (16 rows per wave)
  const int rr = lane / CPR, c = lane % CPR;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int row = w * 16 + i * RPI + rr;
    int s = s0 + row; s = s < S ? s : S - 1;
    const int sc = c ^ swz_row<DH>(row);
    glds16(base + (row_off(s, ld) + (unsigned)(sc * 8)), lds + (w * 16 + i * RPI) * (DH * 2));
  }
}




__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
  bf16x8 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) { r[i] = (short)f2bf(a[i]); r[4 + i] = (short)f2bf(b[i]); }
  return r;
}

__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }  // raw v_exp_f32 (inputs <= 0 here)


// 64 rows x DH tile staged by NW waves (rows s0..s0+63 clamped to S-1) -> LDS [64][DH], swizzled.
// BUILTIN = true: the compiler-tracked form (the 32-query forward kernel, two waves per SIMD, measures 4% faster with it: its waits
// sit where the tile has landed anyway and the asm form's M0 save/restore costs issue slots; profiles/r05_attn_lds_dma.json).
template <int DH, int NW, bool BUILTIN = false>
__device__ __forceinline__ void stage_rows_n(char* lds, const bf16_t* base, int64_t ld, int s0, int S, int w, int lane) {
  constexpr int CPR = DH / 8, RPI = 64 / CPR, RPW = 64 / NW, NI = RPW / RPI;
  static_assert(NI >= 1, "too many waves for this tile");
  const int rr = lane / CPR, c = lane % CPR;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int row = w * RPW + i * RPI + rr;
    int s = s0 + row; s = s < S ? s : S - 1;
    const int sc = c ^ swz_row<DH>(row);
    const bf16_t* g = base + (row_off(s, ld) + (unsigned)(sc * 8));
    if constexpr (BUILTIN) glds16b(g, lds + (w * RPW + i * RPI) * (DH * 2));
    else glds16(g, lds + (w * RPW + i * RPI) * (DH * 2));
  }
}

// 1-D grid -> (seq block, head, batch) with an XCD-aware bijection: hardware dispatches workgroup i to XCD i % 8, so the
// virtual id walks each XCD through a CONTIGUOUS range of (batch, head, block) triples: all blocks of one head run on
// one XCD and its K/V (or Q/dO) tiles are fetched into one L2 instead of eight (PMC: fabric-side fetch per launch was
// 6x the tensor bytes with the plain blockIdx.x-fastest order).
__device__ __forceinline__ void attn_block_coord(int nx, int H, int& xb, int& h, int& b) {
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  const int v = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  xb = v % nx;
  const int hb = v / nx;
  h = hb % H;
  b = hb / H;
}

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

// Bisection levers for the run-to-run differences of the fused dQ epilogue (tools/nondet_bisect.py, profiles/r05_nondeterminism.md):
// -DQFX_NRB_PACK1 = one-instruction packing in norm_rope_bwd_row; -DQFX_NRB_FENCE=<bitmask> = 32 idle states + a scheduling barrier
// at point <bit> of the epilogue.  Both off in the product build.
#ifndef QFX_NRB_FENCE
#define QFX_NRB_FENCE 0
#endif
#ifndef QFX_NRB_NOPS
#define QFX_NRB_NOPS 1
#endif
#ifndef QFX_NRB_OPQ
// bitmask: make a group of intermediates opaque to the SLP vectoriser (no v_pk_*_f32 across it); zero instructions.  Bit 0 (the
// products acc * out_scale) is ON in the product build: it is the one group whose packed form was needed for the run-to-run
// differences of round 4 (profiles/r05_nondeterminism.md: 11 / 11 launches differ with it packed, 0 / 11 and 0 / 15 with it opaque;
// -fno-slp-vectorize likewise 0 / 15) -- the packed v_pk_mul_f32 vdst, s[scale:scale+1], v[accumulator pair] read MFMA accumulator
// pairs directly.
#define QFX_NRB_OPQ 1
#endif
#define NRB_OPQ4(bit, a, b, c, d)                                                          \
  do {                                                                                     \
    if constexpr (((QFX_NRB_OPQ) >> (bit)) & 1) {                                          \
      asm volatile("" : "+v"(a)); asm volatile("" : "+v"(b)); asm volatile("" : "+v"(c)); asm volatile("" : "+v"(d)); \
    }                                                                                      \
  } while (0)
#define NRB_FENCE(bit)                                                  \
  do {                                                                  \
    if constexpr (((QFX_NRB_FENCE) >> (bit)) & 1) {                     \
      __builtin_amdgcn_sched_barrier(0);                                \
      asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");                \
      __builtin_amdgcn_sched_barrier(0);                                \
    }                                                                   \
  } while (0)

// Rank-r down projection of ONE 16-row fragment held in the accumulator layout (ABI 6, qfx_head_lora): lane (g, li) owns row li,
// x[d] = the packed bf16 pairs of columns 16 d + 4 g + {0,1 | 2,3} of this head's dh columns -- exactly what the epilogues store.
//   part[h][row][c0 + j] = sum_n x[row][n] * (W_hi + W_lo)[j][h*dh + n]
// as D[i = j][col = row] = W_frag[i][k] X_frag[k][col] on the MFMA: the k-slots of a 32-deep step are the lane's own eight values
// of two adjacent d blocks (k = 8 g + r -> column 16 (2 ks) + 4 g + r, k = 8 g + 4 + r -> 16 (2 ks + 1) + 4 g + r); the weight
// operand comes from qfx_lora_pack's head-fragment image in exactly that order: one 16-byte load per lane, 1 KiB per wave.
// `frow` = the fragment's first row within its sample (a multiple of 16; with T % 16 == 0 a fragment is all text or all image).
template <int DH>
__device__ __forceinline__ void head_lora_frag(const qfx_head_lora& hl, int h, int T, int frow, int64_t jrow, bool row_ok,
                                               const u32x2 (&x)[DH / 16], int g, int li) {
  if (hl.part == nullptr) return;                                  // block-uniform
  NRB_FENCE(7);
  const bf16_t* wp = hl.w_pk[frow >= T ? 0 : 1];                   // wave-uniform
  if (wp == nullptr) return;
  constexpr int KS = DH / 32;
  const int nfs = hl.R >> 4;
#pragma unroll
  for (int nf = 0; nf < 2; ++nf) {
    if (nf >= nfs) break;
    const bf16_t* wf = wp + ((int64_t)(h * nfs + nf) * KS * 2 * 64 + 16 * g + li) * 8;
    u32x4 ah[KS], al[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { ah[ks] = *(const u32x4*)(wf + (ks * 2) * 512); al[ks] = *(const u32x4*)(wf + (ks * 2 + 1) * 512); }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const u32x4 bx = {x[2 * ks][0], x[2 * ks][1], x[2 * ks + 1][0], x[2 * ks + 1][1]};
      acc = MFMA(__builtin_bit_cast(bf16x8, ah[ks]), __builtin_bit_cast(bf16x8, bx), acc);
      acc = MFMA(__builtin_bit_cast(bf16x8, al[ks]), __builtin_bit_cast(bf16x8, bx), acc);
    }
    if (row_ok) *(f32x4*)(hl.part + (int64_t)h * hl.part_hstride + jrow * hl.ld_part + hl.c0 + nf * 16 + 4 * g) = acc;
  }
}

// Store one 16-row fragment held as packed bf16 pairs in the accumulator layout (lane (g, li): row li, u[d] = columns 16 d + 4 g .. + 3)
// as whole 16-byte pieces: v_permlane16_swap hands the two lanes of a (g, g ^ 1) pair each other's half of a 32-column block, so a lane
// stores 8 adjacent columns -- DH / 32 dwordx4 per lane instead of DH / 16 dwordx2 (the store tail of these kernels is store-ISSUE
// bound, microarch guide: per-lane dwordx2 at a row stride; half the instructions, same bytes).  `rowp` = the row's first column of
// this head; every lane of the wave must call (the exchange is cross-lane), `ok` masks the store.  wide == false (a base or row
// stride that is not 16-byte aligned): the 8-byte form.
template <int DH>
__device__ __forceinline__ void store_frag(bf16_t* rowp, const u32x2 (&u)[DH / 16], int g, bool ok, bool wide) {
  if (wide) {
    const int c0 = 16 * (g & 1) + 4 * (g & 2);
#pragma unroll
    for (int p = 0; p < DH / 32; ++p) {
      NRB_FENCE(5);
      const auto s0 = __builtin_amdgcn_permlane16_swap(u[2 * p][0], u[2 * p + 1][0], false, false);
      const auto s1 = __builtin_amdgcn_permlane16_swap(u[2 * p][1], u[2 * p + 1][1], false, false);
      NRB_FENCE(6);
      if (ok) *(u32x4*)(rowp + 32 * p + c0) = (u32x4){s0[0], s1[0], s0[1], s1[1]};
    }
  } else if (ok) {
#pragma unroll
    for (int d = 0; d < DH / 16; ++d) *(u32x2*)(rowp + d * 16 + 4 * g) = u[d];
  }
}
__device__ __forceinline__ bool rows_16b(const void* base, int64_t ld_elems) {
#if defined(QFX_ATTN_NARROW_STORE)      // A/B lever (tools/build_variants.py): the 8-byte store tail of rounds 1-3
  return false;
#else
  return (((uintptr_t)base | (uintptr_t)(ld_elems * 2)) & 15) == 0;
#endif
}


// Backward of QK RMSNorm + RoPE on one gradient row held in the 16x16 accumulator layout (lane (g, li): row li of fragment f,
// columns 16 d + 4 g + r): the arithmetic of qk_norm_rope_kernel<DH, true> (qfx_elem.hip) on the bf16-rounded attention gradient
//   dn = [rbf](rbf(dy * conj(rope)) * w) ;  xh = x * rstd ;  out = (dn - xh * mean(dn * xh)) * rstd
// with the row statistics folded over the four lane groups by two shuffles.  x = the saved pre-norm row, out packed bf16 per d.
template <int DH>
__device__ __forceinline__ void norm_rope_bwd_row(const f32x4 (&acc)[DH / 16][2], int f, float out_scale, const bf16_t* xrow,
                                                  const float* rrow, const bf16_t* wrow, float eps, int flags, u32x2 (&out)[DH / 16]) {
  constexpr int DF = DH / 16;
  float xh[DF][4], dn[DF][4];
  float ss = 0.f;
  NRB_FENCE(0);
#pragma unroll
  for (int d = 0; d < DF; ++d) {
    const u32x2 ux = *(const u32x2*)(xrow + d * 16);
    xh[d][0] = __uint_as_float(ux[0] << 16); xh[d][1] = __uint_as_float(ux[0] & 0xffff0000u);
    xh[d][2] = __uint_as_float(ux[1] << 16); xh[d][3] = __uint_as_float(ux[1] & 0xffff0000u);
    NRB_OPQ4(4, xh[d][0], xh[d][1], xh[d][2], xh[d][3]);
#pragma unroll
    for (int r = 0; r < 4; ++r) ss += xh[d][r] * xh[d][r];
  }
  ss += __shfl_xor(ss, 16);
  ss += __shfl_xor(ss, 32);
  const float rstd = rsqrtf(ss / (float)DH + eps);
  float dot = 0.f;
  NRB_FENCE(1);
#pragma unroll
  for (int d = 0; d < DF; ++d) {
    const f32x4 cs = *(const f32x4*)(rrow + d * 16);            // (cos, sin) of the pairs (16 d + 4 g)/2 and +1
    const u32x2 uw = *(const u32x2*)(wrow + d * 16);
    const float w0 = __uint_as_float(uw[0] << 16), w1 = __uint_as_float(uw[0] & 0xffff0000u);
    const float w2 = __uint_as_float(uw[1] << 16), w3 = __uint_as_float(uw[1] & 0xffff0000u);
    float e0 = rbf(acc[d][f][0] * out_scale), e1 = rbf(acc[d][f][1] * out_scale);
    float e2 = rbf(acc[d][f][2] * out_scale), e3 = rbf(acc[d][f][3] * out_scale);
    NRB_OPQ4(0, e0, e1, e2, e3);
    float d0 = rbf(e0 * cs[0] + e1 * cs[1]), d1 = rbf(-e0 * cs[1] + e1 * cs[0]);     // dy * conj(f)
    float d2 = rbf(e2 * cs[2] + e3 * cs[3]), d3 = rbf(-e2 * cs[3] + e3 * cs[2]);
    NRB_OPQ4(1, d0, d1, d2, d3);
    dn[d][0] = (flags & 1) ? d0 * w0 : rbf(d0 * w0);
    dn[d][1] = (flags & 1) ? d1 * w1 : rbf(d1 * w1);
    dn[d][2] = (flags & 1) ? d2 * w2 : rbf(d2 * w2);
    dn[d][3] = (flags & 1) ? d3 * w3 : rbf(d3 * w3);
    NRB_OPQ4(2, dn[d][0], dn[d][1], dn[d][2], dn[d][3]);
    if constexpr (((QFX_NRB_FENCE) >> 9) & 1) {
      asm volatile("s_nop 1" : "+v"(dn[d][0]), "+v"(dn[d][1]), "+v"(dn[d][2]), "+v"(dn[d][3]));
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { xh[d][r] *= rstd; dot += dn[d][r] * xh[d][r]; }
    NRB_OPQ4(3, xh[d][0], xh[d][1], xh[d][2], xh[d][3]);
  }
  NRB_FENCE(2);
  dot += __shfl_xor(dot, 16);
  dot += __shfl_xor(dot, 32);
  dot /= (float)DH;
  NRB_FENCE(3);
  // pack2bf_scalar, not pack2bf (round 4), and QFX_NRB_OPQ bit 0 (round 5).  With the one-instruction packing, hipcc's SLP vectoriser
  // re-shapes the whole function into v_pk_*_f32 pairs, and the dQ kernel then came out different from run to run: in ~0.3 % of the
  // 16-row fragments ONE column -- an odd r of lane group g = 3, i.e. the HIGH register of a packed pair, lanes 48-63 -- is wrong in
  // all 16 rows BEFORE the row statistics are formed (every other column then moves by an ulp through `dot`).  Round 5 bisection
  // (tools/nondet_bisect.py, tools/hazard_probe/, profiles/r05_nondeterminism.md): not a missing wait (-amdgpu-waitcnt-forcezero: still
  // 11 / 11), not a fixed-distance hazard (32 idle states at 11 places, 2-16 between producer and consumer: still differs), not the
  // store tail or the head-LoRA MFMAs (off: still differs); gone with -fno-slp-vectorize and gone when ONLY the products
  // acc * out_scale are kept scalar.  The instruction pairs replayed in isolation (2e9 checks each, with VMEM returns, SALU rewrites
  // of the unused SGPR half and a partner wave's MFMAs) never fail: the trigger needs this kernel's surroundings and was not reduced
  // further.  Both guards stay; tests/test_kernels_gpu.py::test_attention_kernels_are_bit_reproducible watches all three kernels.
#pragma unroll
  for (int d = 0; d < DF; ++d) {
#if defined(QFX_NRB_PACK1)
    float o0 = (dn[d][0] - xh[d][0] * dot) * rstd, o1 = (dn[d][1] - xh[d][1] * dot) * rstd;
    float o2 = (dn[d][2] - xh[d][2] * dot) * rstd, o3 = (dn[d][3] - xh[d][3] * dot) * rstd;
    NRB_OPQ4(5, o0, o1, o2, o3);
    if constexpr (((QFX_NRB_FENCE) >> 8) & 1) {       // producers (possibly v_pk_*_f32) | 2 idle states | consumers (v_cvt_pk_bf16_f32)
      asm volatile("s_nop %c4" : "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3) : "i"(QFX_NRB_NOPS));
    }
    out[d][0] = pack2bf(o0, o1);
    out[d][1] = pack2bf(o2, o3);
    if constexpr (((QFX_NRB_FENCE) >> 10) & 1) {      // consumers | 2 idle states | the next d's producers (WAR on the cvt's sources)
      asm volatile("s_nop 1" : "+v"(out[d][0]), "+v"(out[d][1]));
    }
#else
    out[d][0] = pack2bf_scalar((dn[d][0] - xh[d][0] * dot) * rstd, (dn[d][1] - xh[d][1] * dot) * rstd);
    out[d][1] = pack2bf_scalar((dn[d][2] - xh[d][2] * dot) * rstd, (dn[d][3] - xh[d][3] * dot) * rstd);
#endif
  }
  NRB_FENCE(4);
}

// The same backward with its operands taken in two steps (round 5, one-wave-per-SIMD kernels): left to itself hipcc requests the RoPE and
// weight pieces of a row two at a time between the arithmetic that uses them -- eight dependent memory round trips per 16-row fragment,
// ~6000 cycles each fragment with nothing else on the SIMD to cover them (dQ epilogue 31 k cycles against 7 k without the fusion,
// tools/attn64_timing.py).  nrb_load() requests all 24 pieces of a fragment at once (a scheduling barrier keeps the requests together
// and ahead of what follows); the caller requests fragment f + 1 before it works on fragment f.
template <int DH> struct NrbOps { u32x2 x[DH / 16]; f32x4 cs[DH / 16]; u32x2 w[DH / 16]; };
template <int DH>
__device__ __forceinline__ void nrb_load(NrbOps<DH>& o, const bf16_t* xrow, const float* rrow, const bf16_t* wrow) {
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int d = 0; d < DH / 16; ++d) o.x[d] = *(const u32x2*)(xrow + d * 16);
#pragma unroll
  for (int d = 0; d < DH / 16; ++d) o.cs[d] = *(const f32x4*)(rrow + d * 16);
#pragma unroll
  for (int d = 0; d < DH / 16; ++d) o.w[d] = *(const u32x2*)(wrow + d * 16);
  __builtin_amdgcn_sched_barrier(0);
}
template <int DH>
__device__ __forceinline__ void norm_rope_bwd_ops(const f32x4 (&acc)[DH / 16], float out_scale, const NrbOps<DH>& o, float eps, int flags,
                                                  u32x2 (&out)[DH / 16]) {
  constexpr int DF = DH / 16;
  float xh[DF][4], dn[DF][4];
  float ss = 0.f;
#pragma unroll
  for (int d = 0; d < DF; ++d) {
    xh[d][0] = __uint_as_float(o.x[d][0] << 16); xh[d][1] = __uint_as_float(o.x[d][0] & 0xffff0000u);
    xh[d][2] = __uint_as_float(o.x[d][1] << 16); xh[d][3] = __uint_as_float(o.x[d][1] & 0xffff0000u);
#pragma unroll
    for (int r = 0; r < 4; ++r) ss += xh[d][r] * xh[d][r];
  }
  ss += __shfl_xor(ss, 16);
  ss += __shfl_xor(ss, 32);
  const float rstd = rsqrtf(ss / (float)DH + eps);
  float dot = 0.f;
#pragma unroll
  for (int d = 0; d < DF; ++d) {
    const f32x4 cs = o.cs[d];
    const float w0 = __uint_as_float(o.w[d][0] << 16), w1 = __uint_as_float(o.w[d][0] & 0xffff0000u);
    const float w2 = __uint_as_float(o.w[d][1] << 16), w3 = __uint_as_float(o.w[d][1] & 0xffff0000u);
    float e0 = rbf(acc[d][0] * out_scale), e1 = rbf(acc[d][1] * out_scale);
    float e2 = rbf(acc[d][2] * out_scale), e3 = rbf(acc[d][3] * out_scale);
    NRB_OPQ4(0, e0, e1, e2, e3);                                   // see norm_rope_bwd_row: these products stay scalar
    const float d0 = rbf(e0 * cs[0] + e1 * cs[1]), d1 = rbf(-e0 * cs[1] + e1 * cs[0]);     // dy * conj(f)
    const float d2 = rbf(e2 * cs[2] + e3 * cs[3]), d3 = rbf(-e2 * cs[3] + e3 * cs[2]);
    dn[d][0] = (flags & 1) ? d0 * w0 : rbf(d0 * w0);
    dn[d][1] = (flags & 1) ? d1 * w1 : rbf(d1 * w1);
    dn[d][2] = (flags & 1) ? d2 * w2 : rbf(d2 * w2);
    dn[d][3] = (flags & 1) ? d3 * w3 : rbf(d3 * w3);
#pragma unroll
    for (int r = 0; r < 4; ++r) { xh[d][r] *= rstd; dot += dn[d][r] * xh[d][r]; }
  }
  dot += __shfl_xor(dot, 16);
  dot += __shfl_xor(dot, 32);
  dot /= (float)DH;
#pragma unroll
  for (int d = 0; d < DF; ++d) {
    out[d][0] = pack2bf_scalar((dn[d][0] - xh[d][0] * dot) * rstd, (dn[d][1] - xh[d][1] * dot) * rstd);
    out[d][1] = pack2bf_scalar((dn[d][2] - xh[d][2] * dot) * rstd, (dn[d][3] - xh[d][3] * dot) * rstd);
  }
}

}  // namespace

