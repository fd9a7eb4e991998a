// qfx_attn.hip -- joint [text|image] non-causal flash attention for gfx950, forward + backward.
//
// MFMA 16x16x32 bf16 everywhere.  Every tile lives in LDS ROW-major ([tokens][dh], LDS-DMA, double-buffered); operands that
// contract over dh are read with ds_read_b128, operands that contract over TOKENS (V^T for PV, K^T for dQ, Q^T / dO^T for
// dK / dV) come from the same tiles through the gfx950 transpose read ds_read_b64_tr_b16 -- no transposed copy exists in HBM.
// Scores are computed "swapped" (S^T = K Q^T, i.e. D[i=key][j=query]) so that a lane owns ONE
// query column: softmax statistics are lane-local plus two cross-lane steps, and the bf16-packed
// P registers are directly the B operand of the PV MFMA under the key permutation
//     pi(g, j) = 16*(2t) + 4g + j (j<4) ; 16*(2t+1) + 4g + (j-4) (j>=4)
// which the V^T fragment read applies too (a consistent k-permutation leaves a contraction intact).
// Tiles arrive by LDS-DMA with the bank swizzle applied on the global source address.
//
// lse2 is the log2-domain logsumexp of (scale * q.k + mask): P = exp2(scale*log2e * s + mask*log2e - lse2).
#include "qfx_attn_common.h"

#include <atomic>
#include <cstring>
#include <mutex>

namespace qfxi {      // qfx_attn64.hip, qfx_attn_bwd1.hip
int launch_attn_fwd64(const qfx_attn_args* a, hipStream_t stream);
int launch_attn_fwd64p(const qfx_attn_args* a, hipStream_t stream);
int launch_attn_bwd_dq64(const qfx_attn_args* a, hipStream_t stream);
int launch_attn_bwd1(const qfx_attn_args* a, hipStream_t stream);
int attn_bwd1_grid(const qfx_attn_args* a, int* nkb_out);
}

namespace {

// =============================================================================================
// forward: block = 128 queries (4 waves x 32), loop over 64-key tiles, K/V^T double-buffered in LDS
// NW = waves per block (32 queries each).  8 waves / 256 queries, one block per CU, is used when it quantises better onto the
// 256 CUs (S = 2432: 10 x 24 = 240 blocks, 94 % of the CUs, against 456 blocks on 512 half-CU slots = 89 %) and halves the K/V
// LDS-DMA traffic per query; 4 waves / 128 queries, two blocks per CU, otherwise.
#ifndef ATTN_DEFER_MAX
#define ATTN_DEFER_MAX 8.0f     // log2 units; -DATTN_DEFER_MAX=0.0f = the eager running maximum of rounds 1-3
#endif
template <int DH, int NW>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void attn_fwd_kernel(const qfx_attn_args a) {
  constexpr int KC = DH / 32, DF = DH / 16;
  constexpr int TB = 64 * DH * 2;  // bytes per tile (K tile and V^T tile are the same size)
  __shared__ __attribute__((aligned(16))) char smem[4 * TB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  int xb, h, b;
  attn_block_coord((a.S + 32 * NW - 1) / (32 * NW), a.H, xb, h, b);
  const int q0 = xb * (32 * NW) + w * 32;
  const int S = a.S;

  const bf16_t* Kb = a.K + (int64_t)b * S * a.ldk + h * DH;
  const bf16_t* Vb = a.V + (int64_t)b * S * a.ldv + h * DH;

  bf16x8 qf[2][KC];
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    int q = q0 + f * 16 + li; q = q < S ? q : S - 1;
    const bf16_t* qp = a.Q + ((int64_t)b * S + q) * a.ldq + h * DH + 8 * g;
#pragma unroll
    for (int kk = 0; kk < KC; ++kk) qf[f][kk] = *(const bf16x8*)(qp + kk * 32);
  }
  f32x4 oacc[DF][2];
#pragma unroll
  for (int d = 0; d < DF; ++d) { oacc[d][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; oacc[d][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  float mrow[2] = {-INFINITY, -INFINITY}, lrow[2] = {0.f, 0.f};
  const float c2 = a.scale * LOG2E;
  const float* maskb = a.key_mask ? a.key_mask + (int64_t)b * S : nullptr;

  // lane-constant LDS byte offsets of the fragment reads (tile/fragment index only adds an immediate)
  int koff[KC];
#pragma unroll
  for (int kk = 0; kk < KC; ++kk) koff[kk] = li * (DH * 2) + (((kk * 4 + g) ^ swz_row<DH>(li)) << 4);
  // V^T fragments come from the ROW-major V tile through the transpose read (see read_trfrag): toff(d) = toff0 ^ (d << 5)
  const int tr1 = 4 * g + (li >> 2);
  const int toff0 = tr1 * (DH * 2) + ((((li & 3) >> 1) ^ swz_row<DH>(tr1)) << 4) + (li & 1) * 8;
  const int ntiles = (S + 63) / 64;
  f32x4 sacc[4][2];
  auto qk = [&](const char* sK) {
#pragma unroll
    for (int kf = 0; kf < 4; ++kf) { sacc[kf][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; sacc[kf][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    // k-chunk outer, key fragment inner: 8 independent accumulators between two MFMAs of the same chain
#pragma unroll
    for (int kk = 0; kk < KC; ++kk) {
#pragma unroll
      for (int kf = 0; kf < 4; ++kf) {
        const bf16x8 kfr = *(const bf16x8*)(sK + koff[kk] + kf * (16 * DH * 2));
        sacc[kf][0] = MFMA(kfr, qf[0][kk], sacc[kf][0]);
        sacc[kf][1] = MFMA(kfr, qf[1][kk], sacc[kf][1]);
      }
    }
  };
  auto sm_pv = [&](const char* sV, int j0) {
    // online softmax in the log2 domain (lane owns query column li of each q-fragment; keys 16kf+4g+r).
    // Masking work only where it can matter: the ragged last tile or an additive key mask.
    const bool need_mask = (j0 + 64 > S) || (maskb != nullptr);   // wave-uniform
    float mx[2] = {-INFINITY, -INFINITY};
    if (need_mask) {
#pragma unroll
      for (int kf = 0; kf < 4; ++kf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = j0 + kf * 16 + 4 * g + r;
          const bool ok = key < S;
          const float mk = (maskb && ok) ? maskb[key] * LOG2E : 0.f;
#pragma unroll
          for (int f = 0; f < 2; ++f) {
            const float sv = ok ? sacc[kf][f][r] * c2 + mk : -INFINITY;
            sacc[kf][f][r] = sv;
            mx[f] = fmaxf(mx[f], sv);
          }
        }
    } else {
      // common case: the maximum is taken over the RAW scores (scale > 0 commutes with max) and the scale rides in the FMA that
      // feeds v_exp -- one VALU op per score less than scaling first
#pragma unroll
      for (int kf = 0; kf < 4; ++kf)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int f = 0; f < 2; ++f) mx[f] = fmaxf(mx[f], sacc[kf][f][r]);
    }
    const float cs = need_mask ? 1.0f : c2;   // scores are already scaled (and masked) on the masked path
    bf16x8 pb[2][2];
    // Lazy reference maximum: the exponent reference mrow (and with it O and l) moves only when some score of this wave's tile exceeds
    // it by more than 2^8 -- P = exp2(s - mrow) <= 256 otherwise, exact in the bf16 / fp32 ranges involved, and O / l = softmax(s) V
    // whatever reference is used (lse2 = mrow + log2 l likewise).  The test is lane-local (each lane's own 16 scores of a query row
    // against that row's reference, one wave vote); the cross-lane row maximum is only formed when the reference has to move.  On
    // typical data the reference settles within the first tiles and both the reduction and the 64-register rescale of O leave the
    // loop.  (-inf - -inf = NaN compares false: the first tile and fully masked rows take the update path.)
    if (!__all(mx[0] * cs - mrow[0] <= ATTN_DEFER_MAX && mx[1] * cs - mrow[1] <= ATTN_DEFER_MAX)) {
      float alpha[2];
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        float m = mx[f];
        {   // cross-lane maximum over the four 16-lane groups on the VALU (v_permlane16_swap / v_permlane32_swap): no LDS crossbar trip
          const uint32_t u0 = __float_as_uint(m);
          const auto r0 = __builtin_amdgcn_permlane16_swap(u0, u0, false, false);
          m = fmaxf(__uint_as_float(r0[0]), __uint_as_float(r0[1]));
          const uint32_t u1 = __float_as_uint(m);
          const auto r1 = __builtin_amdgcn_permlane32_swap(u1, u1, false, false);
          m = fmaxf(__uint_as_float(r1[0]), __uint_as_float(r1[1]));
        }
        const float mnew = fmaxf(mrow[f], m * cs);
        alpha[f] = fexp2(mrow[f] - ((mnew == -INFINITY) ? 0.f : mnew));
        mrow[f] = mnew;
        lrow[f] *= alpha[f];
      }
#pragma unroll
      for (int d = 0; d < DF; ++d)
#pragma unroll
        for (int r = 0; r < 4; ++r) { oacc[d][0][r] *= alpha[0]; oacc[d][1][r] *= alpha[1]; }
    }
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const float msafe = (mrow[f] == -INFINITY) ? 0.f : mrow[f];
      float ps = 0.f;
#pragma unroll
      for (int kf = 0; kf < 4; ++kf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = fexp2(fmaf(sacc[kf][f][r], cs, -msafe));
          sacc[kf][f][r] = p;
          ps += p;
        }
      lrow[f] += ps;
      pb[f][0] = pack8(sacc[0][f], sacc[1][f]);
      pb[f][1] = pack8(sacc[2][f], sacc[3][f]);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int d = 0; d < DF; ++d) {
        const int o = (toff0 ^ (d << 5)) + t * (32 * DH * 2);
        const bf16x4 lo = __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((QFX_AS3 bf16x4v*)(sV + o)));
        const bf16x4 hi = __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((QFX_AS3 bf16x4v*)(sV + o + 16 * DH * 2)));
        bf16x8 vfr;
        vfr[0] = lo[0]; vfr[1] = lo[1]; vfr[2] = lo[2]; vfr[3] = lo[3];
        vfr[4] = hi[0]; vfr[5] = hi[1]; vfr[6] = hi[2]; vfr[7] = hi[3];
        oacc[d][0] = MFMA(vfr, pb[0][t], oacc[d][0]);
        oacc[d][1] = MFMA(vfr, pb[1][t], oacc[d][1]);
      }
  };
  stage_rows_n<DH, NW, true>(smem, Kb, a.ldk, 0, S, w, lane);
  stage_rows_n<DH, NW, true>(smem + TB, Vb, a.ldv, 0, S, w, lane);
  __syncthreads();
#if defined(QFX_ATTN_TIMING)
  uint64_t tq = 0, tp = 0, tb = 0;
#endif
  for (int jt = 0; jt < ntiles; ++jt) {
    const char* sK = smem + (jt & 1) * 2 * TB;
    const char* sV = sK + TB;
    if (jt + 1 < ntiles) {
      char* nK = smem + ((jt + 1) & 1) * 2 * TB;
      stage_rows_n<DH, NW, true>(nK, Kb, a.ldk, (jt + 1) * 64, S, w, lane);
      stage_rows_n<DH, NW, true>(nK + TB, Vb, a.ldv, (jt + 1) * 64, S, w, lane);
    }
#if defined(QFX_ATTN_TIMING)
    const uint64_t t0 = __builtin_readcyclecounter();
    qk(sK);
    asm volatile("s_nop 0" ::: "memory");
    const uint64_t t1 = __builtin_readcyclecounter();
    sm_pv(sV, jt * 64);
    const uint64_t t2 = __builtin_readcyclecounter();
    __syncthreads();
    const uint64_t t3 = __builtin_readcyclecounter();
    tq += t1 - t0; tp += t2 - t1; tb += t3 - t2;
#else
    qk(sK);
    sm_pv(sV, jt * 64);
    __syncthreads();
#endif
  }
#if defined(QFX_ATTN_TIMING)
  if (lane == 0 && blockIdx.x < 16) {
    float* dbg = a.lse2 + ((int64_t)a.B * a.H) * a.S_pad;      // caller over-allocates lse2 by 16*8*4 floats in the timing build
    dbg[(blockIdx.x * 8 + w) * 4 + 0] = (float)tq; dbg[(blockIdx.x * 8 + w) * 4 + 1] = (float)tp;
    dbg[(blockIdx.x * 8 + w) * 4 + 2] = (float)tb; dbg[(blockIdx.x * 8 + w) * 4 + 3] = (float)ntiles;
  }
#endif
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    float l = lrow[f];
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const int q = q0 + f * 16 + li;
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    u32x2 u[DF];
#pragma unroll
    for (int d = 0; d < DF; ++d) {
      u[d][0] = pack2bf(oacc[d][f][0] * inv, oacc[d][f][1] * inv);
      u[d][1] = pack2bf(oacc[d][f][2] * inv, oacc[d][f][3] * inv);
    }
    store_frag<DH>(a.O + ((int64_t)b * S + (q < S ? q : S - 1)) * a.ldo + h * DH, u, g, q < S, rows_16b(a.O, a.ldo));
    if (q < S && g == 0) a.lse2[((int64_t)b * a.H + h) * a.S_pad + q] = mrow[f] + log2f(l);
    // rank-r down projection of the out-projection adapter on the rows just produced (ABI 6)
    if (q0 + f * 16 < S) head_lora_frag<DH>(a.hl[0], h, a.T, q0 + f * 16, (int64_t)b * S + (q < S ? q : S - 1), q < S, u, g, li);
  }
}

// =============================================================================================
// dsum[b,h,s] = sum_d dO[b,s,h,d] * O[b,s,h,d]
template <int DH>
__global__ __launch_bounds__(256) void attn_prep_kernel(const qfx_attn_args a) {
  constexpr int LPI = DH / 8;
  const int sub = threadIdx.x % LPI;
  const int64_t item = (int64_t)blockIdx.x * (256 / LPI) + threadIdx.x / LPI;
  const int64_t n = (int64_t)a.B * a.S * a.H;
  const int64_t it = item < n ? item : n - 1;
  const int64_t tok = it / a.H;
  const int h = (int)(it % a.H);
  const u32x4 uo = *(const u32x4*)(a.O + tok * a.ldo + h * DH + sub * 8);
  const u32x4 ud = *(const u32x4*)(a.dO + tok * a.lddo + h * DH + sub * 8);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    s += __uint_as_float(uo[i] << 16) * __uint_as_float(ud[i] << 16);
    s += __uint_as_float(uo[i] & 0xffff0000u) * __uint_as_float(ud[i] & 0xffff0000u);
  }
#pragma unroll
  for (int o = 1; o < LPI; o <<= 1) s += __shfl_xor(s, o);
  if (sub == 0 && item < n) {
    const int bb = (int)(tok / a.S), ss = (int)(tok % a.S);
    a.dsum[((int64_t)bb * a.H + h) * a.S_pad + ss] = s;
  }
}


// =============================================================================================
// dQ: block = 128 queries (4 waves x 32), loop over 64-key tiles (K, V row tiles + K^T column tile)
template <int DH, int NW>   // NW as in attn_fwd_kernel
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void attn_bwd_dq_kernel(const qfx_attn_args a) {
  constexpr int KC = DH / 32, DF = DH / 16;
  constexpr int TB = 64 * DH * 2;
  __shared__ __attribute__((aligned(16))) char smem[4 * TB];   // 2 x [K | V] row tiles (double-buffered LDS-DMA)
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  int xb, h, b;
  attn_block_coord((a.S + 32 * NW - 1) / (32 * NW), a.H, xb, h, b);
  const int q0 = xb * (32 * NW) + w * 32;
  const int S = a.S;
  const bf16_t* Kb = a.K + (int64_t)b * S * a.ldk + h * DH;
  const bf16_t* Vb = a.V + (int64_t)b * S * a.ldv + h * DH;

  // DMA sources as uniform base + 32-bit lane offset (rows past S clamp to S-1)
  constexpr int CPR = DH / 8, RPI = 64 / CPR, RPW = 64 / NW, NIS = RPW / RPI;
  auto stage = [&](int jt, int buf) {
    const int j0 = jt * 64;
    char* dK_ = smem + buf * 2 * TB;
#pragma unroll
    for (int i = 0; i < NIS; ++i) {
      const int row = w * RPW + i * RPI + lane / CPR;
      const int sc8 = ((lane % CPR) ^ swz_row<DH>(row)) * 8;
      int sr = j0 + row; sr = sr < S ? sr : S - 1;
      glds16(Kb + (row_off(sr, a.ldk) + (unsigned)sc8), dK_ + (w * RPW + i * RPI) * (DH * 2));
      glds16(Vb + (row_off(sr, a.ldv) + (unsigned)sc8), dK_ + TB + (w * RPW + i * RPI) * (DH * 2));
    }
  };
  stage(0, 0);

  bf16x8 qf[2][KC], dof[2][KC];
  float lse[2], dsm[2];
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    int q = q0 + f * 16 + li; q = q < S ? q : S - 1;
    const bf16_t* qp = a.Q + ((int64_t)b * S + q) * a.ldq + h * DH + 8 * g;
    const bf16_t* dp = a.dO + ((int64_t)b * S + q) * a.lddo + h * DH + 8 * g;
#pragma unroll
    for (int kk = 0; kk < KC; ++kk) { qf[f][kk] = *(const bf16x8*)(qp + kk * 32); dof[f][kk] = *(const bf16x8*)(dp + kk * 32); }
    lse[f] = a.lse2[((int64_t)b * a.H + h) * a.S_pad + q];
    // dsum[q] = sum_d dO[q,d] * O[q,d] is computed HERE (the lane already holds its 32 dO elements of row q) and published
    // for the dK/dV kernel, which runs after this one: no separate qfx_attn_bwd_prep launch or pass over O / dO is needed.
    const bf16_t* op = a.O + ((int64_t)b * S + q) * a.ldo + h * DH + 8 * g;
    float part = 0.f;
#pragma unroll
    for (int kk = 0; kk < KC; ++kk) {
      const bf16x8 ov = *(const bf16x8*)(op + kk * 32);
#pragma unroll
      for (int j = 0; j < 8; ++j) part += bf2f((bf16_t)dof[f][kk][j]) * bf2f((bf16_t)ov[j]);
    }
    part += __shfl_xor(part, 16);
    part += __shfl_xor(part, 32);
    dsm[f] = part;
    if (g == 0 && q0 + f * 16 + li < S) a.dsum[((int64_t)b * a.H + h) * a.S_pad + q] = part;
  }
#if !defined(QFX_ATTN_NO_CLAIM)
#pragma unroll
  for (int f = 0; f < 2; ++f) {      // round 6: claim the prologue's loads before the tile loop (see attn_bwd_dkv_kernel)
#pragma unroll
    for (int kk = 0; kk < KC; ++kk) { asm volatile("" : "+v"(qf[f][kk])); asm volatile("" : "+v"(dof[f][kk])); }
    asm volatile("" : "+v"(lse[f]), "+v"(dsm[f]));
  }
#endif
  f32x4 dq[DF][2];
#pragma unroll
  for (int d = 0; d < DF; ++d) { dq[d][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; dq[d][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  const float c2 = a.scale * LOG2E;
  const float* maskb = a.key_mask ? a.key_mask + (int64_t)b * S : nullptr;

  // lane-constant LDS byte offsets: koff(kk) = koff0 ^ (kk << 6), toff(d) = toff0 ^ (d << 5) (see the dK/dV kernel)
  const int koff0 = li * (DH * 2) + ((g ^ swz_row<DH>(li)) << 4);
  const int tr1 = 4 * g + (li >> 2);
  const int toff0 = tr1 * (DH * 2) + ((((li & 3) >> 1) ^ swz_row<DH>(tr1)) << 4) + (li & 1) * 8;
  auto koff = [&](int kk) { return koff0 ^ (kk << 6); };
  auto trfrag = [&](const char* tile, int d, int t) {
    const int o = (toff0 ^ (d << 5)) + t * (32 * DH * 2);
    const bf16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((QFX_AS3 bf16x4v*)(tile + o));
    const bf16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((QFX_AS3 bf16x4v*)(tile + o + 16 * DH * 2));
    bf16x8 r;
    const bf16x4 l4 = __builtin_bit_cast(bf16x4, lo), h4 = __builtin_bit_cast(bf16x4, hi);
    r[0] = l4[0]; r[1] = l4[1]; r[2] = l4[2]; r[3] = l4[3]; r[4] = h4[0]; r[5] = h4[1]; r[6] = h4[2]; r[7] = h4[3];
    return r;
  };
  const int ntiles = (S + 63) / 64;
  for (int jt = 0; jt < ntiles; ++jt) {
    const int j0 = jt * 64;
    const int buf = jt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (jt + 1 < ntiles) stage(jt + 1, buf ^ 1);
    const char* sK = smem + buf * 2 * TB;
    const char* sV = sK + TB;
    const bool need_mask = (j0 + 64 > S) || (maskb != nullptr);   // wave-uniform
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f32x4 sa[2][2], da[2][2];  // [kf local][qf]
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        const int kf = 2 * t + k2;
        sa[k2][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; sa[k2][1] = sa[k2][0]; da[k2][0] = sa[k2][0]; da[k2][1] = sa[k2][0];
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) {
          const bf16x8 kfr = *(const bf16x8*)(sK + koff(kk) + kf * (16 * DH * 2));
          const bf16x8 vfr = *(const bf16x8*)(sV + koff(kk) + kf * (16 * DH * 2));
          sa[k2][0] = MFMA(kfr, qf[0][kk], sa[k2][0]);
          sa[k2][1] = MFMA(kfr, qf[1][kk], sa[k2][1]);
          da[k2][0] = MFMA(vfr, dof[0][kk], da[k2][0]);
          da[k2][1] = MFMA(vfr, dof[1][kk], da[k2][1]);
        }
        if (need_mask) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = j0 + kf * 16 + 4 * g + r;
            const bool ok = key < S;
            const float mk = (maskb && ok) ? maskb[key] * LOG2E : 0.f;
#pragma unroll
            for (int f = 0; f < 2; ++f) {
              const float p = ok ? fexp2(sa[k2][f][r] * c2 + mk - lse[f]) : 0.f;
              sa[k2][f][r] = ok ? p * (da[k2][f][r] - dsm[f]) : 0.f;
            }
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int f = 0; f < 2; ++f) {
              const float p = fexp2(sa[k2][f][r] * c2 - lse[f]);
              sa[k2][f][r] = p * (da[k2][f][r] - dsm[f]);
            }
        }
      }
      const bf16x8 ds0 = pack8(sa[0][0], sa[1][0]);
      const bf16x8 ds1 = pack8(sa[0][1], sa[1][1]);
#pragma unroll
      for (int d = 0; d < DF; ++d) {
        const bf16x8 ktf = trfrag(sK, d, t);
        dq[d][0] = MFMA(ktf, ds0, dq[d][0]);
        dq[d][1] = MFMA(ktf, ds1, dq[d][1]);
      }
    }
  }
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const int q = q0 + f * 16 + li;
    const int qc = q < S ? q : S - 1;       // rows past S compute on row S-1 (the shuffles below need every lane) and are not stored
    bf16_t* op = a.dQ + ((int64_t)b * S + qc) * a.lddq + h * DH;
    const bool wide = rows_16b(a.dQ, a.lddq);
    if (a.qk_saved) {       // block-uniform: d(pre-norm q) straight from the accumulators (QK RMSNorm + RoPE backward fused here)
      u32x2 u[DF];
      norm_rope_bwd_row<DH>(dq, f, a.scale, a.qk_saved + ((int64_t)b * S + qc) * a.ld_saved + h * DH + 4 * g,
                            a.rope + (int64_t)b * a.rope_bstride + ((int64_t)qc * (DH / 2) + 2 * g) * 2,
                            (qc < a.T ? a.wq_txt : a.wq_img) + 4 * g, a.norm_eps, a.norm_flags, u);
      store_frag<DH>(op, u, g, q < S, wide);
      if (q0 + f * 16 < S) head_lora_frag<DH>(a.hl[1], h, a.T, q0 + f * 16, (int64_t)b * S + qc, q < S, u, g, li);   // v_q = d(pre-norm q) (s B_q)^T
    } else {
      u32x2 u[DF];
#pragma unroll
      for (int d = 0; d < DF; ++d) {
        u[d][0] = pack2bf(dq[d][f][0] * a.scale, dq[d][f][1] * a.scale);
        u[d][1] = pack2bf(dq[d][f][2] * a.scale, dq[d][f][3] * a.scale);
      }
      store_frag<DH>(op, u, g, q < S, wide);
    }
  }
}

// =============================================================================================
// staging helpers for NW-wave blocks (64-row / 64-column tiles)

__device__ __forceinline__ bf16x8 cat8(const bf16x4& a, const bf16x4& b) {
  bf16x8 r;
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
  r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
  return r;
}

// =============================================================================================
// Transposed fragment of a ROW-major [64][DH] tile through the hardware transpose read (ds_read_b64_tr_b16):
// lane (g, li) receives X[q][16*df + li] for the 8 rows q = {32t+4g+j, 32t+16+4g+j}, j<4 -- the pi order of the packed
// P / dS registers.  Source lane 16g + 4j' + m supplies the address of X[q0+j'][16*df+4m .. +3]; the instruction hands
// lane (g, li) element li&3 of source lane 4j + (li>>2), j = 0..3 (mapping verified by qfx_debug_tr_read).
template <int DH>
__device__ __forceinline__ bf16x8 read_trfrag(const char* tile, int df, int t, int g, int li) {
  const int col = 16 * df + 4 * (li & 3);
  const int chunk = col >> 3, off = (col & 7) * 2;
  const int r1 = 32 * t + 4 * g + (li >> 2), r2 = r1 + 16;
  const bf16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((QFX_AS3 bf16x4v*)(tile + r1 * (DH * 2) + ((chunk ^ swz_row<DH>(r1)) << 4) + off));
  const bf16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((QFX_AS3 bf16x4v*)(tile + r2 * (DH * 2) + ((chunk ^ swz_row<DH>(r2)) << 4) + off));
  return cat8(__builtin_bit_cast(bf16x4, lo), __builtin_bit_cast(bf16x4, hi));
}

// =============================================================================================
// dK, dV: block = 256 keys (8 waves x 32), loop over 64-query tiles of Q and dO (row-major, DOUBLE-BUFFERED by LDS-DMA:
// tile it+1 streams in under the MFMAs of tile it, one barrier per tile).  The contractions over queries
// (dV = P^T dO, dK = dS^T Q) read their A operands from the same row-major tiles with the transpose read, so no
// transposed copies of Q / dO exist in HBM or LDS.  lse2 / dsum of the tile ride along by 4-byte LDS-DMA.
// K fragments live in registers, V fragments are re-read from a resident LDS copy of the block's V rows.
template <int DH>
__global__ __launch_bounds__(512, 1) void attn_bwd_dkv_kernel(const qfx_attn_args a) {
  constexpr int KC = DH / 32, DF = DH / 16, NW = 8;
  constexpr int TB = 64 * DH * 2;
  __shared__ __attribute__((aligned(16))) char smem[8 * TB + 1024];   // V(4 tiles of 64 keys) | 2 x [Q | dO] | 2 x [lse2 | dsum]
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  int xb, h, b;
  attn_block_coord((a.S + 255) / 256, a.H, xb, h, b);
  const int kb = xb * 256;
  const int key0 = kb + w * 32;
  const int S = a.S;
  const bf16_t* Qb = a.Q + (int64_t)b * S * a.ldq + h * DH;
  const bf16_t* dOb = a.dO + (int64_t)b * S * a.lddo + h * DH;
  const bf16_t* Vb = a.V + (int64_t)b * S * a.ldv + h * DH;
  const float* lseb = a.lse2 + ((int64_t)b * a.H + h) * a.S_pad;
  const float* dsb = a.dsum + ((int64_t)b * a.H + h) * a.S_pad;
  char* sV = smem;
  char* sStage = smem + 4 * TB;
  char* sStat = smem + 8 * TB;
  const char* myV = sV + (w >> 1) * TB;    // 64-key tile holding this wave's 32 keys (fragments (w&1)*2 + {0,1})

  // DMA sources as uniform base + 32-bit lane offset (rows past S clamp to S-1); nothing 64-bit stays live per lane
  constexpr int CPR = DH / 8, RPI = 64 / CPR, RPW = 64 / NW, NIS = RPW / RPI;
  auto stage = [&](int it, int buf) {
    const int i0 = it * 64;
    char* dQ_ = sStage + buf * 2 * TB;
    // the lane-constant parts of the DMA source offsets are RE-DERIVED from the lane id at every call (the empty asm hides the
    // value's origin from loop-invariant code motion): hoisted, they get spilled -- the kernel sits at the 256-VGPR limit -- and a
    // scratch reload costs an s_waitcnt vmcnt(0) at the top of every iteration (155-158 us against 165-168 at S = 2432)
    int ln = lane;
    asm volatile("" : "+v"(ln));
#pragma unroll
    for (int i = 0; i < NIS; ++i) {
      const int row = w * RPW + i * RPI + ln / CPR;
      const int sc8 = ((ln % CPR) ^ swz_row<DH>(row)) * 8;
      int sr = i0 + row; sr = sr < S ? sr : S - 1;
      glds16(Qb + (row_off(sr, a.ldq) + (unsigned)sc8), dQ_ + (w * RPW + i * RPI) * (DH * 2));
      glds16(dOb + (row_off(sr, a.lddo) + (unsigned)sc8), dQ_ + TB + (w * RPW + i * RPI) * (DH * 2));
    }
    if (w == 0) glds4(lseb + i0 + ln, sStat + buf * 512);
    if (w == 1) glds4(dsb + i0 + ln, sStat + buf * 512 + 256);
  };
#pragma unroll
  for (int i = 0; i < 4; ++i) stage_rows_n<DH, NW>(sV + i * TB, Vb, a.ldv, kb + 64 * i, S, w, lane);
  stage(0, 0);

  bf16x8 kf[2][KC];
  bool keyok[2];
  int mykey[2];
  float mk[2];
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    int k = key0 + f * 16 + li;
    keyok[f] = k < S;
    k = keyok[f] ? k : S - 1;
    mykey[f] = k;
    const bf16_t* kp = a.K + ((int64_t)b * S + k) * a.ldk + h * DH + 8 * g;
#pragma unroll
    for (int kk = 0; kk < KC; ++kk) kf[f][kk] = *(const bf16x8*)(kp + kk * 32);
    mk[f] = (a.key_mask && keyok[f]) ? a.key_mask[(int64_t)b * S + k] * LOG2E : 0.f;
  }
  // Round 6: claim the prologue's register loads BEFORE the tile loop.  hipcc does not see the hand-placed s_waitcnt vmcnt(0) at the top
  // of the loop; left pending in its model, the K fragment loads got counted waits (vmcnt(10) ... vmcnt(3)) at their first use INSIDE
  // the loop, every iteration, right behind the 4-5 LDS-DMA pieces of the next tile -- vmcnt(3) then waits for the oldest of those.
#if !defined(QFX_ATTN_NO_CLAIM)
#pragma unroll
  for (int f = 0; f < 2; ++f) {
#pragma unroll
    for (int kk = 0; kk < KC; ++kk) asm volatile("" : "+v"(kf[f][kk]));
    asm volatile("" : "+v"(mk[f]));
  }
#endif
  f32x4 dk[DF][2], dv[DF][2];
#pragma unroll
  for (int d = 0; d < DF; ++d)
#pragma unroll
    for (int f = 0; f < 2; ++f) { dk[d][f] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[d][f] = dk[d][f]; }
  const float c2 = a.scale * LOG2E;

  // lane-constant LDS byte offsets (tile / fragment indices only add immediates): row fragments per k-chunk,
  // transposed fragments per 16-wide d block
  // koff(kk) = koff0 ^ (kk << 6) and toff(d) = toff0 ^ (d << 5): the chunk index enters the address only through an XOR
  // with the row swizzle, so the k-chunk / d-block bits can be XOR-ed in afterwards (one VGPR each instead of 4 + 8).
  const int koff0 = li * (DH * 2) + ((g ^ swz_row<DH>(li)) << 4);
  const int tr1 = 4 * g + (li >> 2);   // + 32 t (+16): multiples of 16 leave the swizzle unchanged
  const int toff0 = tr1 * (DH * 2) + ((((li & 3) >> 1) ^ swz_row<DH>(tr1)) << 4) + (li & 1) * 8;
  auto koff = [&](int kk) { return koff0 ^ (kk << 6); };
  auto toff = [&](int d) { return toff0 ^ (d << 5); };
  const char* vbase = myV + (w & 1) * 2 * (16 * DH * 2);
  auto trfrag = [&](const char* tile, int d, int t) {
    const bf16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((QFX_AS3 bf16x4v*)(tile + toff(d) + t * (32 * DH * 2)));
    const bf16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((QFX_AS3 bf16x4v*)(tile + toff(d) + t * (32 * DH * 2) + 16 * DH * 2));
    return cat8(__builtin_bit_cast(bf16x4, lo), __builtin_bit_cast(bf16x4, hi));
  };

  const int ntiles = (S + 63) / 64;
  for (int it = 0; it < ntiles; ++it) {
    const int i0 = it * 64;
    const int buf = it & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile `it` (and of V) have landed
    __syncthreads();                                    // everyone's have; everyone is done reading tile it-1
    if (it + 1 < ntiles) stage(it + 1, buf ^ 1);
    const char* sQ = sStage + buf * 2 * TB;
    const char* sdO = sQ + TB;
    const char* sL = sStat + buf * 512;
    const bool tail = i0 + 64 > S;   // wave-uniform: only the last tile masks query rows
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      u32x4 pbu[2], dsu[2];   // packed P / dS of this 32-query half per key fragment: directly the MFMA B operands
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2) {
        __builtin_amdgcn_sched_barrier(0);
        const int qfi = 2 * t + q2;
        f32x4 sa[2], da[2];
        sa[0] = (f32x4){0.f, 0.f, 0.f, 0.f}; sa[1] = sa[0]; da[0] = sa[0]; da[1] = sa[0];
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) {
          const bf16x8 qa = *(const bf16x8*)(sQ + koff(kk) + qfi * (16 * DH * 2));
          const bf16x8 oa = *(const bf16x8*)(sdO + koff(kk) + qfi * (16 * DH * 2));
#pragma unroll
          for (int f = 0; f < 2; ++f) {
            sa[f] = MFMA(qa, kf[f][kk], sa[f]);                                             // D[i=q][j=key]
            da[f] = MFMA(oa, *(const bf16x8*)(vbase + koff(kk) + f * (16 * DH * 2)), da[f]);
          }
        }
        const int ql = qfi * 16 + 4 * g;
        const f32x4 l4 = *(const f32x4*)(sL + ql * 4);
        const f32x4 s4 = *(const f32x4*)(sL + 256 + ql * 4);
        if (tail) {
          const int qb = i0 + ql;
#pragma unroll
          for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const bool ok = (qb + r) < S;
              const float p = ok ? fexp2(sa[f][r] * c2 + mk[f] - l4[r]) : 0.f;
              da[f][r] = ok ? p * (da[f][r] - s4[r]) : 0.f;
              sa[f][r] = p;
            }
        } else if (a.key_mask == nullptr) {     // block-uniform common case: the key mask term is zero, -lse rides in the FMA (one VALU op per score less)
#pragma unroll
          for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float p = fexp2(fmaf(sa[f][r], c2, -l4[r]));
              da[f][r] = p * (da[f][r] - s4[r]);
              sa[f][r] = p;
            }
        } else {
#pragma unroll
          for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float p = fexp2(sa[f][r] * c2 + (mk[f] - l4[r]));
              da[f][r] = p * (da[f][r] - s4[r]);
              sa[f][r] = p;
            }
        }
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          pbu[f][2 * q2] = pack2bf(sa[f][0], sa[f][1]); pbu[f][2 * q2 + 1] = pack2bf(sa[f][2], sa[f][3]);
          dsu[f][2 * q2] = pack2bf(da[f][0], da[f][1]); dsu[f][2 * q2 + 1] = pack2bf(da[f][2], da[f][3]);
        }
      }
      const bf16x8 pb0 = __builtin_bit_cast(bf16x8, pbu[0]), pb1 = __builtin_bit_cast(bf16x8, pbu[1]);
      const bf16x8 ds0 = __builtin_bit_cast(bf16x8, dsu[0]), ds1 = __builtin_bit_cast(bf16x8, dsu[1]);
#pragma unroll
      for (int d = 0; d < DF; ++d) {
        const bf16x8 ot = trfrag(sdO, d, t);
        const bf16x8 qt = trfrag(sQ, d, t);
        dv[d][0] = MFMA(ot, pb0, dv[d][0]);     // D[i=dv][j=key]
        dv[d][1] = MFMA(ot, pb1, dv[d][1]);
        dk[d][0] = MFMA(qt, ds0, dk[d][0]);
        dk[d][1] = MFMA(qt, ds1, dk[d][1]);
      }
    }
  }
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    bf16_t* kp = a.dK + ((int64_t)b * S + mykey[f]) * a.lddk + h * DH;
    bf16_t* vp = a.dV + ((int64_t)b * S + mykey[f]) * a.lddv + h * DH;
    const bool wide_k = rows_16b(a.dK, a.lddk), wide_v = rows_16b(a.dV, a.lddv);
    if (a.qk_saved) {       // block-uniform: d(pre-norm k) straight from the accumulators (rows past S ride along on row S-1, unstored)
      u32x2 u[DF];
      norm_rope_bwd_row<DH>(dk, f, a.scale, a.qk_saved + ((int64_t)b * S + mykey[f]) * a.ld_saved + a.H * DH + h * DH + 4 * g,
                            a.rope + (int64_t)b * a.rope_bstride + ((int64_t)mykey[f] * (DH / 2) + 2 * g) * 2,
                            (mykey[f] < a.T ? a.wk_txt : a.wk_img) + 4 * g, a.norm_eps, a.norm_flags, u);
      store_frag<DH>(kp, u, g, keyok[f], wide_k);
      if (key0 + f * 16 < S) {      // fragment-uniform: v_k = d(pre-norm k) (s B_k)^T, v_v = dV (s B_v)^T (ABI 6)
        head_lora_frag<DH>(a.hl[2], h, a.T, key0 + f * 16, (int64_t)b * S + mykey[f], keyok[f], u, g, li);
        if (a.hl[3].part != nullptr) {
#pragma unroll
          for (int d = 0; d < DF; ++d) { u[d][0] = pack2bf(dv[d][f][0], dv[d][f][1]); u[d][1] = pack2bf(dv[d][f][2], dv[d][f][3]); }
          head_lora_frag<DH>(a.hl[3], h, a.T, key0 + f * 16, (int64_t)b * S + mykey[f], keyok[f], u, g, li);
        }
      }
    }
    u32x2 u[DF];
    if (!a.qk_saved) {
#pragma unroll
      for (int d = 0; d < DF; ++d) {
        u[d][0] = pack2bf(dk[d][f][0] * a.scale, dk[d][f][1] * a.scale);
        u[d][1] = pack2bf(dk[d][f][2] * a.scale, dk[d][f][3] * a.scale);
      }
      store_frag<DH>(kp, u, g, keyok[f], wide_k);
    }
#pragma unroll
    for (int d = 0; d < DF; ++d) { u[d][0] = pack2bf(dv[d][f][0], dv[d][f][1]); u[d][1] = pack2bf(dv[d][f][2], dv[d][f][3]); }
    store_frag<DH>(vp, u, g, keyok[f], wide_v);
  }
}

// Waves per block of the forward kernel.  Two independent 4-wave blocks per CU hide each other's barriers and are faster per
// unit of work (S = 8576: 837 vs 776 TF/s), so the 8-wave / one-block-per-CU form is chosen only where it fills the last round of
// the 256 CUs clearly better (S = 2432: 240 blocks = 94 % vs 456 blocks on 512 half-CU slots = 89 %: 669 -> 710 TF/s).  The dQ
// kernel measured slower with 8 waves in both cases and always uses 4.
// Kernel-selection policy: read from the environment ONCE (first launch), changed through qfx_attn_tune (getenv per launch
// is neither cheap nor safe against a host thread that edits the environment while another one launches).
struct AttnPolicy {
  std::atomic<int> fwd64{-1};       // -1 by shape, 0 = 32-query kernels, 1 = 64-query kernel, 2 = its pipelined form
  std::atomic<int> dq64{-1};        // -1 by shape, 0 / 1
  std::atomic<int> fwd_waves{0};    // 0 by shape, 4 / 8
};
AttnPolicy g_pol;
std::once_flag g_pol_once;
int parse_fwd64(const char* v) { return !strcmp(v, "auto") ? -1 : !strcmp(v, "0") ? 0 : !strcmp(v, "1") ? 1 : !strcmp(v, "1p") ? 2 : -2; }
int parse_dq64(const char* v) { return !strcmp(v, "auto") ? -1 : !strcmp(v, "0") ? 0 : !strcmp(v, "1") ? 1 : -2; }
void policy_from_env() {
  std::call_once(g_pol_once, [] {
    if (const char* e = getenv("QFX_ATTN_FWD64")) { const int v = parse_fwd64(e); if (v != -2) g_pol.fwd64 = v; }
    if (const char* e = getenv("QFX_ATTN_DQ64")) { const int v = parse_dq64(e); if (v != -2) g_pol.dq64 = v; }
    if (const char* e = getenv("QFX_ATTN_FWD_WAVES")) { const int v = atoi(e); if (v == 4 || v == 8) g_pol.fwd_waves = v; }
  });
}

// Waves per block of the forward kernel.  Two independent 4-wave blocks per CU hide each other's barriers and are faster per
// unit of work (S = 8576: 837 vs 776 TF/s), so the 8-wave / one-block-per-CU form is chosen only where it fills the last round of
// the 256 CUs clearly better (S = 2432: 240 blocks = 94 % vs 456 blocks on 512 half-CU slots = 89 %: 669 -> 710 TF/s).  The dQ
// kernel measured slower with 8 waves in both cases and always uses 4.
int pick_waves(const qfx_attn_args* a) {
  const int forced = g_pol.fwd_waves.load();   // A/B lever: 4 or 8
  if (forced == 4 || forced == 8) return forced;
  const long hb = (long)a->H * a->B;
  const long b4 = (long)((a->S + 127) / 128) * hb, b8 = (long)((a->S + 255) / 256) * hb;
  constexpr long NCU = QFX_NUM_CU_TOTAL;
  const double e4 = (double)b4 / (double)(((b4 + 2 * NCU - 1) / (2 * NCU)) * 2 * NCU), e8 = (double)b8 / (double)(((b8 + NCU - 1) / NCU) * NCU);
  return e8 > 1.04 * e4 ? 8 : 4;
}

// The 64-query forward runs one 256-query block per CU: it wins where those blocks fill their last round of CUs (S = 2432: 240 blocks,
// 83 vs 91 us; S = 4608 / 4864: 0.84 / 0.89 of two rounds, 258 vs 278 / 276 vs 315 us) and loses where they do not (S = 3584: 336 blocks =
// 0.66 of two rounds, 208 vs 193 us; S = 1280: 120 blocks, 44 vs 34 us; S = 8576: 0.80 of four rounds, 931 vs 874 us) --
// profiles/r05_attn_fwd64.json.
// Whole-round test shared by the 64-query kernels: their 256-query blocks run one per CU.
bool fills_rounds64(const qfx_attn_args* a) {
  constexpr long NCU = QFX_NUM_CU_TOTAL;
  const long hb = (long)a->H * a->B;
  const long b64 = (long)((a->S + 255) / 256) * hb, b4 = (long)((a->S + 127) / 128) * hb;
  const double e64 = (double)b64 / (double)(((b64 + NCU - 1) / NCU) * NCU);
  const double e4 = (double)b4 / (double)(((b4 + 2 * NCU - 1) / (2 * NCU)) * 2 * NCU);
  const double eo = e64 > e4 ? e64 : e4;      // the 32-query kernels: 8 waves x 256 queries quantise like the 64-query blocks, 4 x 128 like b4
  return e64 > 0.82 && e64 >= eo - 0.02;
}
int pick_fwd64(const qfx_attn_args* a) {      // 0 = 32-query kernels, 1 = 64-query, 2 = pipelined 64-query
  const int f = g_pol.fwd64.load();
  if (f >= 0) return f;
  return fills_rounds64(a) ? 1 : 0;
}
// dQ on 64-query waves (qfx_attn64.hip): 118 vs 136 us at S = 2432, 1312 vs 1326 us at S = 8576 (profiles/r05_attn_dq64.json); same policy
bool pick_dq64(const qfx_attn_args* a) {
  const int f = g_pol.dq64.load();
  return f >= 0 ? f == 1 : fills_rounds64(a);
}

int check_common(const qfx_attn_args* a) {
  if (!a || a->B <= 0 || a->S <= 0 || a->H <= 0 || (a->S_pad % 64) || a->S_pad < a->S) return QFX_EINVAL;
  if (a->dh != 64 && a->dh != 128) return QFX_EUNSUPPORTED;
  // the tile staging addresses rows as 32-bit element offsets from a per-(batch, head) base, built with 24-bit multiplies
  const int64_t lds[] = {a->ldq, a->ldk, a->ldv, a->lddo};
  for (int64_t ld : lds)
    if (ld < 0 || ld >= (1 << 24) || a->S >= (1 << 24) || (int64_t)a->S * ld >= (1LL << 32)) return QFX_EUNSUPPORTED;
  return QFX_OK;
}

// ABI 6: fused rank-r down projections of the slots [first, last] (see qfx_head_lora)
int check_head_lora(const qfx_attn_args* a, int first, int last) {
  for (int i = first; i <= last; ++i) {
    const qfx_head_lora& hl = a->hl[i];
    if (!hl.part) continue;
    if (hl.R != 16 && hl.R != 32) return QFX_EUNSUPPORTED;
    if ((a->T % 16) || a->T < 0 || (hl.ld_part % 4) || (hl.c0 % 4) || hl.c0 < 0 || hl.c0 + hl.R > hl.ld_part ||
        ((uintptr_t)hl.part % 16) || hl.part_hstride < (int64_t)a->B * a->S * hl.ld_part)
      return QFX_EINVAL;
    if (i > 0 && !a->qk_saved) return QFX_EINVAL;        // the backward slots project d(PRE-norm q / k)
    for (int s = 0; s < 2; ++s)
      if ((uintptr_t)hl.w_pk[s] % 16) return QFX_EINVAL;
  }
  return QFX_OK;
}

}  // namespace

extern "C" int qfx_attn_fwd(const qfx_attn_args* a, void* stream) {
  int rc = check_common(a);
  if (rc) return rc;
  if (!a->Q || !a->K || !a->V || !a->O || !a->lse2 || (a->ldq % 8) || (a->ldk % 8) || (a->ldv % 8) || (a->ldo % 4)) return QFX_EINVAL;
  if ((rc = check_head_lora(a, 0, 0))) return rc;
  // dh = 128: 64-query waves, one per SIMD, on the 32x32x16 MFMA with hand-allocated accumulator registers (qfx_attn64.hip, round 5)
  // where its 256-query blocks fill whole rounds of the 256 CUs (pick_fwd64; forced either way through qfx_attn_tune / QFX_ATTN_FWD64).
  // Two 64-query forms exist: query blocks skewed by half a tile (default; 84.5 us at S = 2432) and a continuous pipeline over 32-key
  // sub-tiles ("1p"; 88.0 us: a lone wave issues one instruction per ~6.4 cycles whatever their placement, and the pipeline needs ~30
  // more of them per tile -- profiles/r05_attn_fwd64.json)
  policy_from_env();
  if (a->dh == 128) {
    const int f64 = pick_fwd64(a);
    if (f64 == 2) return qfxi::launch_attn_fwd64p(a, (hipStream_t)stream);
    if (f64 == 1) return qfxi::launch_attn_fwd64(a, (hipStream_t)stream);
  }
  const int nw = pick_waves(a);
  dim3 grid(((a->S + 32 * nw - 1) / (32 * nw)) * a->H * a->B);
  if (a->dh == 128) {
    if (nw == 8) hipLaunchKernelGGL((attn_fwd_kernel<128, 8>), grid, dim3(512), 0, (hipStream_t)stream, *a);
    else hipLaunchKernelGGL((attn_fwd_kernel<128, 4>), grid, dim3(256), 0, (hipStream_t)stream, *a);
  } else {
    if (nw == 8) hipLaunchKernelGGL((attn_fwd_kernel<64, 8>), grid, dim3(512), 0, (hipStream_t)stream, *a);
    else hipLaunchKernelGGL((attn_fwd_kernel<64, 4>), grid, dim3(256), 0, (hipStream_t)stream, *a);
  }
  QFX_CHECK_LAUNCH();
  return QFX_OK;
}

extern "C" int qfx_attn_bwd_prep(const qfx_attn_args* a, void* stream) {
  int rc = check_common(a);
  if (rc) return rc;
  if (!a->O || !a->dO || !a->dsum || (a->ldo % 8) || (a->lddo % 8)) return QFX_EINVAL;
  const int64_t n = (int64_t)a->B * a->S * a->H;
  const int ipb = 256 / (a->dh / 8);
  dim3 grid((unsigned)((n + ipb - 1) / ipb));
  if (a->dh == 128) hipLaunchKernelGGL(attn_prep_kernel<128>, grid, dim3(256), 0, (hipStream_t)stream, *a);
  else hipLaunchKernelGGL(attn_prep_kernel<64>, grid, dim3(256), 0, (hipStream_t)stream, *a);
  QFX_CHECK_LAUNCH();
  return QFX_OK;
}

extern "C" int qfx_attn_bwd_dq(const qfx_attn_args* a, void* stream) {
  int rc = check_common(a);
  if (rc) return rc;
  if (a->qk_saved && (!a->rope || !a->wq_txt || !a->wq_img || (a->ld_saved % 4) || a->T < 0)) return QFX_EINVAL;
  if (!a->Q || !a->K || !a->V || !a->O || !a->dO || !a->lse2 || !a->dsum || !a->dQ) return QFX_EINVAL;
  if ((a->ldq % 8) || (a->ldk % 8) || (a->ldv % 8) || (a->ldo % 8) || (a->lddo % 8) || (a->lddq % 4)) return QFX_EINVAL;
  if ((rc = check_head_lora(a, 1, 1))) return rc;
  policy_from_env();
  // dh = 128: the 64-query kernel of qfx_attn64.hip where its blocks fill whole rounds (forced either way: qfx_attn_tune / QFX_ATTN_DQ64)
  if (a->dh == 128 && pick_dq64(a)) return qfxi::launch_attn_bwd_dq64(a, (hipStream_t)stream);
  const int nw = 4;   /* see pick_waves */
  dim3 grid(((a->S + 32 * nw - 1) / (32 * nw)) * a->H * a->B);
  if (a->dh == 128) {
    if (nw == 8) hipLaunchKernelGGL((attn_bwd_dq_kernel<128, 8>), grid, dim3(512), 0, (hipStream_t)stream, *a);
    else hipLaunchKernelGGL((attn_bwd_dq_kernel<128, 4>), grid, dim3(256), 0, (hipStream_t)stream, *a);
  } else {
    if (nw == 8) hipLaunchKernelGGL((attn_bwd_dq_kernel<64, 8>), grid, dim3(512), 0, (hipStream_t)stream, *a);
    else hipLaunchKernelGGL((attn_bwd_dq_kernel<64, 4>), grid, dim3(256), 0, (hipStream_t)stream, *a);
  }
  QFX_CHECK_LAUNCH();
  return QFX_OK;
}

extern "C" int qfx_attn_bwd_dkv(const qfx_attn_args* a, void* stream) {
  int rc = check_common(a);
  if (rc) return rc;
  if (a->qk_saved && (!a->rope || !a->wk_txt || !a->wk_img || (a->ld_saved % 4) || a->T < 0)) return QFX_EINVAL;
  if (!a->Q || !a->K || !a->V || !a->dO || !a->lse2 || !a->dsum || !a->dK || !a->dV) return QFX_EINVAL;
  if ((a->ldq % 8) || (a->ldk % 8) || (a->ldv % 8) || (a->lddo % 8) || (a->lddk % 4) || (a->lddv % 4)) return QFX_EINVAL;
  if ((rc = check_head_lora(a, 2, 3))) return rc;
  dim3 grid(((a->S + 255) / 256) * a->H * a->B);
  if (a->dh == 128) hipLaunchKernelGGL(attn_bwd_dkv_kernel<128>, grid, dim3(512), 0, (hipStream_t)stream, *a);
  else hipLaunchKernelGGL(attn_bwd_dkv_kernel<64>, grid, dim3(512), 0, (hipStream_t)stream, *a);
  QFX_CHECK_LAUNCH();
  return QFX_OK;
}

extern "C" int qfx_attn_bwd_fused_workspace(const qfx_attn_args* a, int64_t* acc_bytes, int64_t* turn_bytes) {
  if (acc_bytes) *acc_bytes = 0;
  if (turn_bytes) *turn_bytes = 0;
  int rc = check_common(a);
  if (rc) return rc;
  if (qfxi::attn_bwd1_grid(a, nullptr) <= 0) return QFX_EUNSUPPORTED;
  const int64_t tiles = (int64_t)a->B * a->H * ((a->S + 63) / 64);
  if (acc_bytes) *acc_bytes = tiles * 64 * 128 * 4;
  if (turn_bytes) *turn_bytes = tiles * 4;
  return QFX_OK;
}

extern "C" int qfx_attn_bwd_fused(const qfx_attn_args* a, void* stream) {
  int rc = check_common(a);
  if (rc) return rc;
  if (a->dh != 128) return QFX_EUNSUPPORTED;
  if (a->qk_saved && (!a->rope || !a->wq_txt || !a->wq_img || !a->wk_txt || !a->wk_img || (a->ld_saved % 4) || a->T < 0)) return QFX_EINVAL;
  if (!a->Q || !a->K || !a->V || !a->O || !a->dO || !a->lse2 || !a->dsum || !a->dQ || !a->dK || !a->dV || !a->dq_acc || !a->dq_turn) return QFX_EINVAL;
  if ((a->ldq % 8) || (a->ldk % 8) || (a->ldv % 8) || (a->ldo % 8) || (a->lddo % 8) || (a->lddq % 4) || (a->lddk % 4) || (a->lddv % 4)) return QFX_EINVAL;
  if (((uintptr_t)a->dq_acc % 16) || ((uintptr_t)a->dq_turn % 4)) return QFX_EINVAL;
  if ((rc = check_head_lora(a, 1, 3))) return rc;
  if ((rc = qfx_attn_bwd_prep(a, stream))) return rc;      // dsum = rowsum(dO * O): every key block needs it for every query tile
  return qfxi::launch_attn_bwd1(a, (hipStream_t)stream);
}

extern "C" int qfx_attn_tune(const char* spec) {
  policy_from_env();
  if (!spec || !*spec) return QFX_OK;
  int f64 = g_pol.fwd64.load(), d64 = g_pol.dq64.load(), fw = g_pol.fwd_waves.load();
  char buf[256];
  strncpy(buf, spec, sizeof(buf) - 1);
  buf[sizeof(buf) - 1] = 0;
  for (char* tok = strtok(buf, ","); tok; tok = strtok(nullptr, ",")) {
    char* eq = strchr(tok, '=');
    if (!eq) return QFX_EINVAL;
    *eq = 0;
    const char* v = eq + 1;
    if (!strcmp(tok, "fwd64")) { f64 = parse_fwd64(v); if (f64 == -2) return QFX_EINVAL; }
    else if (!strcmp(tok, "dq64")) { d64 = parse_dq64(v); if (d64 == -2) return QFX_EINVAL; }
    else if (!strcmp(tok, "fwd_waves")) { fw = atoi(v); if (fw != 0 && fw != 4 && fw != 8) return QFX_EINVAL; }
    else return QFX_EINVAL;
  }
  g_pol.fwd64 = f64; g_pol.dq64 = d64; g_pol.fwd_waves = fw;      // parsed completely before anything changes
  return QFX_OK;
}
