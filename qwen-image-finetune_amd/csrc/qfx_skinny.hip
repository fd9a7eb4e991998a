// qfx_skinny.hip -- the HBM-bound rank-r LoRA pieces: down projection (MFMA, K split over the
// four waves of a block), weight-gradient outer products (VALU + fp32 atomics) and operand packing.
#include "qfx_common.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#ifndef QFX_GRAD_HANDOFF
#define QFX_GRAD_HANDOFF 8      // 8 = chunk partials added by a second launch (product); 0 = by the last block of the strip, with fences: see lora_grad_kernel
#endif
#ifndef QFX_GRAD_NT_X
#define QFX_GRAD_NT_X 0      // lora_grad: non-temporal loads of the token-side operand (A/B lever)
#endif
namespace {

// ---------------------------------------------------------------------------------------------
// U[M,R] = X[M,K] (W_hi+W_lo)[R,K]^T, fp32 accumulate.  Block = 16 rows of X x 8 waves; the waves
// interleave the K/32 MFMA steps (adjacent waves read adjacent 64 B of each row), two steps in flight
// per wave, LDS reduce at the end.  Outputs: U fp32, the packed bf16 K-extension image for the GEMM,
// and the transposed hi/lo image Ut[R][M] that the weight-gradient kernel contracts over tokens.
// Batched launches: up to QFX_MAX_BATCH independent problems of the same rank share one grid (these kernels are one
// memory round trip long, so every separate launch costs ~2 us of dispatch gap plus its own latency floor).  The
// argument block is read from the kernarg segment (scalar loads); indexing the by-value copy would go through scratch.
#define QFX_AS4 __attribute__((address_space(4)))
struct DownBatch { qfx_lora_down_args a[QFX_MAX_BATCH]; int start[QFX_MAX_BATCH + 1]; int n; };
struct GradBatch { qfx_lora_grad_args a[QFX_MAX_BATCH]; int start[QFX_MAX_BATCH + 1]; int n; };

template <int NF, int RB>   // RB = 16-row groups per block: every block streams ALL of W (hi+lo) from L2, so two groups per block
                            // halve that traffic (for R=48 it was 6x the X bytes and the actual bound of the kernel)
__global__ __launch_bounds__(512) void lora_down_kernel(const DownBatch batch_by_value) {
  constexpr int NW = 8;
  __shared__ float red[NW][RB * NF * 256];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const QFX_AS4 DownBatch& kb = *(const QFX_AS4 DownBatch*)__builtin_amdgcn_kernarg_segment_ptr();
  int pi = 0;
#pragma unroll
  for (int i = 1; i < QFX_MAX_BATCH; ++i)
    if (i < kb.n && (int)blockIdx.x >= kb.start[i]) pi = i;
  const QFX_AS4 qfx_lora_down_args& p = kb.a[pi];
  const int m0 = ((int)blockIdx.x - kb.start[pi]) * (16 * RB);

  const bf16_t* xrow[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    int mr = m0 + rb * 16 + li;
    mr = mr < p.M ? mr : p.M - 1;
    xrow[rb] = p.X + remap_row(mr, p.rows_per_batch, p.x_batch_rows, p.x_row_off) * p.ldx + 8 * g;
  }
  const bf16_t* wh[NF];
  const bf16_t* wl[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    wh[nf] = p.W_hi + (int64_t)(nf * 16 + li) * p.ldw + 8 * g;
    wl[nf] = p.W_lo + (int64_t)(nf * 16 + li) * p.ldw + 8 * g;
  }
  f32x4 acc[RB][NF];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int j = 0; j < NF; ++j) acc[rb][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // The kernel is a latency chain, not a bandwidth stream (~12 k-steps per wave): ALL X fragments of a chunk are requested up
  // front (one HBM latency per chunk); the weight fragments are L2 hits and are double-buffered one step ahead of the MFMAs.
  const int nks = p.K / 32;
  constexpr int CHK = RB == 1 ? 12 : 6;
  for (int base = w; base < nks; base += NW * CHK) {
    bf16x8 xs[CHK][RB];
#pragma unroll
    for (int i = 0; i < CHK; ++i) {
      const int ks = base + i * NW;
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) xs[i][rb] = ks < nks ? *(const bf16x8*)(xrow[rb] + ks * 32) : (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
    }
    if (p.xq != nullptr) {   // MX-FP8 image of X (block-uniform): the 32 columns of a k-step are one MX block held by the 4 lane groups
#pragma unroll
      for (int i = 0; i < CHK; ++i) {
        const int ks = base + i * NW;
        if (ks < nks) {      // wave-uniform
#pragma unroll
          for (int rb = 0; rb < RB; ++rb) {
            float v[8];
            float amax = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[j] = bf2f((bf16_t)xs[i][rb][j]); amax = fmaxf(amax, fabsf(v[j])); }
            amax = fmaxf(amax, __shfl_xor(amax, 16));
            amax = fmaxf(amax, __shfl_xor(amax, 32));
            int eb;
            const u32x2 qv = mx_quant8(v, amax, eb);
            const int m = m0 + rb * 16 + li;
            if (m < p.M) {
              *(u32x2*)(p.xq + (int64_t)m * p.ldxq + ks * 32 + 8 * g) = qv;
              const int kb = p.xq_kb0 + ks;
              if (g == 0) p.xs[((int64_t)(kb >> 2) * p.xs_rows + m) * 4 + (kb & 3)] = (uint8_t)eb;
            }
          }
        }
      }
    }
    // weight fragments: L2 hits streamed through a register ring DEPTH k-steps deep (all of the chunk for one fragment column:
    // with a single step of look-ahead the CHK dependent L2 round trips were most of the launch)
    constexpr int DEPTH = (NF == 1 ? CHK : (NF == 2 ? 6 : 3)) < CHK ? (NF == 1 ? CHK : (NF == 2 ? 6 : 3)) : CHK;
    bf16x8 rh[DEPTH][NF], rl[DEPTH][NF];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int ks = base + d * NW;
      if (ks < nks) {
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) { rh[d][nf] = *(const bf16x8*)(wh[nf] + ks * 32); rl[d][nf] = *(const bf16x8*)(wl[nf] + ks * 32); }
      }
    }
#pragma unroll
    for (int i = 0; i < CHK; ++i) {
      if (base + i * NW < nks) {
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
#pragma unroll
          for (int rb = 0; rb < RB; ++rb) {
            acc[rb][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xs[i][rb], rh[i % DEPTH][nf], acc[rb][nf], 0, 0, 0);
            acc[rb][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xs[i][rb], rl[i % DEPTH][nf], acc[rb][nf], 0, 0, 0);
          }
      }
      const int ksn = base + (i + DEPTH) * NW;
      if (i + DEPTH < CHK && ksn < nks) {
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) { rh[i % DEPTH][nf] = *(const bf16x8*)(wh[nf] + ksn * 32); rl[i % DEPTH][nf] = *(const bf16x8*)(wl[nf] + ksn * 32); }
      }
    }
  }
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[w][((rb * NF + nf) * 64 + lane) * 4 + r] = acc[rb][nf][r];
  __syncthreads();
  for (int e = tid; e < RB * NF * 256; e += 512) {
    float v = 0.f;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) v += red[ww][e];
    const int r = e & 3, ln = (e >> 2) & 63, fi = e >> 8;
    const int rb = fi / NF, nf = fi % NF;
    const int m = m0 + rb * 16 + 4 * (ln >> 4) + r;   // D[i = 4g+r][j = lane&15]
    const int j = nf * 16 + (ln & 15);
    if (m >= p.M) continue;
    if (p.U) p.U[(int64_t)m * p.ldu + j] = v;
    const bf16_t hi = f2bf(v);
    const bf16_t lo = f2bf(v - bf2f(hi));
    if (p.ext) {
      bf16_t* e0 = p.ext + (int64_t)m * p.ld_ext + (j / p.group_R) * p.group_stride + (j % p.group_R);
      e0[0] = hi;
      e0[p.group_R] = lo;
      e0[2 * p.group_R] = hi;
    }
    if (p.Ut_hi) {
      p.Ut_hi[(int64_t)j * p.ld_ut + m] = hi;
      p.Ut_lo[(int64_t)j * p.ld_ut + m] = lo;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Fused LayerNorm+modulate forward + LoRA down projection.  Block = 16 rows x 8 waves; wave w owns the 32-column k-steps
// w, w+8, ... of the 16 rows as MFMA A fragments (lane (g, li): row li, columns 32 ks + 8 g .. +8), so the normalised, modulated
// row block never leaves the registers between the LayerNorm and the rank-r contraction: one pass over x instead of
// LayerNorm (read x, write y) + lora_down (read y).  Row statistics: per-lane partial sums -> the 4 lane groups of a row by
// shuffles -> the 8 waves through LDS (two rounds: mean, then centred second moment, as the stand-alone kernel).
struct LnDownBatch { qfx_ln_down_args a[2]; int start[3]; int n; };

template <int NF>
__global__ __launch_bounds__(512) void ln_down_kernel(const LnDownBatch batch_by_value) {
  constexpr int NW = 8, CH = 12;              // CH k-steps per wave: D <= 8 * 12 * 32 = 3072
  __shared__ float red[NW][NF * 256];
  __shared__ float stat[2][NW][16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const QFX_AS4 LnDownBatch& kb = *(const QFX_AS4 LnDownBatch*)__builtin_amdgcn_kernarg_segment_ptr();
  const int pi = (kb.n > 1 && (int)blockIdx.x >= kb.start[1]) ? 1 : 0;
  const QFX_AS4 qfx_ln_down_args& p = kb.a[pi];
  const int m0 = ((int)blockIdx.x - kb.start[pi]) * 16;
  const int D = p.ln.D, rows = p.ln.rows;
  const int nks = D / 32;
  int mr = m0 + li;
  const bool row_ok = mr < rows;
  mr = row_ok ? mr : rows - 1;
  const int b = mr / p.ln.rows_per_batch;
  const bf16_t* xrow = p.ln.x + (int64_t)mr * D + 8 * g;
  const bf16_t* scr = p.ln.scale + (int64_t)b * p.ln.mod_bstride + 8 * g;
  const bf16_t* shr = p.ln.shift + (int64_t)b * p.ln.mod_bstride + 8 * g;

  // ---- the wave's slice of the 16 rows, requested up front (one HBM latency)
  bf16x8 xs[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int ks = w + i * NW;
    xs[i] = ks < nks ? *(const bf16x8*)(xrow + ks * 32) : (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < CH; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) s += bf2f((bf16_t)xs[i][j]);        // absent k-steps contribute zeros
  s += __shfl_xor(s, 16);
  s += __shfl_xor(s, 32);
  if (g == 0) stat[0][w][li] = s;
  __syncthreads();
  float mean = 0.f;
#pragma unroll
  for (int ww = 0; ww < NW; ++ww) mean += stat[0][ww][li];
  mean /= (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    if (w + i * NW < nks) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = bf2f((bf16_t)xs[i][j]) - mean; q += d * d; }
    }
  }
  q += __shfl_xor(q, 16);
  q += __shfl_xor(q, 32);
  if (g == 0) stat[1][w][li] = q;
  __syncthreads();
  float var = 0.f;
#pragma unroll
  for (int ww = 0; ww < NW; ++ww) var += stat[1][ww][li];
  const float rstd = rsqrtf(var / (float)D + p.ln.eps);

  // ---- modulate (rounding points of the stand-alone kernel), store y, keep the bf16 fragments for the MFMAs
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int ks = w + i * NW;
    if (ks < nks) {
      const bf16x8 sc = *(const bf16x8*)(scr + ks * 32);
      const bf16x8 sh = *(const bf16x8*)(shr + ks * 32);
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float ln = rbf((bf2f((bf16_t)xs[i][j]) - mean) * rstd);
        const float t1 = rbf(1.0f + bf2f((bf16_t)sc[j]));
        o[j] = (short)f2bf(rbf(ln * t1) + bf2f((bf16_t)sh[j]));
      }
      xs[i] = o;
      if (row_ok) *(bf16x8*)(p.ln.y + (int64_t)mr * D + 8 * g + ks * 32) = o;
      if (p.ln.yq) {   // MX-FP8 image of y (block-uniform): the 32 columns of this k-step are one MX block held by the 4 lane groups
        float v[8];
        float amax = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[j] = bf2f((bf16_t)o[j]); amax = fmaxf(amax, fabsf(v[j])); }
        amax = fmaxf(amax, __shfl_xor(amax, 16));
        amax = fmaxf(amax, __shfl_xor(amax, 32));
        int eb;
        const u32x2 qv = mx_quant8(v, amax, eb);
        if (row_ok) {
          *(u32x2*)(p.ln.yq + (int64_t)mr * p.ln.ldyq + 8 * g + ks * 32) = qv;
          if (g == 0) p.ln.ys[((int64_t)(ks >> 2) * p.ln.ys_rows + mr) * 4 + (ks & 3)] = (uint8_t)eb;
        }
      }
    }
  }
  if (p.W_hi == nullptr) return;             // plain LayerNorm rows (block-uniform)

  // ---- rank-r contraction of the wave's k-steps (weights are L2 hits, double-buffered one step ahead)
  const bf16_t* wh[NF];
  const bf16_t* wl[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    wh[nf] = p.W_hi + (int64_t)(nf * 16 + li) * p.ldw + 8 * g;
    wl[nf] = p.W_lo + (int64_t)(nf * 16 + li) * p.ldw + 8 * g;
  }
  // fragment image (block-uniform): fragment (ks, nf) is one lane-linear 1 KiB piece; row-major otherwise
  const bf16_t* fr_h = p.W_fr ? p.W_fr + lane * 8 : nullptr;
  const bf16_t* fr_l = p.W_fr ? fr_h + (int64_t)NF * 16 * D : nullptr;
  auto ldw = [&](const bf16_t* rowp, const bf16_t* frp, int ks, int nf) {
    return frp ? *(const bf16x8*)(frp + (ks * NF + nf) * 512) : *(const bf16x8*)(rowp + ks * 32);
  };
  f32x4 acc[NF];
#pragma unroll
  for (int j = 0; j < NF; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // weight fragments stream from L2 through a 3-deep register ring (three k-steps in flight: with one step of look-ahead the
  // twelve dependent L2 round trips were most of the kernel)
  constexpr int DEPTH = 3;
  bf16x8 wr_h[DEPTH][NF], wr_l[DEPTH][NF];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    const int ks = w + d * NW;
    if (ks < nks) {
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) { wr_h[d][nf] = ldw(wh[nf], fr_h, ks, nf); wr_l[d][nf] = ldw(wl[nf], fr_l, ks, nf); }
    }
  }
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    if (w + i * NW < nks) {
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        acc[nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xs[i], wr_h[i % DEPTH][nf], acc[nf], 0, 0, 0);
        acc[nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xs[i], wr_l[i % DEPTH][nf], acc[nf], 0, 0, 0);
      }
    }
    const int ksn = w + (i + DEPTH) * NW;
    if (i + DEPTH < CH && ksn < nks) {
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) { wr_h[i % DEPTH][nf] = ldw(wh[nf], fr_h, ksn, nf); wr_l[i % DEPTH][nf] = ldw(wl[nf], fr_l, ksn, nf); }
    }
  }
#pragma unroll
  for (int nf = 0; nf < NF; ++nf)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[w][(nf * 64 + lane) * 4 + r] = acc[nf][r];
  __syncthreads();
  for (int e = tid; e < NF * 256; e += 512) {
    float v = 0.f;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) v += red[ww][e];
    const int r = e & 3, ln = (e >> 2) & 63, nf = e >> 8;
    const int m = m0 + 4 * (ln >> 4) + r;   // D[i = 4g+r][j = lane&15]
    const int j = nf * 16 + (ln & 15);
    if (m >= rows) continue;
    const bf16_t hi = f2bf(v);
    const bf16_t lo = f2bf(v - bf2f(hi));
    if (p.ext) {
      bf16_t* e0 = p.ext + (int64_t)m * p.ld_ext + (j / p.group_R) * p.group_stride + (j % p.group_R);
      e0[0] = hi;
      e0[p.group_R] = lo;
      e0[2 * p.group_R] = hi;
    }
    if (p.Ut_hi) {
      p.Ut_hi[(int64_t)j * p.ld_ut + m] = hi;
      p.Ut_lo[(int64_t)j * p.ld_ut + m] = lo;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// G[j,k] += scale * sum_m (Vt_hi+Vt_lo)[j,m] X[m,k]  -- LoRA dA / dB.  Contraction over TOKENS: the
// token-major X tile goes through LDS and comes back as MFMA B fragments with ds_read_b64_tr_b16
// (gfx950 hardware transpose read): within a 16-lane group, lane s supplies 8 bytes
// {row 8g+4h+(s>>2), cols 4(s&3)..+3} and receives {rows 8g+4h+0..3, col s}.
// grid = (K/128, token chunks of GRAD_CH); block = 4 waves x 32 columns; fp32 atomics into G.
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4v;
constexpr int GX_ROWB = 288;  // padded LDS row stride (bytes) of the [32 tokens][128 cols] tile

#ifndef QFX_GRAD_CH
#define QFX_GRAD_CH 512
#endif
constexpr int GRAD_CH = QFX_GRAD_CH;   // tokens per block: 4x fewer device-scope fp32 atomics per output element than 128 (the atomics
                               // of the M/CH partial sums, not the 15 MB stream, bounded the 128-token version)
// G[j, k] += scale * D[j, k] for the 16 NF x 128 tile a block holds in MFMA layout (D[i = rank 4g+r][j = col li] per 16 x 16 fragment)
template <int NF>
__device__ __forceinline__ void grad_update(const QFX_AS4 qfx_lora_grad_args& p, const f32x4 (&acc)[NF][2], int k0, int w, int g, int li, bool plain) {
#pragma unroll
  for (int nf = 0; nf < NF; ++nf)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = nf * 16 + 4 * g + r;
      const int grp = j / p.group_R, jj = j % p.group_R;
      if (jj >= p.r_valid) continue;
      float* G = grp == 0 ? p.G : (grp == 1 ? p.G1 : p.G2);
#pragma unroll
      for (int cf = 0; cf < 2; ++cf) {
        const int k = k0 + w * 32 + cf * 16 + li;
        if (k >= p.K) continue;
        float* gp = G + (int64_t)jj * p.g_sr + (int64_t)k * p.g_sc;
        if (plain) *gp += acc[nf][cf][r] * p.out_scale;
        else unsafeAtomicAdd(gp, acc[nf][cf][r] * p.out_scale);
      }
    }
}

template <int NF>
__global__ __launch_bounds__(256) void lora_grad_kernel(const GradBatch batch_by_value) {
  constexpr int CH = GRAD_CH;
  __shared__ __attribute__((aligned(16))) char sX[2][32 * GX_ROWB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const QFX_AS4 GradBatch& kb = *(const QFX_AS4 GradBatch*)__builtin_amdgcn_kernarg_segment_ptr();
  int pi = 0;
#pragma unroll
  for (int i = 1; i < QFX_MAX_BATCH; ++i)
    if (i < kb.n && (int)blockIdx.y >= kb.start[i]) pi = i;
  const QFX_AS4 qfx_lora_grad_args& p = kb.a[pi];
  const int k0 = blockIdx.x * 128;
  if (k0 >= p.K) return;   // grid.x is sized for the widest problem of the batch
  const int mb = ((int)blockIdx.y - kb.start[pi]) * CH;
  const int nsteps = ((p.M - mb < CH ? p.M - mb : CH) + 31) / 32;   // 32 tokens per step
  const int nquads = (nsteps + 3) / 4;

  // staging: thread -> (token row tid/16 + 16*it, 16-byte chunk tid%16)
  const int srow = tid >> 4, sch = tid & 15;
  int kc = k0 + sch * 8;
  kc = kc < p.K ? kc : 0;
  // X rows are requested a whole quad (4 steps = 128 tokens) ahead of their use; the LDS tile is double-buffered.
  u32x4 st[4][2], stn[4][2];
  auto gload = [&](int quad, u32x4 (&dst)[4][2]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        int m = mb + (quad * 4 + j) * 32 + it * 16 + srow;
        m = m < p.M ? m : p.M - 1;
        const bf16_t* src = p.X + remap_row(m, p.rows_per_batch, p.x_batch_rows, p.x_row_off) * p.ldx + kc;
#if QFX_GRAD_NT_X
        dst[j][it] = __builtin_nontemporal_load((const u32x4*)src);      // every X element is read once per launch: keep it out of the GEMMs' L2
#else
        dst[j][it] = *(const u32x4*)src;
#endif
      }
  };
  f32x4 acc[NF][2];
#pragma unroll
  for (int i = 0; i < NF; ++i) { acc[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[i][1] = acc[i][0]; }

  // rank-side fragments (L2 hits) are fetched one step ahead of the MFMAs that use them
  bf16x8 ah[2][NF], al[2][NF];
  auto vload = [&](int step, int slot) {
    const int mtok = mb + step * 32 + 8 * g;   // tokens mtok..mtok+7 are this lane group's k-slots
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      const int64_t ro = (int64_t)(nf * 16 + li) * p.ldvt + mtok;
      ah[slot][nf] = *(const bf16x8*)(p.Vt_hi + ro);
      al[slot][nf] = *(const bf16x8*)(p.Vt_lo + ro);
    }
  };
  gload(0, st);
  vload(0, 0);
  int buf = 0;
  for (int q = 0; q < nquads; ++q) {
    if (q + 1 < nquads) gload(q + 1, stn);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int s = q * 4 + j;
      if (s < nsteps) {   // block-uniform
#pragma unroll
        for (int it = 0; it < 2; ++it) *(u32x4*)(&sX[buf][(it * 16 + srow) * GX_ROWB + sch * 16]) = st[j][it];
        __syncthreads();   // tile s visible; every wave is past its reads of tile s-2 (same buffer)
        if (s + 1 < nsteps) vload(s + 1, (j + 1) & 1);
        const char* tile = sX[buf];
        bf16x8 b[2];
#pragma unroll
        for (int cf = 0; cf < 2; ++cf) {
          const int colb = (w * 32 + cf * 16 + 4 * (li & 3)) * 2;
          const char* a0 = tile + (8 * g + (li >> 2)) * GX_ROWB + colb;
          const bf16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((QFX_AS3 bf16x4v*)(a0));
          const bf16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((QFX_AS3 bf16x4v*)(a0 + 4 * GX_ROWB));
          const bf16x4 l4 = __builtin_bit_cast(bf16x4, lo), h4 = __builtin_bit_cast(bf16x4, hi);
          b[cf][0] = l4[0]; b[cf][1] = l4[1]; b[cf][2] = l4[2]; b[cf][3] = l4[3];
          b[cf][4] = h4[0]; b[cf][5] = h4[1]; b[cf][6] = h4[2]; b[cf][7] = h4[3];
        }
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
#pragma unroll
          for (int cf = 0; cf < 2; ++cf) {
            acc[nf][cf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[j & 1][nf], b[cf], acc[nf][cf], 0, 0, 0);
            acc[nf][cf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[j & 1][nf], b[cf], acc[nf][cf], 0, 0, 0);
          }
        buf ^= 1;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { st[j][0] = stn[j][0]; st[j][1] = stn[j][1]; }
  }
  // D[i = rank 4g+r][j = col li]
  const int nchunk = (p.M + CH - 1) / CH;
  if (p.ws != nullptr && nchunk > 1) {
    // ---- deterministic form (ABI 7): partial tile -> per-problem scratch, added in CHUNK ORDER (the sum does not depend on who
    // finished when).  QFX_GRAD_HANDOFF = 8 (product): by a second launch, lora_grad_reduce_kernel -- the kernel boundary is the
    // hand-off, this block is done.  = 0 (round 6, first version): by the LAST block of the 128-column strip in this launch (plain slab
    // stores, every wave drains them, ONE agent-scope release by lane 0, relaxed ticket, the last arriver acquires once and reads plain
    // -- cdna_hip_programming.md section 5): correct, but every block's release is a buffer_wbl2 of the XCD's WHOLE L2 next to the main
    // stream's GEMMs: +1.4 ms per step (profiles/r06_grad_handoff.json; fence-free forms with sc0 sc1 stores, also with a read-back
    // before the ticket, hand over stale slabs under the 60-block step: the same file).
    const int chunk = (int)blockIdx.y - kb.start[pi];
    const int nstrip = (p.K + 127) / 128;      // the scratch is sized per problem (qfx_lora_grad_ws_floats), not per launch grid
    float* slab = p.ws + ((int64_t)chunk * nstrip + blockIdx.x) * (NF * 2048) + tid * 4;
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int cf = 0; cf < 2; ++cf) *(f32x4*)(slab + (nf * 2 + cf) * 1024) = acc[nf][cf];
    if constexpr ((QFX_GRAD_HANDOFF & 8) != 0) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* flag = (int*)&sX[0][0];
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (ROCm 7.2 may drop the wait behind buffer_wbl2: restated where it cannot)
      const int t = __hip_atomic_fetch_add(p.ws_count + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = t == nchunk - 1;
      if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __hip_atomic_store(p.ws_count + blockIdx.x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // zero again for the next launch
      }
      *flag = last;
    }
    __syncthreads();
    if (*flag == 0) return;
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int cf = 0; cf < 2; ++cf) acc[nf][cf] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < nchunk; ++c) {
      const float* src = p.ws + ((int64_t)c * nstrip + blockIdx.x) * (NF * 2048) + tid * 4;
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int cf = 0; cf < 2; ++cf) acc[nf][cf] += *(const f32x4*)(src + (nf * 2 + cf) * 1024);
    }
  }
  grad_update<NF>(p, acc, k0, w, g, li, p.ws != nullptr);      // (scratch given: one writer per element and launch, no atomic needed -- a single chunk included)
}

// Second launch of the deterministic form (QFX_GRAD_HANDOFF = 8): block (strip, problem) adds the token-chunk partials of the first
// launch in chunk order and updates G.  Same thread -> element map as lora_grad_kernel; the kernel boundary is the hand-off.
template <int NF>
__global__ __launch_bounds__(256) void lora_grad_reduce_kernel(const GradBatch batch_by_value) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const QFX_AS4 GradBatch& kb = *(const QFX_AS4 GradBatch*)__builtin_amdgcn_kernarg_segment_ptr();
  const QFX_AS4 qfx_lora_grad_args& p = kb.a[blockIdx.y];
  const int k0 = blockIdx.x * 128;
  const int nchunk = (p.M + GRAD_CH - 1) / GRAD_CH;
  if (k0 >= p.K || p.ws == nullptr || nchunk < 2) return;
  const int nstrip = (p.K + 127) / 128;
  f32x4 acc[NF][2];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf)
#pragma unroll
    for (int cf = 0; cf < 2; ++cf) acc[nf][cf] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < nchunk; ++c) {      // chunk order: the sum does not depend on who finished when
    const float* src = p.ws + ((int64_t)c * nstrip + blockIdx.x) * (NF * 2048) + tid * 4;
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int cf = 0; cf < 2; ++cf) acc[nf][cf] += *(const f32x4*)(src + (nf * 2 + cf) * 1024);
  }
  grad_update<NF>(p, acc, k0, w, g, li, true);
}

// ---------------------------------------------------------------------------------------------
// Second half of a down projection fused into an attention epilogue (ABI 6, qfx_head_lora): U[m, j] = sum over the H per-head slabs
// in head order (fixed order: deterministic), then qfx_lora_down's outputs -- the K-extension image [U_hi | U_lo | U_hi] per group
// and the transposed hi / lo split.  One thread per (row, column); a block covers 256 / R rows.
struct HeadReduceBatch { qfx_lora_head_reduce_args a[2]; int start[3]; int n; };

__global__ __launch_bounds__(256) void lora_head_reduce_kernel(const HeadReduceBatch kb) {
  const int pi = (kb.n > 1 && (int)blockIdx.x >= kb.start[1]) ? 1 : 0;
  const qfx_lora_head_reduce_args& p = pi ? kb.a[1] : kb.a[0];
  const int R = p.R;
  const int rpb_ = 256 / R;                           // rows per block (R in {16, 32, 48, 64, 96}: 16 .. 2 rows)
  const int tid = threadIdx.x;
  if (tid >= rpb_ * R) return;
  const int m = ((int)blockIdx.x - kb.start[pi]) * rpb_ + tid / R, j = tid % R;
  if (m >= p.M) return;
  const int64_t jrow = remap_row(m, p.rows_per_batch, p.x_batch_rows, p.x_row_off);
  const float* src = p.part + jrow * p.ld_part + j;
  // 24 slabs requested at a time (independent loads: ONE L2 round trip for the DiT's 24 heads), added in head order: a
  // one-load-per-iteration loop is H dependent round trips (24 us per launch at H = 24 -- more than the qfx_lora_down launch this
  // replaces), eight at a time three
  constexpr int HC = 24;
  float v = 0.f;
  for (int h0 = 0; h0 < p.H; h0 += HC) {
    float t[HC];
#pragma unroll
    for (int i = 0; i < HC; ++i) t[i] = h0 + i < p.H ? src[(int64_t)(h0 + i) * p.part_hstride] : 0.f;
#pragma unroll
    for (int i = 0; i < HC; ++i) v += t[i];
  }
  const bf16_t hi = f2bf(v);
  const bf16_t lo = f2bf(v - bf2f(hi));
  if (p.ext) {
    bf16_t* e0 = p.ext + (int64_t)m * p.ld_ext + (j / p.group_R) * p.group_stride + (j % p.group_R);
    e0[0] = hi;
    e0[p.group_R] = lo;
    e0[2 * p.group_R] = hi;
  }
  if (p.Ut_hi) {
    p.Ut_hi[(int64_t)j * p.ld_ut + m] = hi;
    p.Ut_lo[(int64_t)j * p.ld_ut + m] = lo;
  }
}

// debug: dump the lane mapping of ds_read_b64_tr_b16 (lane l supplies elements 4l..4l+3 of `in`)
__global__ void tr_probe_kernel(const bf16_t* in, bf16_t* out) {
  __shared__ __attribute__((aligned(16))) bf16_t lds[256];
  const int l = threadIdx.x;
  for (int i = 0; i < 4; ++i) lds[l * 4 + i] = in[l * 4 + i];
  __syncthreads();
  const bf16x4v v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((QFX_AS3 bf16x4v*)(lds + l * 4));
  const bf16x4 r = __builtin_bit_cast(bf16x4, v);
  for (int i = 0; i < 4; ++i) out[l * 4 + i] = (bf16_t)r[i];
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lora_pack_kernel(const qfx_lora_pack_args* descs) {
  // One thread per input column k (A side) / output row n (B side): it owns the whole Kext-wide row of the K-extension image
  // (WeT[k,:] / We[n,:] -- written as contiguous 16-byte pieces) and one element of each of the Rp split rows (coalesced across the
  // wave).  The previous element-per-thread form scattered 2-byte stores at row stride: 0.40 ms per step for 240 adapters.
  const qfx_lora_pack_args d = descs[blockIdx.y];
  const int stride = gridDim.x * blockDim.x;
  const int t0 = blockIdx.x * blockDim.x + threadIdx.x;
  const int Rp = d.Rp, Kext = d.Kext;
  // head-fragment images (ABI 6): position of column c's element of row j -- see qfx_lora_pack_args
  auto hl_off = [&](int c, int j, int sel) {
    const int h = c / d.hl_dh, dd = c - h * d.hl_dh, ks = dd >> 5, db = (dd >> 4) & 1, g = (dd >> 2) & 3, r = dd & 3;
    return ((((int64_t)(h * (Rp >> 4) + (j >> 4)) * (d.hl_dh >> 5) + ks) * 2 + sel) * 64 + 16 * g + (j & 15)) * 8 + 4 * db + r;
  };
  for (int k = t0; k < d.K; k += stride) {
    bf16_t* wt = d.WeT + (int64_t)k * d.ld_wet;
    for (int j0 = 0; j0 < Rp; j0 += 8) {
      u32x4 vh, vl;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint32_t ph = 0, pl = 0;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int j = j0 + 2 * q + e;
          const float a = j < d.r ? d.A[(int64_t)j * d.K + k] : 0.f;
          const bf16_t hi = f2bf(a), lo = f2bf(a - bf2f(hi));
          d.A_hi[(int64_t)j * d.ld_a + k] = hi;
          d.A_lo[(int64_t)j * d.ld_a + k] = lo;
          if (d.A_hl) { d.A_hl[hl_off(k, j, 0)] = hi; d.A_hl[hl_off(k, j, 1)] = lo; }
          if (d.A_fr) {
            const int ja = d.fr_row0 + j;
            const int64_t o = ((int64_t)((k >> 5) * d.fr_nf + (ja >> 4)) * 64 + 16 * ((k >> 3) & 3) + (ja & 15)) * 8 + (k & 7);
            d.A_fr[o] = hi; d.A_fr[(int64_t)d.fr_nf * 16 * d.K + o] = lo;
          }
          ph |= (uint32_t)hi << (16 * e); pl |= (uint32_t)lo << (16 * e);
        }
        vh[q] = ph; vl[q] = pl;
      }
      *(u32x4*)(wt + j0) = vh; *(u32x4*)(wt + Rp + j0) = vh; *(u32x4*)(wt + 2 * Rp + j0) = vl;
    }
    for (int c = 3 * Rp; c < Kext; c += 8) *(u32x4*)(wt + c) = (u32x4){0u, 0u, 0u, 0u};
  }
  for (int n = t0; n < d.N; n += stride) {
    bf16_t* we = d.We + (int64_t)n * d.ld_we;
    const float* brow = d.B + (int64_t)n * d.r;
    for (int j0 = 0; j0 < Rp; j0 += 8) {
      u32x4 vh, vl;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint32_t ph = 0, pl = 0;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int j = j0 + 2 * q + e;
          const float b = j < d.r ? d.scale * brow[j] : 0.f;
          const bf16_t hi = f2bf(b), lo = f2bf(b - bf2f(hi));
          d.Bt_hi[(int64_t)j * d.ld_bt + n] = hi;
          d.Bt_lo[(int64_t)j * d.ld_bt + n] = lo;
          if (d.Bt_hl) { d.Bt_hl[hl_off(n, j, 0)] = hi; d.Bt_hl[hl_off(n, j, 1)] = lo; }
          ph |= (uint32_t)hi << (16 * e); pl |= (uint32_t)lo << (16 * e);
        }
        vh[q] = ph; vl[q] = pl;
      }
      *(u32x4*)(we + j0) = vh; *(u32x4*)(we + Rp + j0) = vh; *(u32x4*)(we + 2 * Rp + j0) = vl;
    }
    for (int c = 3 * Rp; c < Kext; c += 8) *(u32x4*)(we + c) = (u32x4){0u, 0u, 0u, 0u};
  }
}

}  // namespace

namespace {
int check_down(const qfx_lora_down_args* a) {
  if (!a->X || !a->W_hi || !a->W_lo) return QFX_EINVAL;
  if (a->M <= 0 || a->K <= 0 || (a->K % 32) || (a->ldx % 8) || (a->ldw % 8) || (a->R % 16) || a->R <= 0) return QFX_EINVAL;
  if (a->ext && (a->group_R <= 0 || (a->R % a->group_R))) return QFX_EINVAL;
  if (a->Ut_hi && (!a->Ut_lo || a->ld_ut < a->M)) return QFX_EINVAL;
  if (a->rows_per_batch <= 0) return QFX_EINVAL;
  if (a->xq && (!a->xs || (a->K % 128) || (a->ldxq % 8) || a->ldxq < a->K || a->xs_rows < a->M || (a->xq_kb0 % 4) || a->xq_kb0 < 0)) return QFX_EINVAL;
  return QFX_OK;
}
int check_grad(const qfx_lora_grad_args* a) {
  if (!a->Vt_hi || !a->Vt_lo || !a->X || !a->G) return QFX_EINVAL;
  if (a->M <= 0 || a->K <= 0 || (a->K % 8) || (a->ldx % 8) || a->R <= 0 || (a->R % 16) || a->rows_per_batch <= 0) return QFX_EINVAL;
  if ((a->ldvt % 8) || a->ldvt < ((a->M + 31) / 32) * 32) return QFX_EINVAL;  /* rows must be zero-padded to a multiple of 32 tokens */
  if (a->group_R <= 0 || (a->group_R % 16) || (a->R % a->group_R) || a->R / a->group_R > 3) return QFX_EINVAL;
  if (a->R / a->group_R > 1 && !a->G1) return QFX_EINVAL;
  if (a->R / a->group_R > 2 && !a->G2) return QFX_EINVAL;
  if (a->ws && (!a->ws_count || ((uintptr_t)a->ws % 16) || ((uintptr_t)a->ws_count % 4) || a->ws_floats < qfx_lora_grad_ws_floats(a->M, a->K, a->R))) return QFX_EINVAL;
  return QFX_OK;
}
}  // namespace

extern "C" int qfx_lora_down_batch(const qfx_lora_down_args* list, int32_t n, void* stream) {
  if (!list || n <= 0 || n > QFX_MAX_BATCH) return QFX_EINVAL;
  DownBatch b;
  int blocks16 = 0;
  for (int i = 0; i < n; ++i) blocks16 += (list[i].M + 15) / 16;
  // two 16-row groups per block only when that still leaves >= ~200 blocks (measured: 384 -> 192 blocks 23.8 -> 18.5 us,
  // but 128 -> 64 blocks 20.7 -> 23.7 us: the kernel is a latency chain and needs the CUs covered)
  const int rb = (blocks16 >= 300 && list[0].R <= 48) ? 2 : 1;
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    const int rc = check_down(&list[i]);
    if (rc) return rc;
    if (list[i].R != list[0].R) return QFX_EINVAL;   /* one MFMA fragment count per launch */
    b.a[i] = list[i];
    b.start[i] = blocks;
    blocks += (list[i].M + 16 * rb - 1) / (16 * rb);
  }
  for (int i = n; i <= QFX_MAX_BATCH; ++i) b.start[i] = blocks;
  b.n = n;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(blocks), block(512);
#define QFX_DOWN(NF) \
  do { if (rb == 2) hipLaunchKernelGGL((lora_down_kernel<NF, 2>), grid, block, 0, s, b); \
       else hipLaunchKernelGGL((lora_down_kernel<NF, 1>), grid, block, 0, s, b); } while (0)
  switch (list[0].R / 16) {
    case 1: QFX_DOWN(1); break;
    case 2: QFX_DOWN(2); break;
    case 3: QFX_DOWN(3); break;
    case 4: hipLaunchKernelGGL((lora_down_kernel<4, 1>), grid, block, 0, s, b); break;
    case 6: hipLaunchKernelGGL((lora_down_kernel<6, 1>), grid, block, 0, s, b); break;
    default: return QFX_EUNSUPPORTED;
  }
#undef QFX_DOWN
  QFX_CHECK_LAUNCH();
  return QFX_OK;
}

extern "C" int qfx_ln_down_fwd(const qfx_ln_down_args* list, int32_t n, void* stream) {
  if (!list || n <= 0 || n > 2) return QFX_EINVAL;
  LnDownBatch b;
  int blocks = 0, R = 0;
  for (int i = 0; i < n; ++i) {
    const qfx_ln_down_args& a = list[i];
    if (!a.ln.x || !a.ln.shift || !a.ln.scale || !a.ln.y || a.ln.rows <= 0 || a.ln.rows_per_batch <= 0) return QFX_EINVAL;
    if (a.ln.D <= 0 || (a.ln.D % 256) || a.ln.D > 3072 || (a.ln.mod_bstride % 8)) return QFX_EUNSUPPORTED;
    if (a.ln.yq && (!a.ln.ys || (a.ln.ldyq % 8) || a.ln.ldyq < a.ln.D || a.ln.ys_rows < a.ln.rows)) return QFX_EINVAL;
    if (a.W_hi) {
      if (!a.W_lo || (a.ldw % 8) || a.R <= 0 || (a.R % 16) || a.R > 48) return QFX_EUNSUPPORTED;
      if (R && a.R != R) return QFX_EINVAL;
      R = a.R;
      if (a.ext && (a.group_R <= 0 || (a.R % a.group_R))) return QFX_EINVAL;
      if (a.Ut_hi && (!a.Ut_lo || a.ld_ut < a.ln.rows)) return QFX_EINVAL;
      if ((uintptr_t)a.W_fr % 16) return QFX_EINVAL;
    }
    b.a[i] = a;
    b.start[i] = blocks;
    blocks += (a.ln.rows + 15) / 16;
  }
  for (int i = n; i <= 2; ++i) b.start[i] = blocks;
  if (n == 1) b.a[1] = list[0];
  b.n = n;
  hipStream_t s = (hipStream_t)stream;
  switch (R / 16) {
    case 0: case 1: hipLaunchKernelGGL(ln_down_kernel<1>, dim3(blocks), dim3(512), 0, s, b); break;
    case 2: hipLaunchKernelGGL(ln_down_kernel<2>, dim3(blocks), dim3(512), 0, s, b); break;
    default: hipLaunchKernelGGL(ln_down_kernel<3>, dim3(blocks), dim3(512), 0, s, b); break;
  }
  QFX_CHECK_LAUNCH();
  return QFX_OK;
}

extern "C" int qfx_lora_head_reduce(const qfx_lora_head_reduce_args* list, int32_t n, void* stream) {
  if (!list || n <= 0 || n > 2) return QFX_EINVAL;
  HeadReduceBatch b;
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    const qfx_lora_head_reduce_args& a = list[i];
    if (!a.part || a.H <= 0 || a.M <= 0 || a.R <= 0 || a.R > 256 || (a.R % 16) || a.ld_part < a.R || a.group_R <= 0 || (a.R % a.group_R) ||
        a.rows_per_batch <= 0 || (!a.ext && !a.Ut_hi) || ((a.Ut_hi == nullptr) != (a.Ut_lo == nullptr)))
      return QFX_EINVAL;
    // the kernel writes Ut[j * ld_ut + m], ext[m * ld_ext + group * group_stride + {0, 1, 2} * group_R + j'] and reads
    // part[h * part_hstride + row * ld_part + j] unchecked: a descriptor that does not cover them is an error, not a memory stomp
    if (a.Ut_hi && a.ld_ut < a.M) return QFX_EINVAL;
    if (a.ext && (a.group_stride < 0 || a.ld_ext < (int64_t)(a.R / a.group_R - 1) * a.group_stride + 3 * a.group_R)) return QFX_EINVAL;
    {
      const int64_t nb = (a.M + a.rows_per_batch - 1) / a.rows_per_batch;
      const int64_t last_row = a.x_batch_rows == 0 ? (int64_t)a.M - 1
                                                     : (nb - 1) * a.x_batch_rows + a.x_row_off + ((a.M - 1) % a.rows_per_batch);
      if (a.x_row_off < 0 || a.x_batch_rows < 0 || a.part_hstride < (last_row + 1) * (int64_t)a.ld_part) return QFX_EINVAL;
    }
    b.a[i] = a;
    b.start[i] = blocks;
    const int rpb_ = 256 / a.R;
    blocks += (a.M + rpb_ - 1) / rpb_;
  }
  for (int i = n; i <= 2; ++i) b.start[i] = blocks;
  if (n == 1) b.a[1] = list[0];
  b.n = n;
  hipLaunchKernelGGL(lora_head_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, b);
  QFX_CHECK_LAUNCH();
  return QFX_OK;
}

extern "C" int qfx_lora_down(const qfx_lora_down_args* a, void* stream) {
  if (!a) return QFX_EINVAL;
  return qfx_lora_down_batch(a, 1, stream);
}

extern "C" int qfx_lora_grad_batch(const qfx_lora_grad_args* list, int32_t n, void* stream) {
  if (!list || n <= 0 || n > QFX_MAX_BATCH) return QFX_EINVAL;
  GradBatch b;
  int chunks = 0, kmax = 0;
  for (int i = 0; i < n; ++i) {
    const int rc = check_grad(&list[i]);
    if (rc) return rc;
    if (list[i].R != list[0].R) return QFX_EINVAL;
    b.a[i] = list[i];
    b.start[i] = chunks;
    chunks += (list[i].M + GRAD_CH - 1) / GRAD_CH;
    kmax = list[i].K > kmax ? list[i].K : kmax;
  }
  for (int i = n; i <= QFX_MAX_BATCH; ++i) b.start[i] = chunks;
  b.n = n;
  dim3 grid((kmax + 127) / 128, chunks);
  hipStream_t s = (hipStream_t)stream;
  switch (list[0].R / 16) {
    case 1: hipLaunchKernelGGL(lora_grad_kernel<1>, grid, dim3(256), 0, s, b); break;
    case 2: hipLaunchKernelGGL(lora_grad_kernel<2>, grid, dim3(256), 0, s, b); break;
    case 3: hipLaunchKernelGGL(lora_grad_kernel<3>, grid, dim3(256), 0, s, b); break;
    case 4: hipLaunchKernelGGL(lora_grad_kernel<4>, grid, dim3(256), 0, s, b); break;
    case 6: hipLaunchKernelGGL(lora_grad_kernel<6>, grid, dim3(256), 0, s, b); break;
    default: return QFX_EUNSUPPORTED;
  }
  QFX_CHECK_LAUNCH();
#if (QFX_GRAD_HANDOFF & 8) != 0
  bool any = false;
  for (int i = 0; i < n; ++i) any = any || (list[i].ws && list[i].M > GRAD_CH);
  if (any) {
    dim3 rgrid((kmax + 127) / 128, n);
    switch (list[0].R / 16) {
      case 1: hipLaunchKernelGGL(lora_grad_reduce_kernel<1>, rgrid, dim3(256), 0, s, b); break;
      case 2: hipLaunchKernelGGL(lora_grad_reduce_kernel<2>, rgrid, dim3(256), 0, s, b); break;
      case 3: hipLaunchKernelGGL(lora_grad_reduce_kernel<3>, rgrid, dim3(256), 0, s, b); break;
      case 4: hipLaunchKernelGGL(lora_grad_reduce_kernel<4>, rgrid, dim3(256), 0, s, b); break;
      default: hipLaunchKernelGGL(lora_grad_reduce_kernel<6>, rgrid, dim3(256), 0, s, b); break;
    }
    QFX_CHECK_LAUNCH();
  }
#endif
  return QFX_OK;
}

extern "C" int64_t qfx_lora_grad_ws_floats(int32_t M, int32_t K, int32_t R) {
  if (M <= 0 || K <= 0 || R <= 0 || (R % 16)) return 0;
  const int64_t chunks = (M + GRAD_CH - 1) / GRAD_CH;
  return chunks > 1 ? chunks * ((K + 127) / 128) * (R / 16) * 2048 : 0;
}

extern "C" int qfx_lora_grad(const qfx_lora_grad_args* a, void* stream) {
  if (!a) return QFX_EINVAL;
  return qfx_lora_grad_batch(a, 1, stream);
}

extern "C" int qfx_lora_pack(const qfx_lora_pack_args* descs, int32_t n, int32_t max_dim, void* stream) {
  if (!descs || n <= 0 || max_dim <= 0) return QFX_EINVAL;
  int bx = (max_dim + 255) / 256;   /* one thread per input column / output row */
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(lora_pack_kernel, dim3(bx, n), dim3(256), 0, (hipStream_t)stream, descs);
  QFX_CHECK_LAUNCH();
  return QFX_OK;
}

extern "C" int qfx_debug_tr_read(const uint16_t* in, uint16_t* out, void* stream) {
  if (!in || !out) return QFX_EINVAL;
  hipLaunchKernelGGL(tr_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, in, out);
  QFX_CHECK_LAUNCH();
  return QFX_OK;
}
