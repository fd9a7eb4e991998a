// qfx_elem.hip -- the HBM-bound row kernels of the DiT block: LayerNorm+modulate (fwd/bwd),
// RMSNorm, modulation GEMV, QK-RMSNorm+RoPE (fwd/bwd), head transposes, criterion, clip+AdamW.
// All bf16 traffic is 16 bytes per lane; one wave owns one row so reductions are shuffle-only.
// Rounding points replicate the reference's bf16 eager graph (every torch op rounds its output).
#include "qfx_common.h"
#include <hip/hip_fp16.h>
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace {

constexpr int MAXP = 8;  // 8 passes x 64 lanes x 8 elems = rows up to 4096 wide

__device__ __forceinline__ void ld8(const bf16_t* p, float (&v)[8]) {
  const u32x4 u = *(const u32x4*)p;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = __uint_as_float(u[i] << 16);
    v[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u);
  }
}
// the same for a stream that is read ONCE per launch (non-temporal: does not displace what other kernels keep in the caches)
__device__ __forceinline__ void ld8_nt(const bf16_t* p, float (&v)[8]) {
  const u32x4 u = __builtin_nontemporal_load((const u32x4*)p);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = __uint_as_float(u[i] << 16);
    v[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ void st8(bf16_t* p, const float (&v)[8]) {
  u32x4 u;
#pragma unroll
  for (int i = 0; i < 4; ++i) u[i] = pack2bf(v[2 * i], v[2 * i + 1]);
  *(u32x4*)p = u;
}

// MX-FP8 image of 8 consecutive output columns of a row-per-wave kernel (lane l holds columns 8l' .. 8l'+7: an MX block is 4
// adjacent lanes).  `o` are the fp32 values that st8 rounds to bf16: the image is taken from the ROUNDED values, like a separate
// qfx_quant_mxfp8 pass over the bf16 tensor.  Every lane of the 4-lane group must call it (shuffles).
__device__ __forceinline__ void mx_store8(const float (&o)[8], uint8_t* dst, uint8_t* sc, int sc_rows, int row, int col, int lane) {
  float v[8];
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) { v[i] = rbf(o[i]); amax = fmaxf(amax, fabsf(v[i])); }
  amax = fmaxf(amax, __shfl_xor(amax, 1));
  amax = fmaxf(amax, __shfl_xor(amax, 2));
  int eb;
  const u32x2 q = mx_quant8(v, amax, eb);
  *(u32x2*)dst = q;
  const int kb = col >> 5;
  if ((lane & 3) == 0) sc[((int64_t)(kb >> 2) * sc_rows + row) * 4 + (kb & 3)] = (uint8_t)eb;
}

// ---------------------------------------------------------------- LayerNorm + modulate, forward
// Batched: up to QFX_MAX_LN_BATCH problems (e.g. the image and the text stream of a block) share one launch; a wave owns
// one row of one problem.  The tiny text-stream problems otherwise pay a full dispatch gap + memory round trip each.
struct LnFwdBatch { qfx_ln_fwd_args a[QFX_MAX_LN_BATCH]; int n; };
struct LnBwdBatch { qfx_ln_bwd_args a[QFX_MAX_LN_BATCH]; int n; };

__global__ __launch_bounds__(256) void ln_mod_fwd_kernel(const LnFwdBatch bt) {
  const int lane = threadIdx.x & 63;
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  int pi = 0;
#pragma unroll
  for (int i = 0; i + 1 < QFX_MAX_LN_BATCH; ++i)
    if (pi == i && i + 1 < bt.n && row >= bt.a[i].rows) { row -= bt.a[i].rows; pi = i + 1; }
  const bf16_t* __restrict__ x; const bf16_t* __restrict__ shift; const bf16_t* __restrict__ scale; bf16_t* __restrict__ y;
  int64_t mod_bstride; int rows, D, rpb; float eps;
  uint8_t* yq; uint8_t* ys; int64_t ldyq; int ys_rows;
  {
    const qfx_ln_fwd_args& q = pi == 0 ? bt.a[0] : (pi == 1 ? bt.a[1] : (pi == 2 ? bt.a[2] : bt.a[3]));
    x = q.x; shift = q.shift; scale = q.scale; y = q.y; mod_bstride = q.mod_bstride; rows = q.rows; D = q.D; rpb = q.rows_per_batch; eps = q.eps;
    yq = q.yq; ys = q.ys; ldyq = q.ldyq; ys_rows = q.ys_rows;
  }
  if (row >= rows) return;
  const int b = row / rpb;
  const bf16_t* xr = x + (int64_t)row * D;
  float v[MAXP][8];
  float s = 0.f;
#pragma unroll
  for (int p = 0; p < MAXP; ++p) {
    const int col = (p * 64 + lane) * 8;
    if (col < D) {
      ld8(xr + col, v[p]);
#pragma unroll
      for (int i = 0; i < 8; ++i) s += v[p][i];
    }
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int p = 0; p < MAXP; ++p) {
    const int col = (p * 64 + lane) * 8;
    if (col < D) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { const float d = v[p][i] - mean; q += d * d; }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
  for (int p = 0; p < MAXP; ++p) {
    const int col = (p * 64 + lane) * 8;
    if (col < D) {
      float sc[8], sh[8], o[8];
      ld8(scale + (int64_t)b * mod_bstride + col, sc);
      ld8(shift + (int64_t)b * mod_bstride + col, sh);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float ln = rbf((v[p][i] - mean) * rstd);
        const float t1 = rbf(1.0f + sc[i]);
        o[i] = rbf(ln * t1) + sh[i];
      }
      st8(y + (int64_t)row * D + col, o);
      if (yq) mx_store8(o, yq + (int64_t)row * ldyq + col, ys, ys_rows, row, col, lane);
    }
  }
}

// ---------------------------------------------------------------- LayerNorm + modulate, backward
template <int NP>   // passes of 512 columns: NP = ceil(D / 512) (register arrays are sized by it)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NP <= 6 ? 3 : 2, NP <= 6 ? 3 : 2))) void ln_mod_bwd_kernel(const LnBwdBatch bt) {
  const int lane = threadIdx.x & 63;
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  int pi = 0;
#pragma unroll
  for (int i = 0; i + 1 < QFX_MAX_LN_BATCH; ++i)
    if (pi == i && i + 1 < bt.n && row >= bt.a[i].rows) { row -= bt.a[i].rows; pi = i + 1; }
  const bf16_t* __restrict__ dy; const bf16_t* __restrict__ x; const bf16_t* __restrict__ scale; const bf16_t* __restrict__ dres;
  const bf16_t* __restrict__ gate; bf16_t* __restrict__ dx; bf16_t* __restrict__ dyg; const float* __restrict__ row_mask;
  int64_t mod_bstride, gate_bstride; int rows, D, rpb; float eps;
  uint8_t* dygq; uint8_t* dygs; int64_t lddygq; int dygs_rows;
  {
    const qfx_ln_bwd_args& q = pi == 0 ? bt.a[0] : (pi == 1 ? bt.a[1] : (pi == 2 ? bt.a[2] : bt.a[3]));
    dy = q.dy; x = q.x; scale = q.scale; dres = q.dres; gate = q.gate; dx = q.dx; dyg = q.dyg; row_mask = q.row_mask;
    mod_bstride = q.mod_bstride; gate_bstride = q.gate_bstride; rows = q.rows; D = q.D; rpb = q.rows_per_batch; eps = q.eps;
    dygq = q.dygq; dygs = q.dygs; lddygq = q.lddygq; dygs_rows = q.dygs_rows;
  }
  if (row >= rows) return;
  const int b = row / rpb;
  const int64_t ro = (int64_t)row * D;
  if (row_mask != nullptr && row_mask[row] == 0.f) {   // padded token: no gradient flows through it
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int col = (p * 64 + lane) * 8;
      if (col < D) {
        *(u32x4*)(dx + ro + col) = z;
        if (dyg) *(u32x4*)(dyg + ro + col) = z;
        if (dyg && dygq) { const float zf[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; mx_store8(zf, dygq + (int64_t)row * lddygq + col, dygs, dygs_rows, row, col, lane); }
      }
    }
    return;
  }
  // All three HBM streams of the row (x, dy, dres) are requested up front and kept as packed bf16 (one memory latency instead of
  // three dependent ones); g = bf16(dy * bf16(1+scale)) is exactly representable in bf16 and is kept packed too.
  u32x4 xr[NP], dyr[NP], drr[NP], gp[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int col = (p * 64 + lane) * 8;
    if (col < D) {
      xr[p] = *(const u32x4*)(x + ro + col);
      dyr[p] = *(const u32x4*)(dy + ro + col);
      if (dres) drr[p] = *(const u32x4*)(dres + ro + col);
    }
  }
  auto lo = [](unsigned u) { return __uint_as_float(u << 16); };
  auto hi = [](unsigned u) { return __uint_as_float(u & 0xffff0000u); };
  float s = 0.f;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int col = (p * 64 + lane) * 8;
    if (col < D) {
#pragma unroll
      for (int i = 0; i < 4; ++i) s += lo(xr[p][i]) + hi(xr[p][i]);
    }
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int col = (p * 64 + lane) * 8;
    if (col < D) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { const float d0 = lo(xr[p][i]) - mean, d1 = hi(xr[p][i]) - mean; q += d0 * d0 + d1 * d1; }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
  float c1 = 0.f, c2 = 0.f;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int col = (p * 64 + lane) * 8;
    if (col < D) {
      const u32x4 sc = *(const u32x4*)(scale + (int64_t)b * mod_bstride + col);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float g0 = rbf(lo(dyr[p][i]) * rbf(1.0f + lo(sc[i])));
        const float g1 = rbf(hi(dyr[p][i]) * rbf(1.0f + hi(sc[i])));
        gp[p][i] = pack2bf(g0, g1);
        c1 += g0 + g1;
        c2 += g0 * ((lo(xr[p][i]) - mean) * rstd) + g1 * ((hi(xr[p][i]) - mean) * rstd);
      }
    }
  }
  c1 = wave_sum(c1) / (float)D;
  c2 = wave_sum(c2) / (float)D;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int col = (p * 64 + lane) * 8;
    if (col < D) {
      float o[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float x0 = (lo(xr[p][i]) - mean) * rstd, x1 = (hi(xr[p][i]) - mean) * rstd;
        const float d0 = rbf((lo(gp[p][i]) - c1 - x0 * c2) * rstd), d1 = rbf((hi(gp[p][i]) - c1 - x1 * c2) * rstd);
        o[2 * i] = dres ? rbf(lo(drr[p][i]) + d0) : d0;
        o[2 * i + 1] = dres ? rbf(hi(drr[p][i]) + d1) : d1;
      }
      st8(dx + ro + col, o);
      if (dyg) {
        float gt[8], og[8];
        ld8(gate + (int64_t)b * gate_bstride + col, gt);
#pragma unroll
        for (int i = 0; i < 8; ++i) og[i] = gt[i] * o[i];
        st8(dyg + ro + col, og);
        if (dygq) mx_store8(og, dygq + (int64_t)row * lddygq + col, dygs, dygs_rows, row, col, lane);
      }
    }
  }
}

#if defined(QFX_LN_BWD_PIPE)
// A/B build (profiles/r05_ln_pipeline.json): the same backward on a PERSISTENT grid (one 4-wave block per CU), every wave
// walking its rows with a two-row software pipeline -- the three HBM streams of row i+1 are requested before the reductions and stores of
// row i.  Same arithmetic, same row -> wave mapping within a 4-row block; the MX-FP8 side output is not supported here.
template <int NP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void ln_mod_bwd_pipe_kernel(const LnBwdBatch bt, int nblk) {
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // the argument block is read from the kernarg segment (scalar loads): pointers into the by-value parameter would copy it to scratch
  typedef const __attribute__((address_space(4))) qfx_ln_bwd_args KA;
  typedef const __attribute__((address_space(4))) LnBwdBatch KB;
  KB& kb = *(KB*)__builtin_amdgcn_kernarg_segment_ptr();
  (void)bt;
  struct Row { KA* q; int row; bool live; };
  auto decode = [&](int blk) {
    int row = blk * 4 + wv, pi = 0;
#pragma unroll
    for (int i = 0; i + 1 < QFX_MAX_LN_BATCH; ++i)
      if (pi == i && i + 1 < kb.n && row >= kb.a[i].rows) { row -= kb.a[i].rows; pi = i + 1; }
    KA* q = &kb.a[pi];
    return Row{q, row, row < q->rows};
  };
  auto lo = [](unsigned u) { return __uint_as_float(u << 16); };
  auto hi = [](unsigned u) { return __uint_as_float(u & 0xffff0000u); };
  auto request = [&](const Row& r, u32x4 (&xr)[NP], u32x4 (&dyr)[NP], u32x4 (&drr)[NP]) {
    if (!r.live) return;
    const int64_t ro = (int64_t)r.row * r.q->D;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int col = (p * 64 + lane) * 8;
      if (col < r.q->D) {
        xr[p] = *(const u32x4*)(r.q->x + ro + col);
        dyr[p] = *(const u32x4*)(r.q->dy + ro + col);
        if (r.q->dres) drr[p] = *(const u32x4*)(r.q->dres + ro + col);
      }
    }
  };
  auto finish = [&](const Row& r, const u32x4 (&xr)[NP], const u32x4 (&dyr)[NP], const u32x4 (&drr)[NP]) {
    if (!r.live) return;
    KA& q = *r.q;
    const int D = q.D, b = r.row / q.rows_per_batch;
    const int64_t ro = (int64_t)r.row * D;
    if (q.row_mask != nullptr && q.row_mask[r.row] == 0.f) {
      const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int col = (p * 64 + lane) * 8;
        if (col < D) { *(u32x4*)(q.dx + ro + col) = z; if (q.dyg) *(u32x4*)(q.dyg + ro + col) = z; }
      }
      return;
    }
    float s = 0.f;
#pragma unroll
    for (int p = 0; p < NP; ++p)
      if ((p * 64 + lane) * 8 < D) {
#pragma unroll
        for (int i = 0; i < 4; ++i) s += lo(xr[p][i]) + hi(xr[p][i]);
      }
    const float mean = wave_sum(s) / (float)D;
    float qq = 0.f;
#pragma unroll
    for (int p = 0; p < NP; ++p)
      if ((p * 64 + lane) * 8 < D) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float d0 = lo(xr[p][i]) - mean, d1 = hi(xr[p][i]) - mean; qq += d0 * d0 + d1 * d1; }
      }
    const float rstd = rsqrtf(wave_sum(qq) / (float)D + q.eps);
    u32x4 gp[NP];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int col = (p * 64 + lane) * 8;
      if (col < D) {
        const u32x4 sc = *(const u32x4*)(q.scale + (int64_t)b * q.mod_bstride + col);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float g0 = rbf(lo(dyr[p][i]) * rbf(1.0f + lo(sc[i])));
          const float g1 = rbf(hi(dyr[p][i]) * rbf(1.0f + hi(sc[i])));
          gp[p][i] = pack2bf(g0, g1);
          c1 += g0 + g1;
          c2 += g0 * ((lo(xr[p][i]) - mean) * rstd) + g1 * ((hi(xr[p][i]) - mean) * rstd);
        }
      }
    }
    c1 = wave_sum(c1) / (float)D;
    c2 = wave_sum(c2) / (float)D;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int col = (p * 64 + lane) * 8;
      if (col < D) {
        float o[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float x0 = (lo(xr[p][i]) - mean) * rstd, x1 = (hi(xr[p][i]) - mean) * rstd;
          const float d0 = rbf((lo(gp[p][i]) - c1 - x0 * c2) * rstd), d1 = rbf((hi(gp[p][i]) - c1 - x1 * c2) * rstd);
          o[2 * i] = q.dres ? rbf(lo(drr[p][i]) + d0) : d0;
          o[2 * i + 1] = q.dres ? rbf(hi(drr[p][i]) + d1) : d1;
        }
        st8(q.dx + ro + col, o);
        if (q.dyg) {
          float gt[8], og[8];
          ld8(q.gate + (int64_t)b * q.gate_bstride + col, gt);
#pragma unroll
          for (int i = 0; i < 8; ++i) og[i] = gt[i] * o[i];
          st8(q.dyg + ro + col, og);
        }
      }
    }
  };
  int blk = blockIdx.x;
  if (blk >= nblk) return;
  u32x4 xa[NP], da[NP], ra[NP], xb[NP], db[NP], rb_[NP];
  Row cur = decode(blk);
  request(cur, xa, da, ra);
  for (;;) {
    const int nb = blk + gridDim.x;
    const bool more = nb < nblk;
    Row nxt = cur;
    if (more) { nxt = decode(nb); request(nxt, xb, db, rb_); }
    __builtin_amdgcn_sched_barrier(0);
    finish(cur, xa, da, ra);
    if (!more) break;
#pragma unroll
    for (int p = 0; p < NP; ++p) { xa[p] = xb[p]; da[p] = db[p]; ra[p] = rb_[p]; }
    cur = nxt; blk = nb;
  }
}
#endif

// ---------------------------------------------------------------- gradients of the modulation vectors (shift, scale, gate)
// grid = (row chunks of 32 per sample, B); block = 4 waves; a wave owns 8 consecutive rows (one row at a time in registers, as in
// the LayerNorm kernels), accumulates its column sums in registers, the four waves combine in LDS and the block issues ONE fp32
// atomic per column and output.  HBM-bound (reads dy, x [, dxo, y] once).
constexpr int MG_RPW = 2;
// Batched: up to QFX_MAX_LN_BATCH problems of one width share a launch (the image and the text stream of a block: the 384-row text
// problem otherwise pays a full dispatch gap + latency chain for 48 blocks of work); blockIdx.x walks the problems' row chunks.
struct ModGradBatch { qfx_mod_grad_args a[QFX_MAX_LN_BATCH]; int start[QFX_MAX_LN_BATCH + 1]; int n; };
template <int NP>
__global__ __launch_bounds__(256) void mod_grad_kernel(const ModGradBatch bt) {
  constexpr int NS = NP <= 6 ? 4 : 2;         // LDS slabs: one per wave (144 KiB at NP = 6; the kernel runs one block per CU anyway), two beyond
  __shared__ float sacc[NS][3][NP * 512];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int b = blockIdx.y;
  int pi = 0;
#pragma unroll
  for (int i = 1; i < QFX_MAX_LN_BATCH; ++i)
    if (i < bt.n && (int)blockIdx.x >= bt.start[i]) pi = i;
  const qfx_mod_grad_args& a = pi == 0 ? bt.a[0] : (pi == 1 ? bt.a[1] : (pi == 2 ? bt.a[2] : bt.a[3]));
  const int bx = (int)blockIdx.x - bt.start[pi];
  if ((int64_t)b * a.rows_per_batch >= a.rows) return;      // grid.y is sized for the problem with the most samples
  const int D = a.D;
  const bool has_gate = a.dgate != nullptr;
  float as[NP][8], ac[NP][8], ag[NP][8];
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int i = 0; i < 8; ++i) { as[p][i] = 0.f; ac[p][i] = 0.f; ag[p][i] = 0.f; }
  // MG_RPW rows per wave, 4 * MG_RPW per block: with 8 rows per wave a 2048-row stream was 64 blocks (a quarter of the CUs), each
  // wave a chain of 8 x 3 dependent memory round trips.  Two rows per wave fill the chip four times over (one global atomic per
  // column and block: 4x the atomics, still < 10 us of them), and all four streams of a row (x, dy, dxo, y) are requested
  // together, packed.  (114 -> 45 us per launch with the plain LDS combine below; 29 -> 10 ms of an all-linear step.)
  const int r_lo = bx * (4 * MG_RPW) + w * MG_RPW;
  for (int rr = 0; rr < MG_RPW; ++rr) {
    const int rl = r_lo + rr;
    if (rl >= a.rows_per_batch) break;            // wave-uniform
    const int64_t row = (int64_t)b * a.rows_per_batch + rl;
    if (row >= a.rows) break;
    if (a.row_mask != nullptr && a.row_mask[row] == 0.f) continue;
    u32x4 gxp[NP], gyp[NP];
    if (has_gate) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int col = (p * 64 + lane) * 8;
        if (col < D) { gxp[p] = *(const u32x4*)(a.dxo + row * a.ld_dxo + col); gyp[p] = *(const u32x4*)(a.y + row * a.ld_y + col); }
      }
    }
    float xv[NP][8], dv[NP][8];
    float s = 0.f;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int col = (p * 64 + lane) * 8;
      if (col < D) {
        ld8(a.x + row * a.ld_x + col, xv[p]);
        ld8(a.dy + row * a.ld_dy + col, dv[p]);
#pragma unroll
        for (int i = 0; i < 8; ++i) s += xv[p][i];
      }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int col = (p * 64 + lane) * 8;
      if (col < D) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float d = xv[p][i] - mean; q += d * d; }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + a.eps);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int col = (p * 64 + lane) * 8;
      if (col < D) {
        // (two rows per wave: the row's terms go straight into the block's LDS sums -- register accumulators cost 144 VGPRs and
        // with them the kernel ran one wave per SIMD)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          as[p][i] += dv[p][i];
          ac[p][i] += dv[p][i] * rbf((xv[p][i] - mean) * rstd);
        }
        if (has_gate) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            ag[p][2 * i] += __uint_as_float(gxp[p][i] << 16) * __uint_as_float(gyp[p][i] << 16);
            ag[p][2 * i + 1] += __uint_as_float(gxp[p][i] & 0xffff0000u) * __uint_as_float(gyp[p][i] & 0xffff0000u);
          }
        }
      }
    }
  }
  // each wave parks its register sums in its OWN LDS slab (plain, independent stores), one barrier, then every thread adds the four
  // slabs for its columns.  No LDS float atomics: one costs ~600 cycles per wave instruction here, conflict-free or not (288 of them
  // were 80 of the kernel's 110 us); no read-modify-write chains either.  Element-major slots -- column (p*64 + lane)*8 + i at
  // i*(NP*64) + p*64 + lane -- keep the 64 lanes of a store on 64 banks.
#pragma unroll
  for (int rnd = 0; rnd < 4 / NS; ++rnd) {
    if (w / NS == rnd) {
      const int sb = w % NS;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int sl = p * 64 + lane;
        if (sl * 8 < D) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int o = i * (NP * 64) + sl;
            sacc[sb][0][o] = (rnd ? sacc[sb][0][o] : 0.f) + as[p][i];
            sacc[sb][1][o] = (rnd ? sacc[sb][1][o] : 0.f) + ac[p][i];
            if (has_gate) sacc[sb][2][o] = (rnd ? sacc[sb][2][o] : 0.f) + ag[p][i];
          }
        }
      }
    }
    __syncthreads();
  }
  for (int c = threadIdx.x; c < D; c += 256) {
    const int slot = (c & 7) * (NP * 64) + (c >> 3);
    float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int sb = 0; sb < NS; ++sb) { t0 += sacc[sb][0][slot]; t1 += sacc[sb][1][slot]; if (has_gate) t2 += sacc[sb][2][slot]; }
    unsafeAtomicAdd(a.dshift + (int64_t)b * a.out_bstride + c, t0);
    unsafeAtomicAdd(a.dscale + (int64_t)b * a.out_bstride + c, t1);
    if (has_gate) unsafeAtomicAdd(a.dgate + (int64_t)b * a.out_bstride + c, t2);
  }
}

__global__ __launch_bounds__(256) void gate_mul_kernel(const bf16_t* __restrict__ dx, const bf16_t* __restrict__ gate,
                                                       int64_t gate_bstride, bf16_t* __restrict__ dyg, int64_t total8,
                                                       int D, int rpb) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total8; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = i * 8;
    const int row = (int)(e / D), col = (int)(e % D);
    float a[8], g[8], o[8];
    ld8(dx + e, a);
    ld8(gate + (int64_t)(row / rpb) * gate_bstride + col, g);
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = a[k] * g[k];
    st8(dyg + e, o);
  }
}

// ---------------------------------------------------------------- RMSNorm with weight (txt_norm)
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                          bf16_t* __restrict__ y, int rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float v[MAXP][8];
  float s = 0.f;
#pragma unroll
  for (int p = 0; p < MAXP; ++p) {
    const int col = (p * 64 + lane) * 8;
    if (col < D) {
      ld8(x + (int64_t)row * D + col, v[p]);
#pragma unroll
      for (int i = 0; i < 8; ++i) s += v[p][i] * v[p][i];
    }
  }
  const float rstd = rsqrtf(wave_sum(s) / (float)D + eps);
#pragma unroll
  for (int p = 0; p < MAXP; ++p) {
    const int col = (p * 64 + lane) * 8;
    if (col < D) {
      float ww[8], o[8];
      ld8(w + col, ww);
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = rbf(v[p][i] * rstd) * ww[i];
      st8(y + (int64_t)row * D + col, o);
    }
  }
}

// ---------------------------------------------------------------- modulation GEMV (pure weight streaming)
// out[mat][b][n] = bf16(sum_k bf16(silu(temb[b][k])) W_mat[n][k] + bias_mat[n]); wave = 4 rows of W.
__global__ __launch_bounds__(256) void mod_gemv_kernel(const bf16_t* __restrict__ temb, int B, int K,
                                                       const bf16_t* const* __restrict__ Ws,
                                                       const bf16_t* const* __restrict__ biases, int N,
                                                       int apply_silu, bf16_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16_t* sT = (bf16_t*)smem_raw;  // [B][K] bf16(silu(temb))
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int i = tid; i < B * K; i += 256) {
    const float t = bf2f(temb[i]);
    sT[i] = apply_silu ? f2bf(t / (1.0f + __expf(-t))) : temb[i];
  }
  __syncthreads();
  const int mat = blockIdx.y;
  const bf16_t* W = Ws[mat];
  const int n0 = (blockIdx.x * 4 + w) * 4;
  if (n0 >= N) return;
  float acc[4][8];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[r][b] = 0.f;
  for (int k = lane * 8; k < K; k += 512) {
    float wv[4][8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + r < N ? n0 + r : N - 1;
      ld8_nt(W + (int64_t)n * K + k, wv[r]);      // 13.6 GB per step read exactly once: non-temporal (-4.6 %: 6.0 -> 6.3 TB/s)
    }
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      if (b < B) {
        float tv[8];
        ld8(sT + b * K + k, tv);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[r][b] = fmaf(wv[r][i], tv[i], acc[r][b]);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int b = 0; b < 8; ++b)
      if (b < B) acc[r][b] = wave_sum(acc[r][b]);
  if (lane == 0) {
    const bf16_t* bias = biases ? biases[mat] : nullptr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + r;
      if (n < N) {
        const float bv = bias ? bf2f(bias[n]) : 0.f;
#pragma unroll
        for (int b = 0; b < 8; ++b)
          if (b < B) out[((int64_t)mat * B + b) * N + n] = f2bf(acc[r][b] + bv);
      }
    }
  }
}

// ---------------------------------------------------------------- transposed modulation GEMV (backward of the frozen AdaLN linears)
// out[b, k] += sum_mat sum_n dy[mat, b, n] * W_mat[n, k]: d(silu(temb)) through the frozen base weights of every modulation linear,
// needed when the conditioning head carries adapters.  One pass over the same 13.6 GB as qfx_mod_gemv: persistent blocks stride
// over (matrix, 4-row group) items, a wave owns one whole weight row at a time (6 KB contiguous) and keeps the [NB, K] partial sums
// in registers (lane l: columns 8l + 512p); four row loads in flight per wave; one LDS combine and one fp32 atomic per (b, k) and
// block at the end.
template <int NB, int NP>
__global__ __launch_bounds__(256) void mod_gemv_t_kernel(const bf16_t* __restrict__ dy, int B, int b0, int N, int K,
                                                         const bf16_t* const* __restrict__ Ws, int nmat, float* __restrict__ out) {
  __shared__ float sacc[NB][NP * 512];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int i = tid; i < NB * NP * 512; i += 256) (&sacc[0][0])[i] = 0.f;
  __syncthreads();
  float acc[NB][NP][8];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[b][p][i] = 0.f;
  const int groups_per_mat = (N + 3) / 4;
  const int64_t nitems = (int64_t)nmat * groups_per_mat;
  // wave-granular grid stride: item = (matrix, group of 4 consecutive rows); the 4 rows are 4 independent loads in flight
  for (int64_t it = (int64_t)blockIdx.x * 4 + w; it < nitems; it += (int64_t)gridDim.x * 4) {
    const int mat = (int)(it / groups_per_mat);
    const int n0 = (int)(it % groups_per_mat) * 4;
    const bf16_t* W = Ws[mat];
    u32x4 wv[4][NP];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + r < N ? n0 + r : N - 1;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int col = (p * 64 + lane) * 8;
        if (col < K) wv[r][p] = *(const u32x4*)(W + (int64_t)n * K + col);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (n0 + r < N) {      // wave-uniform
        float g[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) g[b] = (b0 + b < B) ? bf2f(dy[((int64_t)mat * B + b0 + b) * N + n0 + r]) : 0.f;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const int col = (p * 64 + lane) * 8;
          if (col < K) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float w0 = __uint_as_float(wv[r][p][i] << 16), w1 = __uint_as_float(wv[r][p][i] & 0xffff0000u);
#pragma unroll
              for (int b = 0; b < NB; ++b) { acc[b][p][2 * i] = fmaf(g[b], w0, acc[b][p][2 * i]); acc[b][p][2 * i + 1] = fmaf(g[b], w1, acc[b][p][2 * i + 1]); }
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int col = (p * 64 + lane) * 8;
      if (col < K) {
#pragma unroll
        for (int i = 0; i < 8; ++i) atomicAdd(&sacc[b][col + i], acc[b][p][i]);
      }
    }
  __syncthreads();
  for (int c = tid; c < K; c += 256)
#pragma unroll
    for (int b = 0; b < NB; ++b)
      if (b0 + b < B) unsafeAtomicAdd(out + (int64_t)(b0 + b) * K + c, sacc[b][c]);
}

// ---------------------------------------------------------------- QK RMSNorm + RoPE (in place on qkv)
template <int DH, bool BWD>
__global__ __launch_bounds__(256) void qk_norm_rope_kernel(bf16_t* __restrict__ qkv, bf16_t* __restrict__ saved,
                                                           const float* __restrict__ rope,
                                                           const bf16_t* __restrict__ wq_txt, const bf16_t* __restrict__ wk_txt,
                                                           const bf16_t* __restrict__ wq_img, const bf16_t* __restrict__ wk_img,
                                                           int B, int S, int T, int H, float eps, int flags, int64_t rope_bs) {
  constexpr int LPI = DH / 8;        // lanes per (token, q|k, head) item
  constexpr int IPB = 256 / LPI;     // items per block
  const int sub = threadIdx.x % LPI;
  // item -> (token, q | k, head) in 32-bit arithmetic with TWO divisions (the launcher checks B * S * 2 * H < 2^31): the 64-bit
  // divisions and remainders of rounds 1-3 were most of this kernel's VALU work (VALUBusy 52 % for an HBM-bound pass)
  const unsigned item = blockIdx.x * IPB + threadIdx.x / LPI;
  const unsigned nitems = (unsigned)B * S * 2 * H;
  const bool valid = item < nitems;
  const unsigned it = valid ? item : nitems - 1;
  const unsigned token = it / (unsigned)(2 * H);
  const int rem = (int)(it - token * (unsigned)(2 * H));
  const int which = rem >= H ? 1 : 0, h = rem - which * H;
  const unsigned bidx = token / (unsigned)S;
  const int s = (int)(token - bidx * (unsigned)S);
  const int Dm = H * DH;
  bf16_t* px = qkv + (int64_t)token * 3 * Dm + which * Dm + h * DH + sub * 8;
  bf16_t* ps = saved ? saved + (int64_t)token * 2 * Dm + which * Dm + h * DH + sub * 8 : nullptr;
  const bf16_t* wsel = (s < T) ? (which ? wk_txt : wq_txt) : (which ? wk_img : wq_img);
  float w[8], cs[8];
  ld8(wsel + sub * 8, w);
  {
    const float* rp = rope + (int64_t)bidx * rope_bs + ((int64_t)s * (DH / 2) + sub * 4) * 2;
    const f32x4 c0 = *(const f32x4*)(rp);
    const f32x4 c1 = *(const f32x4*)(rp + 4);
    cs[0] = c0[0]; cs[1] = c0[1]; cs[2] = c0[2]; cs[3] = c0[3];
    cs[4] = c1[0]; cs[5] = c1[1]; cs[6] = c1[2]; cs[7] = c1[3];
  }
  if constexpr (!BWD) {
    float x[8];
    if (flags & 2) {                 // out of place: the producer wrote the pre-norm q,k straight into `saved`
      ld8(ps, x);
    } else {
      ld8(px, x);
      if (ps && valid) st8(ps, x);
    }
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) ss += x[i] * x[i];
#pragma unroll
    for (int o = 1; o < LPI; o <<= 1) ss += __shfl_xor(ss, o);
    const float rstd = rsqrtf(ss / (float)DH + eps);
    float o8[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float t0 = (flags & 1) ? rbf(x[2 * j] * rstd * w[2 * j]) : rbf(rbf(x[2 * j] * rstd) * w[2 * j]);
      const float t1 = (flags & 1) ? rbf(x[2 * j + 1] * rstd * w[2 * j + 1]) : rbf(rbf(x[2 * j + 1] * rstd) * w[2 * j + 1]);
      const float c = cs[2 * j], sn = cs[2 * j + 1];
      o8[2 * j] = t0 * c - t1 * sn;
      o8[2 * j + 1] = t0 * sn + t1 * c;
    }
    if (valid) st8(px, o8);
  } else {
    float dy[8], x[8];
    ld8(px, dy);
    ld8(ps, x);
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) ss += x[i] * x[i];
#pragma unroll
    for (int o = 1; o < LPI; o <<= 1) ss += __shfl_xor(ss, o);
    const float rstd = rsqrtf(ss / (float)DH + eps);
    float dn[8], xh[8];
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float c = cs[2 * j], sn = cs[2 * j + 1];
      const float d0 = rbf(dy[2 * j] * c + dy[2 * j + 1] * sn);     // dy * conj(f)
      const float d1 = rbf(-dy[2 * j] * sn + dy[2 * j + 1] * c);
      dn[2 * j] = (flags & 1) ? d0 * w[2 * j] : rbf(d0 * w[2 * j]);
      dn[2 * j + 1] = (flags & 1) ? d1 * w[2 * j + 1] : rbf(d1 * w[2 * j + 1]);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { xh[i] = x[i] * rstd; dot += dn[i] * xh[i]; }
#pragma unroll
    for (int o = 1; o < LPI; o <<= 1) dot += __shfl_xor(dot, o);
    dot /= (float)DH;
    float o8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o8[i] = (dn[i] - xh[i] * dot) * rstd;
    if (valid) st8(px, o8);
  }
}

// ---------------------------------------------------------------- [B,S,(H,dh)] -> [B,H,dh,S_pad]
__global__ __launch_bounds__(256) void transpose_heads_kernel(const bf16_t* __restrict__ in, int64_t ld_in,
                                                              bf16_t* __restrict__ out, int S, int S_pad, int HD) {
  __shared__ bf16_t tile[64][66];
  const int s0 = blockIdx.x * 64, c0 = blockIdx.y * 64, b = blockIdx.z;
  const int tid = threadIdx.x;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int e = tid + it * 256;  // 512 chunks of 8
    const int r = e >> 3, ch = e & 7;
    const int s = s0 + r;
    u32x4 u = {0u, 0u, 0u, 0u};
    if (s < S) u = *(const u32x4*)(in + ((int64_t)b * S + s) * ld_in + c0 + ch * 8);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      tile[r][ch * 8 + 2 * i] = (bf16_t)(u[i] & 0xffffu);
      tile[r][ch * 8 + 2 * i + 1] = (bf16_t)(u[i] >> 16);
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int e = tid + it * 256;
    const int c = e >> 3, ch = e & 7;  // out row = column c, 8 tokens ch*8..
    u32x4 u;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      u[i] = (uint32_t)tile[ch * 8 + 2 * i][c] | ((uint32_t)tile[ch * 8 + 2 * i + 1][c] << 16);
    *(u32x4*)(out + ((int64_t)b * HD + c0 + c) * S_pad + s0 + ch * 8) = u;
  }
}

// ---------------------------------------------------------------- criterion (weighted-MSE with weight 1)
__global__ __launch_bounds__(256) void mse_kernel(const bf16_t* __restrict__ pred, const bf16_t* __restrict__ target,
                                                  float* __restrict__ loss, bf16_t* __restrict__ dpred, int B, int S_all,
                                                  int S_t, int C, float gscale) {
  __shared__ float red[4];
  const int64_t total = (int64_t)B * S_all * C;
  const float inv = 1.0f / ((float)B * (float)S_t * (float)C);
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int64_t t = i / C;
    const int s = (int)(t % S_all), b = (int)(t / S_all);
    float g = 0.f;
    if (s < S_t) {
      const float d = bf2f(pred[i]) - bf2f(target[((int64_t)b * S_t + s) * C + c]);
      acc += d * d;
      g = 2.0f * d * inv * gscale;
    }
    if (dpred) dpred[i] = f2bf(g);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(loss, (red[0] + red[1] + red[2] + red[3]) * inv);
}

// C % 8 == 0: eight consecutive elements (one 16-byte access each way) per thread and few blocks -- every block ends in ONE atomic
// on the same fp32 scalar, and same-address atomics serialise (~150 ns each: the 512-block scalar form took 78 us for 0.7 MB).
__global__ __launch_bounds__(256) void mse_kernel8(const bf16_t* __restrict__ pred, const bf16_t* __restrict__ target,
                                                   float* __restrict__ loss, bf16_t* __restrict__ dpred, int B, int S_all,
                                                   int S_t, int C, float gscale) {
  __shared__ float red[4];
  const int64_t total8 = (int64_t)B * S_all * C / 8;
  const float inv = 1.0f / ((float)B * (float)S_t * (float)C);
  const int c8 = C / 8;
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total8; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % c8) * 8;
    const int64_t t = i / c8;
    const int s = (int)(t % S_all), b = (int)(t / S_all);
    float g[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (s < S_t) {
      float pv[8], tv[8];
      ld8(pred + i * 8, pv);
      ld8(target + ((int64_t)b * S_t + s) * C + c, tv);
#pragma unroll
      for (int k = 0; k < 8; ++k) { const float d = pv[k] - tv[k]; acc += d * d; g[k] = 2.0f * d * inv * gscale; }
    }
    if (dpred) st8(dpred + i * 8, g);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(loss, (red[0] + red[1] + red[2] + red[3]) * inv);
}

__global__ __launch_bounds__(256) void mse_tw_kernel(const bf16_t* __restrict__ pred, const bf16_t* __restrict__ target,
                                                     const float* __restrict__ tw, float* __restrict__ loss,
                                                     bf16_t* __restrict__ dpred, int B, int S_all, int S_t, int C, float inv_denom,
                                                     float gscale) {
  __shared__ float red[4];
  const int64_t total = (int64_t)B * S_all * C;
  const float invc = 1.0f / (float)C;
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int64_t t = i / C;
    const int s = (int)(t % S_all), b = (int)(t / S_all);
    float g = 0.f;
    if (s < S_t) {
      const float w = tw[(int64_t)b * S_t + s];
      const float d = bf2f(pred[i]) - bf2f(target[((int64_t)b * S_t + s) * C + c]);
      acc += w * d * d * invc;
      g = 2.0f * w * d * invc * inv_denom * gscale;
    }
    if (dpred) dpred[i] = f2bf(g);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(loss, (red[0] + red[1] + red[2] + red[3]) * inv_denom);
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ out) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) acc += g[i] * g[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

// Deterministic form: per-block partial sums in a fixed slot each, folded by ONE block in a fixed order.  Data-parallel replicas
// hold bit-identical gradients after the all-reduce; with the atomic form above the clip coefficient differed in its last bits
// from rank to rank (fp32 atomics commute only approximately) and the replicas drifted apart by ~1e-10 per step.
__global__ __launch_bounds__(256) void sumsq_part_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ part) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) acc += g[i] * g[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void sumsq_fold_kernel(const float* __restrict__ part, int nb, float* __restrict__ out) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) acc += part[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
                                                    float wd, float bc1, float bc2, const float* __restrict__ gnorm_sq,
                                                    float max_norm, float grad_scale) {
  float clip = grad_scale;
  if (gnorm_sq != nullptr && max_norm > 0.f) {
    const float nrm = sqrtf(*gnorm_sq) * grad_scale;
    const float c = max_norm / (nrm + 1e-6f);
    clip *= c < 1.0f ? c : 1.0f;
  }
  const float step = lr / bc1;
  const float rs2 = 1.0f / sqrtf(bc2);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * clip;
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    pi -= step * mi / (sqrtf(vi) * rs2 + eps);
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}

// ---- Prodigy (prodigyopt 1.x, Adam variant) over the flat LoRA buffers: see include/qfx.h.  Host scalars of the package (Python
// float64: d, d_max, d_numerator, d_denom, k) live in a device double[QFX_PRODIGY_STATE] so the step never synchronises.
enum { PS_D = 0, PS_DMAX, PS_NUM, PS_DEN, PS_DHAT, PS_K, PS_ACC_NUM, PS_ACC_DEN, PS_DLR, PS_SKIP };

__global__ void prodigy_begin_kernel(double* __restrict__ st, double lr, double b1, double b2, int use_bc) {
  const double k = st[PS_K];
  const double bc = use_bc ? sqrt(1.0 - pow(b2, k + 1.0)) / (1.0 - pow(b1, k + 1.0)) : 1.0;
  st[PS_DLR] = st[PS_D] * lr * bc;
  st[PS_ACC_NUM] = 0.0;
  st[PS_ACC_DEN] = 0.0;
  st[PS_SKIP] = 0.0;
}

__global__ __launch_bounds__(256) void prodigy_ema_kernel(const float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                          float* __restrict__ v, float* __restrict__ sv, const float* __restrict__ p0,
                                                          int64_t n, double* __restrict__ st, float b1, float b2, float b3, double d0,
                                                          int safeguard, const float* __restrict__ gnorm_sq, float max_norm,
                                                          float grad_scale) {
  __shared__ float red[8];
  float clip = grad_scale;
  if (gnorm_sq != nullptr && max_norm > 0.f) {
    const float nrm = sqrtf(*gnorm_sq) * grad_scale;
    const float c = max_norm / (nrm + 1e-6f);
    clip *= c < 1.0f ? c : 1.0f;
  }
  const double d = st[PS_D], dlr = st[PS_DLR];
  const float am = (float)(d * (1.0 - (double)b1)), av = (float)(d * d * (1.0 - (double)b2));
  const float as = (float)(safeguard ? (d / d0) * d : (d / d0) * dlr);
  float num = 0.f, den = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * clip;
    num += gi * (p0[i] - p[i]);
    m[i] = m[i] * b1 + gi * am;
    v[i] = v[i] * b2 + (av * gi) * gi;
    const float si = sv[i] * b3 + gi * as;
    sv[i] = si;
    den += fabsf(si);
  }
  num = wave_sum(num); den = wave_sum(den);
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = num; red[4 + (threadIdx.x >> 6)] = den; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&st[PS_ACC_NUM], (double)red[0] + (double)red[1] + (double)red[2] + (double)red[3]);
    atomicAdd(&st[PS_ACC_DEN], (double)red[4] + (double)red[5] + (double)red[6] + (double)red[7]);
  }
}

__global__ void prodigy_d_kernel(double* __restrict__ st, double b3, double d0, double d_coef, double growth) {
  const double den = st[PS_ACC_DEN];
  if (den == 0.0) { st[PS_SKIP] = 1.0; return; }   // no progress: the package returns before storing anything
  double d = st[PS_D];
  const double num = st[PS_NUM] * b3 + (d / d0) * st[PS_DLR] * st[PS_ACC_NUM];
  const double d_hat = d_coef * num / den;
  if (d == d0) d = d > d_hat ? d : d_hat;
  double d_max = st[PS_DMAX];
  d_max = d_max > d_hat ? d_max : d_hat;
  const double dg = d * growth;
  d = d_max < dg ? d_max : dg;
  st[PS_NUM] = num; st[PS_DEN] = den; st[PS_D] = d; st[PS_DMAX] = d_max; st[PS_DHAT] = d_hat;
  st[PS_K] += 1.0;
}

__global__ __launch_bounds__(256) void prodigy_apply_kernel(float* __restrict__ p, const float* __restrict__ m, const float* __restrict__ v,
                                                            int64_t n, const double* __restrict__ st, float eps, float wd) {
  if (st[PS_SKIP] != 0.0) return;
  const double dlr = st[PS_DLR];
  const float deps = (float)(st[PS_D] * (double)eps);
  const float adec = (float)(-(double)wd * dlr), astep = (float)(-dlr);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float pi = p[i];
    if (wd != 0.f) pi = pi + pi * adec;
    p[i] = pi + astep * (m[i] / (sqrtf(v[i]) + deps));
  }
}

// ---- runtime helpers: CU-masked side stream + a probe of where blocks run --------------------------------------------------
__global__ void where_kernel(uint32_t* __restrict__ out) {
  if (threadIdx.x == 0) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = xcc;
  }
  // stay resident long enough that concurrently launched blocks have to spread over every allowed CU
  const uint64_t t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < 200000) {}
}

__global__ __launch_bounds__(256) void timestep_embed_kernel(const float* __restrict__ t, int B, int dim, float scale,
                                                             float pre_scale, bf16_t* __restrict__ out) {
  const int half = dim / 2;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * half; i += gridDim.x * blockDim.x) {
    const int b = i / half, j = i % half;
    const float tv = pre_scale != 1.0f ? rbf(rbf(t[b]) * pre_scale) : rbf(t[b]);
    const float f = expf(-9.210340371976184f * (float)j / (float)half);  // ln(10000)
    const float e = scale * (tv * f);
    out[b * dim + j] = f2bf(cosf(e));
    out[b * dim + half + j] = f2bf(sinf(e));
  }
}

__global__ __launch_bounds__(256) void flowmatch_prepare_kernel(const bf16_t* __restrict__ x0, const bf16_t* __restrict__ noise,
                                                                const bf16_t* __restrict__ ctrl, const bf16_t* __restrict__ sigma,
                                                                bf16_t* __restrict__ packed, bf16_t* __restrict__ target,
                                                                int B, int S_t, int S_c, int C, int mode) {
  const int64_t total = (int64_t)B * (S_t + S_c) * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int64_t r = i / C;
    const int s = (int)(r % (S_t + S_c)), b = (int)(r / (S_t + S_c));
    if (s < S_t) {
      const int64_t j = ((int64_t)b * S_t + s) * C + c;
      const float sg = bf2f(sigma[b]), n = bf2f(noise[j]);
      if (mode == 0) {
        const float a = bf2f(x0[j]);
        packed[i] = f2bf(rbf(rbf(1.0f - sg) * a) + rbf(sg * n));
        target[j] = f2bf(n - a);
      } else {   // x0 is fp16 (cache dtype): bf16 * fp16 promotes to fp32 in the reference
        const float a = __half2float(__ushort_as_half(x0[j]));
        packed[i] = f2bf(rbf(1.0f - sg) * a + rbf(sg * n));
        target[j] = f2bf(n - rbf(a));
      }
    } else {
      packed[i] = ctrl[((int64_t)b * S_c + (s - S_t)) * C + c];
    }
  }
}

__global__ __launch_bounds__(256) void add3_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b,
                                                   const bf16_t* __restrict__ c, bf16_t* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float v = rbf(bf2f(a[i]) + bf2f(b[i]));
    if (c) v = v + bf2f(c[i]);
    out[i] = f2bf(v);
  }
}

}  // namespace

extern "C" int qfx_add3_bf16(const uint16_t* a, const uint16_t* b, const uint16_t* c, uint16_t* out, int64_t n, void* stream) {
  if (!a || !b || !out || n <= 0) return QFX_EINVAL;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(add3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, b, c, out, n);
  QFX_CHECK_LAUNCH();
  return QFX_OK;
}

extern "C" int qfx_timestep_embed(const float* t, int32_t B, int32_t dim, float scale, float pre_scale, uint16_t* out, void* stream) {
  if (!t || !out || B <= 0 || dim <= 0 || (dim % 2)) return QFX_EINVAL;
  hipLaunchKernelGGL(timestep_embed_kernel, dim3((B * dim / 2 + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, B, dim, scale, pre_scale, out);
  QFX_CHECK_LAUNCH();
  return QFX_OK;
}

extern "C" int qfx_flowmatch_prepare(const uint16_t* x0, const uint16_t* noise, const uint16_t* ctrl, const uint16_t* sigma,
                                     uint16_t* packed, uint16_t* target, int32_t B, int32_t S_t, int32_t S_c, int32_t C,
                                     int32_t mode, void* stream) {
  if (!x0 || !noise || !sigma || !packed || !target || B <= 0 || S_t <= 0 || S_c < 0 || C <= 0 || (S_c > 0 && !ctrl)) return QFX_EINVAL;
  const int64_t total = (int64_t)B * (S_t + S_c) * C;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(flowmatch_prepare_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x0, noise, ctrl, sigma, packed,
                     target, B, S_t, S_c, C, mode);
  QFX_CHECK_LAUNCH();
  return QFX_OK;
}

extern "C" int qfx_ln_modulate_fwd_batch(const qfx_ln_fwd_args* list, int32_t n, void* stream) {
  if (!list || n <= 0 || n > QFX_MAX_LN_BATCH) return QFX_EINVAL;
  LnFwdBatch bt;
  int rows = 0;
  for (int i = 0; i < n; ++i) {
    const qfx_ln_fwd_args& a = list[i];
    if (!a.x || !a.shift || !a.scale || !a.y || a.rows <= 0 || a.D <= 0 || (a.D % 8) || a.D > MAXP * 512 || a.rows_per_batch <= 0 ||
        (a.mod_bstride % 8))
      return QFX_EINVAL;
    if (a.yq && (!a.ys || (a.D % 128) || (a.ldyq % 8) || a.ldyq < a.D || a.ys_rows < a.rows)) return QFX_EINVAL;
    bt.a[i] = a;
    rows += (a.rows + 3) / 4 * 4;     // problems start on a block boundary (4 rows per block)
    if (i + 1 < n && (a.rows % 4)) return QFX_EINVAL;   /* only the last problem may have a ragged row count */
  }
  for (int i = n; i < QFX_MAX_LN_BATCH; ++i) bt.a[i] = list[0];
  bt.n = n;
  hipLaunchKernelGGL(ln_mod_fwd_kernel, dim3(rows / 4), dim3(256), 0, (hipStream_t)stream, bt);
  QFX_CHECK_LAUNCH();
  return QFX_OK;
}

extern "C" int qfx_ln_modulate_fwd(const uint16_t* x, const uint16_t* shift, const uint16_t* scale, int64_t mod_bstride,
                                   uint16_t* y, int32_t rows, int32_t D, int32_t rows_per_batch, float eps, void* stream) {
  qfx_ln_fwd_args a = {};
  a.x = x; a.shift = shift; a.scale = scale; a.mod_bstride = mod_bstride; a.y = y; a.rows = rows; a.D = D; a.rows_per_batch = rows_per_batch;
  a.eps = eps;
  return qfx_ln_modulate_fwd_batch(&a, 1, stream);
}

extern "C" int qfx_ln_modulate_bwd_batch(const qfx_ln_bwd_args* list, int32_t n, void* stream) {
  if (!list || n <= 0 || n > QFX_MAX_LN_BATCH) return QFX_EINVAL;
  LnBwdBatch bt;
  int rows = 0;
  for (int i = 0; i < n; ++i) {
    const qfx_ln_bwd_args& a = list[i];
    if (!a.dy || !a.x || !a.scale || !a.dx || a.rows <= 0 || a.D <= 0 || (a.D % 8) || a.D > MAXP * 512 || a.rows_per_batch <= 0 ||
        (a.mod_bstride % 8))
      return QFX_EINVAL;
    if (a.dyg && (!a.gate || (a.gate_bstride % 8))) return QFX_EINVAL;
    if (a.dygq && (!a.dyg || !a.dygs || (a.D % 128) || (a.lddygq % 8) || a.lddygq < a.D || a.dygs_rows < a.rows)) return QFX_EINVAL;
    bt.a[i] = a;
    rows += (a.rows + 3) / 4 * 4;
    if (i + 1 < n && (a.rows % 4)) return QFX_EINVAL;
  }
  for (int i = n; i < QFX_MAX_LN_BATCH; ++i) bt.a[i] = list[0];
  bt.n = n;
  int dmax = 0;
  for (int i = 0; i < n; ++i) dmax = list[i].D > dmax ? list[i].D : dmax;
#if defined(QFX_LN_BWD_PIPE)
  bool plain = dmax <= 3072;
  for (int i = 0; i < n; ++i) plain = plain && list[i].dygq == nullptr;
  if (plain) {
    const int nblk = rows / 4;
    const int grid = nblk < QFX_LN_BWD_PIPE ? nblk : QFX_LN_BWD_PIPE;      // QFX_LN_BWD_PIPE = persistent blocks (256 = one per CU)
    hipLaunchKernelGGL(ln_mod_bwd_pipe_kernel<6>, dim3(grid), dim3(256), 0, (hipStream_t)stream, bt, nblk);
    QFX_CHECK_LAUNCH();
    return QFX_OK;
  }
#endif
  if (dmax <= 3072) hipLaunchKernelGGL(ln_mod_bwd_kernel<6>, dim3(rows / 4), dim3(256), 0, (hipStream_t)stream, bt);
  else hipLaunchKernelGGL(ln_mod_bwd_kernel<MAXP>, dim3(rows / 4), dim3(256), 0, (hipStream_t)stream, bt);
  QFX_CHECK_LAUNCH();
  return QFX_OK;
}

extern "C" int qfx_ln_modulate_bwd(const uint16_t* dy, const uint16_t* x, const uint16_t* scale, int64_t mod_bstride,
                                   const uint16_t* dres, const uint16_t* gate, int64_t gate_bstride, uint16_t* dx,
                                   uint16_t* dyg, int32_t rows, int32_t D, int32_t rows_per_batch, float eps, const float* row_mask,
                                   void* stream) {
  qfx_ln_bwd_args a = {};
  a.dy = dy; a.x = x; a.scale = scale; a.mod_bstride = mod_bstride; a.dres = dres; a.gate = gate; a.gate_bstride = gate_bstride;
  a.dx = dx; a.dyg = dyg; a.rows = rows; a.D = D; a.rows_per_batch = rows_per_batch; a.eps = eps; a.row_mask = row_mask;
  return qfx_ln_modulate_bwd_batch(&a, 1, stream);
}

extern "C" int qfx_mod_grad_batch(const qfx_mod_grad_args* list, int32_t n, void* stream) {
  if (!list || n <= 0 || n > QFX_MAX_LN_BATCH) return QFX_EINVAL;
  ModGradBatch bt;
  int blocks = 0, Bmax = 0;
  for (int i = 0; i < n; ++i) {
    const qfx_mod_grad_args* a = &list[i];
    if (!a->dy || !a->x || !a->dshift || !a->dscale) return QFX_EINVAL;
    if (a->rows <= 0 || a->D <= 0 || (a->D % 8) || a->D > MAXP * 512 || a->rows_per_batch <= 0) return QFX_EINVAL;
    if ((a->ld_dy % 8) || (a->ld_x % 8)) return QFX_EINVAL;
    if ((a->dgate != nullptr) != (a->dxo != nullptr) || (a->dgate != nullptr) != (a->y != nullptr)) return QFX_EINVAL;
    if (a->dgate && ((a->ld_dxo % 8) || (a->ld_y % 8))) return QFX_EINVAL;
    if ((a->D + 511) / 512 != (list[0].D + 511) / 512) return QFX_EINVAL;      /* one register-array size per launch */
    bt.a[i] = *a;
    bt.start[i] = blocks;
    blocks += (a->rows_per_batch + 4 * MG_RPW - 1) / (4 * MG_RPW);
    const int B = (a->rows + a->rows_per_batch - 1) / a->rows_per_batch;
    Bmax = B > Bmax ? B : Bmax;
  }
  for (int i = n; i <= QFX_MAX_LN_BATCH; ++i) bt.start[i] = blocks;
  for (int i = n; i < QFX_MAX_LN_BATCH; ++i) bt.a[i] = list[0];
  bt.n = n;
  dim3 grid(blocks, Bmax);
  hipStream_t s = (hipStream_t)stream;
  switch ((list[0].D + 511) / 512) {
    case 1: hipLaunchKernelGGL(mod_grad_kernel<1>, grid, dim3(256), 0, s, bt); break;
    case 2: hipLaunchKernelGGL(mod_grad_kernel<2>, grid, dim3(256), 0, s, bt); break;
    case 3: hipLaunchKernelGGL(mod_grad_kernel<3>, grid, dim3(256), 0, s, bt); break;
    case 4: hipLaunchKernelGGL(mod_grad_kernel<4>, grid, dim3(256), 0, s, bt); break;
    case 5: case 6: hipLaunchKernelGGL(mod_grad_kernel<6>, grid, dim3(256), 0, s, bt); break;
    default: hipLaunchKernelGGL(mod_grad_kernel<8>, grid, dim3(256), 0, s, bt); break;
  }
  QFX_CHECK_LAUNCH();
  return QFX_OK;
}

extern "C" int qfx_mod_grad(const qfx_mod_grad_args* a, void* stream) {
  if (!a) return QFX_EINVAL;
  return qfx_mod_grad_batch(a, 1, stream);
}

extern "C" int qfx_gate_mul(const uint16_t* dx, const uint16_t* gate, int64_t gate_bstride, uint16_t* dyg, int32_t rows,
                            int32_t D, int32_t rows_per_batch, void* stream) {
  if (!dx || !gate || !dyg || rows <= 0 || D <= 0 || (D % 8) || (gate_bstride % 8) || rows_per_batch <= 0) return QFX_EINVAL;
  const int64_t total8 = (int64_t)rows * D / 8;
  int blocks = (int)((total8 + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(gate_mul_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dx, gate, gate_bstride, dyg, total8, D,
                     rows_per_batch);
  QFX_CHECK_LAUNCH();
  return QFX_OK;
}

extern "C" int qfx_rmsnorm_fwd(const uint16_t* x, const uint16_t* w, uint16_t* y, int32_t rows, int32_t D, float eps,
                               void* stream) {
  if (!x || !w || !y || rows <= 0 || D <= 0 || (D % 8) || D > MAXP * 512) return QFX_EINVAL;
  hipLaunchKernelGGL(rmsnorm_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, w, y, rows, D, eps);
  QFX_CHECK_LAUNCH();
  return QFX_OK;
}

extern "C" int qfx_mod_gemv(const uint16_t* temb, int32_t B, int32_t K, const uint16_t* const* W, const uint16_t* const* bias,
                            int32_t nmat, int32_t N, int32_t apply_silu, uint16_t* out, void* stream) {
  if (!temb || !W || !out || B <= 0 || B > 8 || K <= 0 || (K % 8) || nmat <= 0 || N <= 0) return QFX_EINVAL;
  const size_t lds = (size_t)B * K * 2;
  if (lds > 64 * 1024) return QFX_EUNSUPPORTED;
  hipLaunchKernelGGL(mod_gemv_kernel, dim3((N + 15) / 16, nmat), dim3(256), lds, (hipStream_t)stream, temb, B, K, W, bias, N,
                     apply_silu, out);
  QFX_CHECK_LAUNCH();
  return QFX_OK;
}

extern "C" int qfx_mod_gemv_t(const uint16_t* dy, int32_t B, int32_t N, int32_t K, const uint16_t* const* W, int32_t nmat,
                              float* out, void* stream) {
  if (!dy || !W || !out || B <= 0 || B > 8 || N <= 0 || K <= 0 || (K % 8) || K > 3072 || nmat <= 0) return QFX_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int64_t items = (int64_t)nmat * ((N + 3) / 4);
  int grid = (int)((items + 3) / 4);
  if (grid > 4 * 256) grid = 4 * 256;   // 4 blocks per CU (256 CUs)
  const int np = (K + 511) / 512;
  for (int b0 = 0; b0 < B; b0 += 2) {   // two samples' partial sums per pass over the weights (registers); B <= 2 is one pass
    const bool two = b0 + 1 < B;
#define QFX_GT(NB_, NP_) hipLaunchKernelGGL((mod_gemv_t_kernel<NB_, NP_>), dim3(grid), dim3(256), 0, s, dy, B, b0, N, K, W, nmat, out)
    if (np <= 2) { if (two) QFX_GT(2, 2); else QFX_GT(1, 2); }
    else if (np <= 4) { if (two) QFX_GT(2, 4); else QFX_GT(1, 4); }
    else { if (two) QFX_GT(2, 6); else QFX_GT(1, 6); }
#undef QFX_GT
  }
  QFX_CHECK_LAUNCH();
  return QFX_OK;
}

static int launch_qk(bool bwd, uint16_t* qkv, uint16_t* saved, const float* rope, const uint16_t* wq_txt,
                     const uint16_t* wk_txt, const uint16_t* wq_img, const uint16_t* wk_img, int32_t B, int32_t S,
                     int32_t T, int32_t H, int32_t dh, float eps, int32_t flags, int64_t rope_bs, void* stream) {
  if (!qkv || !rope || !wq_txt || !wk_txt || !wq_img || !wk_img || B <= 0 || S <= 0 || T < 0 || T > S || H <= 0) return QFX_EINVAL;
  if (bwd && !saved) return QFX_EINVAL;
  const int64_t nitems = (int64_t)B * S * 2 * H;
  if (nitems >= (1LL << 31) - 64) return QFX_EUNSUPPORTED;      // the kernel indexes items in 32 bits
  hipStream_t s = (hipStream_t)stream;
  if (dh == 128) {
    const int ipb = 256 / 16;
    dim3 grid((unsigned)((nitems + ipb - 1) / ipb));
    if (bwd) hipLaunchKernelGGL((qk_norm_rope_kernel<128, true>), grid, dim3(256), 0, s, qkv, saved, rope, wq_txt, wk_txt, wq_img, wk_img, B, S, T, H, eps, flags, rope_bs);
    else hipLaunchKernelGGL((qk_norm_rope_kernel<128, false>), grid, dim3(256), 0, s, qkv, saved, rope, wq_txt, wk_txt, wq_img, wk_img, B, S, T, H, eps, flags, rope_bs);
  } else if (dh == 64) {
    const int ipb = 256 / 8;
    dim3 grid((unsigned)((nitems + ipb - 1) / ipb));
    if (bwd) hipLaunchKernelGGL((qk_norm_rope_kernel<64, true>), grid, dim3(256), 0, s, qkv, saved, rope, wq_txt, wk_txt, wq_img, wk_img, B, S, T, H, eps, flags, rope_bs);
    else hipLaunchKernelGGL((qk_norm_rope_kernel<64, false>), grid, dim3(256), 0, s, qkv, saved, rope, wq_txt, wk_txt, wq_img, wk_img, B, S, T, H, eps, flags, rope_bs);
  } else {
    return QFX_EUNSUPPORTED;
  }
  QFX_CHECK_LAUNCH();
  return QFX_OK;
}

extern "C" int qfx_qk_norm_rope_fwd(uint16_t* qkv, uint16_t* saved, const float* rope, const uint16_t* wq_txt,
                                    const uint16_t* wk_txt, const uint16_t* wq_img, const uint16_t* wk_img, int32_t B,
                                    int32_t S, int32_t T, int32_t H, int32_t dh, float eps, int32_t flags, int64_t rope_bstride,
                                    void* stream) {
  if ((flags & 2) && !saved) return QFX_EINVAL;
  return launch_qk(false, qkv, saved, rope, wq_txt, wk_txt, wq_img, wk_img, B, S, T, H, dh, eps, flags, rope_bstride, stream);
}
extern "C" int qfx_qk_norm_rope_bwd(uint16_t* dqkv, const uint16_t* saved, const float* rope, const uint16_t* wq_txt,
                                    const uint16_t* wk_txt, const uint16_t* wq_img, const uint16_t* wk_img, int32_t B,
                                    int32_t S, int32_t T, int32_t H, int32_t dh, float eps, int32_t flags, int64_t rope_bstride,
                                    void* stream) {
  return launch_qk(true, dqkv, (uint16_t*)saved, rope, wq_txt, wk_txt, wq_img, wk_img, B, S, T, H, dh, eps, flags, rope_bstride, stream);
}

extern "C" int qfx_transpose_heads(const uint16_t* in, int64_t ld_in, uint16_t* out, int32_t B, int32_t S, int32_t S_pad,
                                   int32_t H, int32_t dh, void* stream) {
  if (!in || !out || B <= 0 || S <= 0 || S_pad < S || (S_pad % 64) || ((H * dh) % 64) || (ld_in % 8)) return QFX_EINVAL;
  hipLaunchKernelGGL(transpose_heads_kernel, dim3(S_pad / 64, H * dh / 64, B), dim3(256), 0, (hipStream_t)stream, in, ld_in,
                     out, S, S_pad, H * dh);
  QFX_CHECK_LAUNCH();
  return QFX_OK;
}

extern "C" int qfx_mse_loss_fwd_bwd(const uint16_t* pred, const uint16_t* target, float* loss, uint16_t* dpred, int32_t B,
                                    int32_t S_all, int32_t S_t, int32_t C, float gscale, void* stream) {
  if (!pred || !target || !loss || B <= 0 || S_all <= 0 || S_t <= 0 || S_t > S_all || C <= 0) return QFX_EINVAL;
  const int64_t total = (int64_t)B * S_all * C;
  // the 8-wide form issues 16-byte accesses on all three tensors: offset views (a row- or column-sliced target) keep the scalar form
  const bool al16 = (((uintptr_t)pred | (uintptr_t)target | (uintptr_t)dpred) & 15) == 0;
  if (C % 8 == 0 && al16) {
    int blocks = (int)((total / 8 + 255) / 256);
    if (blocks > 128) blocks = 128;
    hipLaunchKernelGGL(mse_kernel8, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pred, target, loss, dpred, B, S_all, S_t, C, gscale);
    QFX_CHECK_LAUNCH();
    return QFX_OK;
  }
  int blocks = (int)((total + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(mse_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pred, target, loss, dpred, B, S_all, S_t, C,
                     gscale);
  QFX_CHECK_LAUNCH();
  return QFX_OK;
}

extern "C" int qfx_mse_token_weighted_fwd_bwd(const uint16_t* pred, const uint16_t* target, const float* token_w, float* loss,
                                              uint16_t* dpred, int32_t B, int32_t S_all, int32_t S_t, int32_t C, float inv_denom,
                                              float gscale, void* stream) {
  if (!pred || !target || !token_w || !loss || B <= 0 || S_all <= 0 || S_t <= 0 || S_t > S_all || C <= 0) return QFX_EINVAL;
  const int64_t total = (int64_t)B * S_all * C;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(mse_tw_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pred, target, token_w, loss, dpred, B, S_all,
                     S_t, C, inv_denom, gscale);
  QFX_CHECK_LAUNCH();
  return QFX_OK;
}

extern "C" int qfx_sumsq(const float* g, int64_t n, float* out, void* stream) {
  if (!g || !out || n <= 0) return QFX_EINVAL;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, n, out);
  QFX_CHECK_LAUNCH();
  return QFX_OK;
}

extern "C" int qfx_sumsq_det(const float* g, int64_t n, float* out, float* partials, int32_t nslots, void* stream) {
  if (!g || !out || !partials || n <= 0 || nslots <= 0) return QFX_EINVAL;
  int64_t blocks = (n + 255) / 256;
  if (blocks > nslots) blocks = nslots;
  hipLaunchKernelGGL(sumsq_part_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, n, partials);
  hipLaunchKernelGGL(sumsq_fold_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, (int)blocks, out);
  QFX_CHECK_LAUNCH();
  return QFX_OK;
}

extern "C" int qfx_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                              float eps, float weight_decay, float bias_corr1, float bias_corr2, const float* gnorm_sq,
                              float max_norm, float grad_scale, void* stream) {
  if (!p || !g || !m || !v || n <= 0) return QFX_EINVAL;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps,
                     weight_decay, bias_corr1, bias_corr2, gnorm_sq, max_norm, grad_scale);
  QFX_CHECK_LAUNCH();
  return QFX_OK;
}

extern "C" int qfx_prodigy_init_state(double* state, double d0, void* stream) {
  if (!state || !(d0 > 0.0)) return QFX_EINVAL;
  double h[QFX_PRODIGY_STATE] = {0};
  h[PS_D] = d0; h[PS_DMAX] = d0; h[PS_DHAT] = d0;
  if (hipMemcpyAsync(state, h, sizeof(h), hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) return QFX_EINVAL;
  return hipStreamSynchronize((hipStream_t)stream) == hipSuccess ? QFX_OK : QFX_EINVAL;   // h is a stack buffer
}

extern "C" int qfx_prodigy_step(const qfx_prodigy_args* a, void* stream) {
  if (!a || !a->p || !a->g || !a->exp_avg || !a->exp_avg_sq || !a->s || !a->p0 || !a->state || a->n <= 0) return QFX_EINVAL;
  if (!(a->beta1 > 0.f) || !(a->d0 > 0.f) || a->lr < 0.f) return QFX_EINVAL;
  if (a->weight_decay != 0.f && !a->decouple) return QFX_EUNSUPPORTED;   // coupled decay: not used by any reference config
  if (a->lr == 0.f) return QFX_OK;   // warm-up step 0: the package creates its state and returns; k does not advance
  hipStream_t s = (hipStream_t)stream;
  const double b3 = a->beta3 > 0.f ? (double)a->beta3 : sqrt((double)a->beta2);
  int blocks = (int)((a->n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(prodigy_begin_kernel, dim3(1), dim3(1), 0, s, a->state, (double)a->lr, (double)a->beta1, (double)a->beta2,
                     a->use_bias_correction);
  hipLaunchKernelGGL(prodigy_ema_kernel, dim3(blocks), dim3(256), 0, s, a->p, a->g, a->exp_avg, a->exp_avg_sq, a->s, a->p0, a->n,
                     a->state, a->beta1, a->beta2, (float)b3, (double)a->d0, a->safeguard_warmup, a->gnorm_sq, a->max_norm,
                     a->grad_scale);
  hipLaunchKernelGGL(prodigy_d_kernel, dim3(1), dim3(1), 0, s, a->state, b3, (double)a->d0, (double)a->d_coef,
                     (double)a->growth_rate);
  hipLaunchKernelGGL(prodigy_apply_kernel, dim3(blocks), dim3(256), 0, s, a->p, a->exp_avg, a->exp_avg_sq, a->n, a->state, a->eps,
                     a->weight_decay);
  QFX_CHECK_LAUNCH();
  return QFX_OK;
}

extern "C" int qfx_stream_create_cu_masked(int32_t n_cus, void** stream_out) {
  if (!stream_out || n_cus <= 0 || n_cus > QFX_NUM_CU_TOTAL) return QFX_EINVAL;
  uint32_t mask[QFX_NUM_CU_TOTAL / 32] = {0};
  for (int i = 0; i < n_cus; ++i) mask[i >> 5] |= 1u << (i & 31);   // driver order: consecutive bits walk the XCDs first
  hipStream_t s = nullptr;
  const hipError_t e = hipExtStreamCreateWithCUMask(&s, QFX_NUM_CU_TOTAL / 32, mask);
  if (e != hipSuccess) return -1000 - (int)e;
  *stream_out = (void*)s;
  return QFX_OK;
}

extern "C" int qfx_stream_destroy(void* stream) {
  if (!stream) return QFX_EINVAL;
  const hipError_t e = hipStreamDestroy((hipStream_t)stream);
  return e == hipSuccess ? QFX_OK : -1000 - (int)e;
}

extern "C" int qfx_debug_where(uint32_t* out, int32_t n_blocks, void* stream) {
  if (!out || n_blocks <= 0) return QFX_EINVAL;
  hipLaunchKernelGGL(where_kernel, dim3(n_blocks), dim3(64), 0, (hipStream_t)stream, out);
  QFX_CHECK_LAUNCH();
  return QFX_OK;
}

extern "C" int qfx_abi_version(void) { return QFX_ABI_VERSION; }
extern "C" const char* qfx_build_arch(void) { return "gfx950"; }
