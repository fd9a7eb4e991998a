// qfx_skinny.hip -- the HBM-bound rank-r LoRA pieces: down projection (MFMA, K split over the
// four waves of a block), weight-gradient outer products (VALU + fp32 atomics) and operand packing.
#include "qfx_common.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace {

// ---------------------------------------------------------------------------------------------
// U[M,R] = X[M,K] (W_hi+W_lo)[R,K]^T, fp32 accumulate.  Block = 32 rows of X, 4 waves interleave
// the K/32 MFMA steps (adjacent waves read adjacent 64 B of each row), LDS reduce at the end.
template <int NF>
__global__ __launch_bounds__(256) void lora_down_kernel(const qfx_lora_down_args p) {
  constexpr int MF = 2;
  __shared__ float red[4][MF * NF * 256];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int m0 = blockIdx.x * (16 * MF);

  const bf16_t* xrow[MF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) {
    int m = m0 + mf * 16 + li;
    m = m < p.M ? m : p.M - 1;
    xrow[mf] = p.X + remap_row(m, p.rows_per_batch, p.x_batch_rows, p.x_row_off) * p.ldx + 8 * g;
  }
  const bf16_t* wh[NF];
  const bf16_t* wl[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    wh[nf] = p.W_hi + (int64_t)(nf * 16 + li) * p.ldw + 8 * g;
    wl[nf] = p.W_lo + (int64_t)(nf * 16 + li) * p.ldw + 8 * g;
  }
  f32x4 acc[MF][NF];
#pragma unroll
  for (int i = 0; i < MF; ++i)
#pragma unroll
    for (int j = 0; j < NF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nks = p.K / 32;
  for (int ks = w; ks < nks; ks += 4) {
    const int k = ks * 32;
    bf16x8 x[MF];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) x[mf] = *(const bf16x8*)(xrow[mf] + k);
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      const bf16x8 h = *(const bf16x8*)(wh[nf] + k);
      const bf16x8 l = *(const bf16x8*)(wl[nf] + k);
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[mf], h, acc[mf][nf], 0, 0, 0);
        acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[mf], l, acc[mf][nf], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int mf = 0; mf < MF; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[w][((mf * NF + nf) * 64 + lane) * 4 + r] = acc[mf][nf][r];
  __syncthreads();
  for (int e = tid; e < MF * NF * 256; e += 256) {
    const float v = red[0][e] + red[1][e] + red[2][e] + red[3][e];
    const int r = e & 3, ln = (e >> 2) & 63, fn = e >> 8;
    const int nf = fn % NF, mf = fn / NF;
    const int m = m0 + mf * 16 + 4 * (ln >> 4) + r;   // D[i = 4g+r][j = lane&15]
    const int j = nf * 16 + (ln & 15);
    if (m >= p.M) continue;
    if (p.U) p.U[(int64_t)m * p.ldu + j] = v;
    if (p.ext) {
      const bf16_t hi = f2bf(v);
      const bf16_t lo = f2bf(v - bf2f(hi));
      bf16_t* e0 = p.ext + (int64_t)m * p.ld_ext + (j / p.group_R) * p.group_stride + (j % p.group_R);
      e0[0] = hi;
      e0[p.group_R] = lo;
      e0[2 * p.group_R] = hi;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// G[j,k] += sum_m V[m,j] X[m,k].  grid = (K/512, M/64, R/16); thread = 2 columns x 16 ranks.
__global__ __launch_bounds__(256) void lora_grad_kernel(const qfx_lora_grad_args p) {
  constexpr int MC = 64;
  __shared__ __attribute__((aligned(16))) float sV[MC][16];
  const int tid = threadIdx.x;
  const int jb = blockIdx.z * 16;
  const int mb = blockIdx.y * MC;
  const int k = blockIdx.x * 512 + tid * 2;
  for (int e = tid; e < MC * 16; e += 256) {
    const int r = e >> 4, c = e & 15;
    const int m = mb + r;
    sV[r][c] = (m < p.M && jb + c < p.R) ? p.V[(int64_t)m * p.ldv + jb + c] : 0.f;
  }
  __syncthreads();
  if (k >= p.K) return;
  float a0[16], a1[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) { a0[j] = 0.f; a1[j] = 0.f; }
  const int mend = (p.M - mb) < MC ? (p.M - mb) : MC;
  for (int r = 0; r < mend; ++r) {
    const int64_t row = remap_row(mb + r, p.rows_per_batch, p.x_batch_rows, p.x_row_off);
    const uint32_t xx = *(const uint32_t*)(p.X + row * p.ldx + k);
    const float x0 = __uint_as_float(xx << 16), x1 = __uint_as_float(xx & 0xffff0000u);
#pragma unroll
    for (int j4 = 0; j4 < 4; ++j4) {
      const f32x4 v = *(const f32x4*)(&sV[r][j4 * 4]);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        a0[j4 * 4 + q] = fmaf(v[q], x0, a0[j4 * 4 + q]);
        a1[j4 * 4 + q] = fmaf(v[q], x1, a1[j4 * 4 + q]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    if (jb + j < p.r_valid) {
      float* gp = p.G + (int64_t)(jb + j) * p.g_sr + (int64_t)k * p.g_sc;
      unsafeAtomicAdd(gp, a0[j] * p.out_scale);
      unsafeAtomicAdd(gp + p.g_sc, a1[j] * p.out_scale);
    }
  }
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lora_pack_kernel(const qfx_lora_pack_args* descs) {
  const qfx_lora_pack_args d = descs[blockIdx.y];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int Rp = d.Rp, pad = d.Kext - 3 * d.Rp;
  // A side: A_hi/A_lo [Rp,K], WeT [K,Kext]
  for (int64_t i = t0; i < (int64_t)Rp * d.K; i += stride) {
    const int j = (int)(i / d.K), k = (int)(i % d.K);
    const float a = j < d.r ? d.A[(int64_t)j * d.K + k] : 0.f;
    const bf16_t hi = f2bf(a), lo = f2bf(a - bf2f(hi));
    d.A_hi[(int64_t)j * d.ld_a + k] = hi;
    d.A_lo[(int64_t)j * d.ld_a + k] = lo;
    bf16_t* wt = d.WeT + (int64_t)k * d.ld_wet + j;
    wt[0] = hi; wt[Rp] = hi; wt[2 * Rp] = lo;
  }
  for (int64_t i = t0; i < (int64_t)pad * d.K; i += stride) {
    const int c = (int)(i % pad), k = (int)(i / pad);
    d.WeT[(int64_t)k * d.ld_wet + 3 * Rp + c] = 0;
  }
  // B side: Bt_hi/Bt_lo [Rp,N] = split(s*B^T), We [N,Kext]
  for (int64_t i = t0; i < (int64_t)Rp * d.N; i += stride) {
    const int j = (int)(i / d.N), n = (int)(i % d.N);
    const float b = j < d.r ? d.scale * d.B[(int64_t)n * d.r + j] : 0.f;
    const bf16_t hi = f2bf(b), lo = f2bf(b - bf2f(hi));
    d.Bt_hi[(int64_t)j * d.ld_bt + n] = hi;
    d.Bt_lo[(int64_t)j * d.ld_bt + n] = lo;
    bf16_t* we = d.We + (int64_t)n * d.ld_we + j;
    we[0] = hi; we[Rp] = hi; we[2 * Rp] = lo;
  }
  for (int64_t i = t0; i < (int64_t)pad * d.N; i += stride) {
    const int c = (int)(i % pad), n = (int)(i / pad);
    d.We[(int64_t)n * d.ld_we + 3 * Rp + c] = 0;
  }
}

}  // namespace

extern "C" int qfx_lora_down(const qfx_lora_down_args* a, void* stream) {
  if (!a || !a->X || !a->W_hi || !a->W_lo) return QFX_EINVAL;
  if (a->M <= 0 || a->K <= 0 || (a->K % 32) || (a->ldx % 8) || (a->ldw % 8) || (a->R % 16) || a->R <= 0) return QFX_EINVAL;
  if (a->ext && (a->group_R <= 0 || (a->R % a->group_R))) return QFX_EINVAL;
  if (a->rows_per_batch <= 0) return QFX_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((a->M + 31) / 32), block(256);
  switch (a->R / 16) {
    case 1: hipLaunchKernelGGL(lora_down_kernel<1>, grid, block, 0, s, *a); break;
    case 2: hipLaunchKernelGGL(lora_down_kernel<2>, grid, block, 0, s, *a); break;
    case 3: hipLaunchKernelGGL(lora_down_kernel<3>, grid, block, 0, s, *a); break;
    case 4: hipLaunchKernelGGL(lora_down_kernel<4>, grid, block, 0, s, *a); break;
    case 6: hipLaunchKernelGGL(lora_down_kernel<6>, grid, block, 0, s, *a); break;
    default: return QFX_EUNSUPPORTED;
  }
  QFX_CHECK_LAUNCH();
  return QFX_OK;
}

extern "C" int qfx_lora_grad(const qfx_lora_grad_args* a, void* stream) {
  if (!a || !a->V || !a->X || !a->G) return QFX_EINVAL;
  if (a->M <= 0 || a->K <= 0 || (a->K % 2) || (a->ldx % 2) || a->R <= 0 || a->rows_per_batch <= 0) return QFX_EINVAL;
  dim3 grid((a->K + 511) / 512, (a->M + 63) / 64, (a->R + 15) / 16);
  hipLaunchKernelGGL(lora_grad_kernel, grid, dim3(256), 0, (hipStream_t)stream, *a);
  QFX_CHECK_LAUNCH();
  return QFX_OK;
}

extern "C" int qfx_lora_pack(const qfx_lora_pack_args* descs, int32_t n, int32_t max_dim, void* stream) {
  if (!descs || n <= 0 || max_dim <= 0) return QFX_EINVAL;
  int bx = (max_dim * 16 + 255) / 256;
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(lora_pack_kernel, dim3(bx, n), dim3(256), 0, (hipStream_t)stream, descs);
  QFX_CHECK_LAUNCH();
  return QFX_OK;
}
