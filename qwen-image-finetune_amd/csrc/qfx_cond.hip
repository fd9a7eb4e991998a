// qfx_cond.hip -- adapters on the conditioning head (timestep / guidance / pooled-text embedders, AdaLN modulation linears,
// norm_out.linear): the rank-r side terms of linears that see M = batch rows only.
//
// The frozen base part of these linears is the weight stream of qfx_mod_gemv / qfx_mod_gemv_t (13.6 GB per direction at the
// Qwen size); what is here is the peft arithmetic around it, for a BANK of adapters that share one input x [B, K]:
//     forward   u_a = A_a act(x)            (fp32; act = bf16(silu(.)) or identity, the eager graph's rounding)
//               y_a = bf16(float(y_a) + s_a B_a u_a)           in place on the base output rows (peft lora.Linear.forward)
//     backward  ga = s_a g_a (g bf16, as autograd hands it over)
//               dB_a += ga^T u_a ;  du_a = ga B_a ;  dA_a += du_a^T act(x) ;  dx += du_a A_a     (all fp32)
// Problem sizes are tiny (B <= 8, r <= 64, K <= 3072, N <= 18432, ~120 adapters): the kernels are plain HBM streams over the
// fp32 adapter weights (0.15 GB per direction for "all-linear" at r = 16), one block per (column chunk, adapter).
// Replaces the torch.einsum / autograd evaluation of round 2 (reference call sites: transformer_qwenimage.py:143-156,389-392,
// 430-436,565,664; transformer_flux.py:634-639,729-741 through peft, base_trainer.py:929-941).
#include "qfx_common.h"

namespace {

__device__ __forceinline__ float act_in(bf16_t v, int apply_silu) {
  const float x = bf2f(v);
  return apply_silu ? rbf(x / (1.0f + __expf(-x))) : x;
}

// u[a][b][j] = sum_k A_a[j][k] act(x[b][k]) -- one block per adapter, 256 threads stride over k
__global__ __launch_bounds__(256) void cond_u_kernel(const qfx_cond_lora_args a) {
  __shared__ float red[4][8];
  const int ad = blockIdx.x, t = threadIdx.x;
  const float* A = a.A[ad];
  for (int j = 0; j < a.r; ++j) {
    float acc[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[b] = 0.f;
    for (int k = t; k < a.K; k += 256) {
      const float w = A[(int64_t)j * a.K + k];
#pragma unroll
      for (int b = 0; b < 8; ++b)
        if (b < a.B) acc[b] += w * act_in(a.x[(int64_t)b * a.K + k], a.apply_silu);
    }
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const float s = wave_sum(acc[b]);
      if ((t & 63) == 0) red[t >> 6][b] = s;
    }
    __syncthreads();
    if (t < a.B) a.u[((int64_t)ad * a.B + t) * a.r + j] = red[0][t] + red[1][t] + red[2][t] + red[3][t];
    __syncthreads();
  }
}

// y_a[b][n] = bf16(float(y_a[b][n]) + s_a sum_j B_a[n][j] u_a[b][j]) -- thread per output column
__global__ __launch_bounds__(256) void cond_add_kernel(const qfx_cond_lora_args a) {
  __shared__ float su[8 * 64];
  const int ad = blockIdx.y, t = threadIdx.x;
  for (int i = t; i < a.B * a.r; i += 256) su[i] = a.u[(int64_t)ad * a.B * a.r + i];
  __syncthreads();
  const int n = blockIdx.x * 256 + t;
  if (n >= a.N) return;
  const float* Br = a.Bm[ad] + (int64_t)n * a.r;
  const float s = a.scale[ad];
  bf16_t* y = a.y[ad];
  float acc[8];
#pragma unroll
  for (int b = 0; b < 8; ++b) acc[b] = 0.f;
  for (int j = 0; j < a.r; ++j) {
    const float w = Br[j];
#pragma unroll
    for (int b = 0; b < 8; ++b)
      if (b < a.B) acc[b] += w * su[b * a.r + j];
  }
#pragma unroll
  for (int b = 0; b < 8; ++b)
    if (b < a.B) {
      bf16_t* p = y + (int64_t)b * a.ldy + n;
      *p = f2bf(bf2f(*p) + acc[b] * s);       // peft: (result + lora_B(lora_A(x)) * scaling).to(result dtype)
    }
}

// dB_a[n][j] += sum_b ga[b][n] u_a[b][j] ;  du_a[b][j] += sum_n ga[b][n] B_a[n][j]   (ga = s_a * bf16 gradient)
__global__ __launch_bounds__(256) void cond_bwd_b_kernel(const qfx_cond_lora_args a) {
  __shared__ float su[8 * 64];
  __shared__ float sdu[8 * 64];
  const int ad = blockIdx.y, t = threadIdx.x;
  for (int i = t; i < a.B * a.r; i += 256) { su[i] = a.u[(int64_t)ad * a.B * a.r + i]; sdu[i] = 0.f; }
  __syncthreads();
  const int n = blockIdx.x * 256 + t;
  const bool ok = n < a.N;
  const float s = a.scale[ad];
  const bf16_t* g = a.g[ad];
  float ga[8];
#pragma unroll
  for (int b = 0; b < 8; ++b) ga[b] = (ok && b < a.B) ? bf2f(g[(int64_t)b * a.ldg + n]) * s : 0.f;
  const float* Br = a.Bm[ad] + (int64_t)(ok ? n : 0) * a.r;
  float* dBr = a.dB[ad] + (int64_t)(ok ? n : 0) * a.r;
  for (int j = 0; j < a.r; ++j) {
    const float w = ok ? Br[j] : 0.f;
    float d = 0.f;
#pragma unroll
    for (int b = 0; b < 8; ++b)
      if (b < a.B) {
        d += ga[b] * su[b * a.r + j];
        const float part = wave_sum(ga[b] * w);
        if ((t & 63) == 0) atomicAdd(&sdu[b * a.r + j], part);
      }
    if (ok) dBr[j] += d;       // single writer per row: accumulates across micro-steps without atomics
  }
  __syncthreads();
  for (int i = t; i < a.B * a.r; i += 256) unsafeAtomicAdd(&a.du[(int64_t)ad * a.B * a.r + i], sdu[i]);
}

// dA_a[j][k] += sum_b du_a[b][j] act(x[b][k]) ;  dx[b][k] += sum_j du_a[b][j] A_a[j][k]
__global__ __launch_bounds__(256) void cond_bwd_a_kernel(const qfx_cond_lora_args a) {
  __shared__ float sdu[8 * 64];
  const int ad = blockIdx.y, t = threadIdx.x;
  for (int i = t; i < a.B * a.r; i += 256) sdu[i] = a.du[(int64_t)ad * a.B * a.r + i];
  __syncthreads();
  const int k = blockIdx.x * 256 + t;
  if (k >= a.K) return;
  float xa[8], dxa[8];
#pragma unroll
  for (int b = 0; b < 8; ++b) { xa[b] = b < a.B ? act_in(a.x[(int64_t)b * a.K + k], a.apply_silu) : 0.f; dxa[b] = 0.f; }
  const float* A = a.A[ad];
  float* dA = a.dA[ad];
  for (int j = 0; j < a.r; ++j) {
    const float w = A[(int64_t)j * a.K + k];
    float d = 0.f;
#pragma unroll
    for (int b = 0; b < 8; ++b)
      if (b < a.B) { d += sdu[b * a.r + j] * xa[b]; dxa[b] += sdu[b * a.r + j] * w; }
    dA[(int64_t)j * a.K + k] += d;
  }
  if (a.dx) {
#pragma unroll
    for (int b = 0; b < 8; ++b)
      if (b < a.B) unsafeAtomicAdd(&a.dx[(int64_t)b * a.K + k], dxa[b]);
  }
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = f2bf(in[i]);
}

// dx = bf16( bf16(ds) * silu'(x) ): autograd's silu_backward on a bf16 gradient (opmath fp32, one rounding)
__global__ __launch_bounds__(256) void silu_bwd_kernel(const float* __restrict__ ds, const bf16_t* __restrict__ x, bf16_t* __restrict__ dx,
                                                       int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float t = bf2f(x[i]);
    const float sg = 1.0f / (1.0f + __expf(-t));
    dx[i] = f2bf(rbf(ds[i]) * (sg * (1.0f + t * (1.0f - sg))));
  }
}

int check(const qfx_cond_lora_args* a) {
  if (!a || !a->x || !a->A || !a->Bm || !a->scale || !a->u) return QFX_EINVAL;
  if (a->B <= 0 || a->B > 8 || a->K <= 0 || a->N <= 0 || a->na <= 0 || a->r <= 0 || a->r > 64) return QFX_EINVAL;
  return QFX_OK;
}

}  // namespace

extern "C" int qfx_cond_lora_fwd(const qfx_cond_lora_args* a, void* stream) {
  int rc = check(a);
  if (rc) return rc;
  if (!a->y) return QFX_EINVAL;
  hipLaunchKernelGGL(cond_u_kernel, dim3(a->na), dim3(256), 0, (hipStream_t)stream, *a);
  QFX_CHECK_LAUNCH();
  hipLaunchKernelGGL(cond_add_kernel, dim3((a->N + 255) / 256, a->na), dim3(256), 0, (hipStream_t)stream, *a);
  QFX_CHECK_LAUNCH();
  return QFX_OK;
}

extern "C" int qfx_cond_lora_bwd(const qfx_cond_lora_args* a, void* stream) {
  int rc = check(a);
  if (rc) return rc;
  if (!a->g || !a->dA || !a->dB || !a->du) return QFX_EINVAL;
  hipLaunchKernelGGL(cond_bwd_b_kernel, dim3((a->N + 255) / 256, a->na), dim3(256), 0, (hipStream_t)stream, *a);
  QFX_CHECK_LAUNCH();
  hipLaunchKernelGGL(cond_bwd_a_kernel, dim3((a->K + 255) / 256, a->na), dim3(256), 0, (hipStream_t)stream, *a);
  QFX_CHECK_LAUNCH();
  return QFX_OK;
}

extern "C" int qfx_cast_f32_bf16(const float* in, uint16_t* out, int64_t n, void* stream) {
  if (!in || !out || n <= 0) return QFX_EINVAL;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, out, n);
  QFX_CHECK_LAUNCH();
  return QFX_OK;
}

extern "C" int qfx_silu_bwd(const float* ds, const uint16_t* x, uint16_t* dx, int64_t n, void* stream) {
  if (!ds || !x || !dx || n <= 0) return QFX_EINVAL;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(silu_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, ds, x, dx, n);
  QFX_CHECK_LAUNCH();
  return QFX_OK;
}
