// Common device helpers for the gfx950 kernels of libqfx.  gfx950 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "qfx.h"

typedef uint16_t bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // MFMA A/B fragment (8 bf16 = 4 VGPR)
typedef __attribute__((ext_vector_type(4))) short bf16x4;   // 8 bytes
typedef __attribute__((ext_vector_type(4))) float f32x4;    // MFMA 16x16 C/D fragment
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define QFX_AS1 __attribute__((address_space(1)))
#define QFX_AS3 __attribute__((address_space(3)))

#define QFX_CHECK_LAUNCH()                                   \
  do {                                                       \
    hipError_t e__ = hipGetLastError();                      \
    if (e__ != hipSuccess) return -(1000 + (int)e__);        \
  } while (0)

__device__ __forceinline__ float bf2f(bf16_t u) { return __uint_as_float(((uint32_t)u) << 16); }

// round-to-nearest-even fp32 -> bf16: gfx950 has v_cvt_pk_bf16_f32, reached through __bf16
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
// round a float to bf16 precision, keep as float
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }

__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}

// tanh(u) = 1 - 2/(1 + e^{2u}) on the hardware exp2/rcp (1 ulp each): e overflows to +inf -> rcp 0 -> 1, underflows
// to 0 -> -1.  Absolute error ~1e-7, far below the bf16 rounding every caller applies; the library tanhf costs ~5x
// more VALU work and was 20 % of the fc1 / dgelu GEMM launches.
__device__ __forceinline__ float fast_tanh_f(float u) {
  const float e = __builtin_amdgcn_exp2f(u * 2.8853900817779268f);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ float gelu_tanh_f(float x) {
  // torch gelu(approximate="tanh"): 0.5*x*(1+tanh(sqrt(2/pi)*(x+0.044715x^3)))
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  float u = k0 * (x + k1 * x * x * x);
  return 0.5f * x * (1.0f + fast_tanh_f(u));
}
__device__ __forceinline__ float gelu_tanh_grad_f(float x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  float x2 = x * x;
  float u = k0 * (x + k1 * x * x2);
  float t = fast_tanh_f(u);
  float du = k0 * (1.0f + 3.0f * k1 * x2);
  return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * du;
}

// row remap for joint [text|image] buffers: row(m) = (m / rpb) * batch_rows + off + m % rpb
__device__ __forceinline__ int64_t remap_row(int m, int rpb, int batch_rows, int off) {
  if (batch_rows == 0) return (int64_t)m;
  int b = m / rpb;
  return (int64_t)b * batch_rows + off + (m - b * rpb);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
