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

// two floats -> one dword of bf16 (lo in bits 0-15): ONE v_cvt_pk_bf16_f32.  The scalar form f2bf(lo) | f2bf(hi) << 16 compiles to
// two half-used v_cvt_pk + v_lshlrev + v_or_b32_sdwa (4 VALU issues per dword: 96 of them per tile and wave in the dK/dV loop).
// (pack2bf_scalar: the old form, kept for norm_rope_bwd_row in qfx_attn.hip -- see the note there)
__device__ __forceinline__ uint32_t pack2bf_scalar(float lo, float hi) { return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16); }
typedef __attribute__((ext_vector_type(2))) __bf16 qfx_bf16x2;
typedef __attribute__((ext_vector_type(2))) float qfx_f32x2;
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
#if defined(QFX_PACK2BF_SCALAR)     // the 4-instruction form of rounds 1-3 (A/B lever)
  return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
#else
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((qfx_f32x2){lo, hi}, qfx_bf16x2));
#endif
}

// gelu(approximate="tanh"): 0.5 x (1 + tanh(u)) = x * sigmoid(2u), u = sqrt(2/pi) (x + 0.044715 x^3), evaluated with the
// hardware exp2 / rcp (1 ulp each): s = 1 / (1 + 2^(-2u log2 e)).  e overflows to +inf -> s = 0, underflows to 0 -> s = 1.
// Absolute error ~1e-7, far below the bf16 rounding every caller applies; the library tanhf costs ~5x more VALU work and was
// 20 % of the fc1 / dgelu GEMM launches.  7 VALU ops per element (2 of them transcendental).
__device__ __forceinline__ float gelu_sigmoid2u_f(float x, float x2) {
  // -2 * log2(e) * sqrt(2/pi) = -2.3022082; 0.044715 * that = -0.10294324
  const float t = x * __builtin_fmaf(x2, -0.10294324f, -2.3022082f);     // = -2u log2(e)
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
}
__device__ __forceinline__ float gelu_tanh_f(float x) {
  return x * gelu_sigmoid2u_f(x, x * x);
}
// d/dx [x s(2u)] = s + x s (1 - s) 2u',  2u' = 2 sqrt(2/pi) (1 + 3*0.044715 x^2)
__device__ __forceinline__ float gelu_tanh_grad_f(float x) {
  const float x2 = x * x;
  const float s = gelu_sigmoid2u_f(x, x2);
  const float du2 = __builtin_fmaf(x2, 0.21406445f, 1.5957692f);          // 2 k0 (1 + 3 k1 x^2)
  return __builtin_fmaf(x * du2, s - s * s, s);
}

// row remap for joint [text|image] buffers: row(m) = (m / rpb) * batch_rows + off + m % rpb
__device__ __forceinline__ int64_t remap_row(int m, int rpb, int batch_rows, int off) {
  if (batch_rows == 0) return (int64_t)m;
  if (m < rpb) return (int64_t)(off + m);      // first sample (all rows when B = 1): no integer division on the launch's critical path
  int b = m / rpb;
  return (int64_t)b * batch_rows + off + (m - b * rpb);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// MX-FP8 quantisation of 8 consecutive elements of a 32-element block held by 4 lanes (the caller passes the block maximum over
// those lanes): OCP MX, shared exponent floor(log2(amax)) - 8, e4m3 elements RNE, saturated at +-448 -- the arithmetic of
// quant_mxfp8_kernel, so that producers that quantise on the fly emit the same bytes as a separate pass over their bf16 output.
__device__ __forceinline__ u32x2 mx_quant8(const float (&v)[8], float amax, int& e_biased) {
  int e = (int)((__float_as_uint(amax) >> 23) & 0xff) - 127 - 8;
  if (amax == 0.f) e = -127;
  e = e < -127 ? -127 : e;
  const float inv = __uint_as_float((uint32_t)(127 - e) << 23);
  float q[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) q[i] = fminf(fmaxf(v[i] * inv, -448.f), 448.f);
  uint32_t w0 = 0, w1 = 0;
  w0 = __builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], w0, false);
  w0 = __builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], w0, true);
  w1 = __builtin_amdgcn_cvt_pk_fp8_f32(q[4], q[5], w1, false);
  w1 = __builtin_amdgcn_cvt_pk_fp8_f32(q[6], q[7], w1, true);
  e_biased = e + 127;
  return (u32x2){w0, w1};
}
