// qfx_gemm_fp8.hip -- low-precision trunk: MX-FP8 (OCP e4m3 elements, E8M0 scale per 32 K-elements) base GEMM on the
// block-scaled MFMA of gfx950 (v_mfma_scale_f32_16x16x128_f8f6f4, twice the bf16 matrix rate) + the quantiser that produces
// its operands.  Same contract as qfx_gemm.hip (bias, bf16 mid-rounding, bf16 LoRA K-extension, epilogues).
//
// Tile 128x128, K tile = 128 fp8 bytes per row = the SAME LDS image as the bf16 kernel's 64-element K tile (128-byte rows,
// LDS-DMA with the bank swizzle on the source address), so staging and fragment reads are shared: lane (g = lane>>4, li = lane&15)
// reads the 16-byte chunks g and g+4 of row li -- in the bf16 kernel two k-steps, here the two halves of ONE scaled MFMA operand.
// Hardware K order of that operand (measured, tools/mx_probe): k = (byte/16)*64 + g*16 + byte%16, and the scale supplied by lane
// group g covers hardware k in [32g, 32g+32) = chunks {2g', 2g'+1} of the row with g' = g: i.e. exactly the MX block g of the K
// tile when lane group g holds chunks (g, g+4) ... see the MX-FP8 row of the kernel table in profiles/HISTORY.md section 3.
#include "qfx_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) int v8i;
constexpr int BM = 128, BN = 128;
constexpr int ROWB = 128;                  // bytes per tile row (128 fp8 or 64 bf16)
constexpr int TILE_BYTES = BM * ROWB;      // 16 KiB per operand tile

__device__ __forceinline__ void glds16b(const char* g, char* lds) {
  __builtin_amdgcn_global_load_lds((const QFX_AS1 void*)g, (QFX_AS3 void*)lds, 16, 0, 0);
}

// ------------------------------------------------------------------------------------------------ quantiser
// 4 lanes per MX block (8 elements each), 16 blocks per wave-instruction; grid-stride over blocks.
__global__ __launch_bounds__(256) void quant_mxfp8_kernel(const qfx_quant_args a) {
  const int nbk = a.K / 32;
  const int64_t nblocks = (int64_t)a.M * nbk;
  const int sub = threadIdx.x & 3;
  for (int64_t blk = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 2; blk < nblocks; blk += ((int64_t)gridDim.x * 256) >> 2) {
    const int m = (int)(blk / nbk), kb = (int)(blk % nbk);
    const int64_t xr = remap_row(m, a.rows_per_batch, a.x_batch_rows, a.x_row_off);
    const u32x4 u = *(const u32x4*)(a.X + xr * a.ldx + kb * 32 + sub * 8);
    float v[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(u[i] << 16); v[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u); }
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(v[i]));
    amax = fmaxf(amax, __shfl_xor(amax, 1));
    amax = fmaxf(amax, __shfl_xor(amax, 2));
    // shared exponent: floor(log2(amax)) - 8, from the fp32 exponent field (amax is a bf16 value: normal or zero)
    int e = (int)((__float_as_uint(amax) >> 23) & 0xff) - 127 - 8;
    if (amax == 0.f) e = -127;
    e = e < -127 ? -127 : e;
    const float inv = __uint_as_float((uint32_t)(127 - e) << 23);     // 2^-e  (e in [-127, 119] -> exponent field 8..254)
    uint32_t w0 = 0, w1 = 0;
    float q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = fminf(fmaxf(v[i] * inv, -448.f), 448.f);
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], w0, false);
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], w0, true);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(q[4], q[5], w1, false);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(q[6], q[7], w1, true);
    u32x2 o = {w0, w1};
    *(u32x2*)(a.Q + (int64_t)m * a.ldq + kb * 32 + sub * 8) = o;
    if (sub == 0) a.S[((int64_t)(kb >> 2) * a.M + m) * 4 + (kb & 3)] = (uint8_t)(e + 127);   // tile-major: [K/128][M][4]
  }
}

// ------------------------------------------------------------------------------------------------ GEMM
template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_fp8_kernel(const qfx_gemm_fp8_args q) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];  // [buf][A|B]
  const qfx_gemm_args& p = q.g;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 1, wc = w & 1;

  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int m0 = (swz % tiles_m) * BM;
  const int n0 = (swz / tiles_m) * BN;

  // staging: wave w stages rows [w*32, w*32+32) of both tiles, 8 rows x 128 bytes per DMA instruction
  const int srow = lane >> 3, schunk = lane & 7;
  const char* pa[4];
  const char* pb[4];
  int64_t a_row[4], b_row[4];
  int scb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int lr = w * 32 + i * 8 + srow;
    scb[i] = (schunk ^ ((lr >> 1) & 7)) * 16;  // source BYTE offset inside the 128-byte K tile row
    int gm = m0 + lr; gm = gm < p.M ? gm : p.M - 1;
    int gn = n0 + lr; gn = gn < p.N ? gn : p.N - 1;
    a_row[i] = gm; b_row[i] = gn;
    pa[i] = (const char*)p.A1 + (int64_t)gm * p.lda1 + scb[i];
    pb[i] = (const char*)p.B1 + (int64_t)gn * p.ldb1 + scb[i];
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nt1 = p.K1 / 128, nt2 = p.K2 / 64, nt = nt1 + nt2;
  const int g = lane >> 4, li = lane & 15;

  auto stage = [&](int t) {
    char* sA = smem + (t & 1) * 2 * TILE_BYTES;
    char* sB = sA + TILE_BYTES;
    const int koff = (t < nt1 ? t : t - nt1) * ROWB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16b(pa[i] + koff, sA + (w * 32 + i * 8) * ROWB);
      glds16b(pb[i] + koff, sB + (w * 32 + i * 8) * ROWB);
    }
  };
  // scales of this lane's fragment rows for K tile t: one dword (4 MX blocks) per row, byte g is this lane group's block
  const uint8_t* sap[4];
  const uint8_t* sbp[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int gm = m0 + wr * 64 + i * 16 + li; gm = gm < p.M ? gm : p.M - 1;
    int gn = n0 + wc * 64 + i * 16 + li; gn = gn < p.N ? gn : p.N - 1;
    sap[i] = q.sa + (int64_t)gm * 4;
    sbp[i] = q.sb + (int64_t)gn * 4;
  }
  const int64_t sa_tile = (int64_t)p.M * 4, sb_tile = (int64_t)p.N * 4;      // bytes between K tiles of the tile-major scale arrays
  uint32_t sca[4], scbv[4], scan[4], scbn[4];
  auto load_scales = [&](int t, uint32_t (&sa_)[4], uint32_t (&sb_)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { sa_[i] = *(const uint32_t*)(sap[i] + t * sa_tile); sb_[i] = *(const uint32_t*)(sbp[i] + t * sb_tile); }
  };

  stage(0);
  load_scales(0, sca, scbv);
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    if (t + 1 < nt) {
      if (t + 1 == nt1) {  // switch to the bf16 LoRA K segment
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          pa[i] = (const char*)(p.A2 + a_row[i] * p.lda2) + scb[i];
          pb[i] = (const char*)(p.B2 + b_row[i] * p.ldb2) + scb[i];
        }
      }
      stage(t + 1);
      if (t + 1 < nt1) load_scales(t + 1, scan, scbn);
    }
    const char* sA = smem + (t & 1) * 2 * TILE_BYTES;
    const char* sB = sA + TILE_BYTES;
    auto rd = [&](const char* tile, int row, int kk) {
      return *(const bf16x8*)(tile + row * ROWB + (((kk * 4 + g) ^ ((row >> 1) & 7)) << 4));
    };
    if (t < nt1) {
      v8i a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bf16x8 a0 = rd(sA, wr * 64 + i * 16 + li, 0), a1 = rd(sA, wr * 64 + i * 16 + li, 1);
        const bf16x8 b0 = rd(sB, wc * 64 + i * 16 + li, 0), b1 = rd(sB, wc * 64 + i * 16 + li, 1);
        const u32x4 a0u = __builtin_bit_cast(u32x4, a0), a1u = __builtin_bit_cast(u32x4, a1);
        const u32x4 b0u = __builtin_bit_cast(u32x4, b0), b1u = __builtin_bit_cast(u32x4, b1);
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[i][j] = (int)a0u[j]; a[i][4 + j] = (int)a1u[j]; b[i][j] = (int)b0u[j]; b[i][4 + j] = (int)b1u[j]; }
      }
      int sav[4], sbv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { sav[i] = (int)((sca[i] >> (8 * g)) & 0xffu); sbv[i] = (int)((scbv[i] >> (8 * g)) & 0xffu); }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)   // operands swapped as in the bf16 kernel: lane ends up with 4 consecutive N of one M row
          acc[mi][ni] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b[ni], a[mi], acc[mi][ni], 0, 0, 0, sbv[ni], 0, sav[mi]);
#pragma unroll
      for (int i = 0; i < 4; ++i) { sca[i] = scan[i]; scbv[i] = scbn[i]; }
    } else {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf16x8 a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i] = rd(sA, wr * 64 + i * 16 + li, kk); b[i] = rd(sB, wc * 64 + i * 16 + li, kk); }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[ni], a[mi], acc[mi][ni], 0, 0, 0);
      }
    }
    if (nt2 > 0 && !p.seg2_plain && t == nt1 - 1) {
      // base nn.Linear output is a bf16 tensor in the reference: round (acc + bias) before the LoRA add
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int n = n0 + wc * 64 + ni * 16 + 4 * g;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias != nullptr && n + 3 < p.N) {
          const bf16x4 bb = *(const bf16x4*)(p.bias + n);
#pragma unroll
          for (int r = 0; r < 4; ++r) bv[r] = bf2f((bf16_t)bb[r]);
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[mi][ni][r] = rbf(acc[mi][ni][r] + bv[r]);
      }
    }
    __syncthreads();
  }

  // ---- epilogue: lane holds C[m = ..+li][n = ..+4g+r], r=0..3  (identical to the bf16 128x128 kernel)
  const bool bias_pending = (p.bias != nullptr) && (nt2 == 0 || p.seg2_plain);
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int m = m0 + wr * 64 + mi * 16 + li;
    if (m >= p.M) continue;
    const int bidx = m / p.rows_per_batch;
    const int64_t crow = remap_row(m, p.rows_per_batch, p.c_batch_rows, p.c_row_off);
    const bool keep = p.row_mask == nullptr || p.row_mask[m] != 0.f;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = n0 + wc * 64 + ni * 16 + 4 * g;
      if (n + 3 >= p.N) continue;
      if (!keep) {
        const bf16x4 z = {0, 0, 0, 0};
        *(bf16x4*)(p.C + crow * p.ldc + n) = z;
        if constexpr (EPI == QFX_EPI_GELU) *(bf16x4*)(p.C2 + crow * p.ldc2 + n) = z;
        if constexpr (EPI == QFX_EPI_GATE_RES) { if (p.C2) *(bf16x4*)(p.C2 + (int64_t)m * p.ldc2 + n) = z; }
        continue;
      }
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[mi][ni][r];
      if (bias_pending) {
        const bf16x4 bb = *(const bf16x4*)(p.bias + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += bf2f((bf16_t)bb[r]);
      }
      bf16x4 o;
      if constexpr (EPI == QFX_EPI_NONE) {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (short)f2bf(v[r]);
        *(bf16x4*)(p.C + crow * p.ldc + n) = o;
      } else if constexpr (EPI == QFX_EPI_GELU) {
        bf16x4 o2;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bf16_t h = f2bf(v[r]);
          o[r] = (short)h;
          o2[r] = (short)f2bf(gelu_tanh_f(bf2f(h)));
        }
        *(bf16x4*)(p.C + crow * p.ldc + n) = o;
        *(bf16x4*)(p.C2 + crow * p.ldc2 + n) = o2;
      } else if constexpr (EPI == QFX_EPI_GATE_RES) {
        const bf16x4 gt = *(const bf16x4*)(p.gate + (int64_t)bidx * p.gate_bstride + n);
        const bf16x4 rs = *(const bf16x4*)(p.aux + (p.aux_unmapped ? (int64_t)m : crow) * p.ldaux + n);
        bf16x4 yo;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float y = rbf(v[r]);
          yo[r] = (short)f2bf(y);
          const float gy = rbf(bf2f((bf16_t)gt[r]) * y);
          o[r] = (short)f2bf(bf2f((bf16_t)rs[r]) + gy);
        }
        *(bf16x4*)(p.C + crow * p.ldc + n) = o;
        if (p.C2) *(bf16x4*)(p.C2 + (int64_t)m * p.ldc2 + n) = yo;
      } else {  // QFX_EPI_DGELU
        const bf16x4 hx = *(const bf16x4*)(p.aux + (p.aux_unmapped ? (int64_t)m : crow) * p.ldaux + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float y = rbf(v[r]);
          o[r] = (short)f2bf(y * gelu_tanh_grad_f(bf2f((bf16_t)hx[r])));
        }
        *(bf16x4*)(p.C + crow * p.ldc + n) = o;
      }
    }
  }
}

}  // namespace

extern "C" int qfx_quant_mxfp8(const qfx_quant_args* a, void* stream) {
  if (!a || !a->X || !a->Q || !a->S) return QFX_EINVAL;
  if (a->M <= 0 || a->K <= 0 || (a->K % 128) || (a->ldx % 8) || (a->ldq % 16) || a->rows_per_batch <= 0) return QFX_EINVAL;
  const int64_t nblocks = (int64_t)a->M * (a->K / 32);
  int64_t grid = (nblocks * 4 + 255) / 256;
  if (grid > 16384) grid = 16384;
  hipLaunchKernelGGL(quant_mxfp8_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, *a);
  QFX_CHECK_LAUNCH();
  return QFX_OK;
}

extern "C" int qfx_gemm_mxfp8(const qfx_gemm_fp8_args* a, void* stream) {
  if (!a) return QFX_EINVAL;
  const qfx_gemm_args& g = a->g;
  if (!g.A1 || !g.B1 || !g.C || !a->sa || !a->sb) return QFX_EINVAL;
  if (g.M <= 0 || g.N <= 0 || g.K1 <= 0 || (g.K1 % 128) || (g.K2 % 64) || g.K2 < 0) return QFX_EINVAL;
  if ((g.N % 4) || (g.lda1 % 16) || (g.ldb1 % 16) || (g.ldc % 4)) return QFX_EINVAL;
  if (g.K2 > 0 && (!g.A2 || !g.B2 || (g.lda2 % 8) || (g.ldb2 % 8))) return QFX_EINVAL;
  if (g.rows_per_batch <= 0 || g.a_batch_rows != 0) return QFX_EINVAL;
  if (g.epi == QFX_EPI_GELU && (!g.C2 || (g.ldc2 % 4))) return QFX_EINVAL;
  if (g.epi == QFX_EPI_GATE_RES && (!g.gate || !g.aux || (g.ldaux % 4))) return QFX_EINVAL;
  if (g.epi == QFX_EPI_DGELU && (!g.aux || (g.ldaux % 4))) return QFX_EINVAL;
  if (g.epi < 0 || g.epi > 3) return QFX_EUNSUPPORTED;
  const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
  // large problems: warp-specialised persistent kernel (256x128 tiles); small ones keep the 128x128 kernel
  if (((g.M + 255) / 256) * ((g.N + 127) / 128) >= 160 && !g.seg2_plain && (g.N % 8) == 0 && (g.ldc % 8) == 0 &&
      !(g.epi == QFX_EPI_GELU && (g.ldc2 % 8)) && !((g.epi == QFX_EPI_GATE_RES || g.epi == QFX_EPI_DGELU) && (g.ldaux % 8)) &&
      !(g.epi == QFX_EPI_GATE_RES && ((g.gate_bstride % 8) || (g.C2 && (g.ldc2 % 8)))))
    return qfx_gemm_mxfp8_grouped(a, 1, stream);
  if (a->cq || a->cq_only) return QFX_EUNSUPPORTED;   // the quantising epilogue lives in the persistent kernel only
  hipStream_t s = (hipStream_t)stream;
  switch (g.epi) {
    case QFX_EPI_NONE: hipLaunchKernelGGL(gemm_fp8_kernel<QFX_EPI_NONE>, dim3(tiles), dim3(256), 0, s, *a); break;
    case QFX_EPI_GELU: hipLaunchKernelGGL(gemm_fp8_kernel<QFX_EPI_GELU>, dim3(tiles), dim3(256), 0, s, *a); break;
    case QFX_EPI_GATE_RES: hipLaunchKernelGGL(gemm_fp8_kernel<QFX_EPI_GATE_RES>, dim3(tiles), dim3(256), 0, s, *a); break;
    default: hipLaunchKernelGGL(gemm_fp8_kernel<QFX_EPI_DGELU>, dim3(tiles), dim3(256), 0, s, *a); break;
  }
  QFX_CHECK_LAUNCH();
  return QFX_OK;
}
