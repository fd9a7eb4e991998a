// Ablation lab (NOT product): the shipped 256x128x64 / 8-wave / 3-stage kernel with parts removed, to find what bounds it.
// ABL bits: 1 = no DMA in the K loop, 2 = no LDS fragment reads in the loop, 4 = no s_barrier, 8 = no MFMA, 16 = no epilogue store
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ void glds16(const bf16_t* g, char* lds) {
  __builtin_amdgcn_global_load_lds((const AS1 void*)g, (AS3 void*)lds, 16, 0, 0);
}
constexpr int BM = 256, BN = 128, BK = 64, NST = 3;
constexpr int STAGE = (BM + BN) * BK * 2;

template <int ABL>
__global__ __launch_bounds__(512, 1) void kabl(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, bf16_t* __restrict__ C,
                                               int M, int N, int K, int GM_) {
  constexpr int GM = 8;
  __shared__ __attribute__((aligned(16))) char smem[NST * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 1, wc = w & 1;
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const int nwg = (ABL & 32) ? tiles_m * tiles_n : gridDim.x;
  for (int bid = blockIdx.x; bid < nwg; bid += gridDim.x) {
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  int tm, tn;
  {
    const int per = GM * tiles_n; const int grp_ = swz / per; const int first = grp_ * GM;
    const int gsz = (tiles_m - first) < GM ? (tiles_m - first) : GM;
    const int in = swz - grp_ * per;
    tm = first + in % gsz; tn = in / gsz;
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const int srow = lane >> 3, schunk = lane & 7;
  const bf16_t* pa[4]; const bf16_t* pb[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int lr = w * 32 + i * 8 + srow;
    int gm = m0 + lr; gm = gm < M ? gm : M - 1;
    pa[i] = A + (int64_t)gm * K + (schunk ^ ((lr >> 1) & 7)) * 8;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int lr = w * 16 + i * 8 + srow;
    int gn = n0 + lr; gn = gn < N ? gn : N - 1;
    pb[i] = B + (int64_t)gn * K + (schunk ^ ((lr >> 1) & 7)) * 8;
  }
  const int nt = K / BK;
  auto stageA = [&](int t, int buf, int i) { glds16(pa[i] + t * BK, smem + buf * STAGE + (w * 32 + i * 8) * 128); };
  auto stageB = [&](int t, int buf, int i) { glds16(pb[i] + t * BK, smem + buf * STAGE + BM * 128 + (w * 16 + i * 8) * 128); };
  auto stage = [&](int t, int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) stageA(t, buf, i);
#pragma unroll
    for (int i = 0; i < 2; ++i) stageB(t, buf, i);
  };
  const int g = lane >> 4, li = lane & 15;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto rdA = [&](const char* sA, int kk, int mi) {
    const int row = wr * 64 + mi * 16 + li; const int chunk = kk * 4 + g;
    return *(const bf16x8*)(sA + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
  };
  auto rdB = [&](const char* sB, int kk, int ni) {
    const int row = wc * 64 + ni * 16 + li; const int chunk = kk * 4 + g;
    return *(const bf16x8*)(sB + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
  };
  stage(0, 0);
  if (nt > 1) stage(1, 1);
  int buf = 0;
  bf16x8 a0[4], b0[4], a1[4], b1[4];
  if constexpr (ABL & 2) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) { a0[i] = rdA(smem, 0, i); b0[i] = rdB(smem + BM * 128, 0, i); a1[i] = rdA(smem, 1, i); b1[i] = rdB(smem + BM * 128, 1, i); }
  }
  for (int t = 0; t < nt; ++t) {
    if constexpr (!(ABL & 1)) { if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    else { if (t == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    if constexpr (!(ABL & 4)) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int nb = buf + 2; nb = nb >= NST ? nb - NST : nb;
    const bool pre = (ABL & 1) ? false : (t + 2 < nt);
    const char* sA = smem + buf * STAGE; const char* sB = sA + BM * 128;
    if constexpr (!(ABL & 2)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { a0[i] = rdA(sA, 0, i); b0[i] = rdB(sB, 0, i); }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      if constexpr (!(ABL & 8)) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[ni], a0[mi], acc[mi][ni], 0, 0, 0);
      }
      if constexpr (!(ABL & 2)) { a1[mi] = rdA(sA, 1, mi); b1[mi] = rdB(sB, 1, mi); }
      if (pre) { stageA(t + 2, nb, mi); if (mi < 2) stageB(t + 2, nb, mi); }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (!(ABL & 8)) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[ni], a1[mi], acc[mi][ni], 0, 0, 0);
    } else {
      // keep the reads alive
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) { acc[mi][0][0] += (float)(a0[mi][0] + b0[mi][0] + a1[mi][0] + b1[mi][0]); }
    }
    buf = buf + 1 == NST ? 0 : buf + 1;
  }
  if constexpr (ABL & 16) {
    float s = 0.f;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) s += acc[mi][ni][0] + acc[mi][ni][1] + acc[mi][ni][2] + acc[mi][ni][3];
    if (s == 123.456f) C[0] = 1;
    continue;
  }
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int m = m0 + wr * 64 + mi * 16 + li;
    if (m >= M) continue;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = n0 + wc * 64 + ni * 16 + 4 * g;
      if (n + 3 >= N) continue;
      bf16x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (short)f2bf(acc[mi][ni][r]);
      *(bf16x4*)(C + (int64_t)m * N + n) = o;
    }
  }
  if constexpr (ABL & 32) __syncthreads();
  }
}

#define CASE(x) case x: hipLaunchKernelGGL(kabl<x>, dim3(((x) & 32) ? (tiles < GRID ? tiles : GRID) : tiles), dim3(512), 0, s, a, b, c, M, N, K, 8); break;
extern "C" int lab_gemm(int var, const void* A, const void* B, void* C, int M, int N, int K, int GRID, void* stream) {
  const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  hipStream_t s = (hipStream_t)stream;
  const bf16_t* a = (const bf16_t*)A; const bf16_t* b = (const bf16_t*)B; bf16_t* c = (bf16_t*)C;
  switch (var) {
    CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(7) CASE(8) CASE(9) CASE(16) CASE(17) CASE(19) CASE(23) CASE(24) CASE(26) CASE(32) CASE(48)
    default: return -1;
  }
  return (int)hipGetLastError();
}
