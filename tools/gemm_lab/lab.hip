// GEMM tuning lab (NOT part of the product): variants of the 256x128x64 3-stage kernel, C = A B^T.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ void glds16(const bf16_t* g, char* lds) {
  __builtin_amdgcn_global_load_lds((const AS1 void*)g, (AS3 void*)lds, 16, 0, 0);
}
constexpr int BM = 256, BN = 128, BK = 64, NST = 3;
constexpr int STAGE = (BM + BN) * BK * 2;

// VAR: 0 baseline 16x16x32 compiler order; 1 = software-pipelined/interleaved 16x16; 2 = 1 + setprio;
//      3 = 32x32x16 compiler order; 4 = 32x32x16 interleaved; 5 = 4 + setprio
template <int VAR>
__global__ __launch_bounds__(512, 1) void k256(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, bf16_t* __restrict__ C,
                                               int M, int N, int K, int GM) {
  __shared__ __attribute__((aligned(16))) char smem[NST * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 1, wc = w & 1;
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  int tm, tn;
  if (GM <= 0) { tm = swz % tiles_m; tn = swz / tiles_m; }
  else {
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

  if constexpr (VAR <= 2 || VAR == 6) {
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
    if constexpr (VAR == 6) {
      // ping-pong: waves 0-3 (group 0) and 4-7 (group 1) sit one per SIMD each; in every phase one group issues
      // MFMAs while the other reads fragments, so the matrix pipe of each SIMD always has exactly one feeder.
      const int grp = w >> 2;
      bf16x8 fa[2][4], fb[2][4];
      if (nt > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      for (int t = 0; t <= nt; ++t) {
        int nb = buf + 2; nb = nb >= NST ? nb - NST : nb;
        const bool pre = t + 2 < nt;
        const char* sA = smem + buf * STAGE; const char* sB = sA + BM * 128;
        // ---- even phase
        if (grp == 0) {
          if (t < nt) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
              for (int i = 0; i < 4; ++i) { fa[kk][i] = rdA(sA, kk, i); fb[kk][i] = rdB(sB, kk, i); }
          }
          if (pre) stage(t + 2, nb);
        } else {
          if (t > 0) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
              for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[kk][ni], fa[kk][mi], acc[mi][ni], 0, 0, 0);
                if (pre && kk == 0) { stageA(t + 2, nb, mi); if (mi < 2) stageB(t + 2, nb, mi); }
                __builtin_amdgcn_sched_barrier(0);
              }
          } else if (pre) stage(t + 2, nb);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // ---- odd phase
        if (t < nt) {
          if (grp == 0) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
              for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[kk][ni], fa[kk][mi], acc[mi][ni], 0, 0, 0);
          } else {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
              for (int i = 0; i < 4; ++i) { fa[kk][i] = rdA(sA, kk, i); fb[kk][i] = rdB(sB, kk, i); }
          }
          __builtin_amdgcn_sched_barrier(0);
          if (pre) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
        }
        buf = buf + 1 == NST ? 0 : buf + 1;
      }
    } else
    for (int t = 0; t < nt; ++t) {
      if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      int nb = buf + 2; nb = nb >= NST ? nb - NST : nb;
      const bool pre = t + 2 < nt;
      const char* sA = smem + buf * STAGE; const char* sB = sA + BM * 128;
      if constexpr (VAR == 0) {
        if (pre) stage(t + 2, nb);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          bf16x8 a[4], b[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) { a[i] = rdA(sA, kk, i); b[i] = rdB(sB, kk, i); }
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[ni], a[mi], acc[mi][ni], 0, 0, 0);
        }
      } else {
        bf16x8 a0[4], b0[4], a1[4], b1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { a0[i] = rdA(sA, 0, i); b0[i] = rdB(sB, 0, i); }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (VAR == 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[ni], a0[mi], acc[mi][ni], 0, 0, 0);
          // interleave: next k-step fragments and the DMA of tile t+2 behind these MFMAs
          a1[mi] = rdA(sA, 1, mi); b1[mi] = rdB(sB, 1, mi);
          if (pre) { stageA(t + 2, nb, mi); if (mi < 2) stageB(t + 2, nb, mi); }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[ni], a1[mi], acc[mi][ni], 0, 0, 0);
        if constexpr (VAR == 2) __builtin_amdgcn_s_setprio(0);
      }
      buf = buf + 1 == NST ? 0 : buf + 1;
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
  } else {
    const int g2 = lane >> 5, l32 = lane & 31;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto rdA = [&](const char* sA, int ks, int mi) {
      const int row = wr * 64 + mi * 32 + l32; const int chunk = ks * 2 + g2;
      return *(const bf16x8*)(sA + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
    };
    auto rdB = [&](const char* sB, int ks, int ni) {
      const int row = wc * 64 + ni * 32 + l32; const int chunk = ks * 2 + g2;
      return *(const bf16x8*)(sB + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
    };
    stage(0, 0);
    if (nt > 1) stage(1, 1);
    int buf = 0;
    for (int t = 0; t < nt; ++t) {
      if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      int nb = buf + 2; nb = nb >= NST ? nb - NST : nb;
      const bool pre = t + 2 < nt;
      const char* sA = smem + buf * STAGE; const char* sB = sA + BM * 128;
      if constexpr (VAR == 3) {
        if (pre) stage(t + 2, nb);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          bf16x8 a[2], b[2];
#pragma unroll
          for (int i = 0; i < 2; ++i) { a[i] = rdA(sA, ks, i); b[i] = rdB(sB, ks, i); }
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[ni], a[mi], acc[mi][ni], 0, 0, 0);
        }
      } else {
        bf16x8 a[4][2], b[4][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) { a[0][i] = rdA(sA, 0, i); b[0][i] = rdB(sB, 0, i); }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (VAR == 5) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          if (ks < 3) {
#pragma unroll
            for (int i = 0; i < 2; ++i) { a[ks + 1][i] = rdA(sA, ks + 1, i); b[ks + 1][i] = rdB(sB, ks + 1, i); }
          }
          if (pre) {
            if (ks == 0) { stageA(t + 2, nb, 0); stageA(t + 2, nb, 1); }
            if (ks == 1) { stageA(t + 2, nb, 2); stageA(t + 2, nb, 3); }
            if (ks == 2) { stageB(t + 2, nb, 0); stageB(t + 2, nb, 1); }
          }
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[ks][ni], a[ks][mi], acc[mi][ni], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (VAR == 5) __builtin_amdgcn_s_setprio(0);
      }
      buf = buf + 1 == NST ? 0 : buf + 1;
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int m = m0 + wr * 64 + mi * 32 + l32;
      if (m >= M) continue;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + wc * 64 + ni * 32 + 8 * q + 4 * g2;
          if (n + 3 >= N) continue;
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (short)f2bf(acc[mi][ni][q * 4 + r]);
          *(bf16x4*)(C + (int64_t)m * N + n) = o;
        }
    }
  }
}

// VAR 7: 4 waves (2x2), each 128x64 (8x4 fragments), ONE wave per SIMD, 512-register budget, everything interleaved
__global__ __launch_bounds__(256, 1) void k256w4(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, bf16_t* __restrict__ C,
                                                 int M, int N, int K, int GM) {
  __shared__ __attribute__((aligned(16))) char smem[NST * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 1, wc = w & 1;
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  int tm, tn;
  if (GM <= 0) { tm = swz % tiles_m; tn = swz / tiles_m; }
  else {
    const int per = GM * tiles_n; const int grp_ = swz / per; const int first = grp_ * GM;
    const int gsz = (tiles_m - first) < GM ? (tiles_m - first) : GM;
    const int in = swz - grp_ * per;
    tm = first + in % gsz; tn = in / gsz;
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const int srow = lane >> 3, schunk = lane & 7;
  const bf16_t* pa[8]; const bf16_t* pb[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int lr = w * 64 + i * 8 + srow;
    int gm = m0 + lr; gm = gm < M ? gm : M - 1;
    pa[i] = A + (int64_t)gm * K + (schunk ^ ((lr >> 1) & 7)) * 8;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int lr = w * 32 + i * 8 + srow;
    int gn = n0 + lr; gn = gn < N ? gn : N - 1;
    pb[i] = B + (int64_t)gn * K + (schunk ^ ((lr >> 1) & 7)) * 8;
  }
  const int nt = K / BK;
  auto stageA = [&](int t, int buf, int i) { glds16(pa[i] + t * BK, smem + buf * STAGE + (w * 64 + i * 8) * 128); };
  auto stageB = [&](int t, int buf, int i) { glds16(pb[i] + t * BK, smem + buf * STAGE + BM * 128 + (w * 32 + i * 8) * 128); };
  const int g = lane >> 4, li = lane & 15;
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto rdA = [&](const char* sA, int kk, int mi) {
    const int row = wr * 128 + mi * 16 + li; const int chunk = kk * 4 + g;
    return *(const bf16x8*)(sA + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
  };
  auto rdB = [&](const char* sB, int kk, int ni) {
    const int row = wc * 64 + ni * 16 + li; const int chunk = kk * 4 + g;
    return *(const bf16x8*)(sB + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
  };
#pragma unroll
  for (int i = 0; i < 8; ++i) stageA(0, 0, i);
#pragma unroll
  for (int i = 0; i < 4; ++i) stageB(0, 0, i);
  if (nt > 1) {
#pragma unroll
    for (int i = 0; i < 8; ++i) stageA(1, 1, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) stageB(1, 1, i);
  }
  int buf = 0;
  for (int t = 0; t < nt; ++t) {
    if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int nb = buf + 2; nb = nb >= NST ? nb - NST : nb;
    const bool pre = t + 2 < nt;
    const char* sA = smem + buf * STAGE; const char* sB = sA + BM * 128;
    bf16x8 a0[8], b0[4], a1[8], b1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) b0[i] = rdB(sB, 0, i);
#pragma unroll
    for (int i = 0; i < 8; ++i) a0[i] = rdA(sA, 0, i);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[ni], a0[mi], acc[mi][ni], 0, 0, 0);
      a1[mi] = rdA(sA, 1, mi);
      if (mi < 4) b1[mi] = rdB(sB, 1, mi);
      if (pre) { stageA(t + 2, nb, mi); if (mi < 4) stageB(t + 2, nb, mi); }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[ni], a1[mi], acc[mi][ni], 0, 0, 0);
    buf = buf + 1 == NST ? 0 : buf + 1;
  }
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) {
    const int m = m0 + wr * 128 + mi * 16 + li;
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
}

// VAR 8: 256x256x64 tile, 4 waves (2x2) each 128x128 (8x8 fragments, 256 accumulator registers), one wave per SIMD,
// 2-stage LDS-DMA (64 KiB per stage).  Less global->LDS fill per flop than 256x128.
__global__ __launch_bounds__(256, 1) void k256sq(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, bf16_t* __restrict__ C,
                                                 int M, int N, int K, int GM) {
  constexpr int BN2 = 256, ST2 = (BM + BN2) * BK * 2;
  __shared__ __attribute__((aligned(16))) char smem[2 * ST2];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 1, wc = w & 1;
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN2 - 1) / BN2;
  int tm, tn;
  if (GM <= 0) { tm = swz % tiles_m; tn = swz / tiles_m; }
  else {
    const int per = GM * tiles_n; const int grp_ = swz / per; const int first = grp_ * GM;
    const int gsz = (tiles_m - first) < GM ? (tiles_m - first) : GM;
    const int in = swz - grp_ * per;
    tm = first + in % gsz; tn = in / gsz;
  }
  const int m0 = tm * BM, n0 = tn * BN2;
  const int srow = lane >> 3, schunk = lane & 7;
  const bf16_t* pa[8]; const bf16_t* pb[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int lr = w * 64 + i * 8 + srow;
    int gm = m0 + lr; gm = gm < M ? gm : M - 1;
    int gn = n0 + lr; gn = gn < N ? gn : N - 1;
    pa[i] = A + (int64_t)gm * K + (schunk ^ ((lr >> 1) & 7)) * 8;
    pb[i] = B + (int64_t)gn * K + (schunk ^ ((lr >> 1) & 7)) * 8;
  }
  const int nt = K / BK;
  auto stageA = [&](int t, int buf, int i) { glds16(pa[i] + t * BK, smem + buf * ST2 + (w * 64 + i * 8) * 128); };
  auto stageB = [&](int t, int buf, int i) { glds16(pb[i] + t * BK, smem + buf * ST2 + BM * 128 + (w * 64 + i * 8) * 128); };
  const int g = lane >> 4, li = lane & 15;
  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto rdA = [&](const char* sA, int kk, int mi) {
    const int row = wr * 128 + mi * 16 + li; const int chunk = kk * 4 + g;
    return *(const bf16x8*)(sA + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
  };
  auto rdB = [&](const char* sB, int kk, int ni) {
    const int row = wc * 128 + ni * 16 + li; const int chunk = kk * 4 + g;
    return *(const bf16x8*)(sB + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
  };
#pragma unroll
  for (int i = 0; i < 8; ++i) { stageA(0, 0, i); stageB(0, 0, i); }
  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const bool pre = t + 1 < nt;
    const char* sA = smem + buf * ST2; const char* sB = sA + BM * 128;
    bf16x8 a0[8], b0[8], a1[8], b1[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { b0[i] = rdB(sB, 0, i); a0[i] = rdA(sA, 0, i); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
#pragma unroll
      for (int ni = 0; ni < 8; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[ni], a0[mi], acc[mi][ni], 0, 0, 0);
      a1[mi] = rdA(sA, 1, mi); b1[mi] = rdB(sB, 1, mi);
      if (pre) { stageA(t + 1, buf ^ 1, mi); stageB(t + 1, buf ^ 1, mi); }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
      for (int ni = 0; ni < 8; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[ni], a1[mi], acc[mi][ni], 0, 0, 0);
  }
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) {
    const int m = m0 + wr * 128 + mi * 16 + li;
    if (m >= M) continue;
#pragma unroll
    for (int ni = 0; ni < 8; ++ni) {
      const int n = n0 + wc * 128 + ni * 16 + 4 * g;
      if (n + 3 >= N) continue;
      bf16x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (short)f2bf(acc[mi][ni][r]);
      *(bf16x4*)(C + (int64_t)m * N + n) = o;
    }
  }
}

extern "C" int lab_gemm(int var, const void* A, const void* B, void* C, int M, int N, int K, int GM, void* stream) {
  const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  hipStream_t s = (hipStream_t)stream;
  const bf16_t* a = (const bf16_t*)A; const bf16_t* b = (const bf16_t*)B; bf16_t* c = (bf16_t*)C;
  switch (var) {
    case 0: hipLaunchKernelGGL(k256<0>, dim3(tiles), dim3(512), 0, s, a, b, c, M, N, K, GM); break;
    case 1: hipLaunchKernelGGL(k256<1>, dim3(tiles), dim3(512), 0, s, a, b, c, M, N, K, GM); break;
    case 2: hipLaunchKernelGGL(k256<2>, dim3(tiles), dim3(512), 0, s, a, b, c, M, N, K, GM); break;
    case 3: hipLaunchKernelGGL(k256<3>, dim3(tiles), dim3(512), 0, s, a, b, c, M, N, K, GM); break;
    case 4: hipLaunchKernelGGL(k256<4>, dim3(tiles), dim3(512), 0, s, a, b, c, M, N, K, GM); break;
    case 5: hipLaunchKernelGGL(k256<5>, dim3(tiles), dim3(512), 0, s, a, b, c, M, N, K, GM); break;
    case 6: hipLaunchKernelGGL(k256<6>, dim3(tiles), dim3(512), 0, s, a, b, c, M, N, K, GM); break;
    case 7: hipLaunchKernelGGL(k256w4, dim3(tiles), dim3(256), 0, s, a, b, c, M, N, K, GM); break;
    case 8: { const int t2 = ((M + BM - 1) / BM) * ((N + 255) / 256); hipLaunchKernelGGL(k256sq, dim3(t2), dim3(256), 0, s, a, b, c, M, N, K, GM); } break;
    default: return -1;
  }
  return (int)hipGetLastError();
}
