// qfx_gemm.hip -- bf16 MFMA GEMM for gfx950 with fused LoRA K-extension and epilogues.
//
//   C[M,N] = A1[M,K1] B1[N,K1]^T (+ A2[M,K2] B2[N,K2]^T) + bias  -> epilogue
//
// Both operands are K-contiguous ("TN"): the forward uses the nn.Linear weight [N,K] as is, the dX
// GEMMs use a transposed copy of the frozen weight kept resident in HBM (288 GB makes that free).
// The LoRA side branch rides along as a second K segment: A2 = [u_hi|u_lo|u_hi] (rank-r down
// projection split in two bf16), B2 = [sB_hi|sB_hi|sB_lo], i.e. an fp32-accurate rank-r update
// for 3r extra K columns of MFMA work instead of a separate HBM-bound pass over y.
//
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16x32 fragments.
// Global->LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction), double buffered,
// one barrier per K tile.  LDS image is lane-linear (DMA constraint) so the bank swizzle
// chunk' = chunk ^ ((row>>1)&7) is applied on the SOURCE address and again on the ds_read_b128.
// MFMA operands are swapped (D' = B A^T) so each lane ends up with 4 consecutive N for one M row:
// 8-byte bf16x4 stores / bias / gate / residual accesses in the epilogue.
// blockIdx -> tile mapping is XCD-aware (each XCD walks a contiguous chunk of the tile list).
#include "qfx_common.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <type_traits>

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile

__device__ __forceinline__ void glds16(const bf16_t* g, char* lds) {
  __builtin_amdgcn_global_load_lds((const QFX_AS1 void*)g, (QFX_AS3 void*)lds, 16, 0, 0);
}

// LDS-DMA piece issued from an asm statement: invisible to hipcc's wait-count pass (with the builtin it guards EVERY later LDS read of
// the wave with s_waitcnt vmcnt(0)); completion is tracked by hand-counted s_waitcnt vmcnt(N) (see the dGELU epilogue of gemm256_kernel)
__device__ __forceinline__ void glds16_asm(const bf16_t* g, char* lds) {
  const uint32_t l = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(l) : "memory");
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const qfx_gemm_args p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];  // [buf][A|B]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 1, wc = w & 1;

  // ---- XCD-aware tile mapping (bijective for any grid size)
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int m0 = (swz % tiles_m) * BM;
  const int n0 = (swz / tiles_m) * BN;

  // ---- staging addresses: wave w stages rows [w*32, w*32+32) of both tiles, 8 rows per DMA
  const int srow = lane >> 3;  // 0..7 within the 8-row group
  const int schunk = lane & 7;
  const bf16_t* pa[4];
  const bf16_t* pb[4];
  int64_t a_row[4], b_row[4];
  int sc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int lr = w * 32 + i * 8 + srow;
    sc[i] = (schunk ^ ((lr >> 1) & 7)) * 8;  // source column (elements) inside the 64-wide K tile
    int gm = m0 + lr; gm = gm < p.M ? gm : p.M - 1;
    int gn = n0 + lr; gn = gn < p.N ? gn : p.N - 1;
    a_row[i] = gm; b_row[i] = gn;
    pa[i] = p.A1 + remap_row(gm, p.rows_per_batch, p.a_batch_rows, p.a_row_off) * p.lda1 + sc[i];
    pb[i] = p.B1 + (int64_t)gn * p.ldb1 + sc[i];
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nt1 = p.K1 / BK, nt2 = p.K2 / BK, nt = nt1 + nt2;
  const int g = lane >> 4, li = lane & 15;

  auto stage = [&](int t) {
    char* sA = smem + (t & 1) * 2 * TILE_BYTES;
    char* sB = sA + TILE_BYTES;
    const int koff = (t < nt1 ? t : t - nt1) * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16(pa[i] + koff, sA + (w * 32 + i * 8) * (BK * 2));
      glds16(pb[i] + koff, sB + (w * 32 + i * 8) * (BK * 2));
    }
  };

  stage(0);
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    if (t + 1 < nt) {
      if (t + 1 == nt1) {  // switch to the LoRA K segment
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          pa[i] = p.A2 + a_row[i] * p.lda2 + sc[i];
          pb[i] = p.B2 + b_row[i] * p.ldb2 + sc[i];
        }
      }
      stage(t + 1);
    }
    const char* sA = smem + (t & 1) * 2 * TILE_BYTES;
    const char* sB = sA + TILE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 a[4], b[4];
      const int chunk = kk * 4 + g;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const int row = wr * 64 + mi * 16 + li;
        a[mi] = *(const bf16x8*)(sA + row * (BK * 2) + ((chunk ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int row = wc * 64 + ni * 16 + li;
        b[ni] = *(const bf16x8*)(sB + row * (BK * 2) + ((chunk ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[ni], a[mi], acc[mi][ni], 0, 0, 0);
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

  // ---- epilogue: lane holds C[m = ..+li][n = ..+4g+r], r=0..3
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
        if (p.C2) *(bf16x4*)(p.C2 + (int64_t)m * p.ldc2 + n) = yo;   // pre-gate linear output (rows UNMAPPED), kept for d(gate)
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


// =============================================================================================
// gemm256: WARP-SPECIALISED and PERSISTENT tiles.  One 640-thread block per CU:
//   * waves 0..7  = compute: ds_read_b128 + MFMA 16x16x32 + epilogue only;
//   * waves 8..9  = loaders: all LDS-DMA (global_load_lds_dwordx4) for the block, running up to two K tiles ahead
//                   through the stage ring and straight across output-tile boundaries.
// One s_barrier per K tile joins the two roles ("tile f landed" + "everyone is done with tile f-1").  The compute
// waves never wait on vmcnt, so their epilogue stores drain under the next tile's main loop, and the next tile's
// first stages are already in LDS when the epilogue ends.  Measured (tools/gemm_lab/ws.hip vs abl.hip, warm,
// paired): +12..21 % on the DiT shapes over the same tile with DMA issued from the compute waves.
// Grid = whole rounds over <= 256 CUs (multiple of 8 so that bid % 8 stays the XCD); grouped launch: up to
// QFX_MAX_GROUPS independent problems (image/text streams, q/k/v) share one tile list.
//
// Tile geometries (BMT x TN, K tile 64; compute waves WRN x WCN, each MI x NI fragments of 16 x 16):
//   256 x 128  4 x 2 waves of 64 x 64   3-stage ring (48 KiB)   rotated K loop            -- rounds 1-3
//   256 x 256  2 x 4 waves of 128 x 64  2-stage ring (64 KiB)   streaming K loop (+9 % per flop: a third less LDS-DMA, a
//                                                                 quarter less fragment traffic)
//   160 x 192  2 x 4 waves of 80 x 48   3-stage ring (44 KiB)   rotated K loop            -- round 4
// Why 160 rows: the DiT's M is 2048 image + 384 text rows with DIFFERENT weights per stream.  With 256-row tiles a B = 1 step runs
// 8 + 2 M-tiles (the second text tile half empty: 5 % padding) x 24 N-tiles = 240 tiles on 256 CUs (6 % idle) -- every narrow
// launch is ONE round of one 32768-element tile per CU.  160 rows give 13 + 3 = 16 M-tiles x 16 N-tiles of 192 = exactly 256
// tiles of 30720 elements (-6.25 % per CU; 2 % padding on the image stream).  Measured in the step (profiles/r04_gemm_tiles.json):
// -1 .. -2.6 % per narrow launch (the 80 x 48 wave tile reads 8 fragments per 15 MFMAs and runs ~3.5 % below the 64 x 64 one per
// flop), whole step 99.7 -> 99.1 ms with every narrow launch on it.
// The wide counterparts 160 x 384 (two full rounds for N = 12288; 120 accumulators leave the allocator four registers short: scratch
// reloads inside the K loop) and 160 x 256 (three rounds) were built and measured +13 % / +24 % against 256 x 256: removed.
// The launcher picks the geometry per launch from rounds x relative tile time (qfx_gemm_grouped).
#ifndef QFX_GEMM_NLD
#define QFX_GEMM_NLD 4
#endif
#ifndef QFX_GEMM_PF_DIST
#define QFX_GEMM_PF_DIST 0      // K tiles an L2 prefetch runs ahead of the K loop (0 = off), see the compute waves
#endif
#ifndef QFX_GEMM_AUX_DMA
#define QFX_GEMM_AUX_DMA 1      // bit 0: d(GELU) epilogue, bit 1: gate + residual epilogue (measured SLOWER: profiles/r06_gemm_aux_landing.json) -- the aux rows of the next pass come in by LDS-DMA instead of a load inside the pass (0 = rounds 1-5)
#endif
#ifndef QFX_GEMM_NT_SAVED
#define QFX_GEMM_NT_SAVED 0     // non-temporal stores for outputs that are only read in the backward: bit 0 the GELU epilogue's pre-activation, bit 1 the gate epilogue's pre-gate output
#endif
#ifndef QFX_GEMM_KSTAGGER
#define QFX_GEMM_KSTAGGER 0     // K tiles between the starting points of neighbouring tiles' K loops (0 = every tile starts at k = 0), see the loader waves
#endif
constexpr int NLD = QFX_GEMM_NLD;                 // loader waves
constexpr int WS_THREADS = 512 + 64 * NLD;
constexpr int STG_BYTES = 2048;                   // per compute wave: 16 rows x 64 bf16 staging for the epilogue
constexpr int QFX_NUM_CU = 256;                   // MI355X
template <int BMT, int TN> struct TileCfg {
  static constexpr bool WIDE = TN >= 256;           // streaming K loop
  // K depth of a ring stage.  The wide tile's operands are 64 KB per 64-deep K tile: two such stages hold ONE tile in flight.
  // -DQFX_GEMM_WIDE_BK32 builds it with four 32-deep stages instead (32 KB each, three in flight, one barrier per 32 MFMAs and
  // wave, 64-byte stage rows): measured 4 % SLOWER (148.0 -> 154.1 / 136.6 -> 142.4 us, profiles/r04_gemm_operand_stream.json) --
  // the wide tile's loop period is not "DMA latency + transfer" after all; kept as an A/B lever.
#if defined(QFX_GEMM_WIDE_BK32)
  static constexpr int BKT = WIDE ? 32 : 64;
#else
  static constexpr int BKT = 64;
#endif
  static constexpr int STAGE = (BMT + TN) * BKT * 2;
#if defined(QFX_GEMM_ABL_NST2)     // ablation: one K tile in flight on the narrow tiles too (how much of the K loop is DMA latency?)
  static constexpr int NST = 2;
#else
  static constexpr int NST = WIDE ? (BKT == 32 ? 4 : 2) : 3;
#endif
  static constexpr int WRN = (BMT == 256 && TN == 128) ? 4 : 2;     // compute waves along M
  static constexpr int WCN = 8 / WRN;               // ... along N
  static constexpr int MI = BMT / WRN / 16;         // 16-row MFMA fragments per wave
  static constexpr int NI = TN / WCN / 16;          // 16-column fragments per wave
  static constexpr int NG = NI == 6 ? 3 : NI;       // fragments per epilogue staging pass (<= 64 columns = 128 B per staged row)
  static constexpr int NGRP = NI / NG;
  static_assert(BMT % (WRN * 16) == 0 && TN % (WCN * 16) == 0 && NI % NG == 0, "tile / wave layout");
  static_assert(NST * STAGE + 8 * STG_BYTES <= 160 * 1024, "LDS budget");
  // round 6: per compute wave, a landing buffer for the NEXT epilogue pass's aux rows (16 rows x 16 NG columns bf16), filled by LDS-DMA
  static constexpr int AUXW = 512 * NG;
  static constexpr bool AUX_FITS = NST * STAGE + 8 * STG_BYTES + 8 * AUXW <= 160 * 1024;
};

struct GroupedArgs {
  qfx_gemm_args g[QFX_MAX_GROUPS];
  int tile_start[QFX_MAX_GROUPS + 1];
  int n;
  // MX-FP8 instantiation only: tile-major E8M0 scale arrays of A1 / B1 per group ([K/128][M][4] / [K/128][N][4]); the fp8 operands
  // themselves travel through the SAME loader code as bf16 ones (g.A1 / g.B1 are byte pointers, lda1 / ldb1 / K1 are given in
  // 2-byte units: a 128-byte fp8 K tile is indistinguishable from a 64-element bf16 K tile until it reaches the matrix pipe)
  const uint8_t* sa[QFX_MAX_GROUPS];
  const uint8_t* sb[QFX_MAX_GROUPS];
  // optional MX-FP8 image of the output the NEXT GEMM consumes (qfx_gemm_fp8_args.cq): fp8 bytes, tile-major scales, row stride
  // (bytes), scale rows, "skip the bf16 copy"
  uint8_t* cq[QFX_MAX_GROUPS];
  uint8_t* cs[QFX_MAX_GROUPS];
  int64_t ldcq[QFX_MAX_GROUPS];
  int cq_rows[QFX_MAX_GROUPS];
  int cq_only[QFX_MAX_GROUPS];
  // same-XCD split-K (gemm256_kernel<..., SPLIT = true>): tile_start counts WORK ITEMS (two per tile: item 2t = the upper K half, the
  // producer of an fp32 partial tile; item 2t + 1 = the lower half + LoRA segment + epilogue); ws = partial tiles
  // [tile][compute wave][mi][ni][lane] x 4 floats, wflag = one word per (tile, compute wave), kh_bias = K tiles the consumer's half is
  // shorter by (it also runs the LoRA segment, the exchange and the epilogue)
  float* ws;
  unsigned* wflag;
  int kh_bias;
  int gm;            // supertile height in M-tiles (tile_coord)
};

// The argument block is read straight from the kernarg segment (constant address space, scalar loads): indexing the
// by-value parameter with the run-time group id would make the compiler copy all of it to scratch.
#define QFX_AS4 __attribute__((address_space(4)))
typedef const QFX_AS4 GroupedArgs KGroupedArgs;
typedef const QFX_AS4 qfx_gemm_args KArgs;

template <int BMT, int TN, bool SPLIT = false>
__device__ __forceinline__ void tile_coord(KGroupedArgs& ga, int nwg, int bid, int& gi, int& m0, int& n0, int* pin = nullptr, int* pgsz = nullptr,
                                           int* phalf = nullptr, int* pgt = nullptr) {
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  gi = 0;
#pragma unroll
  for (int i = 1; i < QFX_MAX_GROUPS; ++i)
    if (i < ga.n && swz >= ga.tile_start[i]) gi = i;
  int lt = swz - ga.tile_start[gi];
  if constexpr (SPLIT) {          // two work items per tile, neighbours in the XCD's contiguous range (nwg % 16 == 0: never across XCDs)
    if (phalf) *phalf = 1 - (lt & 1);
    if (pgt) *pgt = swz >> 1;
    lt >>= 1;
  }
  // supertile order: consecutive tile ids walk 8 M-tiles before the next N-tile, so the ~32 tiles an XCD runs at
  // a time form an 8(M) x 4(N) patch that shares A rows and B rows in that XCD's L2.
  const int M = ga.g[gi].M, N = ga.g[gi].N;
  const int tiles_m = (M + BMT - 1) / BMT, tiles_n = (N + TN - 1) / TN;
  // M-tiles a supertile walks before the next N-tile: the XCD patch is GM x (32 / GM) tiles.  8 x 4 moves the fewest operand rows per
  // XCD; the six-problem q/k/v launch (13 + 3 M-tiles per stream, K = 3072) measures 3 % faster on 4 x 8 (profiles/r05_gemm_stream_path.json),
  // every other class 0.3-1 % slower: chosen per launch (qfx_gemm_grouped), -DQFX_GEMM_GM forces one value (A/B builds).
#if defined(QFX_GEMM_GM)
  constexpr int GM = QFX_GEMM_GM;
#else
  const int GM = ga.gm;
#endif
  const int per = GM * tiles_n, sg = lt / per, first = sg * GM;
  const int gsz = (tiles_m - first) < GM ? (tiles_m - first) : GM;
  const int in = lt - sg * per;
  m0 = (first + in % gsz) * BMT;
  n0 = (in / gsz) * TN;
  if (pin) *pin = in;
  if (pgsz) *pgsz = gsz;
}

typedef __attribute__((ext_vector_type(8))) int v8i32;

template <int EPI, int BMT, int TN, bool FP8 = false, bool SPLIT = false>
__global__ __launch_bounds__(WS_THREADS, 1) void gemm256_kernel(const GroupedArgs ga_by_value) {
  static_assert(!(FP8 && !(BMT == 256 && TN == 128)), "the MX-FP8 instantiation uses the 256x128 tile (the streaming loops have no registers for 8-VGPR operands)");
  using TC = TileCfg<BMT, TN>;
  static_assert(!SPLIT || (TC::WIDE && TC::BKT == 64 && !FP8), "split-K runs on the 256x256 bf16 tile");
  constexpr int STAGE_BYTES = TC::STAGE, NSTAGE = TC::NST, MI = TC::MI, NI = TC::NI, NG = TC::NG, NGRP = TC::NGRP;
  constexpr int BKT = TC::BKT;     // K depth of a ring stage (64; 32 on the wide tile)
  static_assert(!(FP8 && BKT != 64), "the MX-FP8 operands need 64-element (128-byte) stage rows");
  constexpr int WCOLS = 16 * NI;   // columns per compute wave
  // dGELU epilogue with its aux rows by LDS-DMA (round 6; see the epilogue)
  constexpr bool AUXDMA = (QFX_GEMM_AUX_DMA != 0) && !FP8 && !SPLIT && TC::AUX_FITS && (EPI == QFX_EPI_DGELU || (EPI == QFX_EPI_GATE_RES && (QFX_GEMM_AUX_DMA & 2)));
  __shared__ __attribute__((aligned(16))) char smem[NSTAGE * STAGE_BYTES + 8 * STG_BYTES + (AUXDMA ? 8 * TC::AUXW : 0)];
  KGroupedArgs& ga = *(KGroupedArgs*)__builtin_amdgcn_kernarg_segment_ptr();  // == ga_by_value (sole explicit argument)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwg = ga.tile_start[QFX_MAX_GROUPS];

  if (w >= 8) {
    // ================================================================ loader waves
    const int lw = w - 8;
    constexpr int CPR = BKT / 8, RPP = 64 / CPR;      // 16-byte chunks per stage row, rows per 1 KiB DMA piece (8 x 128 B or 16 x 64 B)
    constexpr int NA = (BMT / RPP) / NLD, NB = (TN / RPP) / NLD;   // DMA pieces per K stage per loader wave
    static_assert(NLD % 2 == 0 && (BMT / RPP) % NLD == 0 && (TN / RPP) % NLD == 0 && (NSTAGE - 1) * (NA + NB) < 64, "loader split / vmcnt immediate");
    const int srow = lane / CPR, schunk = lane % CPR;
    // source column incl. the bank swizzle, the same for every piece of a loader wave.  128-byte rows: chunk ^ ((row >> 1) & 7) (piece
    // parity = lw & 1 for an even number of loader waves).  64-byte rows: chunk ^ (-(row >> 2) & 3) -- the four rows r, r+4, r+8, r+12
    // that share a 64-byte slot of the 256-byte bank row then take four different chunks in every ds_read_b128 lane group.
    const int sc = BKT == 64 ? (schunk ^ (((lw & 1) * 4 + (srow >> 1)) & 7)) * 8 : (schunk ^ ((0 - (srow >> 2)) & 3)) * 8;
    const bf16_t* pa[NA];
    const bf16_t* pb[NB];
    int ibid = blockIdx.x, it = 0, int1 = 0, intt = 0, ist = 0;
    int im0 = 0, in0 = 0, iM = 1, iN = 1;          // issue cursor's tile (rows are recomputed at the LoRA-segment switch)
    const bf16_t* iA2 = nullptr; const bf16_t* iB2 = nullptr;
    int ilda2 = 0, ildb2 = 0;
    int ikt0 = 0;                                   // K tile the base segment starts at (split-K: the work item's half; QFX_GEMM_KSTAGGER: wraps around)
    auto setp = [&](int bid) {
      int gi;
      int ihalf = 0;
#if QFX_GEMM_KSTAGGER > 0
      int tin_, tgsz_;
      tile_coord<BMT, TN, SPLIT>(ga, nwg, bid, gi, im0, in0, &tin_, &tgsz_, &ihalf);
#else
      tile_coord<BMT, TN, SPLIT>(ga, nwg, bid, gi, im0, in0, nullptr, nullptr, &ihalf);
#endif
      KArgs& p = ga.g[gi];
      int1 = p.K1 / BKT; intt = int1 + p.K2 / BKT;
      if constexpr (SPLIT) {                        // consumer: K tiles [0, kh) + the LoRA segment; producer: [kh, K1 / BKT)
        const int kh = int1 / 2 - ga.kh_bias;
        if (ihalf) { ikt0 = kh; int1 -= kh; intt = int1; }
        else { ikt0 = 0; int1 = kh; intt = kh + p.K2 / BKT; }
      }
#if QFX_GEMM_KSTAGGER > 0
      // The ~32 tiles an XCD runs at a time share A panels 4 ways and B panels 8 ways and walk K in lockstep: every operand line is a
      // compulsory L2 miss that ALL its sharers wait for together.  Starting the base segment QFX_GEMM_KSTAGGER K tiles apart by tile
      // parity (fp32 sums are re-ordered, nothing else changes) lets half the sharers find the line already resident.  Measured
      // (profiles/r05_gemm_kstagger.json): 1-2 % SLOWER at 2 / 4 / 8 K tiles on every launch class -- lockstep sharers merge their
      // misses, the stream is bandwidth- not latency-bound.  A/B lever only (default 0: this code is not compiled).
      if constexpr (!FP8) { ikt0 = ((((tin_ % tgsz_) & 1) + 2 * ((tin_ / tgsz_) & 1)) * QFX_GEMM_KSTAGGER) % int1; }
#endif
      iA2 = p.A2; iB2 = p.B2; ilda2 = p.lda2; ildb2 = p.ldb2; iM = p.M; iN = p.N;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        int gm = im0 + (lw + i * NLD) * RPP + srow; gm = gm < p.M ? gm : p.M - 1;
        pa[i] = p.A1 + remap_row(gm, p.rows_per_batch, p.a_batch_rows, p.a_row_off) * p.lda1 + sc;
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        int gn = in0 + (lw + i * NLD) * RPP + srow; gn = gn < p.N ? gn : p.N - 1;
        pb[i] = p.B1 + (int64_t)gn * p.ldb1 + sc;
      }
    };
    bool more = true;
#if defined(QFX_GEMM_ABL_DMA_TO_VGPR)
    u32x4 abl_sink = {0u, 0u, 0u, 0u};
#endif
    auto issue = [&]() {
      if (it == int1) {  // first K tile of the LoRA segment (A2 rows are never remapped)
#pragma unroll
        for (int i = 0; i < NA; ++i) {
          int gm = im0 + (lw + i * NLD) * RPP + srow; gm = gm < iM ? gm : iM - 1;
          pa[i] = iA2 + (int64_t)gm * ilda2 + sc;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          int gn = in0 + (lw + i * NLD) * RPP + srow; gn = gn < iN ? gn : iN - 1;
          pb[i] = iB2 + (int64_t)gn * ildb2 + sc;
        }
      }
      char* sA = smem + ist * STAGE_BYTES;
      char* sB = sA + BMT * BKT * 2;
#if QFX_GEMM_KSTAGGER > 0
      int kt_ = it < int1 ? it + ikt0 : it - int1;
      if (it < int1 && kt_ >= int1) kt_ -= int1;
      const int koff = kt_ * BKT;
#else
      const int koff = (it < int1 ? it + (SPLIT ? ikt0 : 0) : it - int1) * BKT;
#endif
#if defined(QFX_GEMM_ABL_HALF_DMA)   // ablation (results are garbage): every other DMA piece -- is the K loop bound by the L2 -> LDS stream?
#pragma unroll
      for (int i = 0; i < NA; i += 2) glds16(pa[i] + koff, sA + (lw + i * NLD) * 1024);
#pragma unroll
      for (int i = 0; i < NB; i += 2) glds16(pb[i] + koff, sB + (lw + i * NLD) * 1024);
#elif defined(QFX_GEMM_ABL_DMA_TO_VGPR)   // ablation (garbage results): the same requests as plain 16-byte loads into a register nobody reads -- is the
                                         // stream bound on the L2 -> vector-memory path, or on the LDS side of the LDS-DMA?  (_B: only the weight half)
#pragma unroll
      for (int i = 0; i < NA; ++i) {
#if defined(QFX_GEMM_ABL_DMA_TO_VGPR_B)
        glds16(pa[i] + koff, sA + (lw + i * NLD) * 1024);
#else
        asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(abl_sink) : "v"(pa[i] + koff) : "memory");
#endif
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(abl_sink) : "v"(pb[i] + koff) : "memory");
#else
#pragma unroll
      for (int i = 0; i < NA; ++i) glds16(pa[i] + koff, sA + (lw + i * NLD) * 1024);
#pragma unroll
      for (int i = 0; i < NB; ++i) glds16(pb[i] + koff, sB + (lw + i * NLD) * 1024);
#endif
      ist = ist + 1 == NSTAGE ? 0 : ist + 1;
      if (++it == intt) {
        it = 0; ibid += gridDim.x; more = ibid < nwg;
        if (more) setp(ibid);
      }
    };
    setp(ibid);
    int ahead = 0;  // K tiles issued and not yet handed over
    issue(); ++ahead;
#pragma unroll
    for (int k = 2; k < NSTAGE; ++k)                 // an NSTAGE ring holds NSTAGE - 1 stages ahead
      if (more) { issue(); ++ahead; }
    for (int wbid = blockIdx.x; wbid < nwg; wbid += gridDim.x) {
      int gi, m0, n0, whalf = 0;
      tile_coord<BMT, TN, SPLIT>(ga, nwg, wbid, gi, m0, n0, nullptr, nullptr, &whalf);
      int ntw = ga.g[gi].K1 / BKT + ga.g[gi].K2 / BKT;
      if constexpr (SPLIT) {
        const int n1 = ga.g[gi].K1 / BKT, kh = n1 / 2 - ga.kh_bias;
        ntw = whalf ? n1 - kh : kh + ga.g[gi].K2 / BKT;
      }
      for (int t = 0; t < ntw; ++t) {
        // the oldest K stage in flight must have landed; the ones issued after it (PPS pieces each) may still be in flight
#if defined(QFX_GEMM_ABL_HALF_DMA)
        constexpr int PPS = (NA + 1) / 2 + (NB + 1) / 2;
#else
        constexpr int PPS = NA + NB;
#endif
        if (NSTAGE > 3 && ahead >= 3) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(2 * PPS) : "memory");
        else if (NSTAGE > 2 && ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(PPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        --ahead;
        if (more) { issue(); ++ahead; }   // into the stage every compute wave left before this barrier
      }
    }
#if defined(QFX_GEMM_ABL_DMA_TO_VGPR)
    asm volatile("s_waitcnt vmcnt(0)" :: "v"(abl_sink) : "memory");
#endif
    return;
  }

  // ================================================================== compute waves
  const int wr = w / TC::WCN, wc = w % TC::WCN;
  constexpr int WROWS = 16 * MI;   // rows per compute wave
  char* stg = smem + NSTAGE * STAGE_BYTES + w * STG_BYTES;
  int buf = 0;
  (void)lane;
  // destination of the (uncounted) prefetch loads: ONE register reserved for the whole persistent loop -- hipcc considers an asm
  // load's destination written when the statement ends and would hand the register to something else while the data is in flight
  unsigned pf_sink = 0;
  for (int bid = blockIdx.x; bid < nwg; bid += gridDim.x) {
    int gi, m0, n0, tin, tgsz, half = 0, gtile = 0;
    tile_coord<BMT, TN, SPLIT>(ga, nwg, bid, gi, m0, n0, &tin, &tgsz, &half, &gtile);
    KArgs& p = ga.g[gi];
    // Lane-derived values are re-derived per tile -- here for the K loop, once more for the epilogue -- from a lane id the optimiser
    // cannot see through (v_mbcnt in an asm volatile: neither hoisted nor spilled).  Kept in registers across the whole persistent
    // loop, these lane constants are what the allocator spills when the epilogue is register-hungry, and their reloads land INSIDE
    // the K loop (a scratch load + vmcnt wait per K tile).
    int l0;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l0));
    const int g = l0 >> 4, li = l0 & 15;
    // fragment offsets: the swizzle depends on li only (wave / fragment rows advance in multiples of 16), the k-step flips chunk
    // bit 2, i.e. XOR 64 on the byte offset
    const int swl = (li >> 1) & 7;
    const int offA0 = (wr * WROWS + li) * (BK * 2) + ((g ^ swl) << 4);
    const int offB0 = BMT * BK * 2 + (wc * WCOLS + li) * (BK * 2) + ((g ^ swl) << 4);
    int nt1 = p.K1 / BKT, nt2 = p.K2 / BKT;
    if constexpr (SPLIT) {
      const int kh = nt1 / 2 - ga.kh_bias;
      if (half) { nt1 -= kh; nt2 = 0; } else nt1 = kh;
    }
    const int nt = nt1 + nt2;
    // 64-byte stage rows (32-deep stages of the wide tile): chunk g of row li sits at g ^ (-(li >> 2) & 3), fragments are 1 KiB apart
    const int sw32 = (0 - (li >> 2)) & 3;
    const int offA32 = (wr * WROWS + li) * 64 + ((g ^ sw32) << 4);
    const int offB32 = BMT * 64 + (wc * WCOLS + li) * 64 + ((g ^ sw32) << 4);

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // L2 prefetch.  The K loop is bound by the loaders' L2 -> LDS stream, and that stream by latency: the ring holds at most two K
    // tiles in flight per CU and every K tile is a first touch for the XCD's L2 (an ablation with the compute waves idle still takes
    // 92 % of the launch: profiles/r04_gemm_loader_waves.json).  One dword load per 128-byte line, QFX_GEMM_PF_DIST K tiles ahead,
    // turns the loaders' misses into L2 hits.  The ~32 tiles an XCD runs at a time form an 8 (M) x 4 (N) patch sharing A panels four
    // ways and B panels eight ways, so a tile touches a quarter of its A rows (wave 0) and an eighth of its B rows (wave 1): one
    // wave-instruction per K tile each.  Only a hint: ragged patches leave some lines to the loaders' own misses.  The loads are
    // uncounted (asm) and their results unused.
    const bf16_t* pfp = nullptr;
    if constexpr (QFX_GEMM_PF_DIST > 0 && !FP8) {
      if (w == 0) {
        constexpr int SH = BMT / 4;
        int gm = m0 + ((tin / tgsz) & 3) * SH + (l0 < SH ? l0 : SH - 1); gm = gm < p.M ? gm : p.M - 1;
        pfp = p.A1 + remap_row(gm, p.rows_per_batch, p.a_batch_rows, p.a_row_off) * p.lda1;
      } else if (w == 1) {
        constexpr int SH = TN / 8;
        int gn = n0 + ((tin % tgsz) & 7) * SH + (l0 < SH ? l0 : SH - 1); gn = gn < p.N ? gn : p.N - 1;
        pfp = p.B1 + (int64_t)gn * p.ldb1;
      }
    }
    auto pf = [&](int kt) {
      if constexpr (QFX_GEMM_PF_DIST > 0 && !FP8) {
        if (w < 2 && kt < nt1) {
          asm volatile("global_load_dword %0, %1, off" : "+v"(pf_sink) : "v"(pfp + kt * BKT) : "memory");
        }
      }
    };
#pragma unroll 1
    for (int d = 2; d < QFX_GEMM_PF_DIST; ++d) pf(d);

    // narrow tiles -- rotated K loop: the second k-step of K tile t-1 is issued AFTER barrier t, under the first fragment reads
    // of tile t, so the matrix pipe does not drain while the post-barrier ds_reads are in flight (+2..6 % in the lab).
    // wide tiles -- streaming loop: per k-step the NI B fragments stay resident and the MI A fragments pass through a
    // three-deep ring, two reads ahead of the MFMAs that consume them (the accumulators leave no room to double-buffer).
    auto round_base = [&]() {
      // base nn.Linear output is a bf16 tensor in the reference: round (acc + bias) before the LoRA add
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int n = n0 + wc * WCOLS + ni * 16 + 4 * g;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias != nullptr && n + 3 < p.N) {
          const bf16x4 bb = *(const bf16x4*)(p.bias + n);
#pragma unroll
          for (int r = 0; r < 4; ++r) bv[r] = bf2f((bf16_t)bb[r]);
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[mi][ni][r] = rbf(acc[mi][ni][r] + bv[r]);
      }
    };
    const bool mid_round = nt2 > 0 && !p.seg2_plain;
    // ---- same-XCD split-K hand-off (SPLIT): compute wave w of the producer item hands its fp32 accumulators to compute wave w of the
    // consumer item through the XCD's L2.  Producer: plain stores, s_waitcnt vmcnt(0) (acknowledged by the L2), then the flag.  Consumer:
    // polls the flag (sc0 sc1: past the L1), invalidates the L1 (the words may sit there from an earlier launch), reads the partial tile,
    // adds, clears the flag for the next launch on this stream.  Both items of a tile run in the same round of one grid (<= 256 items,
    // producer first), so the consumer never waits for a block that has not started; the poll gives up after ~2 s instead of hanging.
    auto publish = [&]() {
      int lp;      // a lane id of its own: computed from l0 the addresses below are hoisted above the K passes and cost accumulator spills there
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lp));
      float* dst = ga.ws + ((int64_t)gtile * 8 + w) * (MI * NI * 256) + lp * 4;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) *(f32x4*)(dst + (mi * NI + ni) * 256) = acc[mi][ni];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // every lane writes the same word (no divergent branch around an asm statement: the allocator spills around those)
      const unsigned one = 1u;
      asm volatile("global_store_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" :: "v"(ga.wflag + gtile * 8 + w), "v"(one) : "memory");
    };
    auto fixup = [&]() {
      unsigned* fl = ga.wflag + gtile * 8 + w;
      bool got = false;
      for (int spin = 0; spin < (1 << 21); ++spin) {      // scalar poll past the scalar cache: no vector registers in this loop
        unsigned v;
        asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(fl) : "memory");
        if (v != 0u) { got = true; break; }
        __builtin_amdgcn_s_sleep(8);
      }
      if (!got) __builtin_trap();      // a producer that never shows is a failed launch (HIP error on the stream), not a silently incomplete tile
      asm volatile("buffer_inv sc1" ::: "memory");      // words of the workspace may sit in this CU's L1 from an earlier launch
      int lp;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lp));
      const float* src = ga.ws + ((int64_t)gtile * 8 + w) * (MI * NI * 256) + lp * 4;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {       // one fragment row (4 loads) at a time: more do not fit beside the accumulators
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] += *(const f32x4*)(src + (mi * NI + ni) * 256);
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("" ::: "memory");
      const unsigned zero = 0u;
      asm volatile("global_store_dword %0, %1, off sc0 sc1" :: "v"(fl), "v"(zero) : "memory");
    };
#if defined(QFX_GEMM_ABL_NO_COMPUTE)   // ablation (garbage results): the compute waves keep the barrier protocol only -- how long does the operand stream alone take?
    const bool wave_dead = true;
#else
    const bool wave_dead = m0 + wr * WROWS >= p.M;   // wave-uniform
#endif
    // ---- dGELU epilogue, aux rows by LDS-DMA (round 6).  The epilogue reads one aux row segment per lane and pass (16 per wave and
    // 256 x 256 tile) -- first touches of a 60 MB tensor written a whole forward ago: HBM latency, 128 accumulators leave no registers to
    // request them ahead (profiles/r04_gemm_aux_prefetch.json), the launch ran 180 us against 165 for the GELU epilogue that writes twice
    // as much.  Now lane slot (j, lane) of pass p + 1 is requested one pass ahead by LDS-DMA into a landing buffer of the wave's own
    // (slot-linear: every lane reads back what it asked for; no registers in flight), pass 0 before the K loop.  Completion by counted
    // waits: the compute waves' only vector-memory operations are these requests and the epilogue's stores, ONE per lane slot, in a
    // fixed order (asm statements with memory clobbers pin it) -- see the wait in the epilogue.  Taken where every store of the wave is
    // issued (whole column range inside N, no row mask); everything else keeps the load inside the pass.
    [[maybe_unused]] char* auxb = smem + NSTAGE * STAGE_BYTES + 8 * STG_BYTES + (AUXDMA ? w * TC::AUXW : 0);
    bool use_dma = false;
    // all rows of a wave on the landing-buffer side lie in ONE sample: the C / aux row of row m is m + a wave-uniform delta (no division
    // per lane slot, nothing of the row map held in vector registers)
    [[maybe_unused]] int row_delta = 0;
    [[maybe_unused]] auto aux_issue = [&](int pass, int j, int lnx) {
      constexpr int CHX = 2 * NG, SLOTSX = 16 * CHX;
      const int slot = j * 64 + lnx;
      if (SLOTSX < 128 && slot >= SLOTSX) return;              // (the 48-column groups: 32 live lanes in the second piece)
      const int row = slot / CHX, ch = slot - row * CHX;
      const int mi_ = pass / NGRP, gq_ = pass - mi_ * NGRP;
      int m = m0 + wr * WROWS + mi_ * 16 + row; m = m < p.M ? m : p.M - 1;
      const int n = n0 + wc * WCOLS + ch * 8 + gq_ * (16 * NG);
      const int64_t arow = (int64_t)(m + (p.aux_unmapped ? 0 : row_delta));
      glds16_asm(p.aux + arow * p.ldaux + n, auxb + j * 1024);
    };
    if constexpr (AUXDMA) {
      const int r0 = m0 + wr * WROWS, r1 = (r0 + WROWS < p.M ? r0 + WROWS : p.M) - 1;
      use_dma = !wave_dead && p.row_mask == nullptr && (p.M & 15) == 0 && n0 + wc * WCOLS + WCOLS <= p.N && r1 >= r0 &&
                !(p.bias != nullptr && (nt2 == 0 || p.seg2_plain)) &&                           // (no bias left for the epilogue)
                (p.c_batch_rows == 0 || r0 / p.rows_per_batch == r1 / p.rows_per_batch);        // wave-uniform
      if (use_dma) row_delta = (int)(remap_row(r0, p.rows_per_batch, p.c_batch_rows, p.c_row_off) - r0);
      if constexpr (EPI == QFX_EPI_GATE_RES) use_dma = use_dma && r0 / p.rows_per_batch == r1 / p.rows_per_batch;      // one gate vector
      if (use_dma) { aux_issue(0, 0, l0); aux_issue(0, 1, l0); }
    }
    if constexpr (FP8) {
      // ---- MX-FP8 base segment: per K tile (128 fp8 per row) the four B operands (chunks g and g+4 of their rows = the two halves
      // of one scaled-MFMA operand) stay resident, the A operands stream one fragment row ahead; scales (one dword = the 4 MX
      // blocks of a row and K tile, tile-major: 16 rows = one 64-byte line) are fetched one K tile ahead.
      const char* sap = (const char*)ga.sa[gi];
      const char* sbp = (const char*)ga.sb[gi];
      const int64_t sa_tile = (int64_t)p.M * 4, sb_tile = (int64_t)p.N * 4;
      // 32-bit lane offsets into the (wave-uniform) scale arrays; rows beyond M / N clamp to the last row (their results are never stored)
      const int ar0 = m0 + wr * WROWS + li, br0 = n0 + wc * 64 + li;
      auto aoff = [&](int i) { const int r = ar0 + 16 * i; return (unsigned)((r < p.M ? r : p.M - 1) * 4); };
      auto boff = [&](int i) { const int r = br0 + 16 * i; return (unsigned)((r < p.N ? r : p.N - 1) * 4); };
      uint32_t sca[MI], scb[4];
#pragma unroll
      for (int i = 0; i < MI; ++i) sca[i] = *(const uint32_t*)(sap + aoff(i));
#pragma unroll
      for (int i = 0; i < 4; ++i) scb[i] = *(const uint32_t*)(sbp + boff(i));
      for (int t = 0; t < nt1; ++t) {
        // this lane group's scale bytes of tile t; the dwords of tile t+1 are requested right away into the same registers
        int sav[MI], sbv[4];
#pragma unroll
        for (int i = 0; i < MI; ++i) sav[i] = (int)((sca[i] >> (8 * g)) & 0xffu);
#pragma unroll
        for (int i = 0; i < 4; ++i) sbv[i] = (int)((scb[i] >> (8 * g)) & 0xffu);
        if (t + 1 < nt1) {
          const char* na = sap + (t + 1) * sa_tile;
          const char* nb = sbp + (t + 1) * sb_tile;
#pragma unroll
          for (int i = 0; i < MI; ++i) sca[i] = *(const uint32_t*)(na + aoff(i));
#pragma unroll
          for (int i = 0; i < 4; ++i) scb[i] = *(const uint32_t*)(nb + boff(i));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const char* st = smem + buf * STAGE_BYTES;
        if (wave_dead) { buf = buf + 1 == NSTAGE ? 0 : buf + 1; continue; }   // see the bf16 narrow-tile loop
        v8i32 b[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const u32x4 lo = *(const u32x4*)(st + offB0 + ni * (16 * BK * 2)), hi = *(const u32x4*)(st + ((offB0 + ni * (16 * BK * 2)) ^ 64));
#pragma unroll
          for (int j = 0; j < 4; ++j) { b[ni][j] = (int)lo[j]; b[ni][4 + j] = (int)hi[j]; }
        }
        u32x4 alo = *(const u32x4*)(st + offA0), ahi = *(const u32x4*)(st + (offA0 ^ 64));
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          v8i32 av;
#pragma unroll
          for (int j = 0; j < 4; ++j) { av[j] = (int)alo[j]; av[4 + j] = (int)ahi[j]; }
          if (mi + 1 < MI) {
            alo = *(const u32x4*)(st + offA0 + (mi + 1) * (16 * BK * 2));
            ahi = *(const u32x4*)(st + ((offA0 + (mi + 1) * (16 * BK * 2)) ^ 64));
          }
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b[ni], av, acc[mi][ni], 0, 0, 0, sbv[ni], 0, sav[mi]);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (mid_round && t == nt1 - 1) round_base();
        buf = buf + 1 == NSTAGE ? 0 : buf + 1;
      }
      // ---- bf16 LoRA K-extension tiles (plain loop: they are 1-3 K tiles)
      for (int t = nt1; t < nt; ++t) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const char* st = smem + buf * STAGE_BYTES;
        if (wave_dead) { buf = buf + 1 == NSTAGE ? 0 : buf + 1; continue; }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          bf16x8 a[MI], b[4];
#pragma unroll
          for (int i = 0; i < MI; ++i) a[i] = *(const bf16x8*)(st + ((offA0 + i * (16 * BK * 2)) ^ (kk << 6)));
#pragma unroll
          for (int i = 0; i < 4; ++i) b[i] = *(const bf16x8*)(st + ((offB0 + i * (16 * BK * 2)) ^ (kk << 6)));
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[ni], a[mi], acc[mi][ni], 0, 0, 0);
        }
        buf = buf + 1 == NSTAGE ? 0 : buf + 1;
      }
    } else if constexpr (!TC::WIDE) {
      bf16x8 a0[MI], b0[NI], a1[MI], b1[NI];
      for (int t = 0; t < nt; ++t) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every fragment read of tile t-1 has returned: its stage may be refilled
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const char* st = smem + buf * STAGE_BYTES;
        pf(t + QFX_GEMM_PF_DIST);
        auto rdA = [&](int kk, int mi) { return *(const bf16x8*)(st + ((offA0 + mi * (16 * BK * 2)) ^ (kk << 6))); };
        auto rdB = [&](int kk, int ni) { return *(const bf16x8*)(st + ((offB0 + ni * (16 * BK * 2)) ^ (kk << 6))); };
        // a compute wave whose rows all lie past M (the text stream's last, partly empty tile) keeps
        // the barrier protocol but issues no fragment reads / MFMAs: nothing of it is ever stored, and under the package power
        // cap the saved energy is time (sustained step 99.79 -> 99.25 ms, tools/step_lib_ab.py; invisible in burst timings)
        if (wave_dead) { buf = buf + 1 == NSTAGE ? 0 : buf + 1; continue; }
#pragma unroll
        for (int i = 0; i < MI; ++i) a0[i] = rdA(0, i);
#pragma unroll
        for (int i = 0; i < NI; ++i) b0[i] = rdB(0, i);
        __builtin_amdgcn_sched_barrier(0);
        if (t > 0) {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[ni], a1[mi], acc[mi][ni], 0, 0, 0);
          if (mid_round && t == nt1) round_base();
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[ni], a0[mi], acc[mi][ni], 0, 0, 0);
          a1[mi] = rdA(1, mi);
          if (mi < NI) b1[mi] = rdB(1, mi);
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (NI > MI) {
#pragma unroll
          for (int ni = MI; ni < NI; ++ni) b1[ni] = rdB(1, ni);
        }
        buf = buf + 1 == NSTAGE ? 0 : buf + 1;
      }
      if (!wave_dead) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[ni], a1[mi], acc[mi][ni], 0, 0, 0);
      }
    } else if constexpr (BKT == 32) {
      // wide tile, 32-deep stages: one k-step per stage; the NI B fragments stay resident, the MI A fragments pass through the
      // three-deep register ring as below
      for (int t = 0; t < nt; ++t) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const char* st = smem + buf * STAGE_BYTES;
        pf(t + QFX_GEMM_PF_DIST);
        if (wave_dead) { buf = buf + 1 == NSTAGE ? 0 : buf + 1; continue; }   // see the narrow-tile loop
        const char* pA = st + offA32;
        const char* pB = st + offB32;
        bf16x8 b[NI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) b[ni] = *(const bf16x8*)(pB + ni * 1024);
        bf16x8 fa0 = *(const bf16x8*)(pA), fa1 = *(const bf16x8*)(pA + 1024), fa2;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          if (mi + 2 < MI) fa2 = *(const bf16x8*)(pA + (mi + 2) * 1024);
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[ni], fa0, acc[mi][ni], 0, 0, 0);
          fa0 = fa1; fa1 = fa2;
          __builtin_amdgcn_sched_barrier(0);
        }
        if (mid_round && t == nt1 - 1) round_base();
        buf = buf + 1 == NSTAGE ? 0 : buf + 1;
      }
    } else {
      // SPLIT: the base segment and the LoRA segment run as two passes of this loop with the hand-off between them, so that its
      // temporaries are not live inside the K loop (inside it they cost 450 bytes of scratch per lane)
      auto kpass = [&](int t0, int t1) __attribute__((always_inline)) {
        for (int t = t0; t < t1; ++t) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          const char* st = smem + buf * STAGE_BYTES;
          pf(t + QFX_GEMM_PF_DIST);
          if (wave_dead) { buf ^= 1; continue; }   // see the narrow-tile loop
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            const char* pA = st + (offA0 ^ (kk << 6));
            const char* pB = st + (offB0 ^ (kk << 6));
            bf16x8 b[NI];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) b[ni] = *(const bf16x8*)(pB + ni * (16 * BK * 2));
            bf16x8 fa0 = *(const bf16x8*)(pA), fa1 = *(const bf16x8*)(pA + 16 * BK * 2), fa2;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
              if (mi + 2 < MI) fa2 = *(const bf16x8*)(pA + (mi + 2) * (16 * BK * 2));
#pragma unroll
              for (int ni = 0; ni < NI; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[ni], fa0, acc[mi][ni], 0, 0, 0);
              fa0 = fa1; fa1 = fa2;
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          if constexpr (!SPLIT) { if (mid_round && t == nt1 - 1) round_base(); }
          buf ^= 1;
        }
      };
      if constexpr (SPLIT) {
        kpass(0, nt1);
        if (half == 0) {
#if !defined(QFX_SPLIT_ABL_NOFIX)                 // ablation (garbage results): no hand-off on the consumer side
          if (!wave_dead) fixup();                 // the whole base sum, before it is rounded
#endif
          if (mid_round && !wave_dead) round_base();
          kpass(nt1, nt);
        }
      } else {
        kpass(0, nt);
      }
    }
    if constexpr (SPLIT) {
      if (half) {                    // producer item: the partial tile leaves through the L2, no epilogue
#if !defined(QFX_SPLIT_ABL_NOPUB)                 // ablation: the producer keeps its partial tile to itself
        if (!wave_dead) publish();
#endif
        continue;
      }
    }

    // ---- epilogue (no block barrier): per 16-row pass and group of NG column fragments the wave stages bf16(acc + bias) -- the
    // nn.Linear output, first rounding point of every epilogue -- in its private 2 KiB of LDS (8-byte unit u = nn*4+g of row li
    // stored at u ^ 2*(li>>1): conflict-free writes, 16-byte pairs stay adjacent) and reads it back row-contiguous, so global
    // accesses are row segments of 16 * NG * 2 bytes (128 B; 96 B for the 48-column groups) instead of 8-byte pieces scattered
    // over 16 rows.
    const bool bias_pending = (p.bias != nullptr) && (nt2 == 0 || p.seg2_plain);
    int ln;     // the epilogue's own lane id (see the top of the tile loop)
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    const int ge = ln >> 4, lie = ln & 15;
    float bv[NI][4];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int n = n0 + wc * WCOLS + ni * 16 + 4 * ge;
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[ni][r] = 0.f;
      if (bias_pending && n + 3 < p.N) {
        const bf16x4 bb = *(const bf16x4*)(p.bias + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[ni][r] = bf2f((bf16_t)bb[r]);
      }
    }
    constexpr int CH = 2 * NG;                 // 16-byte chunks per staged row (8, or 6 for the 48-column groups)
    constexpr int SLOTS = 16 * CH;             // chunk slots per staging pass: 128 (two full lane passes) or 96 (64 + 32)
    constexpr bool GATE_CACHE = (NGRP == 1) && !TC::WIDE;   // narrow tiles: the gate vector of the lane's columns stays in registers across the tile (the wide tile has none to spare: 128 accumulators)
    constexpr int NGC = (64 % CH == 0) ? 1 : 2;   // CH = 8: the lane's chunk (lane & 7) is the same in both passes; CH = 6: one vector per pass
    float gt[NGC][8];
    int last_b[NGC];
#pragma unroll
    for (int i = 0; i < NGC; ++i) last_b[i] = -1;
    // MX-FP8 instantiation: the output the next GEMM contracts over (gelu(h) / dh / y) can leave the epilogue quantised -- the 8
    // lanes of a row hold 64 consecutive columns = 2 MX blocks of 4 lanes; same arithmetic as quant_mxfp8_kernel on the
    // bf16-rounded values, so the result is bit-identical to quantising the bf16 tensor in a separate pass.
    uint8_t* cq = nullptr; uint8_t* cs = nullptr; int64_t ldcq = 0; int cq_rows = 0; bool cq_only = false;
    if constexpr (FP8) { cq = ga.cq[gi]; cs = ga.cs[gi]; ldcq = ga.ldcq[gi]; cq_rows = ga.cq_rows[gi]; cq_only = ga.cq_only[gi] != 0; }
    auto quant_store = [&](const u32x4& packed, int64_t crow, int n, int ch) {
      float v[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) { v[2 * q] = __uint_as_float(packed[q] << 16); v[2 * q + 1] = __uint_as_float(packed[q] & 0xffff0000u); }
      float amax = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) amax = fmaxf(amax, fabsf(v[q]));
      amax = fmaxf(amax, __shfl_xor(amax, 1));
      amax = fmaxf(amax, __shfl_xor(amax, 2));
      int e = (int)((__float_as_uint(amax) >> 23) & 0xff) - 127 - 8;
      if (amax == 0.f) e = -127;
      e = e < -127 ? -127 : e;
      const float inv = __uint_as_float((uint32_t)(127 - e) << 23);
      uint32_t w0 = 0, w1 = 0;
      float qv[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) qv[q] = fminf(fmaxf(v[q] * inv, -448.f), 448.f);
      w0 = __builtin_amdgcn_cvt_pk_fp8_f32(qv[0], qv[1], w0, false);
      w0 = __builtin_amdgcn_cvt_pk_fp8_f32(qv[2], qv[3], w0, true);
      w1 = __builtin_amdgcn_cvt_pk_fp8_f32(qv[4], qv[5], w1, false);
      w1 = __builtin_amdgcn_cvt_pk_fp8_f32(qv[6], qv[7], w1, true);
      const u32x2 o = {w0, w1};
      *(u32x2*)(cq + crow * ldcq + n) = o;
      const int kb = n >> 5;
      if ((ch & 3) == 0) cs[((int64_t)(kb >> 2) * cq_rows + crow) * 4 + (kb & 3)] = (uint8_t)(e + 127);
    };
    // read-back slot of this lane in the two lane passes over a staged 16-row group: row, 16-byte chunk, LDS offset, first column
    int srow_[2], sch_[2], soff_[2], ncol_[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int slot = j * 64 + ln;
      srow_[j] = slot / CH; sch_[j] = slot - srow_[j] * CH;
      soff_[j] = srow_[j] * 128 + ((sch_[j] ^ (srow_[j] >> 1)) << 4);
      ncol_[j] = n0 + wc * WCOLS + sch_[j] * 8;
    }
    [[maybe_unused]] u32x4 gvp[NGC];
    if constexpr (AUXDMA) {
      if (use_dma) {
        // (no pending bias on this side: its NI x 4 registers are free during the passes); gate vector(s) of the lane's columns: ONE sample per wave here
        if constexpr (EPI == QFX_EPI_GATE_RES) {
          static_assert(NGRP == 1, "one column group per pass");
          const int bidx = (m0 + wr * WROWS) / p.rows_per_batch;
#pragma unroll
          for (int jc = 0; jc < NGC; ++jc) {
            gvp[jc] = *(const u32x4*)(p.gate + (int64_t)bidx * p.gate_bstride + ncol_[jc]);
            asm volatile("" : "+v"(gvp[jc]));      // (arrived before the first counted wait: hipcc puts its own wait here)
          }
        }
      }
    }
    // The passes exist twice where the landing buffer is compiled in: DMA = true is straight-line code (every lane slot stores: whole
    // 16-row groups inside M, whole column range inside N, no row mask), so the request/store order the counted waits rely on is the
    // program order; DMA = false is the general form.  One wave-uniform branch picks -- a merge INSIDE a pass made hipcc carry the
    // activation row through scratch and wait vmcnt(0) at the join, which drained the request just issued.
    auto epi_passes = [&](auto dma_tag) {
    constexpr bool DMA = decltype(dma_tag)::value;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      if constexpr (DMA) { if (m0 + wr * WROWS + mi * 16 >= p.M) return; }      // wave-uniform: no row of this or any later pass is stored
#pragma unroll
      for (int gq = 0; gq < NGRP; ++gq) {
#pragma unroll
        for (int nn = 0; nn < NG; ++nn) {
          const int ni = gq * NG + nn;
          u32x2 u;
          if constexpr (DMA) {      // (no pending bias on this side: its NI x 4 registers are dead)
            u[0] = pack2bf(acc[mi][ni][0], acc[mi][ni][1]);
            u[1] = pack2bf(acc[mi][ni][2], acc[mi][ni][3]);
          } else {
            u[0] = pack2bf(acc[mi][ni][0] + bv[ni][0], acc[mi][ni][1] + bv[ni][1]);
            u[1] = pack2bf(acc[mi][ni][2] + bv[ni][2], acc[mi][ni][3] + bv[ni][3]);
          }
          *(u32x2*)(stg + lie * 128 + (((nn * 4 + ge) ^ ((lie >> 1) << 1)) << 3)) = u;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (SLOTS < 128 && j * 64 + ln >= SLOTS) continue;
          int row = srow_[j], ch = sch_[j], soff = soff_[j], ncolj = ncol_[j], lnp = ln;
          if constexpr (DMA) {
            // re-derived per lane slot from a lane id hipcc cannot see through: otherwise the addresses of ALL passes are formed ahead
            // (the passes are unrolled), spill, and every reload's own wait drains the request just issued
            asm volatile("" : "+v"(lnp));
            const int slot = j * 64 + lnp;
            row = slot / CH; ch = slot - row * CH;
            soff = row * 128 + ((ch ^ (row >> 1)) << 4);
            ncolj = n0 + wc * WCOLS + ch * 8;
          }
          const u32x4 yv = *(const u32x4*)(stg + soff);
          const int m = m0 + wr * WROWS + mi * 16 + row;
          const int n = ncolj + gq * (16 * NG);
          if constexpr (!DMA) { if (m >= p.M || n + 7 >= p.N) continue; }
          const int64_t crow = DMA ? (int64_t)(m + row_delta) : remap_row(m, p.rows_per_batch, p.c_batch_rows, p.c_row_off);
          if (!DMA && p.row_mask != nullptr && p.row_mask[m] == 0.f) {
            const u32x4 z = {0u, 0u, 0u, 0u};
            *(u32x4*)(p.C + crow * p.ldc + n) = z;
            if constexpr (FP8 && EPI != QFX_EPI_GATE_RES) { if (cq) quant_store(z, crow, n, ch); }
            if constexpr (EPI == QFX_EPI_GELU) *(u32x4*)(p.C2 + crow * p.ldc2 + n) = z;
            if constexpr (EPI == QFX_EPI_GATE_RES) { if (p.C2) *(u32x4*)(p.C2 + (int64_t)m * p.ldc2 + n) = z; }
            continue;
          }
          float y[8];
#pragma unroll
          for (int q = 0; q < 4; ++q) { y[2 * q] = __uint_as_float(yv[q] << 16); y[2 * q + 1] = __uint_as_float(yv[q] & 0xffff0000u); }
          if constexpr (EPI == QFX_EPI_NONE) {
            if (!cq_only) *(u32x4*)(p.C + crow * p.ldc + n) = yv;
            if constexpr (FP8) { if (cq) quant_store(yv, crow, n, ch); }
          } else if constexpr (EPI == QFX_EPI_GELU) {
            u32x4 o2;
#pragma unroll
            for (int q = 0; q < 4; ++q) o2[q] = pack2bf(gelu_tanh_f(y[2 * q]), gelu_tanh_f(y[2 * q + 1]));
            if constexpr ((QFX_GEMM_NT_SAVED & 1) != 0) __builtin_nontemporal_store(yv, (u32x4*)(p.C + crow * p.ldc + n));      // h is next read a whole forward later
            else *(u32x4*)(p.C + crow * p.ldc + n) = yv;
            if (!cq_only) *(u32x4*)(p.C2 + crow * p.ldc2 + n) = o2;
            if constexpr (FP8) { if (cq) quant_store(o2, crow, n, ch); }
          } else if constexpr (EPI == QFX_EPI_GATE_RES) {
            float gl[8];
            const int jc = NGC == 1 ? 0 : j;
            u32x4 rv;
            if constexpr (DMA) {
              // one sample per wave on this side: the gate vector of the lane's columns was fetched before the passes (packed: 4 registers)
#pragma unroll
              for (int q = 0; q < 4; ++q) { gl[2 * q] = __uint_as_float(gvp[jc][q] << 16); gl[2 * q + 1] = __uint_as_float(gvp[jc][q] & 0xffff0000u); }
              // counted wait as in the d(GELU) branch below; with the second output (p.C2) every lane slot stores twice
              const int pass = mi * NGRP + gq;
              if (p.C2 == nullptr) {
                if (pass == 0 && j == 0) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                else if (pass == 0) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
              } else {
                if (pass == 0 && j == 0) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                else if (pass == 0) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
              }
              rv = *(const u32x4*)(auxb + j * 1024 + lnp * 16);
              asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rv) :: "memory");
              constexpr int npass = MI * NGRP;
              aux_issue(pass + 1 < npass ? pass + 1 : npass - 1, j, lnp);
            } else {
            const int bidx = m / p.rows_per_batch;
            if (!GATE_CACHE || bidx != last_b[jc]) {
              const u32x4 gv = *(const u32x4*)(p.gate + (int64_t)bidx * p.gate_bstride + n);
#pragma unroll
              for (int q = 0; q < 4; ++q) { gl[2 * q] = __uint_as_float(gv[q] << 16); gl[2 * q + 1] = __uint_as_float(gv[q] & 0xffff0000u); }
              if (GATE_CACHE) {
#pragma unroll
                for (int q = 0; q < 8; ++q) gt[jc][q] = gl[q];
                last_b[jc] = bidx;
              }
            } else {
#pragma unroll
              for (int q = 0; q < 8; ++q) gl[q] = gt[jc][q];
            }
            // (aux rows are loaded where they are used: requesting them one to MI passes ahead was measured 1-2 % SLOWER on the whole
            // step -- the extra live registers spill, and other waves already cover the round trip: profiles/r04_gemm_aux_prefetch.json)
            rv = *(const u32x4*)(p.aux + (p.aux_unmapped ? (int64_t)m : crow) * p.ldaux + n);
            }
            u32x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float r0 = __uint_as_float(rv[q] << 16), r1 = __uint_as_float(rv[q] & 0xffff0000u);
              o[q] = pack2bf(r0 + rbf(gl[2 * q] * y[2 * q]), r1 + rbf(gl[2 * q + 1] * y[2 * q + 1]));
            }
            *(u32x4*)(p.C + crow * p.ldc + n) = o;
            if (p.C2) {      // pre-gate linear output (rows UNMAPPED), kept for d(gate)
              if constexpr ((QFX_GEMM_NT_SAVED & 2) != 0) __builtin_nontemporal_store(yv, (u32x4*)(p.C2 + (int64_t)m * p.ldc2 + n));
              else *(u32x4*)(p.C2 + (int64_t)m * p.ldc2 + n) = yv;
            }
          } else {  // QFX_EPI_DGELU
            u32x4 hv;
            if constexpr (DMA) {
              {
                // operations this wave issued AFTER the request being awaited (in issue order, one instruction each): pass 0, j = 0: the
                // request of slot (0, 1); pass 0, j = 1: request (1, 0), store (0, 0); every later slot: the other slot's store of the
                // previous pass, the other slot's request, one more store -- three.  VMEM retires in order: vmcnt(N) = "all but the N
                // youngest are done".
                const int pass = mi * NGRP + gq;      // (a constant once the pass loops are unrolled: the branches below fold)
                if (pass == 0 && j == 0) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                else if (pass == 0) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                hv = *(const u32x4*)(auxb + j * 1024 + lnp * 16);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(hv) :: "memory");      // the slot is read: it may be overwritten
                constexpr int npass = MI * NGRP;
                aux_issue(pass + 1 < npass ? pass + 1 : npass - 1, j, lnp);          // (the last pass re-requests itself: the count stays 3)
              }
            } else {
              hv = *(const u32x4*)(p.aux + (p.aux_unmapped ? (int64_t)m : crow) * p.ldaux + n);
            }
            u32x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float h0 = __uint_as_float(hv[q] << 16), h1 = __uint_as_float(hv[q] & 0xffff0000u);
              o[q] = pack2bf(y[2 * q] * gelu_tanh_grad_f(h0), y[2 * q + 1] * gelu_tanh_grad_f(h1));
            }
            if (!cq_only) *(u32x4*)(p.C + crow * p.ldc + n) = o;
            if constexpr (FP8) { if (cq) quant_store(o, crow, n, ch); }
          }
        }
      }
    }
    };
    if constexpr (AUXDMA) {
      if (use_dma) epi_passes(std::true_type{});
      else epi_passes(std::false_type{});
    } else {
      epi_passes(std::false_type{});
    }
  }
  asm volatile("" :: "v"(pf_sink));
}

bool ok256(const qfx_gemm_args* a) {
  if ((a->N % 8) || (a->ldc % 8)) return false;
  if (a->epi == QFX_EPI_GELU && (a->ldc2 % 8)) return false;
  if (a->epi == QFX_EPI_GATE_RES && a->C2 && (a->ldc2 % 8)) return false;
  if ((a->epi == QFX_EPI_GATE_RES || a->epi == QFX_EPI_DGELU) && (a->ldaux % 8)) return false;
  if (a->epi == QFX_EPI_GATE_RES && (a->gate_bstride % 8)) return false;
  return true;
}

int validate(const qfx_gemm_args* a) {
  if (!a->A1 || !a->B1 || !a->C) return QFX_EINVAL;
  if (a->M <= 0 || a->N <= 0 || a->K1 <= 0 || (a->K1 % BK) != 0 || (a->K2 % BK) != 0 || a->K2 < 0) return QFX_EINVAL;
  if ((a->N % 4) != 0 || (a->lda1 % 8) || (a->ldb1 % 8) || (a->ldc % 4)) return QFX_EINVAL;
  if (a->K2 > 0 && (!a->A2 || !a->B2 || (a->lda2 % 8) || (a->ldb2 % 8))) return QFX_EINVAL;
  if (a->rows_per_batch <= 0) return QFX_EINVAL;
  if (a->epi == QFX_EPI_GELU && (!a->C2 || (a->ldc2 % 4))) return QFX_EINVAL;
  if (a->epi == QFX_EPI_GATE_RES && (!a->gate || !a->aux || (a->ldaux % 4))) return QFX_EINVAL;
  if (a->epi == QFX_EPI_DGELU && (!a->aux || (a->ldaux % 4))) return QFX_EINVAL;
  if (a->epi < 0 || a->epi > 3) return QFX_EUNSUPPORTED;
  return QFX_OK;
}

// ---- tile geometry choice ------------------------------------------------------------------------------------------------
// cost(geometry) = rounds over the 256 CUs x relative time of one tile.  The relative tile times are the tile areas scaled by a
// measured per-flop efficiency (256x128 = 1): the 256-wide tile moves a third less LDS-DMA and a quarter less fragment traffic
// per flop (+9 %: round 1); the 160-row tile was measured in round 4 (profiles/r04_gemm_tiles.json).  QFX_GEMM_TILES selects
// the candidate set ("legacy" = the 256-row tiles of rounds 1-3; a comma list of BMTxTN names; default: all); QFX_GEMM_EFF
// overrides the three efficiency factors (A/B experiments).  Both are read once per process.
struct Geo { int bmt, tn; double area, eff; bool on; };
constexpr int NGEO = 3;
struct GeoTable { Geo g[NGEO]; bool split_on; int split_min_k, split_bias; };
// Process-wide tuning state: guarded by a mutex, initialised exactly once from the environment, and every launch works on
// its own SNAPSHOT of the table -- qfx_gemm_tune() from one thread can no longer tear the table under a launch from another
// (side-stream / data-parallel hook threads launch GEMMs concurrently with the main thread).
// The same-XCD split-K policy lives in the same table (it used to be three loose globals read without the lock):
// QFX_GEMM_SPLITK = 0 / 1, QFX_GEMM_SPLITK_MINK = smallest base K taken, QFX_GEMM_SPLITK_BIAS = K tiles the consumer's half is shorter by.
#if defined(QFX_GEMM_SPLITK_DEFAULT_ON)      // A/B builds (tools/build_variants.py)
#define QFX_SPLIT_DEFAULT true
#else
#define QFX_SPLIT_DEFAULT false
#endif
GeoTable g_geo_tab = {{
    {256, 128, 1.0, 1.00, true},
    {256, 256, 2.0, 1.09, true},
    {160, 192, 0.9375, 0.965, true},      // round 4: 0.9375 of the work at ~0.965 of the per-flop speed (profiles/r04_gemm_tiles.json)
}, QFX_SPLIT_DEFAULT, 9216, 3};
std::mutex g_geo_mu;
std::once_flag g_geo_once;

// `tiles`: "legacy" | "all" | comma list of exact BMTxTN names; `eff`: three comma-separated factors.  All-or-nothing: a value
// that does not parse leaves the table untouched and returns QFX_EINVAL.  Caller holds g_geo_mu.
int geo_set_locked(const char* tiles, const char* eff) {
  GeoTable t = g_geo_tab;
  if (tiles && !strncmp(tiles, "splitk", 6)) {      // "splitk=0|1", "splitk_mink=<K>", "splitk_bias=<K tiles>": the split-K policy (A/B lever, tests)
    int v = 0;
    if (sscanf(tiles, "splitk=%d", &v) == 1 && (v == 0 || v == 1)) t.split_on = v != 0;
    else if (sscanf(tiles, "splitk_mink=%d", &v) == 1 && v >= 2048) t.split_min_k = v;
    else if (sscanf(tiles, "splitk_bias=%d", &v) == 1 && v >= 0 && v <= 16) t.split_bias = v;
    else return QFX_EINVAL;
    tiles = nullptr;      // (the efficiency factors of the same call are still parsed below; the table is committed once, at the end)
  }
  if (tiles && *tiles) {
    const std::string v(tiles);
    if (v == "legacy") { for (int i = 0; i < NGEO; ++i) t.g[i].on = i < 2; }
    else if (v == "all") { for (auto& gg : t.g) gg.on = true; }
    else {
      bool on[NGEO] = {false, false, false};
      size_t pos = 0;
      while (pos <= v.size()) {                   // exact comma-split tokens (a substring match would accept "1256x1280")
        const size_t e = v.find(',', pos);
        const std::string tok = v.substr(pos, e == std::string::npos ? std::string::npos : e - pos);
        bool known = false;
        for (int i = 0; i < NGEO; ++i) {
          char name[32];
          snprintf(name, sizeof name, "%dx%d", t.g[i].bmt, t.g[i].tn);
          if (tok == name) { on[i] = true; known = true; }
        }
        if (!known) return QFX_EINVAL;
        if (e == std::string::npos) break;
        pos = e + 1;
      }
      for (int i = 0; i < NGEO; ++i) t.g[i].on = on[i];
    }
  }
  if (eff && *eff) {
    double e[NGEO];
    char tail = 0;
    if (sscanf(eff, "%lf,%lf,%lf%c", &e[0], &e[1], &e[2], &tail) != NGEO) return QFX_EINVAL;
    for (int i = 0; i < NGEO; ++i) if (!(e[i] > 0.1 && e[i] < 10.0)) return QFX_EINVAL;
    for (int i = 0; i < NGEO; ++i) t.g[i].eff = e[i];
  }
  g_geo_tab = t;
  return QFX_OK;
}

void geo_init() {
  std::call_once(g_geo_once, [] {
    std::lock_guard<std::mutex> lk(g_geo_mu);
    const char* tiles = getenv("QFX_GEMM_TILES");
    const char* eff = getenv("QFX_GEMM_EFF");
    if (geo_set_locked(tiles, nullptr) != QFX_OK)
      fprintf(stderr, "libqfx: QFX_GEMM_TILES=\"%s\" not understood (legacy | all | comma list of 256x128,256x256,160x192): ignored\n", tiles);
    if (geo_set_locked(nullptr, eff) != QFX_OK)
      fprintf(stderr, "libqfx: QFX_GEMM_EFF=\"%s\" not understood (three factors in (0.1, 10)): ignored\n", eff);
    if (const char* e = getenv("QFX_GEMM_SPLITK")) g_geo_tab.split_on = e[0] != '0';
    if (const char* e = getenv("QFX_GEMM_SPLITK_MINK")) { const int v = atoi(e); if (v >= 2048) g_geo_tab.split_min_k = v; }
    if (const char* e = getenv("QFX_GEMM_SPLITK_BIAS")) { const int v = atoi(e); if (v >= 0 && v <= 16) g_geo_tab.split_bias = v; }
  });
}

GeoTable geo_snapshot() {
  geo_init();
  std::lock_guard<std::mutex> lk(g_geo_mu);
  return g_geo_tab;
}

// ---- same-XCD split-K workspace: one per stream (launches on a stream are ordered, so a workspace is never shared by two live
// grids), grown on demand, kept for the life of the process
struct SplitWs { float* ws = nullptr; unsigned* flag = nullptr; size_t tiles = 0; };
std::mutex g_ws_mu;
std::map<std::pair<int, hipStream_t>, SplitWs> g_ws;      // (device, stream): the null stream of two devices must not share a workspace
bool split_ws(hipStream_t s, size_t tiles, float** ws, unsigned** flag) {
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return false;   // no allocation inside a graph capture: unsplit
  std::lock_guard<std::mutex> lk(g_ws_mu);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  SplitWs& w = g_ws[std::make_pair(dev, s)];
  if (w.tiles < tiles) {
    // (an old, smaller workspace may still be read by a grid in flight on this stream: it is left allocated, not freed)
    float* nw = nullptr; unsigned* nf = nullptr;
    if (hipMalloc((void**)&nw, tiles * 256 * 256 * sizeof(float)) != hipSuccess) return false;
    if (hipMalloc((void**)&nf, tiles * 8 * sizeof(unsigned)) != hipSuccess) { (void)hipFree(nw); return false; }
    if (hipMemsetAsync(nf, 0, tiles * 8 * sizeof(unsigned), s) != hipSuccess) { (void)hipFree(nw); (void)hipFree(nf); return false; }
    w.ws = nw; w.flag = nf; w.tiles = tiles;
  }
  // every split launch starts from cleared flags (a launch that died mid-way must not leave a stale 'ready' behind)
  if (hipMemsetAsync(w.flag, 0, tiles * 8 * sizeof(unsigned), s) != hipSuccess) return false;
  *ws = w.ws; *flag = w.flag;
  return true;
}

template <int E>
void launch_split(int grid, hipStream_t s, const GroupedArgs& ga) {
  hipLaunchKernelGGL((gemm256_kernel<E, 256, 256, false, true>), dim3(grid), dim3(WS_THREADS), 0, s, ga);
}

template <int E, bool FP8>
void launch_geo(int gi, int grid, hipStream_t s, const GroupedArgs& ga) {
  if constexpr (FP8) {
    hipLaunchKernelGGL((gemm256_kernel<E, 256, 128, true>), dim3(grid), dim3(WS_THREADS), 0, s, ga);
  } else {
    switch (gi) {
      case 0: hipLaunchKernelGGL((gemm256_kernel<E, 256, 128>), dim3(grid), dim3(WS_THREADS), 0, s, ga); break;
      case 1: hipLaunchKernelGGL((gemm256_kernel<E, 256, 256>), dim3(grid), dim3(WS_THREADS), 0, s, ga); break;
      default: hipLaunchKernelGGL((gemm256_kernel<E, 160, 192>), dim3(grid), dim3(WS_THREADS), 0, s, ga); break;
    }
  }
}

template <bool FP8>
int launch_grouped(GroupedArgs& ga, const qfx_gemm_args* probs[], int n, int geo, int epi, hipStream_t s, bool split = false, float* ws = nullptr,
                   unsigned* wflag = nullptr, int split_bias = 0) {
  constexpr int GEO_BMT[NGEO] = {256, 256, 160}, GEO_TN[NGEO] = {128, 256, 192};      // = the table's tile shapes (compile-time constants, never tuned)
  const int bmt = GEO_BMT[geo], tn = GEO_TN[geo];
  const int per_tile = split ? 2 : 1;          // work items per tile
  int tiles = 0;
  for (int i = 0; i < n; ++i) {
    ga.tile_start[i] = tiles;
    tiles += per_tile * ((probs[i]->M + bmt - 1) / bmt) * ((probs[i]->N + tn - 1) / tn);
  }
  for (int i = n; i <= QFX_MAX_GROUPS; ++i) ga.tile_start[i] = tiles;
  ga.n = n;
  // balanced persistent grid: whole rounds over <= 256 CUs, multiple of 8 (one residue class per XCD)
  const int rounds = (tiles + QFX_NUM_CU - 1) / QFX_NUM_CU;
  int grid = (((tiles + rounds - 1) / rounds) + 7) & ~7;
  if (grid > QFX_NUM_CU) grid = QFX_NUM_CU;
  if (grid > tiles) grid = tiles;
  ga.ws = nullptr; ga.wflag = nullptr; ga.kh_bias = 0;
  ga.gm = (n >= 6 && !split) ? 4 : 8;
  if constexpr (!FP8) {
    if (split) {
      ga.ws = ws; ga.wflag = wflag; ga.kh_bias = split_bias;
      switch (epi) {                     // one round: every item has its own CU, the two items of a tile are neighbours in one XCD
        case QFX_EPI_NONE: launch_split<QFX_EPI_NONE>(tiles, s, ga); break;
        case QFX_EPI_GELU: launch_split<QFX_EPI_GELU>(tiles, s, ga); break;
        case QFX_EPI_GATE_RES: launch_split<QFX_EPI_GATE_RES>(tiles, s, ga); break;
        default: launch_split<QFX_EPI_DGELU>(tiles, s, ga); break;
      }
      QFX_CHECK_LAUNCH();
      return QFX_OK;
    }
  }
  switch (epi) {
    case QFX_EPI_NONE: launch_geo<QFX_EPI_NONE, FP8>(geo, grid, s, ga); break;
    case QFX_EPI_GELU: launch_geo<QFX_EPI_GELU, FP8>(geo, grid, s, ga); break;
    case QFX_EPI_GATE_RES: launch_geo<QFX_EPI_GATE_RES, FP8>(geo, grid, s, ga); break;
    default: launch_geo<QFX_EPI_DGELU, FP8>(geo, grid, s, ga); break;
  }
  QFX_CHECK_LAUNCH();
  return QFX_OK;
}

}  // namespace

extern "C" int qfx_gemm_tune(const char* tiles, const char* eff) {
  geo_init();
  std::lock_guard<std::mutex> lk(g_geo_mu);
  return geo_set_locked(tiles, eff);      // both arguments are parsed into ONE temporary table, committed once (all or nothing)
}

extern "C" int qfx_gemm_grouped(const qfx_gemm_args* groups, int32_t n, void* stream) {
  if (!groups || n <= 0 || n > QFX_MAX_GROUPS) return QFX_EINVAL;
  GroupedArgs ga;
  const qfx_gemm_args* probs[QFX_MAX_GROUPS];
  bool n256 = true;
  for (int i = 0; i < n; ++i) {
    const int rc = validate(&groups[i]);
    if (rc) return rc;
    if (groups[i].epi != groups[0].epi) return QFX_EINVAL;
    if (!ok256(&groups[i])) return QFX_EINVAL;  /* 16-byte epilogue accesses */
    n256 = n256 && (groups[i].N % 256) == 0;
    ga.g[i] = groups[i];
    probs[i] = &groups[i];
  }
  const GeoTable geo = geo_snapshot();
  // e.g. B = 1, N = 3072: 240 tiles of 256x128 (one round, cost 1.0) vs 256 tiles of 160x192 (one round, 0.9375 / eff); the q/k/v
  // launch: 720 vs 768 narrow tiles (three rounds either way); N = 12288: 480 tiles of 256x256 (two rounds of 240).
  int best = -1;
  double best_cost = 0.0;
  for (int c = 0; c < NGEO; ++c) {
    const Geo& gg = geo.g[c];
    if (!gg.on) continue;
    if (gg.tn >= 256 && !n256) continue;     // the wide tile keeps whole tiles along N (round-1 contract)
    long t = 0;
    for (int i = 0; i < n; ++i) t += (long)((groups[i].M + gg.bmt - 1) / gg.bmt) * ((groups[i].N + gg.tn - 1) / gg.tn);
    const double cost = (double)((t + QFX_NUM_CU - 1) / QFX_NUM_CU) * gg.area / gg.eff;
    if (best < 0 || cost < best_cost) { best = c; best_cost = cost; }
  }
  if (best < 0) best = 0;
  // Same-XCD 2-way split-K on 256x256 tiles: the narrow launches (N = 3072 at B = 1) are confined to the 160x192 tile because 120
  // tiles of 256x256 fill under half the chip; split along K they are 240 work items -- one round -- and move a third fewer operand
  // bytes per flop (profiles/r05_gemm_splitk.json).  Taken when every problem has whole 256-column tiles and a base K of >= 9216, the
  // items fill one round in pairs that never straddle an XCD (items % 16 == 0), and the round is at least 85 % full.
  if (geo.split_on && n256 && geo.g[1].on) {
    long t256 = 0;
    bool deep = true;
    for (int i = 0; i < n; ++i) {
      t256 += (long)((groups[i].M + 255) / 256) * (groups[i].N / 256);
      deep = deep && groups[i].K1 >= geo.split_min_k && groups[i].K1 / 64 / 2 - geo.split_bias >= 1;      // both halves keep at least one K tile
    }
    const long items = 2 * t256;
    float* ws = nullptr;
    unsigned* wflag = nullptr;
    if (deep && items <= QFX_NUM_CU && items % 16 == 0 && items * 100 >= QFX_NUM_CU * 85 && split_ws((hipStream_t)stream, (size_t)t256, &ws, &wflag))
      return launch_grouped<false>(ga, probs, n, 1, groups[0].epi, (hipStream_t)stream, true, ws, wflag, geo.split_bias);
  }
  return launch_grouped<false>(ga, probs, n, best, groups[0].epi, (hipStream_t)stream);
}

// MX-FP8 operands on the warp-specialised persistent kernel (256x128 tiles): see GroupedArgs for how they reach the loaders.
extern "C" int qfx_gemm_mxfp8_grouped(const qfx_gemm_fp8_args* list, int32_t n, void* stream) {
  if (!list || n <= 0 || n > QFX_MAX_GROUPS) return QFX_EINVAL;
  GroupedArgs ga;
  const qfx_gemm_args* probs[QFX_MAX_GROUPS];
  for (int i = 0; i < n; ++i) {
    qfx_gemm_args g = list[i].g;
    if (!g.A1 || !g.B1 || !g.C || !list[i].sa || !list[i].sb) return QFX_EINVAL;
    if (g.M <= 0 || g.N <= 0 || g.K1 <= 0 || (g.K1 % 128) || (g.K2 % 64) || g.K2 < 0) return QFX_EINVAL;
    if ((g.lda1 % 16) || (g.ldb1 % 16) || g.a_batch_rows != 0 || g.seg2_plain) return QFX_EINVAL;
    g.lda1 /= 2; g.ldb1 /= 2; g.K1 /= 2;          // 2-byte units: the loaders see a bf16 operand of half the K extent
    const int rc = validate(&g);
    if (rc) return rc;
    if (g.epi != list[0].g.epi || !ok256(&g)) return QFX_EINVAL;
    if (list[i].cq) {   // quantised output image: whole 128-column tiles (4 MX blocks), 8-byte stores, scale rows cover every C row
      if (!list[i].cs || g.epi == QFX_EPI_GATE_RES || (g.N % 128) || (list[i].ldcq % 8) || list[i].ldcq < g.N ||
          list[i].cq_rows < (g.c_batch_rows ? (g.M / g.rows_per_batch) * g.c_batch_rows : g.M))
        return QFX_EINVAL;
    } else if (list[i].cq_only) {
      return QFX_EINVAL;
    }
    ga.g[i] = g;
    probs[i] = &ga.g[i];
    ga.sa[i] = list[i].sa; ga.sb[i] = list[i].sb;
    ga.cq[i] = list[i].cq; ga.cs[i] = list[i].cs; ga.ldcq[i] = list[i].ldcq; ga.cq_rows[i] = list[i].cq_rows; ga.cq_only[i] = list[i].cq_only;
  }
  for (int i = n; i < QFX_MAX_GROUPS; ++i) { ga.sa[i] = nullptr; ga.sb[i] = nullptr; ga.cq[i] = nullptr; ga.cs[i] = nullptr; ga.ldcq[i] = 0; ga.cq_rows[i] = 0; ga.cq_only[i] = 0; }
  return launch_grouped<true>(ga, probs, n, 0, list[0].g.epi, (hipStream_t)stream);
}

extern "C" int qfx_gemm_bf16(const qfx_gemm_args* a, void* stream) {
  if (!a) return QFX_EINVAL;
  const int rc = validate(a);
  if (rc) return rc;
  // large problems: persistent tiles; small ones keep the 128x128 kernel (more tiles, 2 blocks per CU)
  const int tiles256 = ((a->M + 255) / 256) * ((a->N + BN - 1) / BN);
  if (tiles256 >= 160 && ok256(a)) return qfx_gemm_grouped(a, 1, stream);
  const int tiles = ((a->M + BM - 1) / BM) * ((a->N + BN - 1) / BN);
  hipStream_t s = (hipStream_t)stream;
  switch (a->epi) {
    case QFX_EPI_NONE: hipLaunchKernelGGL(gemm_kernel<QFX_EPI_NONE>, dim3(tiles), dim3(256), 0, s, *a); break;
    case QFX_EPI_GELU: hipLaunchKernelGGL(gemm_kernel<QFX_EPI_GELU>, dim3(tiles), dim3(256), 0, s, *a); break;
    case QFX_EPI_GATE_RES: hipLaunchKernelGGL(gemm_kernel<QFX_EPI_GATE_RES>, dim3(tiles), dim3(256), 0, s, *a); break;
    default: hipLaunchKernelGGL(gemm_kernel<QFX_EPI_DGELU>, dim3(tiles), dim3(256), 0, s, *a); break;
  }
  QFX_CHECK_LAUNCH();
  return QFX_OK;
}
