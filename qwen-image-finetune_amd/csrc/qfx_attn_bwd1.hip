// qfx_attn_bwd1.hip -- ONE-PASS attention backward for gfx950 (dh = 128): dK, dV AND dQ from a single sweep over the score tiles.
//
// Replaces, for the autograd of the joint SDPA (transformer_qwenimage.py:329-337), the pair qfx_attn_bwd_dq + qfx_attn_bwd_dkv, which
// each recompute S = Q K^T, dP = dO V^T and the exponentials (7 S^2 D units of matrix work, two softmax passes); here 5 units, one pass.
//
//   * work item = (batch, head, 256-key block), 8 waves x 32 keys (the decomposition of attn_bwd_dkv_kernel): a wave keeps dK^T / dV^T
//     of its keys in 128 accumulator registers and its V rows as MFMA operands in 32 more; the block's K rows live in LDS (64 KB,
//     row-major, swizzled) and serve BOTH the S = Q K^T operand reads and the transposed reads of the dQ contraction;
//   * per 64-query tile (Q | dO double-buffered by LDS-DMA, 64 KB): S^T, dP^T, P, dS per wave exactly as before, dV += P^T dO,
//     dK += dS^T Q; the packed bf16 dS (what the dK MFMAs consume) is ALSO written to a [256 keys][64 queries] LDS tile (32 KB); after
//     one barrier every wave contracts a 32 d x 32 q block of dQ^T = K^T dS over all 256 keys of the block (transpose reads of both
//     operands, 32 MFMAs): the fifth unit of matrix work.  LDS = 64 + 64 + 32 KB = all 160 KB; the per-query statistics (lse2, dsum)
//     therefore ride in two registers per lane and reach their consumers by DPP row broadcasts;
//   * the block's partial dQ tile (64 x 128 fp32, 32 KB) is ACCUMULATED ACROSS THE KEY BLOCKS OF A HEAD in a fixed order through a
//     workspace in fragment order (coalesced 1 KB pieces): key block j starts at query tile floor(j ntiles / nkb) and walks the tiles in
//     rotation, so at any time the key blocks of a head work on different tiles; a per-(head, tile) turn counter hands the tile from one
//     key block to the next (arrival order = m, m-1, ..., 0, nkb-1, ..., m+1 with m = the last block that starts at or before the tile;
//     a block only ever waits for a block that passed the tile >= ntiles / nkb iterations earlier).  Same inputs -> same bits: no
//     atomics, no order-dependent sums.  Visibility between CUs (MI355X_MICROARCH, "Valid forms"): payload AND counter are written and
//     read with sc1 accesses only (buffer_load/store ... sc1, global_load/store_dword sc1), the counter store follows
//     s_waitcnt vmcnt(0) + barrier of all eight waves; correct for any placement of the blocks on XCDs;
//   * persistent grid: G = (whole heads per round) x nkb <= 256 blocks, one per CU, items handed out round by round, so every block a
//     turn counter can wait for is resident or becomes resident without anybody's help; spins are bounded (trap, not hang);
//   * the last word on dQ (scale, backward of QK RMSNorm + RoPE, bf16 rounding, fused rank-r projection: the dQ epilogue of the
//     two-pass kernels) is attn_dq_finish_kernel, a streaming pass over the fp32 workspace.
#include "qfx_attn_common.h"
#include <utility>

namespace {

constexpr int DH = 128, KC = DH / 32, DF = DH / 16, NW = 8;
constexpr int TB = 64 * DH * 2;                 // bytes of a 64-row tile
constexpr int DSROW = 128;                      // bytes of a dS row: 64 queries x bf16
constexpr int ACC_TILE = 64 * DH;               // floats of one dQ tile in the workspace
// Ablation switches of the timing builds (tools/build_variants.py x=-DBWD1_ABL=<mask>; wrong results): 1 = no turn wait / accumulated-tile
// load / add, 2 = no workspace stores, 4 = no dQ MFMA loop, 8 = no dS tile writes, 16 = no mid-tile barrier, 32 = no
// rotation (every key block starts at tile 0), 64 = no zeroing of dS for keys past S, 128 = statistics are constants (no DPP).  0 in the product.
#ifndef BWD1_ABL
#define BWD1_ABL 0
#endif
#ifndef BWD1_STORE_AUX
#define BWD1_STORE_AUX 16      // cache policy of the workspace stores: 16 = sc1 (agent scope: what the protocol needs); 0 / 1 / 2 = plain / sc0 / nt timing experiments
#endif
constexpr int SPIN_LIMIT = 1 << 20;      // ~1 s of polling: a turn that never comes is a trap, not a hang

template <class F, int... I> __device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void sfor(F&& f) { sfor_impl(f, std::make_integer_sequence<int, N>{}); }

__device__ __forceinline__ bf16x8 cat8(const bf16x4& a, const bf16x4& b) {
  bf16x8 r;
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
  r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
  return r;
}

// One LDS-DMA piece (1 KiB per wave instruction) with sc1 (served past this CU's L1: the bytes another CU's sc1 store left in memory)
__device__ __forceinline__ void glds16_sc1(const char* g, char* lds) {
  const uint32_t l = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(l) : "memory");
}

// Chunk swizzle of the dS tile (128-byte rows).  A half wave's transpose read covers 8 consecutive rows x 32 bytes (a PAIR of chunks):
// rows of equal parity share the 128-byte half of the 256-byte bank row, so the pair index is XOR-ed with (row >> 1) & 3 (the round-6
// first version XOR-ed the chunk with (row >> 1) & 7: rows r, r + 2 then met in one pair -- 7.0 M conflict cycles per launch, PMC);
// bit 0 separates rows r, r + 8 for the 8-byte writes of a 16-row fragment.
__device__ __forceinline__ int swz_ds(int row) { return (((row >> 1) & 3) << 1) | ((row >> 3) & 1); }

// LDS reads by integer address (what is added after the last lane-dependent operation lands in the instruction's offset field)
__device__ __forceinline__ bf16x8 lds128(uint32_t a) { return *(const QFX_AS3 bf16x8*)a; }
__device__ __forceinline__ bf16x8 ldstr(uint32_t lo, uint32_t hi) {      // two transpose reads = one MFMA operand (see read_trfrag, qfx_attn.hip)
  const bf16x4v l = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((QFX_AS3 bf16x4v*)lo);
  const bf16x4v h = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((QFX_AS3 bf16x4v*)hi);
  return cat8(__builtin_bit_cast(bf16x4, l), __builtin_bit_cast(bf16x4, h));
}

// value held by lane (g, N) for every lane of row g (DPP row_newbcast: one VALU move, usually folded into the consumer)
template <int N> __device__ __forceinline__ float rowb(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x150 + N, 0xf, 0xf, false));
}
// statistics of the four query rows 16 QF + 4 g + r of a tile, from the lane-distributed copy (lane (g, n) holds query
// 16 (n >> 2) + 4 g + (n & 3))
template <int QF> __device__ __forceinline__ f32x4 stat4(float x) {
  return (f32x4){rowb<4 * QF>(x), rowb<4 * QF + 1>(x), rowb<4 * QF + 2>(x), rowb<4 * QF + 3>(x)};
}

__global__ __launch_bounds__(512, 1) void attn_bwd1_kernel(const qfx_attn_args a, const int nkb, const int nitems) {
  __shared__ __attribute__((aligned(1024))) char smem[10 * TB];   // K (4 tiles of 64 keys) | 2 x [Q | dO] | dS [256 keys][64 q]
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int S = a.S, H = a.H;
  const int ntiles = (S + 63) >> 6;
  char* sK = smem;
  char* sStage = smem + 4 * TB;
  char* sDS = smem + 8 * TB;
  // virtual block id: XCD x (hardware: block i runs on XCD i % 8) walks a contiguous range of items -> the key blocks of a head share an L2
  int vid;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    vid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  }
  const float c2 = a.scale * LOG2E;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;                          // LDS byte address of the K block (regions: + 4 TB stage, + 8 TB dS)
  const uint32_t lkw = lds0 + (w >> 1) * TB + (w & 1) * 2 * (16 * DH * 2);  // this wave's 32 key rows inside the resident K block
  const int qh = w >> 2, dq_ = w & 3;                                       // dQ phase: queries 32 qh .., d columns 32 dq_ ..

  constexpr int CPR = DH / 8, RPI = 64 / CPR, RPW = 64 / NW, NIS = RPW / RPI;
  for (int item = vid; item < nitems; item += (int)gridDim.x) {
    const int kbi = item % nkb, hb = item / nkb;
    const int h = hb % H, b = hb / H;
    const int kb0 = kbi * 256, key0 = kb0 + w * 32;
    const bf16_t* Qb = a.Q + (int64_t)b * S * a.ldq + h * DH;
    const bf16_t* dOb = a.dO + (int64_t)b * S * a.lddo + h * DH;
    const bf16_t* Kb = a.K + (int64_t)b * S * a.ldk + h * DH;
    const float* lseb = a.lse2 + ((int64_t)b * H + h) * a.S_pad;
    const float* dsb = a.dsum + ((int64_t)b * H + h) * a.S_pad;
    float* accb = a.dq_acc + ((int64_t)b * H + h) * ntiles * ACC_TILE;
    int* turnb = a.dq_turn + ((int64_t)b * H + h) * ntiles;
    const int64_t accbytes = (int64_t)ntiles * ACC_TILE * 4;
    const __amdgpu_buffer_rsrc_t accr = __builtin_amdgcn_make_buffer_rsrc((void*)accb, 0, accbytes > 0x7fffffff ? 0x7fffffff : (int)accbytes, 0x27000);
    const int st = (BWD1_ABL & 32) ? 0 : (kbi * ntiles) / nkb;          // first query tile of this key block
    const bool ktail = kb0 + 256 > S;             // block-uniform: some keys of this block lie past S

    auto stage = [&](int ti, int buf) {
      const int i0 = ti * 64;
      char* dQ_ = sStage + buf * 2 * TB;
      int ln = lane;
      asm volatile("" : "+v"(ln));      // see attn_bwd_dkv_kernel: keep the lane-constant DMA offsets out of registers across the loop
#pragma unroll
      for (int i = 0; i < NIS; ++i) {
        const int row = w * RPW + i * RPI + ln / CPR;
        const int sc8 = ((ln % CPR) ^ swz_row<DH>(row)) * 8;
        int sr = i0 + row; sr = sr < S ? sr : S - 1;
        glds16(Qb + (row_off(sr, a.ldq) + (unsigned)sc8), dQ_ + (w * RPW + i * RPI) * (DH * 2));
        glds16(dOb + (row_off(sr, a.lddo) + (unsigned)sc8), dQ_ + TB + (w * RPW + i * RPI) * (DH * 2));
      }
    };
    __syncthreads();      // the previous item's LDS reads are over
#pragma unroll
    for (int i = 0; i < 4; ++i) stage_rows_n<DH, NW>(sK + i * TB, Kb, a.ldk, kb0 + 64 * i, S, w, lane);
    const int qstat0 = 16 * (li >> 2) + 4 * g + (li & 3);                    // the query whose statistics this lane carries
    float lse_c = lseb[st * 64 + qstat0], ds_c = dsb[st * 64 + qstat0];      // statistics of the CURRENT tile, one query per lane
    stage(st, 0);

    bf16x8 vf[2][KC];
    float mk[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      int k = key0 + f * 16 + li;
      const bool ok = k < S;
      k = ok ? k : S - 1;
      const bf16_t* vp = a.V + ((int64_t)b * S + k) * a.ldv + h * DH + 8 * g;
#pragma unroll
      for (int kk = 0; kk < KC; ++kk) vf[f][kk] = *(const bf16x8*)(vp + kk * 32);
      mk[f] = (a.key_mask && ok) ? a.key_mask[(int64_t)b * S + k] * LOG2E : 0.f;
    }
    // Claim the prologue's loads BEFORE the tile loop.  hipcc does not see the hand-placed s_waitcnt vmcnt(0) at the top of the loop: left
    // pending in its model, these loads get counted waits (vmcnt(8) ... vmcnt(1)) at their first use INSIDE the loop, in every iteration,
    // just after the next tile's LDS-DMA went out -- and each such wait drains the DMA it was meant to overlap.
#pragma unroll
    for (int f = 0; f < 2; ++f) {
#pragma unroll
      for (int kk = 0; kk < KC; ++kk) asm volatile("" : "+v"(vf[f][kk]));
      asm volatile("" : "+v"(mk[f]));
    }
    f32x4 dk[DF][2], dv[DF][2];
#pragma unroll
    for (int d = 0; d < DF; ++d)
#pragma unroll
      for (int f = 0; f < 2; ++f) { dk[d][f] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[d][f] = dk[d][f]; }

    // arrival order at tile ti: m(ti), m-1, ..., 0, nkb-1, ..., m+1 with m(ti) = ((ti + 1) nkb - 1) / ntiles -> rank = (m - kbi) mod nkb
    int ti = st;
    int mnum = (st + 1) * nkb - 1;
    int m = mnum / ntiles;
    int prev_ti = 0, prev_next = 0;      // tile accumulated in the previous iteration and the counter value that hands it on
    int rel_ti = 0, rel_next = 0;        // ... in the iteration before that: released at this top
    for (int it = 0; it < ntiles; ++it) {
      const int i0 = ti * 64;
      const int buf = it & 1;
      int rank = m - kbi; rank = rank < 0 ? rank + nkb : rank;
      // This wave's DMA pieces of tile `it` (and of K) have landed.  Its four workspace stores of tile it-1 -- the youngest vector-memory
      // operations, issued moments ago, write-through -- stay in flight (VMEM returns in order: vmcnt(4) retires everything older): waiting
      // for them here would expose a full store round trip in every iteration.  They are known complete at the NEXT top (they are older
      // than that iteration's stores), so a tile's turn is handed on two tops after it was accumulated.
      if (it == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if constexpr (BWD1_ABL & 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      __syncthreads();                                    // ... everyone's; everyone is done with the dS tile and the stage of it-1
      if (it > 1 && tid == 0) __hip_atomic_store(turnb + rel_ti, rel_next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      rel_ti = prev_ti; rel_next = prev_next;
      // the statistics requested at the end of the previous iteration are claimed HERE (the compiler's own wait lands where nothing is
      // in flight any more), before the next tile's DMA goes out
      asm volatile("" : "+v"(lse_c), "+v"(ds_c) :: "memory");
      const float dsn_c = -ds_c;
      int nti = ti + 1; nti = nti == ntiles ? 0 : nti;
      int tv = 0;
      if (rank != 0) tv = __hip_atomic_load(turnb + ti, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (it + 1 < ntiles) stage(nti, buf ^ 1);
      // LDS addresses as integers.  Every fragment address is (lane part + region base) ^ (small index << 5 or 6) + compile-time
      // constant: the XOR touches bits 5-7 only and every region base is a multiple of 1 KB, so the base (which alternates with the stage
      // buffer) goes in BEFORE the XOR and everything after it is an instruction immediate -- one VALU operation per fragment address
      // instead of XOR + add.  The lane parts are RE-DERIVED from the lane id in every iteration (the empty asm hides the origin): hoisted
      // out of the tile loop they get spilled -- the kernel sits at the 256-VGPR limit -- and every scratch reload costs an
      // s_waitcnt vmcnt(0) that drains the tile DMA.
      int ln_ = lane;
      asm volatile("" : "+v"(ln_));
      const int g_ = ln_ >> 4, li_ = ln_ & 15;
      const uint32_t koff0 = li_ * (DH * 2) + ((g_ ^ swz_row<DH>(li_)) << 4);
      const int tr1 = 4 * g_ + (li_ >> 2);
      const uint32_t toff0 = tr1 * (DH * 2) + ((((li_ & 3) >> 1) ^ swz_row<DH>(tr1)) << 4) + (li_ & 1) * 8;
      const uint32_t lQ = lds0 + 4 * TB + buf * 2 * TB;            // Q tile of this iteration; dO at + TB
      const uint32_t qb = lQ + koff0;                               // row fragments: (qb ^ (kk << 6)) + qf * 4 KB (+ TB for dO)
      const uint32_t kb_ = lkw + koff0;                             // ... of this wave's K rows: (kb_ ^ (kk << 6)) + f * 4 KB
      const uint32_t tqb = lQ + toff0;                              // transposed fragments: (tqb ^ (d << 5)) + t * 8 KB (+ 4 KB) (+ TB)
      const uint32_t tkb = lds0 + toff0;                            // ... of the K block: + (kc >> 1) * TB + (kc & 1) * 8 KB
      // dS tile: row = key (128-byte rows = 8 chunks of 16 B), chunk' = chunk ^ swz_ds(row).  Writer: lane (g, li) of a 16-key
      // fragment owns key row li and the four queries 16 qf + 4 g + r: 8 bytes at chunk 2 qf + (g >> 1), half g & 1.
      const uint32_t dwb = lds0 + 8 * TB + (w * 32 + li_) * DSROW + ((((g_ >> 1) ^ swz_ds(li_))) << 4) + (g_ & 1) * 8;     // ^ (qf << 5), + f * 2 KB
      // reader (transposed fragment of a 32-key chunk for query fragment qf): rows 4 g + (li >> 2) (+16), column 16 qf + 4 (li & 3)
      const uint32_t drb = lds0 + 8 * TB + tr1 * DSROW + (((((li_ & 3) >> 1)) ^ swz_ds(tr1)) << 4) + (li_ & 1) * 8;        // ^ (qf << 5), + kc * 4 KB
      const bool tail = i0 + 64 > S;   // wave-uniform: only the last tile masks query rows
      // ---- S^T, dP^T, P, dS of this wave's 32 keys x 64 queries; dV += P^T dO, dK += dS^T Q; dS -> LDS
      sfor<2>([&](auto T_) {
        constexpr int t = T_.value;
        u32x4 pbu[2], dsu[2];
        sfor<2>([&](auto Q2_) {
          constexpr int q2 = Q2_.value, qfi = 2 * t + q2;
          __builtin_amdgcn_sched_barrier(0);
          // per-query statistics of this fragment's four rows: -dsum[q] is where the dP chains START (the MFMA takes it as its C operand:
          // the subtraction dP - dsum costs nothing), lse2[q] rides in the FMA that feeds the exponential
          const f32x4 l4 = (BWD1_ABL & 128) ? (f32x4){lse_c, lse_c, lse_c, lse_c} : stat4<qfi>(lse_c);
          const f32x4 s4n = (BWD1_ABL & 128) ? (f32x4){dsn_c, dsn_c, dsn_c, dsn_c} : stat4<qfi>(dsn_c);
          f32x4 sa[2], da[2];
          sa[0] = (f32x4){0.f, 0.f, 0.f, 0.f}; sa[1] = sa[0]; da[0] = s4n; da[1] = s4n;
#pragma unroll
          for (int kk = 0; kk < KC; ++kk) {
            const uint32_t qk_ = (qb ^ (kk << 6)) + qfi * (16 * DH * 2);
            const bf16x8 qa = lds128(qk_);
            const bf16x8 oa = lds128(qk_ + TB);
#pragma unroll
            for (int f = 0; f < 2; ++f) {
              sa[f] = MFMA(qa, lds128((kb_ ^ (kk << 6)) + f * (16 * DH * 2)), sa[f]);      // D[i=q][j=key]
              da[f] = MFMA(oa, vf[f][kk], da[f]);
            }
          }
          if (tail) {
            const int qb_ = i0 + qfi * 16 + 4 * g;
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const bool ok = (qb_ + r) < S;
                const float p = ok ? fexp2(sa[f][r] * c2 + mk[f] - l4[r]) : 0.f;
                da[f][r] = ok ? p * da[f][r] : 0.f;
                sa[f][r] = p;
              }
          } else if (a.key_mask == nullptr) {     // block-uniform common case
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float p = fexp2(fmaf(sa[f][r], c2, -l4[r]));
                da[f][r] = p * da[f][r];
                sa[f][r] = p;
              }
          } else {
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float p = fexp2(sa[f][r] * c2 + (mk[f] - l4[r]));
                da[f][r] = p * da[f][r];
                sa[f][r] = p;
              }
          }
#pragma unroll
          for (int f = 0; f < 2; ++f) {
            pbu[f][2 * q2] = pack2bf(sa[f][0], sa[f][1]); pbu[f][2 * q2 + 1] = pack2bf(sa[f][2], sa[f][3]);
            dsu[f][2 * q2] = pack2bf(da[f][0], da[f][1]); dsu[f][2 * q2 + 1] = pack2bf(da[f][2], da[f][3]);
            u32x2 dsw = {dsu[f][2 * q2], dsu[f][2 * q2 + 1]};
            // keys past S ride along on row S - 1: their dS must not reach dQ (only the LDS copy: their dK / dV rows are never stored)
            if (ktail && !(BWD1_ABL & 64) && key0 + f * 16 + li >= S) dsw = (u32x2){0u, 0u};
            if constexpr (!(BWD1_ABL & 8)) *(QFX_AS3 u32x2*)((dwb ^ (qfi << 5)) + f * (16 * DSROW)) = dsw;
          }
        });
        const bf16x8 pb0 = __builtin_bit_cast(bf16x8, pbu[0]), pb1 = __builtin_bit_cast(bf16x8, pbu[1]);
        const bf16x8 ds0 = __builtin_bit_cast(bf16x8, dsu[0]), ds1 = __builtin_bit_cast(bf16x8, dsu[1]);
#pragma unroll
        for (int d = 0; d < DF; ++d) {
          const uint32_t ta = (tqb ^ (d << 5)) + t * (32 * DH * 2);
          const bf16x8 qt = ldstr(ta, ta + 16 * DH * 2);
          const bf16x8 ot = ldstr(ta + TB, ta + TB + 16 * DH * 2);
          dv[d][0] = MFMA(ot, pb0, dv[d][0]);     // D[i=d][j=key]
          dv[d][1] = MFMA(ot, pb1, dv[d][1]);
          dk[d][0] = MFMA(qt, ds0, dk[d][0]);
          dk[d][1] = MFMA(qt, ds1, dk[d][1]);
        }
      });
      if constexpr (!(BWD1_ABL & 16)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // the dS tile is complete (the next tile's DMA stays in flight)
      if (it + 1 < ntiles) {      // next tile's statistics: requested HERE (their registers are free from the last softmax on) so that they
        // have the whole dQ phase to arrive -- requested at the end of the iteration, the top-of-loop wait exposed their full latency
        // in every iteration (PMC: +50 M parked wave cycles per launch against the two-pass dK/dV kernel)
        const int qstat = nti * 64 + 16 * (li_ >> 2) + 4 * g_ + (li_ & 3);
        lse_c = lseb[qstat];
        ds_c = dsb[qstat];
      }
      // ---- my turn at this tile's dQ accumulator?  (requested at the top; normally long since granted)
      // The tile accumulated so far comes in by LDS-DMA (sc1) into the stage buffer this tile's Q | dO just left -- free until the top of
      // the next iteration -- while the dQ MFMAs run: it never occupies registers beside them (held in registers it was spilled, and
      // every scratch access waits vmcnt(0)).
      const int aoff = ti * (ACC_TILE * 4) + w * 4096;
      char* sOld = sStage + buf * 2 * TB + w * 4096;
      if constexpr (BWD1_ABL & 1) asm volatile("" :: "v"(tv));      // (ablation builds: keep the flag load's register claimed as the product does)
      if (rank != 0 && !(BWD1_ABL & 1)) {
        int spins = 0;
        while (__builtin_amdgcn_readfirstlane(tv) != rank) {
          __builtin_amdgcn_s_sleep(2);
          tv = __hip_atomic_load(turnb + ti, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (++spins > SPIN_LIMIT) __builtin_trap();
        }
        const char* src = (const char*)accb + aoff + ln_ * 16;
#pragma unroll
        for (int fi = 0; fi < 4; ++fi) glds16_sc1(src + fi * 1024, sOld + fi * 1024);
      }
      // ---- dQ^T[32 d][32 q] of this wave over the block's 256 keys
      f32x4 qacc[2][2];
#pragma unroll
      for (int df = 0; df < 2; ++df) { qacc[df][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; qacc[df][1] = qacc[df][0]; }
      {
        const uint32_t tk0 = tkb ^ ((2 * dq_) << 5), tk1 = tkb ^ ((2 * dq_ + 1) << 5);
        const uint32_t dr0 = drb ^ ((2 * qh) << 5), dr1 = drb ^ ((2 * qh + 1) << 5);
#pragma unroll
        for (int kc = 0; kc < ((BWD1_ABL & 4) ? 0 : 8); ++kc) {
          constexpr int HI = 16 * DH * 2;
          const int ko = (kc >> 1) * TB + (kc & 1) * (32 * DH * 2), so = kc * (32 * DSROW);
          const bf16x8 k0 = ldstr(tk0 + ko, tk0 + ko + HI), k1 = ldstr(tk1 + ko, tk1 + ko + HI);
          const bf16x8 s0 = ldstr(dr0 + so, dr0 + so + 16 * DSROW), s1 = ldstr(dr1 + so, dr1 + so + 16 * DSROW);
          qacc[0][0] = MFMA(k0, s0, qacc[0][0]);      // D[i=d][j=q]
          qacc[0][1] = MFMA(k0, s1, qacc[0][1]);
          qacc[1][0] = MFMA(k1, s0, qacc[1][0]);
          qacc[1][1] = MFMA(k1, s1, qacc[1][1]);
        }
      }
      if (rank != 0 && !(BWD1_ABL & 1)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the accumulated tile has landed (so has the next tile's Q | dO)
        const uint32_t lo_ = lQ + w * 4096 + ln_ * 16;
#pragma unroll
        for (int fi = 0; fi < 4; ++fi) {
          const f32x4 o = *(const QFX_AS3 f32x4*)(lo_ + fi * 1024);
          // element by element, on purpose: a vector += lowers to v_pk_add_f32 reading the MFMA accumulator pairs in place -- the pattern
          // behind the run-to-run differences of round 4 (profiles/r05_nondeterminism.md; tools/pk_mfma_scan.py keeps it out)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = qacc[fi >> 1][fi & 1][r];
            asm volatile("" : "+v"(v));
            qacc[fi >> 1][fi & 1][r] = v + o[r];
          }
        }
      }
#pragma unroll
      for (int fi = 0; fi < ((BWD1_ABL & 2) ? 0 : 4); ++fi)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, qacc[fi >> 1][fi & 1]), accr, ln_ * 16, aoff + fi * 1024, BWD1_STORE_AUX);
      prev_ti = ti;
      prev_next = rank == nkb - 1 ? 0 : rank + 1;      // the last key block leaves the counter at zero for the next launch
      // next tile (rotation) and its first arriver
      ti = nti;
      if (ti == 0) { mnum = nkb - 1; m = 0; }
      else { mnum += nkb; if (mnum >= (m + 1) * ntiles) ++m; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      if (ntiles > 1) __hip_atomic_store(turnb + rel_ti, rel_next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(turnb + prev_ti, prev_next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    // ---- dK, dV of this wave's keys (the epilogue of attn_bwd_dkv_kernel)
    bool keyok[2];
    int mykey[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const int k = key0 + f * 16 + li;
      keyok[f] = k < S;
      mykey[f] = keyok[f] ? k : S - 1;
    }
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      bf16_t* kp = a.dK + ((int64_t)b * S + mykey[f]) * a.lddk + h * DH;
      bf16_t* vp = a.dV + ((int64_t)b * S + mykey[f]) * a.lddv + h * DH;
      const bool wide_k = rows_16b(a.dK, a.lddk), wide_v = rows_16b(a.dV, a.lddv);
      if (a.qk_saved) {       // block-uniform: d(pre-norm k) straight from the accumulators (rows past S ride along on row S-1, unstored)
        u32x2 u[DF];
        norm_rope_bwd_row<DH>(dk, f, a.scale, a.qk_saved + ((int64_t)b * S + mykey[f]) * a.ld_saved + a.H * DH + h * DH + 4 * g,
                              a.rope + (int64_t)b * a.rope_bstride + ((int64_t)mykey[f] * (DH / 2) + 2 * g) * 2,
                              (mykey[f] < a.T ? a.wk_txt : a.wk_img) + 4 * g, a.norm_eps, a.norm_flags, u);
        store_frag<DH>(kp, u, g, keyok[f], wide_k);
        if (key0 + f * 16 < S) {      // fragment-uniform: v_k = d(pre-norm k) (s B_k)^T, v_v = dV (s B_v)^T (ABI 6)
          head_lora_frag<DH>(a.hl[2], h, a.T, key0 + f * 16, (int64_t)b * S + mykey[f], keyok[f], u, g, li);
          if (a.hl[3].part != nullptr) {
#pragma unroll
            for (int d = 0; d < DF; ++d) { u[d][0] = pack2bf(dv[d][f][0], dv[d][f][1]); u[d][1] = pack2bf(dv[d][f][2], dv[d][f][3]); }
            head_lora_frag<DH>(a.hl[3], h, a.T, key0 + f * 16, (int64_t)b * S + mykey[f], keyok[f], u, g, li);
          }
        }
      }
      u32x2 u[DF];
      if (!a.qk_saved) {
#pragma unroll
        for (int d = 0; d < DF; ++d) {
          u[d][0] = pack2bf(dk[d][f][0] * a.scale, dk[d][f][1] * a.scale);
          u[d][1] = pack2bf(dk[d][f][2] * a.scale, dk[d][f][3] * a.scale);
        }
        store_frag<DH>(kp, u, g, keyok[f], wide_k);
      }
#pragma unroll
      for (int d = 0; d < DF; ++d) { u[d][0] = pack2bf(dv[d][f][0], dv[d][f][1]); u[d][1] = pack2bf(dv[d][f][2], dv[d][f][3]); }
      store_frag<DH>(vp, u, g, keyok[f], wide_v);
    }
  }
}

// dQ from the accumulated fp32 tiles: one block per (batch, head, 64-query tile), wave w = the tile's 16-row fragment w.  The workspace
// keeps a tile as [8 waves][4 fragments][64 lanes][4] floats in the producers' accumulator layout, which is also the layout
// norm_rope_bwd_row / store_frag / head_lora_frag work on: 8 coalesced 1 KB loads per wave, then the dQ epilogue of the two-pass kernels.
__global__ __launch_bounds__(256) void attn_dq_finish_kernel(const qfx_attn_args a) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int S = a.S, H = a.H;
  const int ntiles = (S + 63) >> 6;
  int bid = blockIdx.x;
  const int ti = bid % ntiles; bid /= ntiles;
  const int h = bid % H, b = bid / H;
  const int q0 = ti * 64 + w * 16;
  if (q0 >= S) return;      // wave-uniform
  const float* acc = a.dq_acc + (((int64_t)b * H + h) * ntiles + ti) * ACC_TILE;
  const int q = q0 + li;
  const int qc = q < S ? q : S - 1;       // rows past S compute on row S-1 (the shuffles need every lane) and are not stored
  bf16_t* op = a.dQ + ((int64_t)b * S + qc) * a.lddq + h * DH;
  const bool wide = rows_16b(a.dQ, a.lddq);
  // every operand of the fragment is requested up front (accumulated tile: 8 coalesced 1 KB pieces; saved pre-norm row, RoPE table,
  // norm weight: nrb_load) -- one memory round trip per wave instead of a chain of them
  NrbOps<DH> ops;
  if (a.qk_saved)       // block-uniform
    nrb_load<DH>(ops, a.qk_saved + ((int64_t)b * S + qc) * a.ld_saved + h * DH + 4 * g,
                 a.rope + (int64_t)b * a.rope_bstride + ((int64_t)qc * (DH / 2) + 2 * g) * 2, (qc < a.T ? a.wq_txt : a.wq_img) + 4 * g);
  f32x4 dq[DF];
#pragma unroll
  for (int d = 0; d < DF; ++d) dq[d] = *(const f32x4*)(acc + ((w >> 1) * 4 + (d >> 1)) * 1024 + ((d & 1) * 2 + (w & 1)) * 256 + lane * 4);
  u32x2 u[DF];
  if (a.qk_saved) {       // d(pre-norm q) (QK RMSNorm + RoPE backward fused here)
    norm_rope_bwd_ops<DH>(dq, a.scale, ops, a.norm_eps, a.norm_flags, u);
    store_frag<DH>(op, u, g, q < S, wide);
    head_lora_frag<DH>(a.hl[1], h, a.T, q0, (int64_t)b * S + qc, q < S, u, g, li);   // v_q = d(pre-norm q) (s B_q)^T
  } else {
#pragma unroll
    for (int d = 0; d < DF; ++d) {
      u[d][0] = pack2bf(dq[d][0] * a.scale, dq[d][1] * a.scale);
      u[d][1] = pack2bf(dq[d][2] * a.scale, dq[d][3] * a.scale);
    }
    store_frag<DH>(op, u, g, q < S, wide);
  }
}

int num_cus() {
  static const int n = [] {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
    return p.multiProcessorCount;
  }();
  return n;
}

}  // namespace

namespace qfxi {

// Persistent grid of the one-pass backward: whole heads per round, every block of a round resident at once.  0 = not supported.
int attn_bwd1_grid(const qfx_attn_args* a, int* nkb_out) {
  if (a->dh != 128 || a->S < 64) return 0;
  const int nkb = (a->S + 255) / 256;
  int cus = num_cus();
  if (cus > QFX_NUM_CU_TOTAL) cus = QFX_NUM_CU_TOTAL;
  const long heads = (long)a->B * a->H;
  long per_round = cus / nkb;
  if (per_round < 1) return 0;
  if (per_round > heads) per_round = heads;
  // whole heads per XCD (grid / 8 a multiple of nkb) where that costs no extra round: a head's K / V / Q / dO then stream through one L2
  const long p8 = per_round - per_round % 8;
  if (p8 >= 8 && (heads + p8 - 1) / p8 == (heads + per_round - 1) / per_round) per_round = p8;
  if (nkb_out) *nkb_out = nkb;
  return (int)(per_round * nkb);
}

int launch_attn_bwd1(const qfx_attn_args* a, hipStream_t stream) {
  int nkb = 0;
  const int G = attn_bwd1_grid(a, &nkb);
  if (G <= 0) return QFX_EUNSUPPORTED;
  const int nitems = a->B * a->H * nkb;
  hipLaunchKernelGGL(attn_bwd1_kernel, dim3(G), dim3(512), 0, stream, *a, nkb, nitems);
  QFX_CHECK_LAUNCH();
  const int ntiles = (a->S + 63) / 64;
  hipLaunchKernelGGL(attn_dq_finish_kernel, dim3((unsigned)(a->B * a->H * ntiles)), dim3(256), 0, stream, *a);
  QFX_CHECK_LAUNCH();
  return QFX_OK;
}

}  // namespace qfxi
