// qfx_attn64.hip -- attention forward with 64-query waves, ONE wave per SIMD, on v_mfma_f32_32x32x16_bf16 (gfx950, dh = 128).
//
// Replaces, for the device work of transformer_qwenimage.py:329-337 (joint SDPA), the 8 x 32-query form of qfx_attn.hip where the
// launcher selects it (qfx_attn_fwd, QFX_ATTN_FWD64).  Structure (cdna_hip_programming.md "4-wave, one-wave-per-SIMD" note):
//   * block = 4 waves x 64 queries = 256 queries of one head; each wave owns a whole SIMD and its 512 registers;
//   * the accumulator half of the register file is allocated BY HAND: a[0:127] = O^T (4 d-blocks x 2 query blocks x 16), a[128:191] = the
//     wave's Q fragments.  Every MFMA is an asm statement that names those registers, so hipcc can neither park Q in the accumulator
//     half and copy it back (8 v_accvgpr_read per MFMA pair: the 142 us prototype of round 3) nor move the O accumulators;
//   * scores are computed swapped (S^T = K Q^T): a lane owns one query column, 32 of its 64 scores per tile (the partner lane l ^ 32 owns
//     the other 32), so the online softmax is lane-local; the packed bf16 P registers are directly the B operand of the PV MFMA
//     under a key permutation that the V^T transpose read applies too -- no cross-lane traffic in the tile loop;
//   * the two 32-query blocks of a wave are SKEWED by half a tile: T1 QK^T(qb 0) | T2 QK^T(qb 1) with softmax(qb 0) in the MFMA gaps |
//     T3 PV(qb 0) with softmax(qb 1) in the gaps | T4 PV(qb 1).  Each "slot" = one 32-cycle MFMA + <= 8 other issues, pinned by
//     sched_barrier; K rows are read twice per tile (once per query block), V^T twice: LDS traffic per flop = the old kernel's,
//     MFMA count half, softmax instructions per score 4.2 instead of 8.4;
//   * K / V tiles by LDS-DMA into a two-deep ring (one barrier per tile); K keeps the b128 swizzle of qfx_attn.hip, V gets its own:
//     a half wave's transpose read covers 4 rows x 64 bytes, chunk' = chunk ^ ((row & 3) << 2) puts the four rows on four different
//     64-byte bank groups;
//   * epilogue: O / l is staged through LDS (row stride 272 B: conflict-free 8-byte writes and reads) and re-read in the 16-row
//     fragment layout of qfx_attn.hip, so the wide stores and the fused rank-r projection (qfx_head_lora) are the SAME code.
// Register audit after every edit (tools/agpr_audit.py): no scratch, no compiler v_accvgpr_* outside the asm statements.
#include "qfx_attn_common.h"
#include <utility>

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

// ablation lever (wrong results): -DQFX_A64_NOEXP replaces the exponentials of the pipelined forward by moves -- what do they cost a lone wave?
#if defined(QFX_A64_NOEXP)
#define A64_EXP "v_mov_b32"
#else
#define A64_EXP "v_exp_f32"
#endif

#ifndef ATTN_DEFER_MAX
#define ATTN_DEFER_MAX 8.0f
#endif

// every statement that writes the hand-allocated half names all of it: the compiler then holds nothing there across our statements
#define A64_CLOB \
  "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", \
      "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", \
      "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", \
      "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", \
      "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", \
      "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", \
      "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", \
      "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", \
      "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", \
      "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", \
      "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", \
      "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", \
      "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", \
      "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", \
      "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", \
      "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", \
      "a253", "a254", "a255"

constexpr int A_O = 0;        // O^T block (db, qb): a[A_O + 16 * (2 db + qb) .. + 15]
constexpr int A_Q = 128;      // Q fragment (qb, ks): a[A_Q + 4 * (8 qb + ks) .. + 3]
constexpr int A_K = 192;      // K fragment (ks, kb) of the current tile: a[A_K + 4 * (2 ks + kb) .. + 3]

template <int R> __device__ __forceinline__ void agpr_write(uint32_t v) {
  asm volatile("v_accvgpr_write_b32 a[%c1], %0" ::"v"(v), "i"(R) : A64_CLOB);
}
template <int R> __device__ __forceinline__ float agpr_read() {
  float x;
  asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(x) : "i"(R));
  return x;
}
// S^T block += K fragment (A: 32 keys x 16 d) x Q fragment (B), BOTH in the accumulator half.  WAIT >= 0: the statement first waits until at
// most WAIT LDS operations are outstanding (the fragment was requested by k_load WAIT + 1 requests ago; counts of hipcc's own LDS
// traffic can only make the wait stricter, never laxer: LDS operations return in order).
template <int KREG, int QREG, int WAIT, bool FIRST> __device__ __forceinline__ void mfma_qk(f32x16& s) {
  if constexpr (FIRST) {
    if constexpr (WAIT >= 0)
      asm volatile("s_waitcnt lgkmcnt(%c5)\n\tv_mfma_f32_32x32x16_bf16 %0, a[%c1:%c2], a[%c3:%c4], 0" : "=&v"(s) : "i"(KREG), "i"(KREG + 3), "i"(QREG), "i"(QREG + 3), "i"(WAIT));
    else
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%c1:%c2], a[%c3:%c4], 0" : "=&v"(s) : "i"(KREG), "i"(KREG + 3), "i"(QREG), "i"(QREG + 3));
  } else {
    if constexpr (WAIT >= 0)
      asm volatile("s_waitcnt lgkmcnt(%c5)\n\tv_mfma_f32_32x32x16_bf16 %0, a[%c1:%c2], a[%c3:%c4], %0" : "+v"(s) : "i"(KREG), "i"(KREG + 3), "i"(QREG), "i"(QREG + 3), "i"(WAIT));
    else
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%c1:%c2], a[%c3:%c4], %0" : "+v"(s) : "i"(KREG), "i"(KREG + 3), "i"(QREG), "i"(QREG + 3));
  }
}
// K fragment LDS -> accumulator half (uncounted by hipcc: the consumer waits, see mfma_qk)
template <int KREG, int OFF> __device__ __forceinline__ void k_load(uint32_t lds_addr) {
  asm volatile("ds_read_b128 a[%c1:%c2], %0 offset:%c3" ::"v"(lds_addr), "i"(KREG), "i"(KREG + 3), "i"(OFF) : A64_CLOB, "memory");
}
// O^T block (accumulator half) += V^T fragment (A: 32 d x 16 keys) x P fragment (B: 16 keys x 32 queries)
template <int OREG> __device__ __forceinline__ void mfma_pv(const bf16x8& vf, const u32x4& pf) {
  asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(vf), "v"(pf), "i"(OREG), "i"(OREG + 15) : A64_CLOB);
}
template <int BREG, bool FIRST> __device__ __forceinline__ void mfma_vb(f32x16& s, const bf16x8& af) {      // D (VGPR) (+)= A (VGPR) x B (AGPR)
  if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], 0" : "=&v"(s) : "v"(af), "i"(BREG), "i"(BREG + 3));
  else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], %0" : "+v"(s) : "v"(af), "i"(BREG), "i"(BREG + 3));
}
__device__ __forceinline__ float vmax3(float a, float b, float c) {      // one instruction, no canonicalising v_max on the asm-produced scores
  float m;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(a), "v"(b), "v"(c));
  return m;
}
__device__ __forceinline__ float fexp2_(float x) { return __builtin_amdgcn_exp2f(x); }

template <class F, int... I> __device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void sfor(F&& f) { sfor_impl(f, std::make_integer_sequence<int, N>{}); }

// chunk' = chunk ^ swz64(row): one 16-byte-chunk swizzle of a [rows][256 B] LDS tile that serves ds_read_b128 fragments (the rows of a
// 16-lane read group on one chunk) AND the transpose read of the 32x32x16 layout (4 rows x 4 chunks per half wave)
__device__ __forceinline__ int swz64(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

// Per-wave B-operand fragments (Q, dO; O for dsum) of a 32-row query block.  A lane of the MFMA layout owns ONE row: loading its
// fragments straight from HBM touches 32 rows per wave instruction (48 such loads took 20 k cycles of the dQ prologue, 10 us).  Instead:
// coalesced 16-byte loads in row order (a wave instruction = 4 whole rows) -> a wave-private 8 KB LDS slab (swz64) -> ds_read_b128 in
// fragment order.  rows_issue() only requests (all tensors of a query block in flight together), rows_to_frags() does the LDS round trip.
__device__ __forceinline__ void rows_issue(const bf16_t* base, int64_t ld, int row0, int S, int lane, u32x4 (&v)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int r = row0 + 4 * i + (lane >> 4); r = r < S ? r : S - 1;
    v[i] = *(const u32x4*)(base + (int64_t)r * ld + (lane & 15) * 8);
  }
}
__device__ __forceinline__ void rows_to_frags(char* slab, int lane, const u32x4 (&v)[8], u32x4 (&f)[8]) {
  const int lj = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = 4 * i + (lane >> 4);
    *(u32x4*)(slab + r * 256 + (((lane & 15) ^ swz64(r)) << 4)) = v[i];
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) f[ks] = *(const u32x4*)(slab + lj * 256 + (((2 * ks + hi) ^ swz64(lj)) << 4));
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the slab is rewritten by the next tensor
}

constexpr int TB = 64 * 128 * 2;          // bytes of a K (or V) tile
constexpr int STG_LD = 272;               // staging row stride of the epilogue (bytes)

__global__ __launch_bounds__(256, 1) void attn_fwd64_kernel(const qfx_attn_args a) {
  constexpr int DH = 128;
  // FOUR [K | V] stages (128 KB).  At the top of tile j a wave waits for all but its 8 youngest LDS-DMA requests (vmcnt(8)): tiles <= j + 1
  // have landed, tile j + 2 (requested in T4 of tile j - 1, where a piece costs ~25 issue cycles against ~40 in T1) stays in flight for
  // another tile time.  After the barrier tile j + 1 is known complete for EVERY wave, so the first K fragments of tile j + 1 are
  // requested at the end of tile j, across the next barrier (with two stages every tile opened with an exposed LDS round trip), and
  // the stage of tile j - 1 is free for tile j + 3.  The epilogue's staging slabs (4 x 17 KB) re-use the ring after a last barrier.
  __shared__ __attribute__((aligned(16))) char smem[8 * TB];
  static_assert(8 * TB >= 4 * 64 * STG_LD, "staging slabs must fit the ring");
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, lj = lane & 31;
  int xb, h, b;
  attn_block_coord((a.S + 255) / 256, a.H, xb, h, b);
  const int S = a.S;
  const int q0 = xb * 256 + w * 64;
  const bool live = q0 < S;                 // wave-uniform: a wave whose rows all lie past S only stages tiles and keeps the barrier protocol
  const bf16_t* Kb = a.K + (int64_t)b * S * a.ldk + h * DH;
  const bf16_t* Vb = a.V + (int64_t)b * S * a.ldv + h * DH;
  const int ntiles = (S + 63) / 64;

  // ---- LDS-DMA staging: wave w brings rows 16 w .. 16 w + 15 of both tiles, 4 pieces of 4 rows (1 KiB) each per tile.  Piece i < 4: K rows
  // 16 w + 4 i ..; piece i >= 4: V rows.  Rows past S clamp to S - 1 (a tile past the last one is staged into the idle buffer and
  // never read: no branch in the tile body).
  auto stage_piece = [&](int jt, int buf, int i) {
    char* dK = smem + buf * 2 * TB;
    const int rr = lane >> 4, c = lane & 15, ii = i & 3;
    const int row = 16 * w + 4 * ii + rr;
    int s = jt * 64 + row; s = s < S ? s : S - 1;
    if (i < 4) glds16a(Kb + (row_off(s, a.ldk) + (unsigned)((c ^ swz_row<DH>(row)) * 8)), dK + (16 * w + 4 * ii) * 256);
    else glds16a(Vb + (row_off(s, a.ldv) + (unsigned)((c ^ (rr << 2)) * 8)), dK + TB + (16 * w + 4 * ii) * 256);
  };
  auto stage = [&](int jt, int buf) {
#pragma unroll
    for (int i = 0; i < 8; ++i) stage_piece(jt, buf, i);
  };
  stage(0, 0);
  stage(1, 1);
  stage(2, 2);

  // ---- Q fragments -> a[128:191] (through the wave's slab in ring stage 3, free until the first tile issues LDS-DMA into it -- after the
  // first barrier), O^T = 0 -> a[0:127]
  if (live) {
    char* slab = smem + 3 * 2 * TB + w * 8192;
    const bf16_t* qbase = a.Q + (int64_t)b * S * a.ldq + h * DH;
    u32x4 qr[2][8];
    rows_issue(qbase, a.ldq, q0, S, lane, qr[0]);
    rows_issue(qbase, a.ldq, q0 + 32, S, lane, qr[1]);
    sfor<2>([&](auto QB) {
      u32x4 qv[8];
      rows_to_frags(slab, lane, qr[QB.value], qv);
      sfor<8>([&](auto KS) {
        sfor<4>([&](auto I) { agpr_write<A_Q + 4 * (8 * QB.value + KS.value) + I.value>(qv[KS.value][I.value]); });
      });
    });
    sfor<128>([&](auto I) { agpr_write<A_O + I.value>(0u); });
  }

  float mrow[2] = {-INFINITY, -INFINITY}, lrow[2] = {0.f, 0.f};
  const float c2 = a.scale * LOG2E;
  const float* maskb = a.key_mask ? a.key_mask + (int64_t)b * S : nullptr;

  // lane-constant LDS offsets.  K fragment (A operand of S^T): row = 32 kb + lj, 16-byte chunk 2 ks + hi under the b128 swizzle.
  const int koff0 = lj * 256 + ((hi ^ swz_row<DH>(lj)) << 4);             // ^ (ks << 5), + kb * 8192
  // V^T fragment (A operand of O^T) by transpose read: 16-lane group G = lane / 16 covers d columns 32 db + 16 (G & 1) .. + 15 and the
  // key rows 16 t + 4 hi + j' (first read) / + 8 (second), j' = (lane & 15) / 4; source lane m = lane & 3 points at columns + 4 m.
  const int jq = (lane & 15) >> 2, mq = lane & 3, gq = (lane >> 4) & 1;
  const int toff0 = (4 * hi + jq) * 256 + ((((2 * gq + (mq >> 1)) ^ (jq << 2))) << 4) + (mq & 1) * 8;   // ^ (db << 6), + t * 4096 (+ 2048)

  f32x16 S0[2], S1[2];       // score blocks of query block 0 / 1: [kb]
  u32x4 P0[4], P1[4];        // packed P, B operand of k-step t = 2 kb + u

#if defined(QFX_A64_TIMING)
  uint64_t tph[6] = {0, 0, 0, 0, 0, 0}, tmark = __builtin_readcyclecounter();      // barrier, T1, T2, T3, T4, rest
#define A64_T(i) do { asm volatile("s_nop 0" ::: "memory"); const uint64_t n_ = __builtin_readcyclecounter(); tph[i] += n_ - tmark; tmark = n_; } while (0)
#else
#define A64_T(i) do { } while (0)
#endif
  int b_cur = 0, b_nxt = 1, b_nn = 2, b_dma = 3;       // ring stages of tile jt, jt + 1 (landed), jt + 2 (landing), jt + 3 (to be requested)
  for (int jt = 0; jt < ntiles; ++jt) {
    // see the ring description at the top: my pieces of tiles <= jt + 1 have landed, tile jt + 2's may stay in flight
    A64_T(5);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    A64_T(0);
    if (!live) {
      stage(jt + 3, b_dma);
      const int t_ = b_cur; b_cur = b_nxt; b_nxt = b_nn; b_nn = b_dma; b_dma = t_;
      continue;
    }
    const char* sK = smem + b_cur * 2 * TB;
    const char* sV = sK + TB;
    const uint32_t kb_next = (uint32_t)(uintptr_t)(smem + b_nxt * 2 * TB);
    const int j0 = jt * 64;
    const bool need_mask = (j0 + 64 > S) || (maskb != nullptr);     // wave-uniform
    // the tile body exists twice: MASKED (ragged last tile / additive key mask: scores are scaled and masked before the maximum,
    // cs = 1) and the plain one of the headline shapes (the scale rides in the FMA that feeds v_exp)
    auto tile = [&](auto MASKED) {
    constexpr bool masked = MASKED.value;
    const float cs = masked ? 1.0f : c2;

    // Fragment plumbing.  K: 16 fragments (ks, kb) are read ONCE per tile into a[192:255] during T1 (requested PFK slots ahead, asm,
    // waited for by the consuming MFMA statement) and serve T1 AND T2.  V^T: 16 fragments (t, db) are read once during T3 (PFV slots
    // ahead, hipcc counts them) into 64 VGPRs and serve T3 AND T4.  LDS traffic per tile and wave: 32 KB (the 32-query kernels: 2 x 32).
    uint32_t kaddr[8], vaddr[4];
    {
      const uint32_t kb0 = (uint32_t)(uintptr_t)sK, vb0 = (uint32_t)(uintptr_t)sV;      // LDS byte addresses (low 32 bits of the generic pointer)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) kaddr[ks] = kb0 + (uint32_t)(koff0 ^ (ks << 5));
#pragma unroll
      for (int db = 0; db < 4; ++db) vaddr[db] = vb0 + (uint32_t)(toff0 ^ (db << 6));
    }
    constexpr int PFK = 6, PFV = 4;
    auto kreq = [&](auto N) {       // request K fragment n = 2 ks + kb
      constexpr int n = N.value;
      if constexpr (n < 16) k_load<A_K + 4 * n, (n & 1) * 8192>(kaddr[n >> 1]);
    };
    bf16x8 vc[16];
    auto vreq = [&](auto N) {       // request V^T fragment n = 4 t + db
      constexpr int n = N.value;
      if constexpr (n < 16) {
        const char* vp = (const char*)(uintptr_t)0;      // LDS address space pointer from the 32-bit address
        (void)vp;
        QFX_AS3 char* p3 = (QFX_AS3 char*)(uintptr_t)vaddr[n & 3];
        const bf16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((QFX_AS3 bf16x4v*)(p3 + (n >> 2) * 4096));
        const bf16x4v hv = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((QFX_AS3 bf16x4v*)(p3 + (n >> 2) * 4096 + 2048));
        const bf16x4 l4 = __builtin_bit_cast(bf16x4, lo), h4 = __builtin_bit_cast(bf16x4, hv);
        bf16x8 r;
        r[0] = l4[0]; r[1] = l4[1]; r[2] = l4[2]; r[3] = l4[3]; r[4] = h4[0]; r[5] = h4[1]; r[6] = h4[2]; r[7] = h4[3];
        vc[n] = r;
      }
    };
    if (jt == 0) sfor<PFK>([&](auto P) { kreq(P); });       // later tiles: requested at the end of the previous tile's T4
    // one slot of QK^T for query block QB: K fragment n = I (k-step ks = I / 2, key block kb = I % 2)
    auto qk_slot = [&](auto QB, auto I, f32x16 (&Sx)[2]) {
      constexpr int n = I.value, ks = n / 2, kb = n % 2;
      if constexpr (QB.value == 0) {
        kreq(std::integral_constant<int, n + PFK>{});
        constexpr int wait = (15 - n) < PFK ? (15 - n) : PFK;          // requests issued after fragment n
        mfma_qk<A_K + 4 * n, A_Q + 4 * ks, wait, ks == 0>(Sx[kb]);
        vreq(I);       // T1 is MFMA-bound with idle issue slots: ALL V^T fragments of the tile are requested here (hipcc counts them;
                       // its own requests younger than a K fragment only make that fragment's wait stricter)
      } else {
        mfma_qk<A_K + 4 * n, A_Q + 4 * (8 + ks), -1, ks == 0>(Sx[kb]);
      }
    };
    // one slot of PV for query block QB: V^T fragment n = I (k-step t = I / 4, d block db = I % 4)
    auto pv_slot = [&](auto QB, auto I, const u32x4 (&Px)[4]) {
      constexpr int n = I.value, t = n / 4, db = n % 4;
      mfma_pv<A_O + 16 * (2 * db + QB.value)>(vc[n], Px[t]);
    };
    // softmax of query block QB spread over 16 slots.  Lane (lj, hi) holds, of query column lj, the scores of keys
    // 32 kb + 8 (r / 4) + 4 hi + r % 4 (r = 0..15): value index k = 16 kb + r.
    float mxa = 0.f, mxb = 0.f, negm = 0.f, lsum = 0.f, lsum2 = 0.f, pe0 = 0.f, pe1 = 0.f;
    auto sm_slot = [&](auto QB, auto I, f32x16 (&Sx)[2], u32x4 (&Px)[4]) {
      constexpr int qb = QB.value, i = I.value;
      if constexpr (i < 2) {
        f32x16& sv = Sx[i];
        if constexpr (masked) {        // off the headline path: scale + additive mask here, -inf for keys past S
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int key0 = j0 + 32 * i + 8 * c + 4 * hi;           // four consecutive keys
            f32x4 mk4 = {0.f, 0.f, 0.f, 0.f};
            if (maskb != nullptr) {
#pragma unroll
              for (int r = 0; r < 4; ++r) mk4[r] = maskb[(key0 + r) < S ? (key0 + r) : S - 1] * LOG2E;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) sv[4 * c + r] = (key0 + r) < S ? sv[4 * c + r] * c2 + mk4[r] : -INFINITY;
          }
        }
        // four v_max3 chains in ONE statement (dependent instructions >= 2 apart; as separate statements hipcc pads every dependent pair)
        float ma, mb_, mc_, md_;
        asm volatile(
            "v_max3_f32 %0, %4, %5, %6\n\tv_max3_f32 %1, %7, %8, %9\n\tv_max3_f32 %2, %10, %11, %12\n\tv_max3_f32 %3, %13, %14, %15\n\t"
            "v_max3_f32 %0, %0, %16, %17\n\tv_max3_f32 %1, %1, %18, %19\n\tv_max_f32 %2, %2, %3\n\tv_max3_f32 %0, %0, %1, %2"
            : "=&v"(ma), "=&v"(mb_), "=&v"(mc_), "=&v"(md_)
            : "v"(sv[0]), "v"(sv[1]), "v"(sv[2]), "v"(sv[3]), "v"(sv[4]), "v"(sv[5]), "v"(sv[6]), "v"(sv[7]), "v"(sv[8]), "v"(sv[9]), "v"(sv[10]),
              "v"(sv[11]), "v"(sv[12]), "v"(sv[13]), "v"(sv[14]), "v"(sv[15]));
        if constexpr (i == 0) mxa = ma; else mxb = ma;
      }
      if constexpr (i == 2) {
        float mx = fmaxf(mxa, mxb);
        // lazy reference maximum (qfx_attn.hip): the reference of a row moves only when a score of this tile exceeds it by > 2^8
        if (__builtin_expect(!__all(mx * cs - mrow[qb] <= ATTN_DEFER_MAX), 0)) {
          const uint32_t u1 = __float_as_uint(mx);
          const auto r1 = __builtin_amdgcn_permlane32_swap(u1, u1, false, false);
          mx = fmaxf(__uint_as_float(r1[0]), __uint_as_float(r1[1]));
          const float mnew = fmaxf(mrow[qb], mx * cs);
          const float alpha = fexp2_(mrow[qb] - ((mnew == -INFINITY) ? 0.f : mnew));
          mrow[qb] = mnew;
          lrow[qb] *= alpha;
          if (jt > 0) {          // O^T(., qb) *= alpha: no MFMA on these accumulators is in flight in T2 (qb 0) / T3 (qb 1)
            sfor<64>([&](auto R) {
              constexpr int reg = A_O + 16 * (2 * (R.value / 16) + qb) + R.value % 16;
              agpr_write<reg>(__float_as_uint(agpr_read<reg>() * alpha));
            });
            asm volatile("s_nop 1");
          }
        }
        negm = (mrow[qb] == -INFINITY) ? 0.f : -mrow[qb];
        lsum = 0.f; lsum2 = 0.f; pe0 = 0.f; pe1 = 0.f;      // pe = 0: the first statement "finishes" a pair that adds nothing
      }
      // pairs of scores -> p = exp2(s * cs - m), row sum, one packed dword of the PV B operand: pair pi (values 2 pi, 2 pi + 1)
      constexpr int np = i < 2 ? 0 : (i == 3 || i == 4) ? 2 : 1;
      constexpr int p0 = i < 2 ? 0 : i == 2 ? 0 : i == 3 ? 1 : i == 4 ? 3 : i;        // i >= 5: pair i
#pragma unroll
      for (int pp = 0; pp < np; ++pp) {
        const int pi = p0 + pp, k = 2 * pi;
        // ONE statement per pair, so that the work stays in THIS slot (as plain C++ hipcc's instruction selection sinks every pair to
        // its consumer, the PV MFMA of the next phase, and the gaps of this phase stay empty), software-pipelined by one pair: the
        // statement starts pair pi (fma, exp) and finishes pair pi - 1 (row sums on two accumulators, packing) -- no instruction
        // depends on one less than three places before it, a lone wave on its SIMD has nobody to hide a dependent-issue stall.
        float t0, t1;
        uint32_t pw;
        asm volatile(
            "v_fma_f32 %5, %7, %9, %10\n\tv_fma_f32 %6, %8, %9, %10\n\tv_add_f32 %1, %1, %3\n\tv_add_f32 %2, %2, %4\n\t"
            "v_cvt_pk_bf16_f32 %0, %3, %4\n\tv_exp_f32 %3, %5\n\tv_exp_f32 %4, %6"
            : "=&v"(pw), "+v"(lsum), "+v"(lsum2), "+v"(pe0), "+v"(pe1), "=&v"(t0), "=&v"(t1)
            : "v"(Sx[k >> 4][k & 15]), "v"(Sx[k >> 4][(k & 15) + 1]), "v"(cs), "v"(negm));
        if (pi > 0) Px[(pi - 1) >> 2][(pi - 1) & 3] = pw;
      }
      if constexpr (i == 15) {      // finish pair 15
        uint32_t pw;
        asm volatile("v_add_f32 %1, %1, %3\n\tv_add_f32 %2, %2, %4\n\tv_cvt_pk_bf16_f32 %0, %3, %4" : "=&v"(pw), "+v"(lsum), "+v"(lsum2) : "v"(pe0), "v"(pe1));
        Px[3][3] = pw;
        lrow[qb] += lsum + lsum2;
      }
    };

    // T1: QK^T of query block 0
    sfor<16>([&](auto I) {
      qk_slot(std::integral_constant<int, 0>{}, I, S0);
      __builtin_amdgcn_sched_barrier(0);
    });
    A64_T(1);
    // T2: QK^T of query block 1, softmax of block 0 in the gaps
    sfor<16>([&](auto I) {
      qk_slot(std::integral_constant<int, 1>{}, I, S1);
      sm_slot(std::integral_constant<int, 0>{}, I, S0, P0);
      __builtin_amdgcn_sched_barrier(0);
    });
    A64_T(2);
    // T3: PV of query block 0, softmax of block 1 in the gaps
    sfor<16>([&](auto I) {
      pv_slot(std::integral_constant<int, 0>{}, I, P0);
      sm_slot(std::integral_constant<int, 1>{}, I, S1, P1);
      __builtin_amdgcn_sched_barrier(0);
    });
    A64_T(3);
    // T4: PV of query block 1
    sfor<16>([&](auto I) {
      pv_slot(std::integral_constant<int, 1>{}, I, P1);
      // the LDS-DMA pieces of tile jt + 3 ride in the gaps of this phase (8 x ~40 issue cycles at the loop head stall a lone wave's MFMAs;
      // T4 has nothing else between its MFMAs)
      if constexpr (I.value < 8) stage_piece(jt + 3, b_dma, I.value);
      // the first K fragments of tile jt + 1 (its stage landed a tile ago) are requested here, across the coming barrier
      if constexpr (I.value >= 16 - PFK) {
        constexpr int n = I.value - (16 - PFK);
        k_load<A_K + 4 * n, (n & 1) * 8192>(kb_next + (uint32_t)(koff0 ^ ((n >> 1) << 5)));
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    A64_T(4);
    };   // tile
    if (need_mask) tile(std::true_type{}); else tile(std::false_type{});
    { const int t_ = b_cur; b_cur = b_nxt; b_nxt = b_nn; b_nn = b_dma; b_dma = t_; }
  }
#if defined(QFX_A64_TIMING)
  if (lane == 0 && blockIdx.x < 16) {      // caller over-allocates lse2 by 16 * 4 * 8 floats in the timing build
    float* dbg = a.lse2 + ((int64_t)a.B * a.H) * a.S_pad + (blockIdx.x * 4 + w) * 8;
    for (int i = 0; i < 6; ++i) dbg[i] = (float)tph[i];
    dbg[6] = (float)ntiles;
  }
#endif
  // every wave is done with the ring and every stray LDS-DMA piece (tiles past the last one are staged, never read) has landed:
  // the ring becomes the staging slabs
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (!live) return;

  // ---- epilogue: O = O^T / l -> bf16 -> staging slab [64 rows][272 B] of this wave -> 16-row fragments of the qfx_attn.hip layout
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the last PV MFMAs have written the accumulators
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the K requests for the tile past the last one
  char* stg = smem + w * (64 * STG_LD);
  float lse_out[2];
  sfor<2>([&](auto QB) {
    constexpr int qb = QB.value;
    float l = lrow[qb];
    l += __shfl_xor(l, 32);
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    lse_out[qb] = mrow[qb] + log2f(l);
    sfor<4>([&](auto DB) {
      sfor<4>([&](auto C) {
        constexpr int reg = A_O + 16 * (2 * DB.value + qb) + 4 * C.value;
        const float o0 = agpr_read<reg>() * inv, o1 = agpr_read<reg + 1>() * inv, o2 = agpr_read<reg + 2>() * inv, o3 = agpr_read<reg + 3>() * inv;
        const u32x2 u = {pack2bf(o0, o1), pack2bf(o2, o3)};
        *(u32x2*)(stg + (32 * qb + lj) * STG_LD + (32 * DB.value + 8 * C.value + 4 * hi) * 2) = u;
      });
    });
    const int q = q0 + 32 * qb + lj;
    if (q < S && hi == 0) a.lse2[((int64_t)b * a.H + h) * a.S_pad + q] = lse_out[qb];
  });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the slab is private to the wave: no block barrier
  const int g = lane >> 4, li = lane & 15;
  const bool wide = rows_16b(a.O, a.ldo);
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    if (q0 + 16 * f >= S) break;                           // wave-uniform
    u32x2 u[DH / 16];
#pragma unroll
    for (int d = 0; d < DH / 16; ++d) u[d] = *(const u32x2*)(stg + (16 * f + li) * STG_LD + (16 * d + 4 * g) * 2);
    const int q = q0 + 16 * f + li;
    const int qc = q < S ? q : S - 1;
    store_frag<DH>(a.O + ((int64_t)b * S + qc) * a.ldo + h * DH, u, g, q < S, wide);
    head_lora_frag<DH>(a.hl[0], h, a.T, q0 + 16 * f, (int64_t)b * S + qc, q < S, u, g, li);
  }
}


// =============================================================================================================================
// Forward, third form (round 5): a CONTINUOUS software pipeline over 32-key sub-tiles.  The skewed kernel above puts the softmax of a
// query block into the gaps of 16 MFMAs (8-9 instructions per gap) and leaves the gaps of the other two phases almost empty; a lone wave
// hides <= 5 single-issue instructions per 32x32x16 MFMA gap.  Here iteration i issues
//   X_i  the 16 QK^T MFMAs of sub-tile i + 1 (8 K fragments, each feeding BOTH query blocks -- no fragment cache needed), then
//   Y_i  the 16 PV MFMAs of sub-tile i (8 V^T fragments, both query blocks),
// with the online softmax of sub-tile i (2 x 16 scores per lane) laid into those 32 gaps: maxima (4 + 4 instructions per query block),
// the lazy-reference vote, ten pair statements (7 instructions) in X, six more and the two finishing statements in the first gaps of
// Y -- P of k-step 0 is complete before Y starts, P of k-step 1 before its first MFMA -- then the LDS-DMA pieces of tile j + 3.  The
// online softmax runs per 32-key sub-tile (reference maximum, row sum and the rare rescale of O^T; the rescale sits in X, where only
// QK^T MFMAs are in flight).  Scores are double-buffered across sub-tiles (2 x 32 registers); barriers, ring and prologue as above.
__global__ __launch_bounds__(256, 1) void attn_fwd64p_kernel(const qfx_attn_args a) {
  constexpr int DH = 128;
  __shared__ __attribute__((aligned(16))) char smem[8 * TB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, lj = lane & 31;
  int xb, h, b;
  attn_block_coord((a.S + 255) / 256, a.H, xb, h, b);
  const int S = a.S;
  const int q0 = xb * 256 + w * 64;
  const bool live = q0 < S;
  const bf16_t* Kb = a.K + (int64_t)b * S * a.ldk + h * DH;
  const bf16_t* Vb = a.V + (int64_t)b * S * a.ldv + h * DH;
  const int ntiles = (S + 63) / 64;

  auto stage_piece = [&](int jt, int buf, int i) {
    char* dK = smem + buf * 2 * TB;
    const int rr = lane >> 4, c = lane & 15, ii = i & 3;
    const int row = 16 * w + 4 * ii + rr;
    int s = jt * 64 + row; s = s < S ? s : S - 1;
    const unsigned sc = (unsigned)((c ^ swz64(row)) * 8);
    if (i < 4) glds16a(Kb + (row_off(s, a.ldk) + sc), dK + (16 * w + 4 * ii) * 256);
    else glds16a(Vb + (row_off(s, a.ldv) + sc), dK + TB + (16 * w + 4 * ii) * 256);
  };
  auto stage = [&](int jt, int buf) {
#pragma unroll
    for (int i = 0; i < 8; ++i) stage_piece(jt, buf, i);
  };
  stage(0, 0);
  stage(1, 1);
  stage(2, 2);

  if (live) {
    char* slab = smem + 3 * 2 * TB + w * 8192;
    const bf16_t* qbase = a.Q + (int64_t)b * S * a.ldq + h * DH;
    u32x4 qr[2][8];
    rows_issue(qbase, a.ldq, q0, S, lane, qr[0]);
    rows_issue(qbase, a.ldq, q0 + 32, S, lane, qr[1]);
    sfor<2>([&](auto QB) {
      u32x4 qv[8];
      rows_to_frags(slab, lane, qr[QB.value], qv);
      sfor<8>([&](auto KS) {
        sfor<4>([&](auto I) { agpr_write<A_Q + 4 * (8 * QB.value + KS.value) + I.value>(qv[KS.value][I.value]); });
      });
    });
    sfor<128>([&](auto I) { agpr_write<A_O + I.value>(0u); });
  }

  float mrow[2] = {-INFINITY, -INFINITY}, lrow[2] = {0.f, 0.f};
  const float c2 = a.scale * LOG2E;
  const float* maskb = a.key_mask ? a.key_mask + (int64_t)b * S : nullptr;

  // lane-constant LDS offsets under swz64 (both tiles): b128 fragment row 32 kb + lj, chunk 2 ks + hi; transposed fragment rows
  // 16 t + 4 hi + j' (+ 8), d columns 32 db + 16 gq + 4 mq
  const int koff0 = lj * 256 + ((hi ^ swz64(lj)) << 4);
  const int jq = (lane & 15) >> 2, mq = lane & 3, gq = (lane >> 4) & 1;
  const int tbase = (4 * hi + jq) * 256 + (mq & 1) * 8;
  const int toffa = tbase + (((2 * gq + (mq >> 1)) ^ ((jq << 2) | hi)) << 4);
  const int toffb = tbase + 2048 + (((2 * gq + (mq >> 1)) ^ ((jq << 2) | (hi ^ 2))) << 4);

  f32x16 Sa[2], Sb[2];       // scores of the current / next sub-tile per query block (roles swap every sub-tile)
  u32x4 Pw[2][2];            // packed P of the current sub-tile: [query block][k-step u]
  constexpr int RK = 4, PFK = 3, RV = 4, PFV = 3;
  bf16x8 kf[RK], vf[RV];
  float mxs[2] = {0.f, 0.f}, negm[2] = {0.f, 0.f}, ls0[2] = {0.f, 0.f}, ls1[2] = {0.f, 0.f}, pe0[2] = {0.f, 0.f}, pe1[2] = {0.f, 0.f};
  float mt[2][4];

  int b_cur = 0, b_nxt = 1, b_nn = 2, b_dma = 3;
  if (live) {
    // tile 0 has landed (16 younger requests -- tiles 1, 2 -- may be in flight); every wave's has
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (live) {      // QK^T of sub-tile 0 -> Sa (no softmax to overlap yet)
    const char* sK0 = smem;
    sfor<8>([&](auto KS) {
      const bf16x8 f = *(const bf16x8*)(sK0 + (koff0 ^ (KS.value << 5)));
      mfma_vb<A_Q + 4 * KS.value, KS.value == 0>(Sa[0], f);
      mfma_vb<A_Q + 4 * (8 + KS.value), KS.value == 0>(Sa[1], f);
    });
    sfor<PFK>([&](auto P) { kf[P.value % RK] = *(const bf16x8*)(sK0 + 8192 + (koff0 ^ (P.value << 5))); });
  }

#if defined(QFX_A64_TIMING)
  uint64_t tph[6] = {0, 0, 0, 0, 0, 0}, tmark = __builtin_readcyclecounter();
#endif
  for (int jt = 0; jt < ntiles; ++jt) {
    A64_T(5);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    A64_T(0);
    if (!live) {
      stage(jt + 3, b_dma);
      const int t_ = b_cur; b_cur = b_nxt; b_nxt = b_nn; b_nn = b_dma; b_dma = t_;
      continue;
    }
    const char* sKc = smem + b_cur * 2 * TB;        // tile jt
    const char* sVc = sKc + TB;
    const char* sKn = smem + b_nxt * 2 * TB;        // tile jt + 1
    const int j0 = jt * 64;
    const bool need_mask = (j0 + 64 > S) || (maskb != nullptr);
    const float cs = need_mask ? 1.0f : c2;

    // one sub-tile iteration: KB = which 32-key half of tile jt is "current"; HASNEXT = a sub-tile follows (QK^T MFMAs exist)
    auto subtile = [&](auto KBc, auto HASNEXT, f32x16 (&Sc)[2], f32x16 (&Sn)[2]) {
      constexpr int kb = KBc.value;
      constexpr bool has_next = HASNEXT.value;
      const char* sKx = kb == 0 ? sKc + 8192 : sKn;           // K rows of sub-tile i + 1: (tile jt, half 1) or (tile jt + 1, half 0)
      const char* sKy = kb == 0 ? sKn : sKn + 8192;           // ... of sub-tile i + 2: its first fragments are requested at the end of Y
      auto kreq = [&](auto N) { if constexpr (N.value < 8 && has_next) kf[N.value % RK] = *(const bf16x8*)(sKx + (koff0 ^ (N.value << 5))); };
      auto kreq_next = [&](auto N) { kf[N.value % RK] = *(const bf16x8*)(sKy + (koff0 ^ (N.value << 5))); };
      auto vreq = [&](auto N) {      // V^T fragment n = 4 u + db of the current sub-tile
        constexpr int n = N.value;
        if constexpr (n < 8) {
          constexpr int t = 2 * kb + (n >> 2), db = n & 3;
          const bf16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((QFX_AS3 bf16x4v*)(sVc + (toffa ^ (db << 6)) + t * 4096));
          const bf16x4v hv = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((QFX_AS3 bf16x4v*)(sVc + (toffb ^ (db << 6)) + t * 4096));
          const bf16x4 l4 = __builtin_bit_cast(bf16x4, lo), h4 = __builtin_bit_cast(bf16x4, hv);
          bf16x8 r;
          r[0] = l4[0]; r[1] = l4[1]; r[2] = l4[2]; r[3] = l4[3]; r[4] = h4[0]; r[5] = h4[1]; r[6] = h4[2]; r[7] = h4[3];
          vf[n % RV] = r;
        }
      };
      // ---- softmax pieces of the current sub-tile (lane (lj, hi): value r = key 8 (r / 4) + 4 hi + r % 4 of the 32-key half)
      auto mask_cur = [&](int qb) {
        f32x16& sv = Sc[qb];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int key0 = j0 + 32 * kb + 8 * c + 4 * hi;
          f32x4 mk4 = {0.f, 0.f, 0.f, 0.f};
          if (maskb != nullptr) {
#pragma unroll
            for (int r = 0; r < 4; ++r) mk4[r] = maskb[(key0 + r) < S ? (key0 + r) : S - 1] * LOG2E;
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) sv[4 * c + r] = (key0 + r) < S ? sv[4 * c + r] * c2 + mk4[r] : -INFINITY;
        }
      };
      auto max_a = [&](auto QB) {
        constexpr int qb = QB.value;
        const f32x16& sv = Sc[qb];
        float c0, c1, c2_, c3;
        asm volatile("v_max3_f32 %0, %4, %5, %6\n\tv_max3_f32 %1, %7, %8, %9\n\tv_max3_f32 %2, %10, %11, %12\n\tv_max3_f32 %3, %13, %14, %15"
                     : "=&v"(c0), "=&v"(c1), "=&v"(c2_), "=&v"(c3)
                     : "v"(sv[0]), "v"(sv[1]), "v"(sv[2]), "v"(sv[3]), "v"(sv[4]), "v"(sv[5]), "v"(sv[6]), "v"(sv[7]), "v"(sv[8]), "v"(sv[9]), "v"(sv[10]), "v"(sv[11]));
        mt[qb][0] = c0; mt[qb][1] = c1; mt[qb][2] = c2_; mt[qb][3] = c3;
      };
      auto max_b = [&](auto QB) {
        constexpr int qb = QB.value;
        const f32x16& sv = Sc[qb];
        float c0 = mt[qb][0], c1 = mt[qb][1], c2_ = mt[qb][2];
        const float c3 = mt[qb][3];
        asm volatile("v_max3_f32 %0, %0, %4, %5\n\tv_max3_f32 %1, %1, %6, %7\n\tv_max_f32 %2, %2, %3\n\tv_max3_f32 %0, %0, %1, %2"
                     : "+v"(c0), "+v"(c1), "+v"(c2_) : "v"(c3), "v"(sv[12]), "v"(sv[13]), "v"(sv[14]), "v"(sv[15]));
        mxs[qb] = c0;
      };
      auto vote = [&](auto QB) {
        constexpr int qb = QB.value;
        float mx = mxs[qb];
        if (__builtin_expect(!__all(mx * cs - mrow[qb] <= ATTN_DEFER_MAX), 0)) {
          const uint32_t u1 = __float_as_uint(mx);
          const auto r1 = __builtin_amdgcn_permlane32_swap(u1, u1, false, false);
          mx = fmaxf(__uint_as_float(r1[0]), __uint_as_float(r1[1]));
          const float mnew = fmaxf(mrow[qb], mx * cs);
          const float alpha = fexp2_(mrow[qb] - ((mnew == -INFINITY) ? 0.f : mnew));
          const bool had = mrow[qb] != -INFINITY;        // (lane-varying; the rescale below is skipped wave-wide only for the first sub-tile)
          (void)had;
          mrow[qb] = mnew;
          lrow[qb] *= alpha;
          if (!(jt == 0 && kb == 0)) {          // O^T(., qb) *= alpha: only QK^T MFMAs are in flight in X
            sfor<64>([&](auto R) {
              constexpr int reg = A_O + 16 * (2 * (R.value / 16) + qb) + R.value % 16;
              agpr_write<reg>(__float_as_uint(agpr_read<reg>() * alpha));
            });
            asm volatile("s_nop 1");
          }
        }
        negm[qb] = (mrow[qb] == -INFINITY) ? 0.f : -mrow[qb];
        ls0[qb] = 0.f; ls1[qb] = 0.f; pe0[qb] = 0.f; pe1[qb] = 0.f;
      };
      auto pair = [&](auto QB, auto PI) {       // starts pair PI (fma, exp), finishes pair PI - 1 (row sums, packing)
        constexpr int qb = QB.value, pi = PI.value, k = 2 * pi;
        float t0, t1, l0 = ls0[qb], l1 = ls1[qb], e0 = pe0[qb], e1 = pe1[qb];
        uint32_t pw;
        asm volatile(
            "v_fma_f32 %5, %7, %9, %10\n\tv_fma_f32 %6, %8, %9, %10\n\tv_add_f32 %1, %1, %3\n\tv_add_f32 %2, %2, %4\n\t"
            "v_cvt_pk_bf16_f32 %0, %3, %4\n\t" A64_EXP " %3, %5\n\t" A64_EXP " %4, %6"
            : "=&v"(pw), "+v"(l0), "+v"(l1), "+v"(e0), "+v"(e1), "=&v"(t0), "=&v"(t1)
            : "v"(Sc[qb][k]), "v"(Sc[qb][k + 1]), "v"(cs), "v"(negm[qb]));
        ls0[qb] = l0; ls1[qb] = l1; pe0[qb] = e0; pe1[qb] = e1;
        if constexpr (pi > 0) Pw[qb][(pi - 1) >> 2][(pi - 1) & 3] = pw;
      };
      auto fin = [&](auto QB) {                 // finishes pair 7
        constexpr int qb = QB.value;
        float l0 = ls0[qb], l1 = ls1[qb];
        const float e0 = pe0[qb], e1 = pe1[qb];
        uint32_t pw;
        asm volatile("v_add_f32 %1, %1, %3\n\tv_add_f32 %2, %2, %4\n\tv_cvt_pk_bf16_f32 %0, %3, %4" : "=&v"(pw), "+v"(l0), "+v"(l1) : "v"(e0), "v"(e1));
        Pw[qb][1][3] = pw;
        lrow[qb] += l0 + l1;
      };
      using Q0 = std::integral_constant<int, 0>;
      using Q1 = std::integral_constant<int, 1>;
      // fillers of gap g (after the MFMA of gap g): X = 0..15, Y = 16..31
      auto filler = [&](auto G) {
        constexpr int g = G.value;
        if constexpr (g == 0) { if (need_mask) { mask_cur(0); mask_cur(1); } max_a(Q0{}); }
        else if constexpr (g == 1) max_b(Q0{});
        else if constexpr (g == 2) max_a(Q1{});
        else if constexpr (g == 3) max_b(Q1{});
        else if constexpr (g == 4) vote(Q0{});
        else if constexpr (g == 5) vote(Q1{});
        else if constexpr (g >= 6 && g <= 21) {
          constexpr int k = g - 6;                                    // pairs (0,0) (1,0) (0,1) (1,1) ... (1,7)
          if constexpr ((k & 1) == 0) pair(Q0{}, std::integral_constant<int, k / 2>{});
          else pair(Q1{}, std::integral_constant<int, k / 2>{});
        }
        else if constexpr (g == 22) fin(Q0{});
        else if constexpr (g == 23) fin(Q1{});
        else if constexpr (g >= 24 && g <= 27) stage_piece(jt + 3, b_dma, 4 * kb + (g - 24));
        else if constexpr (g >= 29 && g < 29 + PFK) kreq_next(std::integral_constant<int, g - 29>{});      // across the sub-tile (and tile) seam
      };
      // ---- X: QK^T of the next sub-tile (K fragment ks -> both query blocks), fragments requested PFK ahead
      sfor<8>([&](auto KS) {
        constexpr int ks = KS.value;
        kreq(std::integral_constant<int, ks + PFK>{});
        if constexpr (has_next) mfma_vb<A_Q + 4 * ks, ks == 0>(Sn[0], kf[ks % RK]);
        filler(std::integral_constant<int, 2 * ks>{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (has_next) mfma_vb<A_Q + 4 * (8 + ks), ks == 0>(Sn[1], kf[ks % RK]);
        if constexpr (ks >= 5) vreq(std::integral_constant<int, ks - 5>{});      // the first V^T fragments of Y, three gaps ahead
        filler(std::integral_constant<int, 2 * ks + 1>{});
        __builtin_amdgcn_sched_barrier(0);
      });
      A64_T(1 + 2 * kb);
      // ---- Y: PV of the current sub-tile (V^T fragment (u, db) -> both query blocks)
      sfor<8>([&](auto N) {
        constexpr int n = N.value, u = n >> 2, db = n & 3;
        vreq(std::integral_constant<int, n + PFV>{});
        mfma_pv<A_O + 16 * (2 * db + 0)>(vf[n % RV], Pw[0][u]);
        filler(std::integral_constant<int, 16 + 2 * n>{});
        __builtin_amdgcn_sched_barrier(0);
        mfma_pv<A_O + 16 * (2 * db + 1)>(vf[n % RV], Pw[1][u]);
        filler(std::integral_constant<int, 17 + 2 * n>{});
        __builtin_amdgcn_sched_barrier(0);
      });
      A64_T(2 + 2 * kb);
    };
    subtile(std::integral_constant<int, 0>{}, std::true_type{}, Sa, Sb);
    if (jt + 1 < ntiles) subtile(std::integral_constant<int, 1>{}, std::true_type{}, Sb, Sa);
    else subtile(std::integral_constant<int, 1>{}, std::false_type{}, Sb, Sa);
    { const int t_ = b_cur; b_cur = b_nxt; b_nxt = b_nn; b_nn = b_dma; b_dma = t_; }
  }
#if defined(QFX_A64_TIMING)
  if (lane == 0 && blockIdx.x < 16) {
    float* dbg = a.lse2 + ((int64_t)a.B * a.H) * a.S_pad + (blockIdx.x * 4 + w) * 8;
    for (int i = 0; i < 6; ++i) dbg[i] = (float)tph[i];
    dbg[6] = (float)ntiles;
  }
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (!live) return;

  // ---- epilogue: O = O^T / l -> bf16 -> staging slab [64 rows][272 B] of this wave -> 16-row fragments of the qfx_attn.hip layout
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the last PV MFMAs have written the accumulators
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the K requests for the tile past the last one
  char* stg = smem + w * (64 * STG_LD);
  float lse_out[2];
  sfor<2>([&](auto QB) {
    constexpr int qb = QB.value;
    float l = lrow[qb];
    l += __shfl_xor(l, 32);
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    lse_out[qb] = mrow[qb] + log2f(l);
    sfor<4>([&](auto DB) {
      sfor<4>([&](auto C) {
        constexpr int reg = A_O + 16 * (2 * DB.value + qb) + 4 * C.value;
        const float o0 = agpr_read<reg>() * inv, o1 = agpr_read<reg + 1>() * inv, o2 = agpr_read<reg + 2>() * inv, o3 = agpr_read<reg + 3>() * inv;
        const u32x2 u = {pack2bf(o0, o1), pack2bf(o2, o3)};
        *(u32x2*)(stg + (32 * qb + lj) * STG_LD + (32 * DB.value + 8 * C.value + 4 * hi) * 2) = u;
      });
    });
    const int q = q0 + 32 * qb + lj;
    if (q < S && hi == 0) a.lse2[((int64_t)b * a.H + h) * a.S_pad + q] = lse_out[qb];
  });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the slab is private to the wave: no block barrier
  const int g = lane >> 4, li = lane & 15;
  const bool wide = rows_16b(a.O, a.ldo);
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    if (q0 + 16 * f >= S) break;                           // wave-uniform
    u32x2 u[DH / 16];
#pragma unroll
    for (int d = 0; d < DH / 16; ++d) u[d] = *(const u32x2*)(stg + (16 * f + li) * STG_LD + (16 * d + 4 * g) * 2);
    const int q = q0 + 16 * f + li;
    const int qc = q < S ? q : S - 1;
    store_frag<DH>(a.O + ((int64_t)b * S + qc) * a.ldo + h * DH, u, g, q < S, wide);
    head_lora_frag<DH>(a.hl[0], h, a.T, q0 + 16 * f, (int64_t)b * S + qc, q < S, u, g, li);
  }
}

// =============================================================================================================================
// dQ with 64-query waves (round 5).  Per 64-key tile and wave: S^T = K Q^T and dP^T = V dO^T (2 x 32 MFMAs), dS = P (dP - dsum) on the VALU,
// dQ^T += K^T dS (32 MFMAs) -- 96 MFMAs against ~290 VALU instructions: 3 per MFMA where the forward has 4.5, so unlike the forward this
// loop can be bound by the matrix pipe.  Accumulator half: a[0:127] = dQ^T (4 d-blocks x 2 query blocks), a[128:191] = Q fragments,
// a[192:255] = dO fragments (both B operands).  Every LDS fragment (K / V rows for S / dP, K^T by transpose read for dQ) feeds TWO MFMAs, one per
// query block -- with one fragment per MFMA the four waves read 320 KB of LDS per tile, more than the 3072 MFMA cycles of a tile can
// deliver (first build: 132 us, 30 % of the wave cycles in s_waitcnt) -- and the tile is skewed by KEY halves instead of query blocks:
//   A  S(., kb 0), dP(., kb 0)     32 MFMAs | 16 K / V fragment reads
//   B  S(., kb 1), dP(., kb 1)     32 MFMAs | 16 fragment reads, dS pairs 0..7 of both query blocks (the kb 0 values)
//   C  dQ(.) += K^T dS, k-steps 0,1 16 MFMAs | 8 K^T fragments, pairs 8..11 (k-step 2)
//   D  k-steps 2,3                  16 MFMAs | 8 K^T fragments, pairs 12..15 before k-step 3, the LDS-DMA pieces of tile jt + 3
// Both tiles use ONE swizzle that serves ds_read_b128 (rows of a 16-lane group on one chunk) and the transpose read (4 rows x 4 chunks
// per half wave): chunk' = chunk ^ (((row & 3) << 2) | ((row >> 2) & 3)).

constexpr int A_DQ = 0, A_QF = 128, A_DO = 192;

__global__ __launch_bounds__(256, 1) void attn_bwd_dq64_kernel(const qfx_attn_args a) {
  constexpr int DH = 128;
  __shared__ __attribute__((aligned(16))) char smem[8 * TB];       // 4 x [K | V] stages (see the forward); epilogue slabs re-use it
  static_assert(8 * TB >= 4 * 64 * 512, "staging slabs must fit the ring");
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, lj = lane & 31;
#if defined(QFX_A64_TIMING)
  const uint64_t t_entry = __builtin_readcyclecounter();
#endif
  int xb, h, b;
  attn_block_coord((a.S + 255) / 256, a.H, xb, h, b);
  const int S = a.S;
  const int q0 = xb * 256 + w * 64;
  const bool live = q0 < S;
  const bf16_t* Kb = a.K + (int64_t)b * S * a.ldk + h * DH;
  const bf16_t* Vb = a.V + (int64_t)b * S * a.ldv + h * DH;
  const int ntiles = (S + 63) / 64;

  auto stage_piece = [&](int jt, int buf, int i) {
    char* dK = smem + buf * 2 * TB;
    const int rr = lane >> 4, c = lane & 15, ii = i & 3;
    const int row = 16 * w + 4 * ii + rr;
    int s = jt * 64 + row; s = s < S ? s : S - 1;
    const unsigned sc = (unsigned)((c ^ swz64(row)) * 8);
    if (i < 4) glds16a(Kb + (row_off(s, a.ldk) + sc), dK + (16 * w + 4 * ii) * 256);
    else glds16a(Vb + (row_off(s, a.ldv) + sc), dK + TB + (16 * w + 4 * ii) * 256);
  };
  auto stage = [&](int jt, int buf) {
#pragma unroll
    for (int i = 0; i < 8; ++i) stage_piece(jt, buf, i);
  };
  stage(0, 0);
  stage(1, 1);
  stage(2, 2);

  // ---- Q, dO fragments -> accumulator half; dsum[q] = sum_d dO[q, d] O[q, d] (computed here, published for dK / dV); lse2[q]
  float nlse[2] = {0.f, 0.f}, dsm[2] = {0.f, 0.f};
  if (live) {
    char* slab = smem + 3 * 2 * TB + w * 8192;      // ring stage 3 is free until tile 0 issues LDS-DMA into it (after the first barrier)
    const bf16_t* qbase = a.Q + (int64_t)b * S * a.ldq + h * DH;
    const bf16_t* dbase = a.dO + (int64_t)b * S * a.lddo + h * DH;
    const bf16_t* obase = a.O + (int64_t)b * S * a.ldo + h * DH;
    sfor<2>([&](auto QB) {
      constexpr int qb = QB.value;
      u32x4 qr[8], dr[8], orw[8], qv[8], dv[8], ov[8];
      rows_issue(qbase, a.ldq, q0 + 32 * qb, S, lane, qr);
      rows_issue(dbase, a.lddo, q0 + 32 * qb, S, lane, dr);
      rows_issue(obase, a.ldo, q0 + 32 * qb, S, lane, orw);
      int q = q0 + 32 * qb + lj; q = q < S ? q : S - 1;
      nlse[qb] = -a.lse2[((int64_t)b * a.H + h) * a.S_pad + q];
      rows_to_frags(slab, lane, qr, qv);
      rows_to_frags(slab, lane, dr, dv);
      rows_to_frags(slab, lane, orw, ov);
      float part = 0.f;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          part += __uint_as_float(dv[ks][j] << 16) * __uint_as_float(ov[ks][j] << 16);
          part += __uint_as_float(dv[ks][j] & 0xffff0000u) * __uint_as_float(ov[ks][j] & 0xffff0000u);
        }
      part += __shfl_xor(part, 32);
      dsm[qb] = part;
      if (hi == 0 && q0 + 32 * qb + lj < S) a.dsum[((int64_t)b * a.H + h) * a.S_pad + q] = part;
      sfor<8>([&](auto KS) {
        sfor<4>([&](auto I) {
          agpr_write<A_QF + 4 * (8 * qb + KS.value) + I.value>(qv[KS.value][I.value]);
          agpr_write<A_DO + 4 * (8 * qb + KS.value) + I.value>(dv[KS.value][I.value]);
        });
      });
    });
    sfor<128>([&](auto I) { agpr_write<A_DQ + I.value>(0u); });
  }
  const float c2 = a.scale * LOG2E;
  const float* maskb = a.key_mask ? a.key_mask + (int64_t)b * S : nullptr;

  // lane-constant LDS offsets under swz64.  b128 fragment of a row tile: row 32 kb + lj, chunk 2 ks + hi.  Transposed fragment: rows
  // 16 t + 4 hi + j' (first read, (row >> 2) & 3 = hi) and + 8 (second, = hi ^ 2), d columns 32 db + 16 gq + 4 mq.
  const int koff0 = lj * 256 + ((hi ^ swz64(lj)) << 4);                   // ^ (ks << 5), + kb * 8192
  const int jq = (lane & 15) >> 2, mq = lane & 3, gq = (lane >> 4) & 1;
  const int tbase = (4 * hi + jq) * 256 + (mq & 1) * 8;
  const int toffa = tbase + (((2 * gq + (mq >> 1)) ^ ((jq << 2) | hi)) << 4);              // ^ (db << 6), + t * 4096
  const int toffb = tbase + 2048 + (((2 * gq + (mq >> 1)) ^ ((jq << 2) | (hi ^ 2))) << 4);

  f32x16 Sq[2][2], Dq[2][2];              // scores and dP: [query block][key block]
  u32x4 Pq[2][4];                         // packed dS of a query block, B operand of k-step t
  int b_cur = 0, b_nxt = 1, b_nn = 2, b_dma = 3;
#if defined(QFX_A64_TIMING)
  uint64_t tph[6] = {0, 0, 0, 0, 0, 0}, tmark = __builtin_readcyclecounter();
  const uint64_t t_loop0 = tmark;
#endif
  for (int jt = 0; jt < ntiles; ++jt) {
    A64_T(5);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    A64_T(0);
    if (!live) {
      stage(jt + 3, b_dma);
      const int t_ = b_cur; b_cur = b_nxt; b_nxt = b_nn; b_nn = b_dma; b_dma = t_;
      continue;
    }
    const char* sK = smem + b_cur * 2 * TB;
    const char* sV = sK + TB;
    const int j0 = jt * 64;
    const bool need_mask = (j0 + 64 > S) || (maskb != nullptr);
    const float cs = need_mask ? 1.0f : c2;

    // K / V fragment m of a key half (m = 0..15: ks = m / 2, even = K, odd = V), through a 6-deep ring requested 5 fragments ahead, two
    // MFMAs (query block 0, 1) per fragment; K^T fragment n (n = 0..15: k-step t = n / 4, d block db = n & 3) through a 4-deep ring
    // requested 3 ahead.  hipcc counts all of these LDS reads.
    constexpr int PF = 5, RING = 6, PFT = 3, RINGT = 4;
    bf16x8 fr[RING], kt[RINGT];
    auto frag = [&](int g) {                // g = 0..31 over both key halves: kb = g / 16
      const int kb = g >> 4, ks = (g & 15) >> 1;
      return *(const bf16x8*)(((g & 1) ? sV : sK) + kb * 8192 + (koff0 ^ (ks << 5)));
    };
    auto ktreq = [&](auto N) {
      constexpr int n = N.value;
      if constexpr (n < 16) {
        const char* pa = sK + (toffa ^ ((n & 3) << 6)) + (n >> 2) * 4096;
        const char* pb = sK + (toffb ^ ((n & 3) << 6)) + (n >> 2) * 4096;
        const bf16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((QFX_AS3 bf16x4v*)pa);
        const bf16x4v hv = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((QFX_AS3 bf16x4v*)pb);
        const bf16x4 l4 = __builtin_bit_cast(bf16x4, lo), h4 = __builtin_bit_cast(bf16x4, hv);
        bf16x8 r;
        r[0] = l4[0]; r[1] = l4[1]; r[2] = l4[2]; r[3] = l4[3]; r[4] = h4[0]; r[5] = h4[1]; r[6] = h4[2]; r[7] = h4[3];
        kt[n % RINGT] = r;
      }
    };
    // one fragment of phase A / B: g = 16 KB + m -> S (even m) or dP (odd m) of BOTH query blocks, key block KB
    auto sd_frag = [&](auto KB, auto M, auto&& mid) {
      constexpr int kb = KB.value, m = M.value, ks = m >> 1, g = 16 * kb + m;
      if constexpr (g + PF < 32) fr[(g + PF) % RING] = frag(g + PF);
      if constexpr ((m & 1) == 0) mfma_vb<A_QF + 4 * ks, ks == 0>(Sq[0][kb], fr[g % RING]);
      else mfma_vb<A_DO + 4 * ks, ks == 0>(Dq[0][kb], fr[g % RING]);
      mid();
      __builtin_amdgcn_sched_barrier(0);
      if constexpr ((m & 1) == 0) mfma_vb<A_QF + 4 * (8 + ks), ks == 0>(Sq[1][kb], fr[g % RING]);
      else mfma_vb<A_DO + 4 * (8 + ks), ks == 0>(Dq[1][kb], fr[g % RING]);
    };
    // dS of one pair of keys (values k, k + 1 of query block qb; k = 16 kb + r): p = exp2(s cs - lse), ds = p (dp - dsum), packed -- in TWO
    // statements of five instructions, one behind each of the two MFMAs of a fragment: a lone wave hides <= 5 single-issue instructions
    // per 32x32x16 MFMA gap (MI355X_MICROARCH "one wave per SIMD"); ten behind the second MFMA and none behind the first measured fully
    // additive (phase B 1769 instead of ~1100 cycles).
    float pt0 = 0.f, pt1 = 0.f, pu0 = 0.f, pu1 = 0.f;
    auto ds_half1 = [&](auto QB, auto PI) {
      constexpr int qb = QB.value, k = 2 * PI.value;
      float t0, t1, u0, u1;        // (locals: clang does not capture a variable that is named only in an asm constraint of a nested generic lambda)
      asm volatile("v_fma_f32 %0, %4, %8, %9\n\tv_fma_f32 %1, %5, %8, %9\n\tv_sub_f32 %2, %6, %10\n\tv_sub_f32 %3, %7, %10\n\tv_exp_f32 %0, %0"
                   : "=&v"(t0), "=&v"(t1), "=&v"(u0), "=&v"(u1)
                   : "v"(Sq[qb][k >> 4][k & 15]), "v"(Sq[qb][k >> 4][(k & 15) + 1]), "v"(Dq[qb][k >> 4][k & 15]), "v"(Dq[qb][k >> 4][(k & 15) + 1]), "v"(cs),
                     "v"(nlse[qb]), "v"(dsm[qb]));
      pt0 = t0; pt1 = t1; pu0 = u0; pu1 = u1;
    };
    auto ds_half2 = [&](auto QB, auto PI) {
      constexpr int qb = QB.value, pi = PI.value;
      uint32_t pw;
      float t0 = pt0, t1 = pt1;
      const float u0 = pu0, u1 = pu1;
      asm volatile("v_exp_f32 %2, %2\n\tv_mul_f32 %1, %1, %3\n\tv_mul_f32 %2, %2, %4\n\tv_cvt_pk_bf16_f32 %0, %1, %2"
                   : "=&v"(pw), "+v"(t0), "+v"(t1) : "v"(u0), "v"(u1));
      Pq[qb][pi >> 2][pi & 3] = pw;
    };
    auto mask_block = [&](f32x16& sv, int kb) {      // off the headline path: scale + additive mask, -inf (=> p = 0) for keys past S
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int key0 = j0 + 32 * kb + 8 * c + 4 * hi;
        f32x4 mk4 = {0.f, 0.f, 0.f, 0.f};
        if (maskb != nullptr) {
#pragma unroll
          for (int r = 0; r < 4; ++r) mk4[r] = maskb[(key0 + r) < S ? (key0 + r) : S - 1] * LOG2E;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) sv[4 * c + r] = (key0 + r) < S ? sv[4 * c + r] * c2 + mk4[r] : -INFINITY;
      }
    };
    // one K^T fragment of phase C / D: dQ^T(db, qb) += K^T(t, db) dS(qb, t) for both query blocks
    auto dq_frag = [&](auto N, auto&& mid) {
      constexpr int n = N.value, t = n >> 2, db = n & 3;
      ktreq(std::integral_constant<int, n + PFT>{});
      mfma_pv<A_DQ + 16 * (2 * db + 0)>(kt[n % RINGT], Pq[0][t]);
      mid();
      __builtin_amdgcn_sched_barrier(0);
      mfma_pv<A_DQ + 16 * (2 * db + 1)>(kt[n % RINGT], Pq[1][t]);
    };

    sfor<PF>([&](auto P) { fr[P.value % RING] = frag(P.value); });
    // A: key half 0; the LDS-DMA pieces of tile jt + 3, one per four MFMAs
    sfor<16>([&](auto M) {
      constexpr int m = M.value;
      sd_frag(std::integral_constant<int, 0>{}, M, [&] { if constexpr (m % 2 == 1) stage_piece(jt + 3, b_dma, m / 2); });
      __builtin_amdgcn_sched_barrier(0);
    });
    A64_T(1);
    // B: key half 1; dS pairs 0..7 of both query blocks, one per fragment (the MFMAs that wrote S / dP of key half 0 are >= 2 back)
    sfor<16>([&](auto M) {
      constexpr int m = M.value;
      using QB = std::integral_constant<int, m & 1>;
      using PI = std::integral_constant<int, m / 2>;
      if constexpr (m == 0) { if (need_mask) { mask_block(Sq[0][0], 0); mask_block(Sq[1][0], 0); } }
      sd_frag(std::integral_constant<int, 1>{}, M, [&] { ds_half1(QB{}, PI{}); });
      ds_half2(QB{}, PI{});
      __builtin_amdgcn_sched_barrier(0);
    });
    A64_T(2);
    // C: k-steps 0, 1 of dQ; dS pairs 8..11 (k-step 2) of both query blocks
    sfor<PFT>([&](auto P) { ktreq(P); });
    sfor<8>([&](auto N) {
      constexpr int n = N.value;
      using QB = std::integral_constant<int, n & 1>;
      using PI = std::integral_constant<int, 8 + n / 2>;
      if constexpr (n == 0) { if (need_mask) { mask_block(Sq[0][1], 1); mask_block(Sq[1][1], 1); } }
      dq_frag(N, [&] { ds_half1(QB{}, PI{}); });
      ds_half2(QB{}, PI{});
      __builtin_amdgcn_sched_barrier(0);
    });
    A64_T(3);
    // D: k-steps 2, 3.  Every MFMA of k-step 3 (fragments 12..15) reads all of pairs 12..15, and those need S / dP of key half 1, complete only
    // at the end of B: 16 pairs for the 12 fragments of C and D's first half.  The four fragments of k-step 2 therefore carry TWO pairs
    // each (9 instructions per gap instead of 5: ~ +200 cycles per tile); the pair state lives in statement-local registers.
    sfor<8>([&](auto N) {
      constexpr int n = N.value;
      if constexpr (n < 4) {
        using P0 = std::integral_constant<int, 12 + n>;
        dq_frag(std::integral_constant<int, 8 + n>{}, [&] { ds_half1(std::integral_constant<int, 0>{}, P0{}); ds_half2(std::integral_constant<int, 0>{}, P0{}); });
        ds_half1(std::integral_constant<int, 1>{}, P0{});
        ds_half2(std::integral_constant<int, 1>{}, P0{});
      } else {
        dq_frag(std::integral_constant<int, 8 + n>{}, [&] {});
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    A64_T(4);
    { const int t_ = b_cur; b_cur = b_nxt; b_nxt = b_nn; b_nn = b_dma; b_dma = t_; }
  }
#if defined(QFX_A64_TIMING)
  if (lane == 0 && blockIdx.x < 16) {      // caller over-allocates dsum by 16 * 4 * 8 floats in the timing build
    float* dbg = a.dsum + ((int64_t)a.B * a.H) * a.S_pad + (blockIdx.x * 4 + w) * 8;
    for (int i = 0; i < 6; ++i) dbg[i] = (float)tph[i];
    dbg[6] = (float)ntiles;
    dbg[7] = (float)(t_loop0 - t_entry);       // prologue
  }
  const uint64_t t_loop1 = __builtin_readcyclecounter();
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (!live) return;

  // ---- epilogue: ALL of dQ^T (fp32) leaves the accumulator half first (the compiler's own MFMAs of the fused rank-r projection below take
  // accumulator registers of their choosing) -> LDS slab of this wave [64 rows][512 B], 16-byte chunks XOR-swizzled by row & 7 -> the
  // 16-row fragment layout of qfx_attn.hip -> its epilogue code (QK-norm + RoPE backward, wide stores, fused projection)
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  char* stg = smem + w * (64 * 512);
  const int g = lane >> 4, li = lane & 15;
  const bool wide = rows_16b(a.dQ, a.lddq);
  // operands of the fused QK-norm + RoPE backward: those of fragment 0 are requested BEFORE the accumulators are staged, those of
  // fragment f + 1 before fragment f is worked on (qfx_attn_common.h: nrb_load)
  NrbOps<DH> nops[2];
  auto nrb_req = [&](int f4, NrbOps<DH>& o) {
    int qc = q0 + 16 * f4 + li; qc = qc < S ? qc : S - 1;
    nrb_load<DH>(o, a.qk_saved + ((int64_t)b * S + qc) * a.ld_saved + h * DH + 4 * g,
                 a.rope + (int64_t)b * a.rope_bstride + ((int64_t)qc * (DH / 2) + 2 * g) * 2, (qc < a.T ? a.wq_txt : a.wq_img) + 4 * g);
  };
  if (a.qk_saved) nrb_req(0, nops[0]);
  sfor<2>([&](auto QB) {
    sfor<4>([&](auto DB) {
      sfor<4>([&](auto C) {
        constexpr int reg = A_DQ + 16 * (2 * DB.value + QB.value) + 4 * C.value;
        const f32x4 v = {agpr_read<reg>(), agpr_read<reg + 1>(), agpr_read<reg + 2>(), agpr_read<reg + 3>()};
        const int row = 32 * QB.value + lj, chunk = 8 * DB.value + 2 * C.value + hi;
        *(f32x4*)(stg + row * 512 + ((chunk ^ (row & 7)) << 4)) = v;
      });
    });
  });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  sfor<4>([&](auto F4) {
    constexpr int f4 = F4.value;
    const int qf0 = q0 + 16 * f4;
    if (qf0 >= S) return;                               // wave-uniform
    f32x4 dq[DH / 16];
    const int row = 16 * f4 + li;
#pragma unroll
    for (int d = 0; d < DH / 16; ++d) dq[d] = *(const f32x4*)(stg + row * 512 + (((4 * d + g) ^ (row & 7)) << 4));
    const int q = qf0 + li;
    const int qc = q < S ? q : S - 1;
    bf16_t* op = a.dQ + ((int64_t)b * S + qc) * a.lddq + h * DH;
    u32x2 u[DH / 16];
    if (a.qk_saved) {
      if constexpr (f4 + 1 < 4) { if (qf0 + 16 < S) nrb_req(f4 + 1, nops[(f4 + 1) & 1]); }
      norm_rope_bwd_ops<DH>(dq, a.scale, nops[f4 & 1], a.norm_eps, a.norm_flags, u);
      store_frag<DH>(op, u, g, q < S, wide);
      head_lora_frag<DH>(a.hl[1], h, a.T, qf0, (int64_t)b * S + qc, q < S, u, g, li);
    } else {
#pragma unroll
      for (int d = 0; d < DH / 16; ++d) {
        u[d][0] = pack2bf(dq[d][0] * a.scale, dq[d][1] * a.scale);
        u[d][1] = pack2bf(dq[d][2] * a.scale, dq[d][3] * a.scale);
      }
      store_frag<DH>(op, u, g, q < S, wide);
    }
  });
#if defined(QFX_A64_TIMING)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0 && blockIdx.x < 16) {
    float* dbg = a.dsum + ((int64_t)a.B * a.H) * a.S_pad + (blockIdx.x * 4 + w) * 8;
    dbg[5] = (float)(__builtin_readcyclecounter() - t_loop1);      // epilogue (overwrites "loop rest")
  }
#endif
}

}  // namespace

namespace qfxi {
// launcher used by qfx_attn_fwd (qfx_attn.hip); arguments are validated there
int launch_attn_fwd64(const qfx_attn_args* a, hipStream_t stream) {
  dim3 grid(((a->S + 255) / 256) * a->H * a->B);
  hipLaunchKernelGGL(attn_fwd64_kernel, grid, dim3(256), 0, stream, *a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? QFX_OK : -(1000 + (int)e);
}
int launch_attn_fwd64p(const qfx_attn_args* a, hipStream_t stream) {
  dim3 grid(((a->S + 255) / 256) * a->H * a->B);
  hipLaunchKernelGGL(attn_fwd64p_kernel, grid, dim3(256), 0, stream, *a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? QFX_OK : -(1000 + (int)e);
}
int launch_attn_bwd_dq64(const qfx_attn_args* a, hipStream_t stream) {
  dim3 grid(((a->S + 255) / 256) * a->H * a->B);
  hipLaunchKernelGGL(attn_bwd_dq64_kernel, grid, dim3(256), 0, stream, *a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? QFX_OK : -(1000 + (int)e);
}
}  // namespace qfxi
