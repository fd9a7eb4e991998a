// Prototype: rank-r down projection U[M,R] = X[M,K] (W_hi + W_lo)[R,K]^T as a split-K kernel whose blocks share their weight slice
// through LDS (8 waves x 16 rows per block): the product kernel gives every 16-row block ALL of W through its own L1 (2-6x the
// X bytes per CU), and a CU moves only ~10-16 B/clk.  Partial sums go to U with fp32 atomics (prototype: timing + correctness).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

template <int NF, int KS>   // KS = k-steps (32 columns) per split, multiple of SUB
__global__ __launch_bounds__(512) void down_sk(const bf16_t* __restrict__ X, int64_t ldx, int M, const bf16_t* __restrict__ Whi,
                                                const bf16_t* __restrict__ Wlo, int64_t ldw, float* __restrict__ U, int ldu, int nsplit) {
  constexpr int SUB = 6, NC = KS / SUB;
  constexpr int WROW = SUB * 64 + 16;                 // padded LDS row (bytes): 36-dword lane stride -> conflict-free b128 reads
  constexpr int ROWS = 2 * NF * 16;                   // hi rows then lo rows
  constexpr int PIECES = ROWS * SUB * 4;              // 16-byte pieces per sub-chunk
  constexpr int NP = (PIECES + 511) / 512;
  __shared__ __attribute__((aligned(16))) char sW[2][ROWS * WROW];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, li = lane & 15;
  const int rb = blockIdx.x / nsplit, sp = blockIdx.x % nsplit;
  const int m0 = rb * 128 + w * 16;
  int mr = m0 + li; mr = mr < M ? mr : M - 1;
  const bf16_t* xrow = X + (int64_t)mr * ldx + (int64_t)sp * KS * 32 + 8 * g;
  bf16x8 xs[KS];
#pragma unroll
  for (int i = 0; i < KS; ++i) xs[i] = *(const bf16x8*)(xrow + i * 32);
  // cooperative weight loader: piece e -> (row, 16-byte chunk) of the sub-chunk
  const bf16_t* wsrc[NP]; int wdst[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    int e = tid + p * 512; e = e < PIECES ? e : PIECES - 1;
    const int row = e / (SUB * 4), ch = e % (SUB * 4);
    const bf16_t* base = row < NF * 16 ? Whi + (int64_t)row * ldw : Wlo + (int64_t)(row - NF * 16) * ldw;
    wsrc[p] = base + (int64_t)sp * KS * 32 + ch * 8;
    wdst[p] = row * WROW + ch * 16;
  }
  u32x4 wreg[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) wreg[p] = *(const u32x4*)(wsrc[p]);
  f32x4 acc[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) acc[nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    char* buf = sW[c & 1];
#pragma unroll
    for (int p = 0; p < NP; ++p) if (tid + p * 512 < PIECES) *(u32x4*)(buf + wdst[p]) = wreg[p];
    __syncthreads();
    if (c + 1 < NC) {
#pragma unroll
      for (int p = 0; p < NP; ++p) wreg[p] = *(const u32x4*)(wsrc[p] + (c + 1) * SUB * 32);
    }
#pragma unroll
    for (int i = 0; i < SUB; ++i)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        const bf16x8 bh = *(const bf16x8*)(buf + (nf * 16 + li) * WROW + (i * 4 + g) * 16);
        const bf16x8 bl = *(const bf16x8*)(buf + (NF * 16 + nf * 16 + li) * WROW + (i * 4 + g) * 16);
        acc[nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xs[c * SUB + i], bh, acc[nf], 0, 0, 0);
        acc[nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xs[c * SUB + i], bl, acc[nf], 0, 0, 0);
      }
  }
#pragma unroll
  for (int nf = 0; nf < NF; ++nf)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + 4 * g + r;
      if (m < M) atomicAdd(U + (int64_t)m * ldu + nf * 16 + li, acc[nf][r]);
    }
}

extern "C" int run_down_sk(const void* X, int64_t ldx, int M, int K, const void* Whi, const void* Wlo, int64_t ldw, int R, float* U, int ldu,
                           void* stream) {
  const int nsplit = 16, rbs = (M + 127) / 128;
  dim3 grid(rbs * nsplit), block(512);
  hipStream_t s = (hipStream_t)stream;
  const bf16_t* x = (const bf16_t*)X; const bf16_t* wh = (const bf16_t*)Whi; const bf16_t* wl = (const bf16_t*)Wlo;
  if (K == 3072 && R == 16) hipLaunchKernelGGL((down_sk<1, 6>), grid, block, 0, s, x, ldx, M, wh, wl, ldw, U, ldu, nsplit);
  else if (K == 3072 && R == 48) hipLaunchKernelGGL((down_sk<3, 6>), grid, block, 0, s, x, ldx, M, wh, wl, ldw, U, ldu, nsplit);
  else if (K == 12288 && R == 16) hipLaunchKernelGGL((down_sk<1, 24>), grid, block, 0, s, x, ldx, M, wh, wl, ldw, U, ldu, nsplit);
  else if (K == 12288 && R == 48) hipLaunchKernelGGL((down_sk<3, 24>), grid, block, 0, s, x, ldx, M, wh, wl, ldw, U, ldu, nsplit);
  else return -1;
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
