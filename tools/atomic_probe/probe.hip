// Lab (NOT product): sustained rate of fp32 global atomic adds in the access pattern a single-pass attention backward would use for
// dQ: every (head, key-block) workgroup adds a [64 x 128] fp32 tile per query tile into the head's dQ accumulator [S, 128].
#include <hip/hip_runtime.h>
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>
#include <stdint.h>
// grid = heads * kblocks; block = 256 threads; acc [heads][S][128] fp32
template <int MODE>
__global__ __launch_bounds__(256) void k(float* acc, int S, int kblocks, int xcd_aware) {
  int bid = blockIdx.x;
  if (xcd_aware) {  // all blocks of a head on one XCD (bid % 8 == xcd): virtual id walks a contiguous range per XCD
    const int nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  }
  const int h = bid / kblocks, kb = bid % kblocks;
  float* a = acc + (size_t)h * S * 128;
  const int t = threadIdx.x;
  const int ntiles = S / 64;
  for (int it = 0; it < ntiles; ++it) {
    const int qt = (it + kb * 3) % ntiles;     // different key blocks walk the query tiles with different phases
    float* tile = a + (size_t)qt * 64 * 128;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      float* p = tile + i * 256 + t;           // 64 consecutive floats per wave instruction
      const float v = 1.0f + (float)i;
      if (MODE == 0) unsafeAtomicAdd(p, v);
      else if (MODE == 1) atomicAdd(p, v);
      else __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
}
extern "C" int run(float* acc, int heads, int S, int kblocks, int mode, int xcd_aware, void* stream) {
  dim3 g(heads * kblocks), b(256);
  if (mode == 0) hipLaunchKernelGGL(k<0>, g, b, 0, (hipStream_t)stream, acc, S, kblocks, xcd_aware);
  else if (mode == 1) hipLaunchKernelGGL(k<1>, g, b, 0, (hipStream_t)stream, acc, S, kblocks, xcd_aware);
  else hipLaunchKernelGGL(k<2>, g, b, 0, (hipStream_t)stream, acc, S, kblocks, xcd_aware);
  return (int)hipGetLastError();
}
