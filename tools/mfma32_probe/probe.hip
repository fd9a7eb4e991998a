// Probe of v_mfma_f32_32x32x16_bf16's operand / result lane mapping (tools/mfma32_probe/run.py compares with a host product).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ void probe(const uint16_t* A /*[32][16]*/, const uint16_t* B /*[32][16] = B^T*/, float* out /*[64][16]*/) {
  const int l = threadIdx.x;
  const bf16x8 a = *(const bf16x8*)(A + (l & 31) * 16 + 8 * (l >> 5));   // lane: row l%32, k = 8*(l/32)..+7
  const bf16x8 b = *(const bf16x8*)(B + (l & 31) * 16 + 8 * (l >> 5));   // lane: col l%32, same k
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int i = 0; i < 16; ++i) out[l * 16 + i] = c[i];
}
extern "C" int run_probe(const uint16_t* A, const uint16_t* B, float* out, void* stream) {
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, (hipStream_t)stream, A, B, out);
  return (int)hipGetLastError();
}
