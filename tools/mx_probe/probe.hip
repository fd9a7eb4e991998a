// Lab (NOT product): lane <-> element mapping of v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 e4m3 operands, E8M0 block scales).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) int v8i;
typedef __attribute__((ext_vector_type(4))) float v4f;
// block b: A bytes [64][32] at A + b*2048 (or shared when a_stride == 0), same for B; scales [64] ints; out D [64][4]
__global__ void probe(const uint8_t* A, int a_stride, const uint8_t* B, int b_stride, const int* sa, int sa_stride, const int* sb, float* D) {
  const int l = threadIdx.x, blk = blockIdx.x;
  const uint8_t* a_ = A + (size_t)blk * a_stride;
  const uint8_t* b_ = B + (size_t)blk * b_stride;
  v8i a, b;
  for (int i = 0; i < 8; ++i) { a[i] = ((const int*)(a_ + l * 32))[i]; b[i] = ((const int*)(b_ + l * 32))[i]; }
  v4f c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sa[blk * sa_stride + l], 0, sb[l]);
  for (int i = 0; i < 4; ++i) D[(size_t)blk * 256 + l * 4 + i] = c[i];
}
extern "C" int run_probe(const void* A, int a_stride, const void* B, int b_stride, const void* sa, int sa_stride, const void* sb, void* D, int nblk, void* stream) {
  hipLaunchKernelGGL(probe, dim3(nblk), dim3(64), 0, (hipStream_t)stream, (const uint8_t*)A, a_stride, (const uint8_t*)B, b_stride,
                     (const int*)sa, sa_stride, (const int*)sb, (float*)D);
  return (int)hipGetLastError();
}
