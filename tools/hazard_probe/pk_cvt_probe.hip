// pk_cvt_probe.hip -- does a VALU that reads the result of a packed-fp32 op (v_pk_mul_f32) ONE instruction later see all of it?
//
// Round 5: the fused dQ epilogue of qfx_attn.hip came out different from run to run when hipcc compiled
//     out = pack2bf((dn - xh * dot) * rstd, ...)
// to   v_pk_mul_f32 v[66:67], ... ; v_pk_mul_f32 v[68:69], ... ; v_cvt_pk_bf16_f32 v66, v66, v67 ; v_cvt_pk_bf16_f32 v67, v68, v69
// (tools/nondet_bisect.py: ONE column -- r = 3 of lane group g = 3, i.e. the HIGH result register, lanes 48-63 -- wrong in all 16 rows
// of a fragment, everything else a consequence).  This probe isolates the instruction pair in inline asm with explicit registers and
// counts wrong lanes per mode:
//   0  pk_mul A ; pk_mul B ; cvt(A) ; cvt(B)            the compiler's sequence (B's consumer one instruction after its producer)
//   1  pk_mul A ; pk_mul B ; s_nop 0 ; cvt(A) ; cvt(B)  one idle state
//   2  pk_mul B ; cvt(B)                                 back to back
//   3  mul ; mul ; mul ; mul ; cvt(A) ; cvt(B)          scalar producers
//   4  pk_mul A ; pk_mul B ; s_nop 7 ; cvt(A) ; cvt(B)  far apart (control)
//   5  pk_mul B ; v_mov (independent) ; v_add_f32 consumer of B.hi    -- is it the consumer type or any VALU?
//   6  8 x global_load_dwordx4 (un-waited, other registers) ; 16 x [mode-0 sequence] ; s_waitcnt -- do VMEM returns landing inside
//      the sequence matter?   7 = the same with the loads waited for first (control)
//   8  the scale as the LOW half of an SGPR pair (op_sel_hi:[0,1]) whose HIGH half is SALU scratch rewritten around the packed ops; 9 = control
// build: hipcc --offload-arch=gfx950 -O2 -o pk_cvt_probe pk_cvt_probe.hip ;  run: ./pk_cvt_probe [blocks] [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

__device__ __forceinline__ uint32_t bf16_rne(float f) {
  uint32_t u = __float_as_uint(f);
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

template <int MODE>
__global__ __launch_bounds__(512) void probe(const float* __restrict__ in, unsigned long long* bad, unsigned long long* lane_hist, int iters, float s) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  float x0 = in[(tid * 4 + 0) & 0xffff], x1 = in[(tid * 4 + 1) & 0xffff], y0 = in[(tid * 4 + 2) & 0xffff], y1 = in[(tid * 4 + 3) & 0xffff];
  unsigned long long nbad = 0;
  if constexpr (MODE >= 10) {
    // modes 10-12: the partner wave of every SIMD (waves 4-7 of the 512-thread block) runs a dense MFMA loop -- its results come back
    // through the same VGPR write port -- while waves 0-3 run the mode-0 / mode-8 sequence.  10: partner on v_mfma 16x16x32 bf16;
    // 11: partner idle (control);  12: partner MFMA, probe = SGPR-pair form (mode 8 without the s7 rewrites)
    if ((threadIdx.x >> 6) >= 4) {
      if (MODE != 11) {
        for (int it = 0; it < iters * 6; ++it)
          asm volatile(".rept 16\n\tv_mfma_f32_16x16x32_bf16 v[40:43], v[24:27], v[28:31], v[40:43]\n\tv_mfma_f32_16x16x32_bf16 v[44:47], v[24:27], v[28:31], v[44:47]\n\t.endr"
                       ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31");
      }
      return;
    }
    for (int it = 0; it < iters; ++it) {
      uint32_t acc0, acc1;
      asm volatile(
          "v_mov_b32 v20, %[s]\n\tv_readfirstlane_b32 s6, %[s]\n\tv_mov_b32 v12, %[x0]\n\tv_mov_b32 v13, %[x1]\n\tv_mov_b32 v14, %[y0]\n\tv_mov_b32 v15, %[y1]\n\t"
          "v_mov_b32 v18, 0\n\tv_mov_b32 v19, 0\n\ts_nop 4\n\t"
          ".rept 48\n\t"
          "v_mov_b32 v10, 0\n\tv_mov_b32 v11, 0\n\tv_mov_b32 v16, 0\n\tv_mov_b32 v17, 0\n\t"
          ".if %[mode] == 12\n\t"
          "v_pk_mul_f32 v[10:11], s[6:7], v[12:13] op_sel_hi:[0,1]\n\tv_pk_mul_f32 v[16:17], s[6:7], v[14:15] op_sel_hi:[0,1]\n\t"
          ".else\n\t"
          "v_pk_mul_f32 v[10:11], v[20:21], v[12:13] op_sel_hi:[0,1]\n\tv_pk_mul_f32 v[16:17], v[20:21], v[14:15] op_sel_hi:[0,1]\n\t"
          ".endif\n\t"
          "v_cvt_pk_bf16_f32 v10, v10, v11\n\tv_cvt_pk_bf16_f32 v11, v16, v17\n\t"
          "v_xor_b32 v18, v18, v10\n\tv_xor_b32 v19, v19, v11\n\t"
          ".endr\n\t"
          "s_nop 4\n\tv_mov_b32 %[o0], v18\n\tv_mov_b32 %[o1], v19\n\t"
          : [o0] "=v"(acc0), [o1] "=v"(acc1)
          : [s] "v"(s), [x0] "v"(x0), [x1] "v"(x1), [y0] "v"(y0), [y1] "v"(y1), [mode] "i"(MODE)
          : "s6", "s7", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21");
      if (acc0 != 0u || acc1 != 0u) {
        ++nbad; atomicAdd(&lane_hist[lane], 1ull);
        if (acc1 & 0xffffu) atomicAdd(&lane_hist[64], 1ull);
        if (acc1 >> 16) atomicAdd(&lane_hist[65], 1ull);
        if (acc0) atomicAdd(&lane_hist[66], 1ull);
      }
      x0 = x0 * 1.0009765625f + 0.37f; y1 = y1 * 0.9990234375f - 0.23f;
      if (fabsf(x0) > 1e6f) x0 = 1.f;
    }
    if (nbad) atomicAdd(bad, nbad);
    return;
  }
  for (int it = 0; it < iters; ++it) {
    uint32_t o0, o1;
    float f5 = 0.f;
    if constexpr (MODE == 8 || MODE == 9) {
      // the epilogue's actual form: v_pk_mul_f32 vdst, s[6:7], v[pair] op_sel_hi:[0,1] -- the scale broadcast from the LOW half of an SGPR
      // pair whose HIGH half (s7) the compiler treats as undefined and re-uses as SALU scratch right around the packed op (mode 8);
      // mode 9 = the same with s7 left alone (control)
      uint32_t acc0, acc1;
      asm volatile(
          "v_mov_b32 v12, %[x0]\n\tv_mov_b32 v13, %[x1]\n\tv_mov_b32 v14, %[y0]\n\tv_mov_b32 v15, %[y1]\n\t"
          "v_readfirstlane_b32 s6, %[s]\n\ts_mov_b32 s7, 0x3039\n\tv_mov_b32 v18, 0\n\tv_mov_b32 v19, 0\n\ts_nop 4\n\t"
          ".rept 48\n\t"
          "v_mov_b32 v10, 0\n\tv_mov_b32 v11, 0\n\tv_mov_b32 v16, 0\n\tv_mov_b32 v17, 0\n\t"
          ".if %[mode] == 8\n\ts_mul_i32 s7, s7, 0x41c64e6d\n\t.endif\n\t"
          "v_pk_mul_f32 v[10:11], s[6:7], v[12:13] op_sel_hi:[0,1]\n\t"
          ".if %[mode] == 8\n\ts_add_i32 s7, s7, 0x3039\n\t.endif\n\t"
          "v_pk_mul_f32 v[16:17], s[6:7], v[14:15] op_sel_hi:[0,1]\n\t"
          ".if %[mode] == 8\n\ts_ashr_i32 s7, s7, 3\n\t.endif\n\t"
          "v_cvt_pk_bf16_f32 v10, v10, v11\n\t"
          "v_cvt_pk_bf16_f32 v11, v16, v17\n\t"
          "v_xor_b32 v18, v18, v10\n\tv_xor_b32 v19, v19, v11\n\t"
          ".endr\n\t"
          "s_nop 4\n\tv_mov_b32 %[o0], v18\n\tv_mov_b32 %[o1], v19\n\t"
          : [o0] "=v"(acc0), [o1] "=v"(acc1)
          : [s] "v"(s), [x0] "v"(x0), [x1] "v"(x1), [y0] "v"(y0), [y1] "v"(y1), [mode] "i"(MODE)
          : "s6", "s7", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19");
      if (acc0 != 0u || acc1 != 0u) {
        ++nbad; atomicAdd(&lane_hist[lane], 1ull);
        if (acc1 & 0xffffu) atomicAdd(&lane_hist[64], 1ull);
        if (acc1 >> 16) atomicAdd(&lane_hist[65], 1ull);
        if (acc0) atomicAdd(&lane_hist[66], 1ull);
      }
    } else if constexpr (MODE == 6 || MODE == 7) {
      uint32_t acc0, acc1;
      const float* lp = in + ((tid * 4 + it * 64) & 0x3fff);
      asm volatile(
          "v_mov_b32 v20, %[s]\n\tv_mov_b32 v12, %[x0]\n\tv_mov_b32 v13, %[x1]\n\tv_mov_b32 v14, %[y0]\n\tv_mov_b32 v15, %[y1]\n\t"
          "v_mov_b32 v18, 0\n\tv_mov_b32 v19, 0\n\t"
          "global_load_dwordx4 v[30:33], %[lp], off\n\tglobal_load_dwordx4 v[34:37], %[lp], off offset:256\n\t"
          "global_load_dwordx4 v[38:41], %[lp], off offset:512\n\tglobal_load_dwordx4 v[42:45], %[lp], off offset:768\n\t"
          "global_load_dwordx4 v[46:49], %[lp], off offset:1024\n\tglobal_load_dwordx4 v[50:53], %[lp], off offset:1280\n\t"
          "global_load_dwordx4 v[54:57], %[lp], off offset:1536\n\tglobal_load_dwordx4 v[58:61], %[lp], off offset:1792\n\t"
          ".if %[mode] == 7\n\ts_waitcnt vmcnt(0)\n\t.endif\n\t"
          ".rept 48\n\t"
          "v_mov_b32 v10, 0\n\tv_mov_b32 v11, 0\n\tv_mov_b32 v16, 0\n\tv_mov_b32 v17, 0\n\t"
          "v_pk_mul_f32 v[10:11], v[20:21], v[12:13] op_sel_hi:[0,1]\n\t"
          "v_pk_mul_f32 v[16:17], v[20:21], v[14:15] op_sel_hi:[0,1]\n\t"
          "v_cvt_pk_bf16_f32 v10, v10, v11\n\t"
          "v_cvt_pk_bf16_f32 v11, v16, v17\n\t"
          "v_xor_b32 v18, v18, v10\n\tv_xor_b32 v19, v19, v11\n\t"
          ".endr\n\t"
          "s_waitcnt vmcnt(0)\n\ts_nop 4\n\tv_mov_b32 %[o0], v18\n\tv_mov_b32 %[o1], v19\n\t"
          : [o0] "=v"(acc0), [o1] "=v"(acc1)
          : [s] "v"(s), [x0] "v"(x0), [x1] "v"(x1), [y0] "v"(y0), [y1] "v"(y1), [lp] "v"(lp), [mode] "i"(MODE)
          : "memory", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v30", "v31", "v32", "v33", "v34", "v35", "v36",
            "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55",
            "v56", "v57", "v58", "v59", "v60", "v61");
      if (acc0 != 0u || acc1 != 0u) {     // 48 identical results XOR to zero
        ++nbad; atomicAdd(&lane_hist[lane], 1ull);
        if (acc1 & 0xffffu) atomicAdd(&lane_hist[64], 1ull);
        if (acc1 >> 16) atomicAdd(&lane_hist[65], 1ull);
        if (acc0) atomicAdd(&lane_hist[66], 1ull);
      }
    } else if constexpr (MODE == 5) {
      asm volatile(
          "v_mov_b32 v20, %[s]\n\tv_mov_b32 v14, %[y0]\n\tv_mov_b32 v15, %[y1]\n\tv_mov_b32 v16, 0\n\tv_mov_b32 v17, 0\n\ts_nop 4\n\t"
          "v_pk_mul_f32 v[16:17], v[20:21], v[14:15] op_sel_hi:[0,1]\n\t"
          "v_mov_b32 v22, v20\n\t"
          "v_add_f32 v23, v17, v17\n\t"
          "s_nop 4\n\tv_mov_b32 %[f5], v23\n\t"
          : [f5] "=v"(f5) : [s] "v"(s), [y0] "v"(y0), [y1] "v"(y1)
          : "v14", "v15", "v16", "v17", "v20", "v21", "v22", "v23");
      if (f5 != 2.f * (s * y1)) { ++nbad; atomicAdd(&lane_hist[lane], 1ull); }
    } else {
      asm volatile(
          "v_mov_b32 v20, %[s]\n\tv_mov_b32 v12, %[x0]\n\tv_mov_b32 v13, %[x1]\n\tv_mov_b32 v14, %[y0]\n\tv_mov_b32 v15, %[y1]\n\t"
          "v_mov_b32 v10, 0\n\tv_mov_b32 v11, 0\n\tv_mov_b32 v16, 0\n\tv_mov_b32 v17, 0\n\ts_nop 4\n\t"
          // ---- producers
          ".if %[mode] == 3\n\t"
          "v_mul_f32 v10, v20, v12\n\tv_mul_f32 v11, v20, v13\n\tv_mul_f32 v16, v20, v14\n\tv_mul_f32 v17, v20, v15\n\t"
          ".elseif %[mode] == 2\n\t"
          "v_mul_f32 v10, v20, v12\n\tv_mul_f32 v11, v20, v13\n\ts_nop 4\n\tv_cvt_pk_bf16_f32 v10, v10, v11\n\ts_nop 4\n\t"
          "v_pk_mul_f32 v[16:17], v[20:21], v[14:15] op_sel_hi:[0,1]\n\t"
          ".else\n\t"
          "v_pk_mul_f32 v[10:11], v[20:21], v[12:13] op_sel_hi:[0,1]\n\t"
          "v_pk_mul_f32 v[16:17], v[20:21], v[14:15] op_sel_hi:[0,1]\n\t"
          ".endif\n\t"
          ".if %[mode] == 1\n\ts_nop 0\n\t.endif\n\t"
          ".if %[mode] == 4\n\ts_nop 7\n\t.endif\n\t"
          // ---- consumers
          ".if %[mode] != 2\n\tv_cvt_pk_bf16_f32 v10, v10, v11\n\t.endif\n\t"
          "v_cvt_pk_bf16_f32 v11, v16, v17\n\t"
          "s_nop 4\n\tv_mov_b32 %[o0], v10\n\tv_mov_b32 %[o1], v11\n\t"
          : [o0] "=v"(o0), [o1] "=v"(o1)
          : [s] "v"(s), [x0] "v"(x0), [x1] "v"(x1), [y0] "v"(y0), [y1] "v"(y1), [mode] "i"(MODE)
          : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v20", "v21");
      const uint32_t e0 = bf16_rne(s * x0) | (bf16_rne(s * x1) << 16);
      const uint32_t e1 = bf16_rne(s * y0) | (bf16_rne(s * y1) << 16);
      if (o0 != e0 || o1 != e1) {
        ++nbad;
        atomicAdd(&lane_hist[lane], 1ull);
        if ((o1 & 0xffffu) != (e1 & 0xffffu)) atomicAdd(&lane_hist[64], 1ull);     // B.lo wrong
        if ((o1 >> 16) != (e1 >> 16)) atomicAdd(&lane_hist[65], 1ull);              // B.hi wrong
        if (o0 != e0) atomicAdd(&lane_hist[66], 1ull);                              // A wrong
      }
    }
    // new data every iteration (cheap LCG on the bit patterns, kept finite)
    x0 = x0 * 1.0009765625f + 0.37f; x1 = x1 * 0.99951171875f - 0.11f; y0 = y0 * 1.00048828125f + 0.59f; y1 = y1 * 0.9990234375f - 0.23f;
    if (fabsf(x0) > 1e6f) x0 = 1.f; if (fabsf(y0) > 1e6f) y0 = 1.f;
  }
  if (nbad) atomicAdd(bad, nbad);
}

template <int MODE>
void run(const float* in, int blocks, int iters, const char* what) {
  unsigned long long *bad, *hist;
  hipMalloc(&bad, 8); hipMalloc(&hist, 8 * 68);
  hipMemset(bad, 0, 8); hipMemset(hist, 0, 8 * 68);
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(512), 0, 0, in, bad, hist, iters, 0.08838834764831845f);
  hipDeviceSynchronize();
  unsigned long long hb, hh[68];
  hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(hh, hist, 8 * 68, hipMemcpyDeviceToHost);
  unsigned long long q[4] = {0, 0, 0, 0};
  for (int l = 0; l < 64; ++l) q[l / 16] += hh[l];
  printf("{\"mode\": %d, \"what\": \"%s\", \"checks\": %llu, \"bad\": %llu, \"bad_by_lane_quarter\": [%llu, %llu, %llu, %llu], \"B_lo_wrong\": %llu, \"B_hi_wrong\": %llu, \"A_wrong\": %llu}\n",
         MODE, what, (unsigned long long)blocks * 512ull * iters, hb, q[0], q[1], q[2], q[3], hh[64], hh[65], hh[66]);
  hipFree(bad); hipFree(hist);
}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 2048, iters = argc > 2 ? atoi(argv[2]) : 2000;
  std::vector<float> h(65536);
  uint32_t st = 12345;
  for (auto& v : h) { st = st * 1664525u + 1013904223u; v = ((int)(st >> 8) % 20001 - 10000) * 1e-3f; }
  float* in; hipMalloc(&in, h.size() * 4); hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  run<0>(in, blocks, iters, "pk_mul A; pk_mul B; cvt A; cvt B");
  run<1>(in, blocks, iters, "pk_mul A; pk_mul B; s_nop 0; cvt A; cvt B");
  run<2>(in, blocks, iters, "pk_mul B; cvt B");
  run<3>(in, blocks, iters, "4 x v_mul; cvt A; cvt B");
  run<4>(in, blocks, iters, "pk_mul A; pk_mul B; s_nop 7; cvt A; cvt B");
  run<5>(in, blocks, iters, "pk_mul B; v_mov; v_add_f32 reads B.hi");
  run<6>(in, blocks, iters / 8, "8 un-waited global_load_dwordx4, then 48 x [pk_mul A; pk_mul B; cvt A; cvt B]");
  run<7>(in, blocks, iters / 8, "the same with the loads waited for first (control)");
  run<8>(in, blocks, iters / 8, "48 x [s_mul s7; pk_mul A <- s[6:7]; s_add s7; pk_mul B <- s[6:7]; s_ashr s7; cvt A; cvt B]");
  run<9>(in, blocks, iters / 8, "the same with s7 left alone (control)");
  run<10>(in, blocks, iters / 8, "waves 0-3: 48 x [pk_mul A; pk_mul B; cvt A; cvt B] while waves 4-7 (same SIMDs) run back-to-back MFMAs");
  run<11>(in, blocks, iters / 8, "the same with the partner waves idle (control)");
  run<12>(in, blocks, iters / 8, "partner MFMAs, probe with the scale in an SGPR pair");
  return 0;
}
