// Issue cost of the vector instructions of the regression kernel's inner loop, alone and next to
// v_mfma_f32_16x16x32_f16, with one or two waves per SIMD (blockDim 256 or 512 on one CU).
// Build and run:  hipcc --offload-arch=gfx950 -O2 -o /tmp/issue_cost tools/probes/issue_cost.hip && /tmp/issue_cost
// Prints cycles per loop iteration (s_memtime ticks at 100 MHz are too coarse: wall clock over many iterations).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NV>
__global__ void probe(float* out, int iters, float seed) {
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(seed + j); b[j] = (_Float16)(seed * 0.5f + j); }
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = seed + threadIdx.x + j;
  unsigned h[8] = {0, 0, 0, 0, 0, 0, 0, 0}, hc[8];
  for (int j = 0; j < 8; ++j) hc[j] = 0x3c003c00u + j;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
      if (MODE & 1) {     // one MFMA (4 independent accumulators round robin)
        if (rep == 0) c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
        if (rep == 1) c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
        if (rep == 2) c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
        if (rep == 3) c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
      }
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int r = (rep * NV + k) & 7;
        if ((MODE >> 1) == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[r]) : "v"(seed));
        if ((MODE >> 1) == 1) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h[r]) : "v"(v[r]), "v"(v[(r + 1) & 7]));
        if ((MODE >> 1) == 2) asm volatile("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "+v"(h[r]) : "v"(v[r]), "v"(hc[r]));
        if ((MODE >> 1) == 3) asm volatile("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(h[r]) : "v"(v[r]), "v"(hc[r]));
        if ((MODE >> 1) == 5) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(v[r]) : "v"(v[(r + 3) & 7]), "v"(seed));
        if ((MODE >> 1) == 6) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(v[r]) : "v"(hc[r]));
        if ((MODE >> 1) == 7) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(v[r]) : "v"(hc[r]));
        if ((MODE >> 1) == 8) asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(h[r]) : "v"(hc[r]), "v"(hc[(r + 1) & 7]));
        if ((MODE >> 1) == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(*(double*)&v[(r & 3) * 2]) : "v"(*(double*)&v[((r + 1) & 3) * 2]));
      }
    }
  }
  float s = c0[0] + c1[1] + c2[2] + c3[3];
  for (int j = 0; j < 8; ++j) s += v[j];
  for (int j = 0; j < 8; ++j) s += (float)h[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int NV>
void run(const char* what, float* d) {
  for (int threads : {256, 512}) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<MODE, NV>), dim3(256), dim3(threads), 0, 0, d, 100, 1.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE, NV>), dim3(256), dim3(threads), 0, 0, d, iters, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: threads / 256 waves each doing iters * 4 groups
    const double ns_per_group = ms * 1e6 / (iters * 4.0);
    printf("%-44s %d wave(s)/SIMD: %.2f ns per group (= per MFMA gap) -> %.1f cycles at 2.4 GHz\n", what, threads / 256,
           ns_per_group, ns_per_group * 2.4);
  }
}

int main() {
  float* d; hipMalloc(&d, 256 * 512 * 4);
  run<1, 0>("MFMA only", d);
  run<0, 4>("4 x v_fma_f32", d);
  run<2, 4>("4 x v_cvt_pk_f16_f32", d);
  run<4, 4>("4 x v_fma_mixlo_f16", d);
  run<6, 4>("4 x v_fma_mixhi_f16", d);
  run<8, 4>("4 x v_pk_fma_f32", d);
  run<10, 4>("4 x v_sub_f32", d);
  run<12, 4>("4 x v_cvt_f32_f16", d);
  run<14, 4>("4 x v_cvt_f32_f16 sdwa (high half)", d);
  run<16, 4>("4 x v_pk_add_f16", d);
  run<0, 8>("8 x v_fma_f32", d);
  run<4, 8>("8 x v_fma_mixlo_f16", d);
  run<1, 1>("MFMA + 1 v_fma_f32", d);
  run<1, 2>("MFMA + 2 v_fma_f32", d);
  run<1, 3>("MFMA + 3 v_fma_f32", d);
  run<1, 4>("MFMA + 4 v_fma_f32", d);
  run<3, 2>("MFMA + 2 v_cvt_pk_f16_f32", d);
  run<5, 2>("MFMA + 2 v_fma_mixlo_f16", d);
  run<5, 3>("MFMA + 3 v_fma_mixlo_f16", d);
  run<11, 2>("MFMA + 2 v_sub_f32", d);
  run<17, 2>("MFMA + 2 v_pk_add_f16", d);
  run<9, 2>("MFMA + 2 v_pk_fma_f32", d);
  return 0;
}
