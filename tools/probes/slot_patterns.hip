// Issue cost of three-slot patterns (one v_mfma_f32_16x16x32_f16 + a few vector instructions per slot), two waves
// per SIMD: candidates for the hi / lo split of a residual pair next to the matrix instructions of the regression
// kernel.  Build and run:  hipcc --offload-arch=gfx950 -O2 -o /tmp/slot_patterns tools/probes/slot_patterns.hip && /tmp/slot_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// op codes: F fma, C cvt_pk_f16_f32, L mixlo, H mixhi, S sub, W cvt_f32_f16 (low half), X cvt_f32_f16 sdwa (high half), A and
template <char OP>
__device__ __forceinline__ void op(float (&v)[8], unsigned (&h)[8], int r, float seed) {
  const int r1 = (r + 1) & 7, r3 = (r + 3) & 7;
  if (OP == 'F') asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(v[r]) : "v"(v[r3]));
  if (OP == 'C') asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h[r]) : "v"(v[r]), "v"(v[r1]));
  if (OP == 'L') asm volatile("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "+v"(h[r]) : "v"(v[r]), "v"(h[r3]));
  if (OP == 'H') asm volatile("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(h[r]) : "v"(v[r]), "v"(h[r3]));
  if (OP == 'S') asm volatile("v_sub_f32 %0, %1, %2" : "=v"(v[r]) : "v"(v[r3]), "v"(v[r1]));
  if (OP == 'W') asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(v[r]) : "v"(h[r3]));
  if (OP == 'X') asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(v[r]) : "v"(h[r3]));
  if (OP == 'A') asm volatile("v_and_b32 %0, %1, %2" : "=v"(h[r]) : "v"(v[r3]), "v"(h[r1]));
}
template <char... OPS> struct Slot {
  static __device__ __forceinline__ void run(float (&v)[8], unsigned (&h)[8], int& r, float seed) {
    ((op<OPS>(v, h, r, seed), r = (r + 1) & 7), ...);
  }
};
template <class S1, class S2, class S3, bool MFMA>
__global__ void probe(float* out, int iters, float seed) {
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(seed + j); b[j] = (_Float16)(seed * 0.5f + j); }
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0;
  float v[8]; unsigned h[8];
  for (int j = 0; j < 8; ++j) { v[j] = seed + threadIdx.x + j; h[j] = 0x3c003c00u + j; }
  for (int it = 0; it < iters; ++it) {
    int r = 0;      // (register indices are compile-time constants: every pattern has at most 8 operations)
    if (MFMA) c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
    S1::run(v, h, r, seed); __builtin_amdgcn_sched_barrier(0);
    if (MFMA) c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
    S2::run(v, h, r, seed); __builtin_amdgcn_sched_barrier(0);
    if (MFMA) c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
    S3::run(v, h, r, seed); __builtin_amdgcn_sched_barrier(0);
  }
  float s = c0[0] + c1[1] + c2[2];
  for (int j = 0; j < 8; ++j) s += v[j] + (float)h[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class S1, class S2, class S3>
void run(const char* what, float* d) {
  for (int mf = 1; mf >= 0; --mf) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    if (mf) hipLaunchKernelGGL((probe<S1, S2, S3, true>), dim3(256), dim3(512), 0, 0, d, 100, 1.f);
    else hipLaunchKernelGGL((probe<S1, S2, S3, false>), dim3(256), dim3(512), 0, 0, d, 100, 1.f);
    hipEventRecord(e0);
    if (mf) hipLaunchKernelGGL((probe<S1, S2, S3, true>), dim3(256), dim3(512), 0, 0, d, iters, 1.f);
    else hipLaunchKernelGGL((probe<S1, S2, S3, false>), dim3(256), dim3(512), 0, 0, d, iters, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s %s: %.2f ns per three slots (both waves of a SIMD)\n", what, mf ? "with MFMAs" : "vector only", ms * 1e6 / iters);
  }
}
int main() {
  float* d; hipMalloc(&d, 256 * 512 * 4);
  run<Slot<>, Slot<>, Slot<>>("nothing", d);
  run<Slot<'C', 'F'>, Slot<'L', 'F'>, Slot<'H'>>("now: [cvt fma] [mixlo fma] [mixhi]", d);
  run<Slot<'C'>, Slot<'L'>, Slot<'H'>>("split only: [cvt] [mixlo] [mixhi]", d);
  run<Slot<'F', 'F'>, Slot<'C'>, Slot<>>("no low piece: [fma fma] [cvt] []", d);
  run<Slot<'C', 'W', 'F'>, Slot<'X', 'S', 'F'>, Slot<'S', 'C'>>("convert back: [cvt cvtlo fma] [cvthi sub fma] [sub cvt]", d);
  run<Slot<'A', 'A', 'F'>, Slot<'S', 'S', 'F'>, Slot<'C', 'C'>>("mask: [and and fma] [sub sub fma] [cvt cvt]", d);
  run<Slot<'A', 'S', 'F'>, Slot<'A', 'S', 'F'>, Slot<'C', 'C'>>("mask: [and sub fma] [and sub fma] [cvt cvt]", d);
  run<Slot<'A', 'A', 'S'>, Slot<'S', 'C', 'F'>, Slot<'C', 'F'>>("mask: [and and sub] [sub cvt fma] [cvt fma]", d);
  run<Slot<'C', 'F'>, Slot<'F', 'L'>, Slot<'H'>>("[cvt fma] [fma mixlo] [mixhi]", d);
  run<Slot<'F', 'C'>, Slot<'L', 'F'>, Slot<'H'>>("[fma cvt] [mixlo fma] [mixhi]", d);
  run<Slot<'F', 'C'>, Slot<'F', 'L'>, Slot<'H'>>("[fma cvt] [fma mixlo] [mixhi]", d);
  run<Slot<'C'>, Slot<'L', 'F'>, Slot<'F', 'H'>>("[cvt] [mixlo fma] [fma mixhi]", d);
  run<Slot<'C'>, Slot<'F', 'L'>, Slot<'F', 'H'>>("[cvt] [fma mixlo] [fma mixhi]", d);
  run<Slot<'C', 'L'>, Slot<'F', 'F'>, Slot<'H'>>("[cvt mixlo] [fma fma] [mixhi]", d);
  run<Slot<'C', 'L'>, Slot<'H', 'F'>, Slot<'F'>>("[cvt mixlo] [mixhi fma] [fma]", d);
  run<Slot<'C', 'L'>, Slot<'H'>, Slot<'F', 'F'>>("[cvt mixlo] [mixhi] [fma fma]", d);
  run<Slot<'C', 'F', 'F'>, Slot<'L'>, Slot<'H'>>("[cvt fma fma] [mixlo] [mixhi]", d);
  run<Slot<'C', 'F'>, Slot<'L'>, Slot<'H', 'F'>>("[cvt fma] [mixlo] [mixhi fma]", d);
  return 0;
}
