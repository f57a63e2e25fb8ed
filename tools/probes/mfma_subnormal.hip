// Does v_mfma_f32_16x16x32_f16 honour f16 SUBNORMAL inputs (|v| < 2^-14), and do the VALU conversions that
// produce the operands (float -> _Float16 cast, v_cvt_pk_f16_f32, v_fma_mix*_f16) keep them?
// Build and run:  hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_subnormal tools/probes/mfma_subnormal.hip && /tmp/mfma_subnormal
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(float a0, float b0, float* out) {
  const int lane = threadIdx.x;
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)0.f; b[j] = (_Float16)0.f; }
  if ((lane >> 4) == 0) { a[0] = (_Float16)a0; b[0] = (_Float16)b0; }   // one product on k = 0
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  unsigned hi, lo;
  asm volatile("v_cvt_pk_f16_f32 %0, %2, %3\n\t"
               "v_fma_mixlo_f16 %1, %2, 1.0, -%0 op_sel_hi:[0,0,1]\n\t"
               "v_fma_mixhi_f16 %1, %3, 1.0, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
               : "=&v"(hi), "=&v"(lo) : "v"(a0), "v"(b0));
  if (lane == 0) {
    out[0] = c[0];
    out[1] = (float)(_Float16)a0;
    out[2] = (float)__builtin_bit_cast(_Float16, (unsigned short)(hi & 0xffff));
    out[3] = (float)__builtin_bit_cast(_Float16, (unsigned short)(lo & 0xffff));
  }
}

int main() {
  float* d; hipMalloc(&d, 16);
  struct { float a, b; const char* what; } cases[] = {
      {ldexpf(1.f, -15), 1024.f, "2^-15 (subnormal) x 1024"},
      {ldexpf(1.5f, -20), 1024.f, "1.5 2^-20 (subnormal) x 1024"},
      {ldexpf(1.f, -24), 16384.f, "2^-24 (smallest subnormal) x 16384"},
      {ldexpf(1.f, -14), 1024.f, "2^-14 (smallest normal) x 1024"},
      {1.f + ldexpf(1.f, -11) + ldexpf(1.f, -26), 1.f, "1 + 2^-11 + 2^-26: lo piece is subnormal-sized"},
  };
  for (auto& cs : cases) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, cs.a, cs.b, d);
    float h[4]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%-48s exact product %.9g  mfma %.9g | cast(a) %.9g  cvt_pk(a) %.9g  mix lo(a) %.9g (a - hi = %.9g)\n", cs.what,
           (double)(float)(_Float16)cs.a * (double)(float)(_Float16)cs.b, h[0], h[1], h[2], h[3], cs.a - h[2]);
  }
  return 0;
}
