// valu_probe.hip — issue cost of the VALU instructions the epilogues are made of (cycles per wave instruction,
// one wave per SIMD and two), from s_memtime around 8 independent chains x 256 repetitions.
// Build: hipcc -O3 --offload-arch=gfx950 tools/valu_probe.hip -o tools/valu_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
template <int OP>
__global__ void k(float* out, unsigned long long* cyc) {
  float v[8];
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p[8];
  for (int i = 0; i < 8; ++i) { v[i] = 0.5f + 0.001f * (threadIdx.x + i); p[i] = (f2){v[i], v[i] + 1.f}; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < 256; ++r) {
#define S_EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
#define S_RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
#define S_EXPH(i) asm volatile("v_exp_f16 %0, %0" : "+v"(v[i]));
#define S_FMA(i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[i]));
#define S_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[i]));
#define S_CVT(i) asm volatile("v_cvt_pk_f16_f32 %0, %0, %0" : "+v"(v[i]));
#define S_MUL(i) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(v[i]));
#define S_SQRT(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(v[i]));
    if (OP == 0) { REP8(S_EXP) }
    if (OP == 1) { REP8(S_RCP) }
    if (OP == 2) { REP8(S_EXPH) }
    if (OP == 3) { REP8(S_FMA) }
    if (OP == 4) { REP8(S_PKFMA) }
    if (OP == 5) { REP8(S_CVT) }
    if (OP == 6) { REP8(S_MUL) }
    if (OP == 7) { REP8(S_SQRT) }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += v[i] + p[i][0] + p[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
  float* out; unsigned long long* cyc;
  CK(hipMalloc(&out, 1 << 22)); CK(hipMalloc(&cyc, 8));
  const char* names[] = {"v_exp_f32", "v_rcp_f32", "v_exp_f16", "v_fma_f32", "v_pk_fma_f32", "v_cvt_pk_f16_f32", "v_mul_f32", "v_sqrt_f32"};
  for (int wps = 1; wps <= 2; ++wps)
    for (int op = 0; op < 8; ++op) {
      dim3 g(256), b(256 * wps);
      switch (op) {
        case 0: hipLaunchKernelGGL(k<0>, g, b, 0, 0, out, cyc); break;
        case 1: hipLaunchKernelGGL(k<1>, g, b, 0, 0, out, cyc); break;
        case 2: hipLaunchKernelGGL(k<2>, g, b, 0, 0, out, cyc); break;
        case 3: hipLaunchKernelGGL(k<3>, g, b, 0, 0, out, cyc); break;
        case 4: hipLaunchKernelGGL(k<4>, g, b, 0, 0, out, cyc); break;
        case 5: hipLaunchKernelGGL(k<5>, g, b, 0, 0, out, cyc); break;
        case 6: hipLaunchKernelGGL(k<6>, g, b, 0, 0, out, cyc); break;
        case 7: hipLaunchKernelGGL(k<7>, g, b, 0, 0, out, cyc); break;
      }
      CK(hipDeviceSynchronize());
      unsigned long long c;
      CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
      printf("%d wave(s)/SIMD  %-18s %6.2f cycles per wave instruction (s_memtime ticks, this wave)  -> %5.2f per SIMD slot\n", wps, names[op],
             (double)c / (256 * 8), (double)c / (256 * 8) / wps);
    }
  return 0;
}
