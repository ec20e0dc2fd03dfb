// isa_probe.hip — small gfx950 questions answered on the device before a kernel depends on them:
//   1. ds_read_b64_tr_b16: which LDS elements reach which lane (the attention PV operand read);
//   2. LDS cycles of that read for candidate V images (bank conflicts of the transpose read);
//   3. MODE.FP16_OVFL: does v_cvt_pk_f16_f32 saturate to ±65504 instead of ±inf when it is set.
// Build: hipcc -O3 --offload-arch=gfx950 tools/isa_probe.hip -o /tmp/isa_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>

#include <vector>

typedef short v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s* lds_v4s;

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e = (x);                                                            \
    if (e != hipSuccess) {                                                         \
      printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__);         \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

__global__ void tr_semantics(short* out, int pattern, int row_stride) {
  __shared__ short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
  int elem;
  if (pattern == 0) elem = lane * 4;                                    // contiguous
  else elem = (g * 4 + (i >> 2)) * row_stride + (i & 3) * 4;            // [4 rows][16 cols] block per group
  v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(lds + elem));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = r[j];
}

// layout 0: [key][64 dims] 128-B rows, plain;  1: same, 32-B segment XOR (key>>1)&3;
// layout 2: [dt][key][16 dims] (32-B rows, contiguous keys);  3: plain ds_read_b64 of a V^T image
// (round-1 layout: [64 d][VS bytes], key-pair XOR (d>>3)<<2) for comparison
__global__ void tr_timing(uint64_t* cyc, int* sink, int layout, int LP) {
  extern __shared__ short smem[];
  for (int i = threadIdx.x; i < LP * 64 + 64 * 16; i += blockDim.x) smem[i] = (short)(i * 7);
  __syncthreads();
  const int lane = threadIdx.x & 63, fr = lane & 15, g = lane >> 4;
  int acc = 0;
  const int NU = LP / 32;
  const int VS = ((LP * 2 - 16 + 255) / 256) * 256 + 16;
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int rep = 0; rep < 8; ++rep)
#pragma unroll 1
    for (int dt = 0; dt < 4; ++dt) {
#pragma unroll 7
      for (int u = 0; u < NU; ++u) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int key = u * 32 + half * 16 + g * 4 + (fr >> 2);
          int byte;
          if (layout == 0) byte = key * 128 + dt * 32 + (fr & 3) * 8;
          else if (layout == 1) byte = key * 128 + ((dt ^ ((key >> 1) & 3)) * 32) + (fr & 3) * 8;
          else if (layout == 2) byte = dt * (LP * 32) + key * 32 + (fr & 3) * 8;
          else {
            const int vx = (dt * 2 + (fr >> 3)) << 2;
            byte = (dt * 16 + fr) * VS + ((((2 * u + half) * 8 + g * 2) ^ vx) << 2);
          }
          v4s r;
          if (layout < 3) r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)((char*)smem + byte));
          else r = *(const v4s*)((const char*)smem + byte);
          acc += r[0] + r[1] + r[2] + r[3];
        }
      }
    }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

__global__ void f16_ovfl(const float* in, uint32_t* out, int n, int set_mode) {
  if (set_mode) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
  typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
  typedef float f2_t __attribute__((ext_vector_type(2)));
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const f2_t v = {in[i], -in[i]};
    out[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, h2_t));
  }
}

static float h2f(uint16_t h) {
  const uint32_t s = (h >> 15) & 1, e = (h >> 10) & 31, m = h & 1023;
  float v;
  if (e == 0) v = ldexpf((float)m, -24);
  else if (e == 31) v = m ? NAN : INFINITY;
  else v = ldexpf((float)(m | 1024), (int)e - 25);
  return s ? -v : v;
}

int main() {
  short* d;
  CK(hipMalloc(&d, 256 * 2));
  std::vector<short> h(256);
  for (int pat = 0; pat < 2; ++pat) {
    hipLaunchKernelGGL(tr_semantics, dim3(1), dim3(64), 0, 0, d, pat, 64);
    CK(hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost));
    printf("ds_read_b64_tr_b16 pattern %d (LDS element index = value; %s):\n", pat,
           pat ? "lane (i,g) addr = row (4g + i/4) * 64 + (i%4)*4" : "lane addr = 4*lane");
    for (int l = 0; l < 64; ++l) {
      printf("  lane %2d:", l);
      for (int j = 0; j < 4; ++j) printf(" %5d", h[l * 4 + j]);
      if (pat) {
        printf("   = (row,col)");
        for (int j = 0; j < 4; ++j) printf(" (%d,%d)", h[l * 4 + j] / 64, h[l * 4 + j] % 64);
      }
      printf("\n");
    }
  }
  // timing
  const int LP = 224, wgs = 512, threads = 256;
  uint64_t* dc;
  int* ds;
  CK(hipMalloc(&dc, wgs * 4 * 8));
  CK(hipMalloc(&ds, wgs * threads * 4));
  std::vector<uint64_t> hc(wgs * 4);
  for (int layout = 0; layout < 4; ++layout) {
    for (int rep = 0; rep < 2; ++rep)
      hipLaunchKernelGGL(tr_timing, dim3(wgs), dim3(threads), 64 * 1024, 0, dc, ds, layout, LP);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hc.data(), dc, hc.size() * 8, hipMemcpyDeviceToHost));
    double s = 0;
    for (auto v : hc) s += (double)v;
    const double per = s / hc.size() / (8.0 * 4 * (LP / 32) * 2);
    printf("layout %d: %.1f cycles per wave-level read (2 WG x 4 waves per CU share the LDS)\n", layout, per);
  }
  // FP16_OVFL
  const float vals[8] = {1.0f, 65504.0f, 65519.0f, 65520.0f, 70000.0f, 1e6f, 3.4e38f, INFINITY};
  float* di;
  uint32_t* dout;
  CK(hipMalloc(&di, sizeof(vals)));
  CK(hipMalloc(&dout, 8 * 4));
  CK(hipMemcpy(di, vals, sizeof(vals), hipMemcpyHostToDevice));
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(f16_ovfl, dim3(1), dim3(64), 0, 0, di, dout, 8, mode);
    uint32_t ho[8];
    CK(hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost));
    printf("v_cvt_pk_f16_f32 with MODE.FP16_OVFL=%d:", mode);
    for (int i = 0; i < 8; ++i) printf("  %g -> (%g, %g)", vals[i], h2f(ho[i] & 0xffff), h2f(ho[i] >> 16));
    printf("\n");
  }
  return 0;
}
