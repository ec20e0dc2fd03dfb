// store_probe.hip — how fast can a CU (and the chip) issue 16-byte-per-lane global stores?
// Each workgroup (512 threads, like the persistent GEMMs) writes `reps` x 128 KiB in the pattern of the GEMM
// epilogue (one instruction = 8 rows x 128 B at a row stride) or as contiguous 1-KiB blocks.
// Build: hipcc -O3 --offload-arch=gfx950 tools/store_probe.hip -o tools/store_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int NT, int PAT>
__global__ __launch_bounds__(512) void k(char* out, size_t ld, int reps, size_t wg_stride, int xcds) {
  if ((int)(blockIdx.x & 7) >= xcds) return;  // only the workgroups of the first `xcds` XCDs store
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32x4 v = {(uint32_t)lane, 1u, 2u, 3u};
  for (int r = 0; r < reps; ++r) {
    char* base = out + (size_t)blockIdx.x * wg_stride + (size_t)r * 256 * ld;  // a 256-row x 512-B "tile"
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      char* p;
      if (PAT == 0) {  // wave tile 128 rows x 128 B at (wave>>2)*128 rows, (wave&3)*128 B; 8 rows per instruction
        const int row = (wave >> 2) * 128 + i * 8 + (lane >> 3);
        p = base + (size_t)row * ld + (wave & 3) * 128 + (lane & 7) * 16;
      } else {  // contiguous: wave writes 16 KiB = rows of 512 B... one instruction = 1 KiB contiguous
        p = base + (size_t)(wave * 16 + i) * 1024 + lane * 16;
      }
      if (NT) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
      else asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    }
  }
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 14;
  char* buf;
  const size_t ld = 4608;  // bytes per row of a [M, 2304] bf16 matrix
  const size_t wg_stride = (size_t)reps * 256 * ld;
  const size_t bytes = 256 * wg_stride + (1 << 20);
  CK(hipMalloc(&buf, bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int xcds : {8, 4, 2, 1}) for (int grid : {256, 64}) for (int pat = 0; pat < 1; ++pat) for (int nt = 0; nt < 2; ++nt) {
    auto launch = [&]() {
      if (pat == 0 && nt == 0) hipLaunchKernelGGL((k<0, 0>), dim3(grid), dim3(512), 0, 0, buf, ld, reps, wg_stride, xcds);
      if (pat == 0 && nt == 1) hipLaunchKernelGGL((k<1, 0>), dim3(grid), dim3(512), 0, 0, buf, ld, reps, wg_stride, xcds);
      if (pat == 1 && nt == 0) hipLaunchKernelGGL((k<0, 1>), dim3(grid), dim3(512), 0, 0, buf, ld, reps, wg_stride, xcds);
      if (pat == 1 && nt == 1) hipLaunchKernelGGL((k<1, 1>), dim3(grid), dim3(512), 0, 0, buf, ld, reps, wg_stride, xcds);
    };
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 5; ++i) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const int active = grid * xcds / 8;
    const double us = ms * 1e3 / 5, by = (double)active * reps * 128 * 1024;
    printf("xcds %d grid %3d pattern %s %s: %8.1f us  %6.2f TB/s  %5.1f B/clk/CU at 2.1 GHz\n", xcds, grid, pat ? "contig " : "8rowx128", nt ? "nt   " : "plain",
           us, by / us / 1e6, by / active / (us * 1e-6 * 2.1e9));
  }
  return 0;
}
