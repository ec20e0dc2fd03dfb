// gemm_bench.hip — standalone A/B harness for the GEMM kernels of mcm_amd/csrc/gemm.hip.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -DMCM_HARNESS [-DMCM_GEMM_TRACE] [-DMCM_GEMM_ABLATE: in-loop ablation bits 1 / 2 / 32] -I mcm_amd/csrc tools/gemm_bench.hip \
//        mcm_amd/csrc/gemm.hip -o /tmp/gemm_bench
// Run:   gemm_bench M N K epi [iters]   → per-variant time / TFLOP/s, max |diff| vs variant 0
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "common.hpp"

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e = (x);                                                                \
    if (e != hipSuccess) {                                                             \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__);     \
      exit(1);                                                                         \
    }                                                                                  \
  } while (0)

static uint16_t f2bf_h(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float bf2f_h(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static float frand(uint64_t& s) {
  s = s * 6364136223846793005ull + 1442695040888963407ull;
  return ((s >> 40) & 0xffffff) / (float)0x800000 - 1.0f;  // uniform [-1,1)
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 100864, N = argc > 2 ? atoi(argv[2]) : 2304,
            K = argc > 3 ? atoi(argv[3]) : 768, epi = argc > 4 ? atoi(argv[4]) : 0,
            iters = argc > 5 ? atoi(argv[5]) : 20;
  if (argc > 6) gemm_set_group_n(atoi(argv[6]));
  if (argc > 7) gemm_set_dbg(atoi(argv[7]));
  setvbuf(stdout, NULL, _IONBF, 0);
  printf("M=%d N=%d K=%d epi=%d\n", M, N, K, epi);
  uint64_t seed = 1;
  std::vector<uint16_t> hx((size_t)M * K), hw((size_t)N * K);
  for (auto& v : hx) v = f2bf_h(frand(seed));
  for (auto& v : hw) v = f2bf_h(frand(seed) * 0.05f);
  std::vector<float> hb(N), hr((size_t)M * N);
  for (auto& v : hb) v = frand(seed) * 0.1f;
  for (auto& v : hr) v = frand(seed);
  uint16_t *dx, *dw;
  float *db, *dr;
  void* dout;
  CK(hipMalloc(&dx, hx.size() * 2));
  CK(hipMalloc(&dw, hw.size() * 2));
  CK(hipMalloc(&db, N * 4));
  CK(hipMalloc(&dr, (size_t)M * N * 4));
  CK(hipMalloc(&dout, (size_t)M * N * 4));
  CK(hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice));
  // LayerNorm-fold side buffers with neutral contents: gamma = 1, (rstd, mean rstd) = (1, 0), c = 0
  void* dfz; float *dfg, *dfc; float2 *dfp, *dfr;
  CK(hipMalloc(&dfz, (size_t)M * N * 2));
  CK(hipMalloc(&dfg, (size_t)N * 4));
  CK(hipMalloc(&dfc, (size_t)N * 4));
  CK(hipMalloc(&dfp, (size_t)(N / 64 + 1) * M * 8));
  CK(hipMalloc(&dfr, (size_t)M * 8));
  {
    std::vector<float> ones(N, 1.0f);
    std::vector<float2> r1(M, make_float2(1.0f, 0.0f));
    CK(hipMemcpy(dfg, ones.data(), (size_t)N * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dfc, 0, (size_t)N * 4));
    CK(hipMemcpy(dfr, r1.data(), (size_t)M * 8, hipMemcpyHostToDevice));
  }
  GemmArgs a{};
  a.x = dx; a.w = dw; a.bias = db; a.out = dout; a.resid = dr;
  a.M = M; a.N = N; a.K = K; a.ldx = K; a.ldo = N;
  const size_t out_elems = (size_t)M * N;
  std::vector<float> ref, got(out_elems);
  std::vector<uint16_t> tmp(out_elems);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int nvar = 10;  // 9 = ping-pong with the LayerNorm-fold epilogue of this EPI (neutral fold data: same results)
  const unsigned vmask = argc > 9 ? (unsigned)strtoul(argv[9], 0, 0) : 0x3ffu;  // variants to time
  double best[nvar] = {0};
  for (int round = 0; round < 3; ++round)
    for (int v = 0; v < nvar; ++v) {
      if (!((vmask >> v) & 1) && !(v == 0 && round == 0)) continue;
      gemm_set_variant(v == 9 ? 5 : v);
      a.fold_z = nullptr; a.fold_g = nullptr; a.fold_part = nullptr; a.fold_rs = nullptr; a.fold_c = nullptr;
      if (v == 9) {
        if (epi == EPI_RESID) { a.fold_z = dfz; a.fold_g = dfg; a.fold_part = dfp; }
        else if (epi != EPI_PATCH) { a.fold_rs = dfr; a.fold_c = dfc; }
        else continue;
      }
      if (round == 0) {  // correctness pass
        CK(hipMemcpy(dr, hr.data(), hr.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemset(dout, 0, out_elems * 4));
        CK(launch_gemm(MCM_PREC_BF16, epi, a, 0));
        CK(hipDeviceSynchronize());
        if (epi == EPI_RESID) {
          CK(hipMemcpy(got.data(), dr, out_elems * 4, hipMemcpyDeviceToHost));
        } else {
          CK(hipMemcpy(tmp.data(), dout, out_elems * 2, hipMemcpyDeviceToHost));
          for (size_t i = 0; i < out_elems; ++i) got[i] = bf2f_h(tmp[i]);
        }
        if (v == 0) {
          ref = got;
          // spot-check variant 0 against a host dot product
          double worst = 0;
          for (int t = 0; t < 64; ++t) {
            const int m = (int)(((uint64_t)t * 7919 * 131) % M), n = (int)(((uint64_t)t * 104729) % N);
            double acc = hb[n];
            for (int k = 0; k < K; ++k) acc += (double)bf2f_h(hx[(size_t)m * K + k]) * bf2f_h(hw[(size_t)n * K + k]);
            if (epi == EPI_GELU) acc = acc / (1.0 + exp(-1.702 * acc));
            if (epi == EPI_RESID) acc += hr[(size_t)m * N + n];
            const double d = fabs(acc - ref[(size_t)m * N + n]);
            if (d > worst) worst = d;
          }
          printf("  variant 0 vs host fp64 spot check: max|d| = %.3e\n", worst);
        } else {
          double worst = 0;
          size_t nbad = 0;
          for (size_t i = 0; i < out_elems; ++i) {
            const double d = fabs((double)got[i] - ref[i]);
            if (d > worst) worst = d;
            if (d > 0) ++nbad;
          }
          printf("  variant %d vs variant 0: max|d| = %.3e, %zu differing elements\n", v, worst, nbad);
        }
      }
      CK(launch_gemm(MCM_PREC_BF16, epi, a, 0));  // warm
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < iters; ++i) CK(launch_gemm(MCM_PREC_BF16, epi, a, 0));
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = 1e3 * ms / iters, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
      if (tf > best[v]) best[v] = tf;
      printf("  round %d variant %d: %9.1f us  %7.1f TFLOP/s\n", round, v, us, tf);
    }
#ifdef MCM_GEMM_TRACE
  {  // per-step cycle stamps of the persistent 256x256 kernel (variant 3), blocks 0 and 77
    const size_t words = (size_t)256 * 8 * 64 * 8;
    uint64_t* dt;
    CK(hipMalloc(&dt, words * 8));
    CK(hipMemset(dt, 0, words * 8));
    GemmArgs b = a;
    b.pos = (const float*)dt;
    gemm_set_variant(argc > 8 ? atoi(argv[8]) : 3);
    gemm_set_dbg((argc > 7 ? atoi(argv[7]) : 0) | 128);
    CK(launch_gemm(MCM_PREC_BF16, epi, b, 0));
    CK(hipDeviceSynchronize());
    std::vector<uint64_t> ht(words);
    CK(hipMemcpy(ht.data(), dt, words * 8, hipMemcpyDeviceToHost));
    const int nk = K / 64;
    const int tv = argc > 8 ? atoi(argv[8]) : 3;
    if (tv == 5) {  // ping-pong kernel: per-phase sums (mid-tile steps | steps carrying an epilogue)
      const uint32_t* u = (const uint32_t*)ht.data();
      for (int blk : {0, 77, 200}) for (int w = 0; w < 8; ++w) {
        const uint32_t* o = u + ((size_t)blk * 8 + w) * 10;
        printf("PP blk %3d wave %d:", blk, w);
        for (int i = 0; i < 2; ++i) {
          const double n = o[i * 5 + 4] ? o[i * 5 + 4] : 1;
          printf("  %s n=%u mem %.0f bar1 %.0f comp %.0f vm+bar2 %.0f | %.0f", i ? "epi-steps" : "mid-steps", o[i * 5 + 4],
                 o[i * 5 + 0] / n, o[i * 5 + 1] / n, o[i * 5 + 2] / n, o[i * 5 + 3] / n,
                 (o[i * 5 + 0] + o[i * 5 + 1] + o[i * 5 + 2] + o[i * 5 + 3]) / n);
        }
        printf("\n");
      }
    } else
    for (int blk : {0, 77}) for (int w : {0, 4}) {
      double d[8] = {0};
      int cnt = 0;
      for (int st = nk; st < 4 * nk && st < 63; ++st) {
        const uint64_t* r = &ht[(((size_t)blk * 8 + w) * 64 + st) * 8];
        const uint64_t* rn = r + 8;
        for (int k = 0; k < 6; ++k) d[k] += (double)(r[k + 1] - r[k]);
        d[6] += (double)(rn[0] - r[6]);
        d[7] += (double)(rn[0] - r[0]);
        ++cnt;
      }
      printf("TRACE blk %3d wave %d (avg over %d steps, incl. tile ends): d0 %.0f  d1 %.0f  d2 %.0f  d3 %.0f  d4 %.0f  d5 %.0f  tail %.0f | step %.0f cyc   [v3: vmwait barrier issueA half0 issueB half1 tail; v5: issue(+epi) reads waits barrier compute vmwait+barrier]\n",
             blk, w, cnt, d[0] / cnt, d[1] / cnt, d[2] / cnt, d[3] / cnt, d[4] / cnt, d[5] / cnt, d[6] / cnt, d[7] / cnt);
      // a few steps in full: mid-tile, the tile's last step (tail = epilogue) and the two after it
      for (int st : {nk + 3, 2 * nk - 2, 2 * nk - 1, 2 * nk, 2 * nk + 1}) {
        if (st >= 62) continue;
        const uint64_t* r = &ht[(((size_t)blk * 8 + w) * 64 + st) * 8];
        printf("      step %2d:", st);
        for (int k = 0; k < 6; ++k) printf(" %5llu", (unsigned long long)(r[k + 1] - r[k]));
        printf(" | tail %llu\n", (unsigned long long)(r[8] - r[6]));
      }
    }
  }
#endif
  printf("BEST M=%d N=%d K=%d epi=%d:", M, N, K, epi);
  for (int v = 0; v < nvar; ++v) printf(" v%d=%.1f", v, best[v]);
  printf(" TFLOP/s\n");
  return 0;
}
