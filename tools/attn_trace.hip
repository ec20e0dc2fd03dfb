// attn_trace.hip — where a workgroup of the fp16 attention kernel spends its life (EXPERIMENTS.md R5.7): s_memtime stamps at
// entry / loads issued / own loads landed / barrier passed / q-block 0 done / q-block 1 done / exit, per wave, for every workgroup
// of one B/16 batch-512 launch (6 144 workgroups), averaged.  The product sources with -DMCM_ATTN_TRACE; nothing of this is
// in the shipped library.
// Build + run (GPU box): hipcc -O3 -std=c++17 --offload-arch=gfx950 -DMCM_ATTN_TRACE -I mcm_amd/csrc tools/attn_trace.hip -o /tmp/attn_trace && /tmp/attn_trace
#include "../mcm_amd/csrc/attention.hip"

#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
  const int nseq = 512, L = 197, heads = 12, D = heads * 64;
  const size_t rows = (size_t)nseq * L;
  std::vector<uint16_t> h(rows * 3 * D);
  uint32_t st = 12345u;
  for (auto& v : h) {  // fp16 values in (-1, 1)
    st = st * 1664525u + 1013904223u;
    const float f = ((st >> 8) & 0xffff) / 32768.0f - 1.0f;
    _Float16 hv = (_Float16)f;
    v = __builtin_bit_cast(uint16_t, hv);
  }
  uint16_t *qkv, *out;
  unsigned long long* tr;
  const size_t nwg = (size_t)nseq * heads, trn = nwg * 8 * 8;
  CK(hipMalloc(&qkv, h.size() * 2));
  CK(hipMalloc(&out, rows * D * 2));
  CK(hipMalloc(&tr, trn * 8));
  CK(hipMemcpy(qkv, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) CK(launch_attention(MCM_PREC_F16, qkv, out, nseq, L, heads, false, 0, nullptr));  // warm
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int rep = 0; rep < 10; ++rep) CK(launch_attention(MCM_PREC_F16, qkv, out, nseq, L, heads, false, 0, nullptr));
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("untraced-buffer launches: %.1f us per launch (stamps compiled in, buffer null)\n", 1e3 * ms / 10);
  CK(hipMemset(tr, 0, trn * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_attn_trace), &tr, sizeof(tr)));
  CK(hipEventRecord(e0));
  CK(launch_attention(MCM_PREC_F16, qkv, out, nseq, L, heads, false, 0, nullptr));
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> t(trn);
  CK(hipMemcpy(t.data(), tr, trn * 8, hipMemcpyDeviceToHost));
  // HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13]; XCC id in bits 32+.
  // Group the workgroups by the CU they ran on (every XCD has its own s_memtime base: only differences inside a CU are used).
  struct Wg { unsigned long long s0, s3, e; double qb0, qb1, issue, land, bar; };
  std::map<unsigned long long, std::vector<Wg>> cus;
  for (size_t w = 0; w < nwg; ++w) {
    const unsigned long long* a = &t[(w * 8 + 0) * 8];
    unsigned long long s0 = ~0ull, s6 = 0;
    for (int wv = 0; wv < 8; ++wv) {
      s0 = std::min(s0, t[(w * 8 + wv) * 8 + 0]);
      s6 = std::max(s6, t[(w * 8 + wv) * 8 + 6]);
    }
    const unsigned long long id = a[7];
    const unsigned long long key = ((id >> 32) << 16) | (((id >> 13) & 7) << 8) | (((id >> 12) & 1) << 4) | ((id >> 8) & 15);
    cus[key].push_back(Wg{s0, a[3], s6, (double)(a[4] - a[3]), (double)(a[5] - a[4]), (double)(a[1] - a[0]), (double)(a[2] - a[1]),
                          (double)(a[3] - a[2])});
  }
  double life = 0, qb0 = 0, qb1 = 0, issue = 0, land = 0, bar = 0, busy1 = 0, span_sum = 0, conc = 0;
  size_t ncu = cus.size();
  for (auto& kv : cus) {
    auto& v = kv.second;
    unsigned long long first = ~0ull, last = 0;
    for (auto& g : v) {
      first = std::min(first, g.s0);
      last = std::max(last, g.e);
      life += (double)(g.e - g.s0);
      qb0 += g.qb0; qb1 += g.qb1; issue += g.issue; land += g.land; bar += g.bar;
    }
    const double span = (double)(last - first);
    span_sum += span;
    double resident = 0;
    for (auto& g : v) resident += (double)(g.e - g.s0);
    conc += resident / span;          // mean number of workgroups resident on this CU over its busy span
    (void)busy1;
  }
  const double n = (double)nwg;
  const double tick_per_us = span_sum / ncu / (1e3 * ms);   // a CU's span is (nearly) the launch: ticks per microsecond
  printf("traced launch %.1f us; CUs seen %zu (%.1f workgroups each); s_memtime: %.0f ticks per us (shader clock %.2f GHz)\n", 1e3 * ms, ncu,
         n / ncu, tick_per_us, tick_per_us / 1e3);
  auto us = [&](double ticks) { return ticks / n / tick_per_us; };
  printf("per workgroup, mean, microseconds (ticks):\n");
  printf("  life: first wave in -> last wave out           %6.2f (%.0f)\n", us(life), life / n);
  printf("  wave 0: issuing its loads (Q + K + V DMA)      %6.2f (%.0f)\n", us(issue), issue / n);
  printf("  wave 0: its loads landed (s_waitcnt vmcnt 0)   %6.2f (%.0f)\n", us(land), land / n);
  printf("  wave 0: barrier (every wave's loads landed)    %6.2f (%.0f)\n", us(bar), bar / n);
  printf("  wave 0: q-block 0 (QK^T, softmax, PV, stores)  %6.2f (%.0f)\n", us(qb0), qb0 / n);
  printf("  wave 0: q-block 1                              %6.2f (%.0f)\n", us(qb1), qb1 / n);
  printf("workgroups resident per CU, averaged over the CU's busy span: %.2f (LDS allows 3)\n", conc / ncu);
  return 0;
}
