// score.hip — the fused scoring tail of the hot loop: cosine similarity against the prompt
// bank, temperature softmax, and the score reduction, one launch, [B] floats out.
//
// Replaces reference utils/detection_util.py:232-248:
//   output = image_features @ text_features.T                      (:232)   skinny GEMM
//   smax   = softmax(output / T)                                   (:236)
//   MCM / max-logit: -max(smax) / -max(output)                     (:234,:248)
//   energy : -T * logsumexp(output / T)                            (:239)
//   entropy: scipy.stats.entropy(smax, axis=1)  (natural log)      (:243)
//   var    : -np.var(smax, axis=1)              (ddof = 0)         (:246)
// and removes the reference's [B,K] softmax D2H copy (:236): the similarities of one image
// live only in LDS.  HBM-bound on the image features ([B,P] fp32, read once); the text bank
// ([K,P] fp32, 2 MB at K=1000) is re-read per image from L2 / Infinity Cache with fully
// coalesced 1-KiB wave reads (a wave owns a prompt, lanes split the feature dim).
// All arithmetic fp32 with fp64 block reductions (the sums of K terms), independent of the
// towers' MFMA precision mode.
#include "common.hpp"

namespace {

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

constexpr int NWV = 16;  // waves per workgroup: the per-prompt dot products are a dependent
                         // load -> fma -> cross-lane chain, so more waves = shorter chain

__device__ __forceinline__ double block_sum_d(double v, double* red, int lane, int wave) {
  v = wave_sum_d(v);
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int i = 0; i < NWV; ++i) t += red[i];
  return t;
}

__global__ __launch_bounds__(NWV * 64) void score_kernel(const float* __restrict__ img,
                                                    const float* __restrict__ text, int K, int P,
                                                    float T, int kind, float* __restrict__ scores) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* f = (float*)smem;        // [P]
  float* sim = f + P;             // [K]
  __shared__ double red[NWV];
  __shared__ float redf[NWV];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int d = tid; d < P; d += NWV * 64) f[d] = img[(size_t)b * P + d];
  __syncthreads();
  for (int k = wave; k < K; k += NWV) {
    const float* t = text + (size_t)k * P;
    float a = 0.f;
    for (int d = lane * 4; d < P; d += 256) {
      const float4 tv = *(const float4*)(t + d);
      const float4 fv = *(const float4*)(f + d);
      a = fmaf(tv.x, fv.x, a);
      a = fmaf(tv.y, fv.y, a);
      a = fmaf(tv.z, fv.z, a);
      a = fmaf(tv.w, fv.w, a);
    }
    a = wave_sum(a);
    if (lane == 0) sim[k] = a;
  }
  __syncthreads();
  float m = -INFINITY;
  for (int k = tid; k < K; k += NWV * 64) m = fmaxf(m, sim[k]);
  m = wave_max(m);
  if (lane == 0) redf[wave] = m;
  __syncthreads();
  m = redf[0];
#pragma unroll
  for (int i = 1; i < NWV; ++i) m = fmaxf(m, redf[i]);
  if (kind == MCM_SCORE_MAX_LOGIT) {
    if (tid == 0) scores[b] = -m;
    return;
  }
  const float mt = m / T;
  double z = 0.0, ez = 0.0;
  for (int k = tid; k < K; k += NWV * 64) {
    const float u = sim[k] / T - mt;
    const float e = expf(u);
    sim[k] = e;
    z += (double)e;
    ez += (double)e * (double)u;
  }
  z = block_sum_d(z, red, lane, wave);
  if (kind == MCM_SCORE_MCM) {
    if (tid == 0) scores[b] = -(float)(1.0 / z);          // max softmax = exp(0)/z
  } else if (kind == MCM_SCORE_ENERGY) {
    if (tid == 0) scores[b] = -(T * (mt + (float)log(z)));
  } else if (kind == MCM_SCORE_ENTROPY) {
    ez = block_sum_d(ez, red, lane, wave);                // H = log z - sum(e*u)/z
    if (tid == 0) scores[b] = (float)(log(z) - ez / z);
  } else {                                                // variance of the fp32 softmax
    const float rz = (float)(1.0 / z);
    double s1 = 0.0;
    for (int k = tid; k < K; k += NWV * 64) {
      const float p = sim[k] * rz;
      sim[k] = p;
      s1 += (double)p;
    }
    const double mean = block_sum_d(s1, red, lane, wave) / (double)K;
    double s2 = 0.0;
    for (int k = tid; k < K; k += NWV * 64) {
      const double c = (double)sim[k] - mean;
      s2 += c * c;
    }
    s2 = block_sum_d(s2, red, lane, wave);
    if (tid == 0) scores[b] = (float)(-(s2 / (double)K));
  }
}

}  // namespace

hipError_t launch_score(const float* img, int B, const float* text, int K, int P, float T, int kind,
                        float* scores, hipStream_t s) {
  if (B <= 0 || K <= 0 || P % 4 || kind < 0 || kind > MCM_SCORE_VAR || !(T > 0.f))
    return hipErrorInvalidValue;
  const int lds = (P + K) * (int)sizeof(float);
  if (lds > 150 * 1024) return hipErrorInvalidValue;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)score_kernel,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL(score_kernel, dim3(B), dim3(NWV * 64), lds, s, img, text, K, P, T, kind, scores);
  return hipGetLastError();
}
