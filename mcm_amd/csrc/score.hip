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

// ---- Mahalanobis baseline (reference utils/detection_util.py:176-207, --score maha) ------------
// The reference computes, per class c,  -0.5 * (f - mu_c) P (f - mu_c)^T  with two torch.mm per class
// and keeps the max; the function returns its negation = min_c 0.5 d_c.  Expanded,
//   d_c = f P f^T  -  f . (P mu_c + P^T mu_c)  +  mu_c P mu_c^T  =  q - W_c . f + k_c,
// so per image the work is one P x P quadratic form and C dot products instead of C quadratic
// forms (C = 1000: 500x less).  W and k are prepared once per (means, precision) pair.  The three
// terms nearly cancel when f is close to a class mean, so they are accumulated in fp64.
__global__ __launch_bounds__(256) void maha_prepare_kernel(const float* __restrict__ means,
                                                           const float* __restrict__ prec, int P,
                                                           double* __restrict__ w, double* __restrict__ k) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* mu = (double*)smem;   // [P]
  __shared__ double red[4];
  const int c = blockIdx.x, tid = threadIdx.x;
  for (int j = tid; j < P; j += 256) mu[j] = (double)means[(size_t)c * P + j];
  __syncthreads();
  double kc = 0.0;
  for (int p = tid; p < P; p += 256) {
    double a = 0.0, b = 0.0;  // (P mu)_p and (P^T mu)_p
    for (int j = 0; j < P; ++j) {
      a += (double)prec[(size_t)p * P + j] * mu[j];
      b += (double)prec[(size_t)j * P + p] * mu[j];
    }
    w[(size_t)c * P + p] = a + b;
    kc += mu[p] * a;
  }
  kc = wave_sum_d(kc);
  if ((tid & 63) == 0) red[tid >> 6] = kc;
  __syncthreads();
  if (tid == 0) k[c] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(NWV * 64) void maha_score_kernel(const float* __restrict__ feats,
                                                              const float* __restrict__ prec,
                                                              const double* __restrict__ w,
                                                              const double* __restrict__ k, int C, int P,
                                                              float* __restrict__ scores) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* f = (double*)smem;  // [P]
  __shared__ double red[NWV];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int j = tid; j < P; j += NWV * 64) f[j] = (double)feats[(size_t)b * P + j];
  __syncthreads();
  double qa = 0.0;  // q = f P f^T: a wave per row of P
  for (int p = wave; p < P; p += NWV) {
    double a = 0.0;
    for (int j = lane; j < P; j += 64) a += (double)prec[(size_t)p * P + j] * f[j];
    a = wave_sum_d(a);
    if (lane == 0) qa += f[p] * a;
  }
  const double q = block_sum_d(lane == 0 ? qa : 0.0, red, lane, wave);
  double best = INFINITY;
  for (int c = wave; c < C; c += NWV) {
    double a = 0.0;
    for (int j = lane; j < P; j += 64) a += w[(size_t)c * P + j] * f[j];
    a = wave_sum_d(a);
    best = fmin(best, q - a + k[c]);
  }
  __syncthreads();
  if (lane == 0) red[wave] = best;
  __syncthreads();
  if (tid == 0) {
    double m = red[0];
    for (int i = 1; i < NWV; ++i) m = fmin(m, red[i]);
    scores[b] = (float)(0.5 * m);
  }
}

}  // namespace

hipError_t launch_maha_prepare(const float* means, const float* prec, int C, int P, double* w, double* c,
                               hipStream_t s) {
  if (C <= 0 || P <= 0 || P > 4096) return hipErrorInvalidValue;
  hipLaunchKernelGGL(maha_prepare_kernel, dim3(C), dim3(256), P * sizeof(double), s, means, prec, P, w, c);
  return hipGetLastError();
}

hipError_t launch_maha_score(const float* feats, int B, const float* prec, const double* w, const double* c,
                             int C, int P, float* scores, hipStream_t s) {
  if (B <= 0 || C <= 0 || P <= 0 || P > 4096) return hipErrorInvalidValue;
  hipLaunchKernelGGL(maha_score_kernel, dim3(B), dim3(NWV * 64), P * sizeof(double), s, feats, prec, w, c, C,
                     P, scores);
  return hipGetLastError();
}

hipError_t launch_score(const float* img, int B, const float* text, int K, int P, float T, int kind,
                        float* scores, hipStream_t s) {
  if (B <= 0 || K <= 0 || P % 4 || kind < 0 || kind > MCM_SCORE_VAR || !(T > 0.f))
    return hipErrorInvalidValue;
  const int lds = (P + K) * (int)sizeof(float);
  if (lds > 150 * 1024) return hipErrorInvalidValue;
  static PerDeviceFlag attr_set;
  if (!attr_set.get()) {
    hipError_t e = hipFuncSetAttribute((const void*)score_kernel,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) return e;
    attr_set.set();
  }
  hipLaunchKernelGGL(score_kernel, dim3(B), dim3(NWV * 64), lds, s, img, text, K, P, T, kind, scores);
  return hipGetLastError();
}
