// metrics.hip — AUROC / AUPR / FPR@recall on the device, so the score
// vectors never leave HBM: three doubles come back instead of 60k floats.
//
// Replaces reference utils/detection_util.py:66-119 (get_measures / fpr_and_fdr_at_recall and
// the sklearn roc_auc_score / average_precision_score calls inside it) with a sort-free
// formulation.  For every example s_i (ID = positive class, :112-113) count
//     tp_i = #{pos >= s_i}   fp_i = #{neg >= s_i}   gt_i = #{neg > s_i}
// by brute force (N^2 fp32 compares: 3.6e9 at 50k+10k, a few hundred microseconds of VALU);
// then, exactly:
//   AUROC = sum_{i in pos} (2*n_neg - fp_i - gt_i) / (2 * n_pos * n_neg)     Mann-Whitney U with
//           ties at 1/2 == the trapezoid area under sklearn's ROC curve;  integer numerator
//   AUPR  = (1/n_pos) * sum_{i in pos} tp_i / (tp_i + fp_i)                  == sklearn's step-wise
//           sum over distinct thresholds of dRecall * Precision (positives sharing a value add up
//           to that threshold's dRecall)
//   FPR   = fp_j / n_neg at the operating point j (one per distinct value >= min(pos)) whose
//           recall tp_j/n_pos is closest to the level, ties resolved towards the LOWEST threshold
//           (the reversed-slice argmin of :100-106 scans from full recall downwards).
// Comparisons are on the fp32 scores (as the reference's float32 arrays), recall / precision
// arithmetic is fp64 like numpy's.
#include "common.hpp"

namespace {

constexpr int CT = 256;     // threads per counting workgroup = examples per workgroup
constexpr int TILE = 4096;  // staged comparison values per LDS tile

__device__ __forceinline__ float ex_at(const float* pos, long n_pos, const float* neg, long i, float sgn) {
  return sgn * (i < n_pos ? pos[i] : neg[i - n_pos]);
}

__global__ __launch_bounds__(CT) void count_kernel(const float* __restrict__ pos, long n_pos,
                                                   const float* __restrict__ neg, long n_neg, float sgn,
                                                   uint32_t* __restrict__ tp, uint32_t* __restrict__ fp,
                                                   uint32_t* __restrict__ gt) {
  __shared__ __attribute__((aligned(16))) float tile[TILE];
  const long n = n_pos + n_neg;
  const long i = (long)blockIdx.x * CT + threadIdx.x;
  const float s = i < n ? ex_at(pos, n_pos, neg, i, sgn) : 0.f;
  uint32_t c_tp = 0, c_fp = 0, c_gt = 0;
  for (long j0 = 0; j0 < n_pos; j0 += TILE) {
    const int m = (int)((n_pos - j0) < TILE ? (n_pos - j0) : TILE);
    __syncthreads();
    for (int j = threadIdx.x; j < TILE; j += CT) tile[j] = j < m ? sgn * pos[j0 + j] : -INFINITY;
    __syncthreads();
#pragma unroll 4
    for (int j = 0; j < TILE; j += 4) {  // wave-uniform address: LDS broadcast reads
      const float4 v = *(const float4*)(tile + j);
      c_tp += (v.x >= s) + (v.y >= s) + (v.z >= s) + (v.w >= s);
    }
  }
  for (long j0 = 0; j0 < n_neg; j0 += TILE) {
    const int m = (int)((n_neg - j0) < TILE ? (n_neg - j0) : TILE);
    __syncthreads();
    for (int j = threadIdx.x; j < TILE; j += CT) tile[j] = j < m ? sgn * neg[j0 + j] : -INFINITY;
    __syncthreads();
#pragma unroll 4
    for (int j = 0; j < TILE; j += 4) {
      const float4 v = *(const float4*)(tile + j);
      c_fp += (v.x >= s) + (v.y >= s) + (v.z >= s) + (v.w >= s);
      c_gt += (v.x > s) + (v.y > s) + (v.z > s) + (v.w > s);
    }
  }
  if (i < n) {
    tp[i] = c_tp;
    fp[i] = c_fp;
    gt[i] = c_gt;
  }
}

constexpr int RT = 1024;  // the final reduction is one workgroup

template <typename T, typename F>
__device__ __forceinline__ T block_reduce(T v, T* red, F op) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = op(v, __shfl_xor(v, o, 64));
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  T t = red[0];
  for (int w = 1; w < RT / 64; ++w) t = op(t, red[w]);
  return t;
}

// out[0..2] = auroc, aupr, fpr;  out[3] = the chosen operating point's threshold (diagnostic)
__global__ __launch_bounds__(RT) void measures_kernel(const float* __restrict__ pos, long n_pos,
                                                      const float* __restrict__ neg, long n_neg, float sgn,
                                                      const uint32_t* __restrict__ tp,
                                                      const uint32_t* __restrict__ fp,
                                                      const uint32_t* __restrict__ gt, double level,
                                                      double* __restrict__ out) {
  __shared__ double red_d[RT / 64];
  __shared__ unsigned long long red_u[RT / 64];
  __shared__ float red_f[RT / 64];
  const long n = n_pos + n_neg;
  const int tid = threadIdx.x;

  unsigned long long u2 = 0;  // 2 * Mann-Whitney U
  double ap = 0.0;
  float minpos = INFINITY;
  for (long i = tid; i < n_pos; i += RT) {
    u2 += 2ull * (unsigned long long)n_neg - fp[i] - gt[i];
    ap += (double)tp[i] / ((double)tp[i] + (double)fp[i]);
    minpos = fminf(minpos, sgn * pos[i]);
  }
  u2 = block_reduce(u2, red_u, [](unsigned long long a, unsigned long long b) { return a + b; });
  ap = block_reduce(ap, red_d, [](double a, double b) { return a + b; });
  minpos = block_reduce(minpos, red_f, [](float a, float b) { return fminf(a, b); });

  // operating point: min |recall - level|, then lowest threshold
  double bd = INFINITY;
  float bs = INFINITY;
  for (long i = tid; i < n; i += RT) {
    const float s = ex_at(pos, n_pos, neg, i, sgn);
    if (!(s >= minpos)) continue;
    const double d = fabs((double)tp[i] / (double)n_pos - level);
    if (d < bd || (d == bd && s < bs)) { bd = d; bs = s; }
  }
  const double gd = block_reduce(bd, red_d, [](double a, double b) { return a < b ? a : b; });
  if (bd != gd) bs = INFINITY;
  const float gs = block_reduce(bs, red_f, [](float a, float b) { return fminf(a, b); });
  // every example at the chosen threshold carries the same fp count
  unsigned long long pick = 0;
  for (long i = tid; i < n; i += RT)
    if (ex_at(pos, n_pos, neg, i, sgn) == gs) pick = fp[i];
  pick = block_reduce(pick, red_u, [](unsigned long long a, unsigned long long b) { return a > b ? a : b; });
  if (tid == 0) {
    out[0] = (double)u2 / (2.0 * (double)n_pos * (double)n_neg);
    out[1] = ap / (double)n_pos;
    out[2] = (double)pick / (double)n_neg;
    out[3] = (double)gs;
  }
}

// Fixed-edge histogram of a score vector (numpy.histogram semantics: bin i = [e_i, e_{i+1}), the last bin
// closed, values outside [e_0, e_nb] dropped).  The per-rank payload of BASELINE.json's "all-gather of
// per-shard score histograms": constant size, summed over ranks by one RCCL all-reduce.
constexpr int HB_MAX = 8192;
__global__ __launch_bounds__(256) void hist_kernel(const float* __restrict__ x, long n,
                                                   const float* __restrict__ edges, int nb,
                                                   unsigned long long* __restrict__ counts) {
  __shared__ float e[HB_MAX + 1];
  __shared__ uint32_t c[HB_MAX];
  for (int i = threadIdx.x; i <= nb; i += 256) e[i] = edges[i];
  for (int i = threadIdx.x; i < nb; i += 256) c[i] = 0;
  __syncthreads();
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float v = x[i];
    if (!(v >= e[0] && v <= e[nb])) continue;
    int lo = 0, hi = nb;  // largest b with e[b] <= v, clamped to the last bin
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (e[mid] <= v) lo = mid; else hi = mid;
    }
    atomicAdd(&c[lo], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nb; i += 256)
    if (c[i]) atomicAdd(&counts[i], (unsigned long long)c[i]);
}

}  // namespace

hipError_t launch_histogram(const float* x, long n, const float* edges, int nb, unsigned long long* counts,
                            hipStream_t s) {
  if (!x || !edges || !counts || n < 0 || nb <= 0 || nb > HB_MAX) return hipErrorInvalidValue;
  hipError_t e = hipMemsetAsync(counts, 0, (size_t)nb * sizeof(unsigned long long), s);
  if (e != hipSuccess || n == 0) return e;
  const long blocks = (n + 255) / 256;
  hipLaunchKernelGGL(hist_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, s, x, n, edges, nb,
                     counts);
  return hipGetLastError();
}

size_t measures_workspace_bytes(long n) { return (size_t)n * 3 * sizeof(uint32_t) + 4 * sizeof(double) + 64; }

hipError_t launch_measures(const float* pos, long n_pos, const float* neg, long n_neg, int negate,
                           double level, void* workspace, double** out_dev, hipStream_t s) {
  if (!pos || !neg || n_pos <= 0 || n_neg <= 0 || n_pos + n_neg > 0x7fffffffL) return hipErrorInvalidValue;
  const long n = n_pos + n_neg;
  double* out = (double*)workspace;  // 4 doubles, then the three count arrays
  uint32_t* tp = (uint32_t*)((char*)workspace + 64);
  uint32_t* fp = tp + n;
  uint32_t* gt = fp + n;
  const float sgn = negate ? -1.f : 1.f;
  hipLaunchKernelGGL(count_kernel, dim3((unsigned)((n + CT - 1) / CT)), dim3(CT), 0, s, pos, n_pos, neg, n_neg,
                     sgn, tp, fp, gt);
  hipLaunchKernelGGL(measures_kernel, dim3(1), dim3(RT), 0, s, pos, n_pos, neg, n_neg, sgn, tp, fp, gt, level,
                     out);
  *out_dev = out;
  return hipGetLastError();
}
