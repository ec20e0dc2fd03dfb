// layernorm.hip — row LayerNorm, fp32 statistics, one 64-lane wave per row.
//
// Replaces nn.LayerNorm of the CLIP towers (HF modeling_clip.py:358,360 layer_norm1/2,
// :605 pre_layrnorm, :608 post_layernorm, :507 final_layer_norm; eps 1e-5 from
// configuration_clip.py).  HBM-bound: reads the fp32 residual row once (16 B/lane),
// keeps it in registers for the exact two-pass mean / centred variance, writes the
// normalised row in the GEMM operand dtype (bf16 or fp32).  In-place (y == x, fp32) is
// safe: a wave has its whole row in registers before it stores.
#include "common.hpp"

namespace {

constexpr int MAXV = 4;  // float4 per lane: D <= 64*4*4 = 1024

template <int OUT>  // OUT = MCM_PREC_F32: fp32 rows; BF16 / F16: packed 16-bit rows
__global__ __launch_bounds__(256) void layernorm_kernel(const float* x,
                                                        const float* __restrict__ g,
                                                        const float* __restrict__ b, void* y,
                                                        int M, int D, float eps, size_t xs,
                                                        size_t ys, int rev, int nt) {
  enter_precision_mode<OUT>();
  const int lane = threadIdx.x & 63;
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  if (rev) row = M - 1 - row;
  const float* xr = x + (size_t)row * xs;
  float4 v[MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int d = (i * 64 + lane) * 4;
    if (d < D) {
      if (nt) {
        typedef float f4_t __attribute__((ext_vector_type(4)));
        const f4_t t = __builtin_nontemporal_load((const f4_t*)(xr + d));
        v[i] = make_float4(t.x, t.y, t.z, t.w);
      } else {
        v[i] = *(const float4*)(xr + d);
      }
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int d = (i * 64 + lane) * 4;
    if (d < D) {
      v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
      q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int d = (i * 64 + lane) * 4;
    if (d < D) {
      const float4 gv = *(const float4*)(g + d);
      const float4 bv = *(const float4*)(b + d);
      float4 o;
      o.x = v[i].x * rstd * gv.x + bv.x;
      o.y = v[i].y * rstd * gv.y + bv.y;
      o.z = v[i].z * rstd * gv.z + bv.z;
      o.w = v[i].w * rstd * gv.w + bv.w;
      if constexpr (OUT != MCM_PREC_F32) {
        uint2 pk;
        pk.x = pack2<OUT>(o.x, o.y);
        pk.y = pack2<OUT>(o.z, o.w);
        *(uint2*)((uint16_t*)y + (size_t)row * ys + d) = pk;
      } else {
        *(float4*)((float*)y + (size_t)row * ys + d) = o;
      }
    }
  }
}

// ---- folded LayerNorm (common.hpp, EPI_RESID_LN): row statistics from the producer's partial sums ------
// stats[M][npart][2] = (sum x, sum x^2) over 64-column slices  ->  rowab[M][2] = (rstd, rstd * mean).
// Partials are fp32 sums of 64 fp32 terms; they are combined in fp64, so the variance's mean^2 subtraction
// loses nothing beyond the partials' own rounding (relative 1e-7, against 5e-4 operand rounding).
__global__ __launch_bounds__(256) void ln_stats_finalize_kernel(const float* __restrict__ stats, int npart, int M,
                                                                int D, float eps, float* __restrict__ rowab) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  const float2* p = (const float2*)(stats + (size_t)m * npart * 2);
  double s1 = 0.0, s2 = 0.0;
  for (int i = 0; i < npart; ++i) {
    const float2 v = p[i];
    s1 += v.x;
    s2 += v.y;
  }
  const double mean = s1 / D;
  double var = s2 / D - mean * mean;
  var = var > 0.0 ? var : 0.0;
  const float rstd = 1.0f / sqrtf((float)var + eps);
  *(float2*)(rowab + 2 * (size_t)m) = make_float2(rstd, rstd * (float)mean);
}

// colsum[n] = sum_k gamma[k] * W[n,k],  bias2[n] = bias[n] + sum_k beta[k] * W[n,k], with W the OPERAND copy
// the MFMAs read (so the folded terms match the products exactly); one wave per output column, fp64 sums.
template <int PREC>
__global__ __launch_bounds__(256) void ln_fold_kernel(const void* __restrict__ w, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ bias,
                                                      int N, int K, float* __restrict__ colsum,
                                                      float* __restrict__ bias2) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= N) return;
  double sg = 0.0, sb = 0.0;
  for (int k = lane; k < K; k += 64) {
    float wv;
    if constexpr (PREC == MCM_PREC_F16) wv = (float)((const _Float16*)w)[(size_t)n * K + k];
    else wv = bf2f(((const uint16_t*)w)[(size_t)n * K + k]);
    sg += (double)gamma[k] * wv;
    sb += (double)beta[k] * wv;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    sg += __shfl_xor(sg, o, 64);
    sb += __shfl_xor(sb, o, 64);
  }
  if (lane == 0) {
    colsum[n] = (float)sg;
    bias2[n] = (float)((bias ? (double)bias[n] : 0.0) + sb);
  }
}

}  // namespace

hipError_t launch_ln_stats_finalize(const float* stats, int npart, int M, int D, float eps, float* rowab,
                                    hipStream_t s) {
  if (!stats || !rowab || npart <= 0 || M <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(ln_stats_finalize_kernel, dim3((M + 255) / 256), dim3(256), 0, s, stats, npart, M, D, eps, rowab);
  return hipGetLastError();
}

hipError_t launch_ln_fold(int prec, const void* w_op, const float* gamma, const float* beta, const float* bias,
                          int N, int K, float* colsum, float* bias2, hipStream_t s) {
  if (prec == MCM_PREC_F16)
    hipLaunchKernelGGL(ln_fold_kernel<MCM_PREC_F16>, dim3((N + 3) / 4), dim3(256), 0, s, w_op, gamma, beta, bias, N, K,
                       colsum, bias2);
  else if (prec == MCM_PREC_BF16)
    hipLaunchKernelGGL(ln_fold_kernel<MCM_PREC_BF16>, dim3((N + 3) / 4), dim3(256), 0, s, w_op, gamma, beta, bias, N, K,
                       colsum, bias2);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launch_layernorm(int prec, const float* x, const float* g, const float* b, void* y,
                            int M, int D, float eps, bool out_f32, hipStream_t s, size_t x_stride,
                            size_t y_stride, bool reverse) {
  if (M <= 0 || D <= 0 || D % 4 || D > 64 * 4 * MAXV) return hipErrorInvalidValue;
  const size_t xs = x_stride ? x_stride : (size_t)D, ys = y_stride ? y_stride : (size_t)D;
  if (xs % 4 || ys % 4) return hipErrorInvalidValue;
  const dim3 grid((M + 3) / 4), block(256);
  const int rev = reverse ? 1 : 0;
  // x is streamed with the non-temporal hint: the 310-MB residual read would otherwise push the
  // 155 MB of LayerNorm output — the next GEMM's X operand — out of L2 / Infinity Cache
  // (measured: GEMM time -4 %, +3.3 % end to end).
  constexpr int nt = 1;
  if (prec == MCM_PREC_BF16 && !out_f32)
    hipLaunchKernelGGL(layernorm_kernel<MCM_PREC_BF16>, grid, block, 0, s, x, g, b, y, M, D, eps, xs, ys, rev, nt);
  else if (prec == MCM_PREC_F16 && !out_f32)
    hipLaunchKernelGGL(layernorm_kernel<MCM_PREC_F16>, grid, block, 0, s, x, g, b, y, M, D, eps, xs, ys, rev, nt);
  else
    hipLaunchKernelGGL(layernorm_kernel<MCM_PREC_F32>, grid, block, 0, s, x, g, b, y, M, D, eps, xs, ys, rev, nt);
  return hipGetLastError();
}
