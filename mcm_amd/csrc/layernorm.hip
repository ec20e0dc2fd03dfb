// layernorm.hip — row LayerNorm, fp32 statistics, one 64-lane wave per row.
//
// Replaces nn.LayerNorm of the CLIP towers (HF modeling_clip.py:358,360 layer_norm1/2,
// :605 pre_layrnorm, :608 post_layernorm, :507 final_layer_norm; eps 1e-5 from
// configuration_clip.py).  HBM-bound: reads the fp32 residual row once (16 B/lane),
// keeps it in registers for the exact two-pass mean / centred variance, writes the
// normalised row in the GEMM operand dtype (bf16 or fp32).  In-place (y == x, fp32) is
// safe: a wave has its whole row in registers before it stores.
#include "ln_row.hpp"

namespace {

constexpr int MAXV = LN_MAXV;  // float4 per lane: D <= 64*4*4 = 1024

template <int OUT, bool X2 = false>  // OUT = MCM_PREC_F32: fp32 rows; BF16 / F16: packed 16-bit rows (X2: split rows, ys counts their elements)
__global__ __launch_bounds__(256) void layernorm_kernel(const float* x,
                                                        const float* __restrict__ g,
                                                        const float* __restrict__ b, void* y,
                                                        int M, int D, float eps, size_t xs,
                                                        size_t ys, int rev, int nt, unsigned int* sat) {
  enter_precision_mode<OUT>();
  const int lane = threadIdx.x & 63;
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  if (rev) row = M - 1 - row;
  const float* xr = x + (size_t)row * xs;
  float4 v[MAXV];
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
    }
  }
  ln_row_apply(v, g, b, D, eps, lane);
  float amax = 0.f;
  if constexpr (OUT != MCM_PREC_F32) ln_row_store<OUT, 0, X2>(v, (uint16_t*)y + (size_t)row * ys, D, lane, amax);
  else ln_row_store<OUT>(v, (float*)y + (size_t)row * ys, D, lane, amax);
  sat_report<OUT>(amax, sat);
}

// (optionally the CLS row of every image =) pre_layrnorm followed by layer 0's layer_norm1 in one pass (HF modeling_clip.py:642 then :370): the fp32 result of
// the first LayerNorm is written back in place (it is the residual stream) AND, still in registers, normalised again
// into the first QKV GEMM's operand.  Same arithmetic as the two launches (the second LayerNorm sees exactly the fp32
// values the first one stores), one 310-MB read of x less.
template <int OUT, bool X2 = false>
__global__ __launch_bounds__(256) void layernorm_pre_kernel(float* x, const float* __restrict__ g0,
                                                            const float* __restrict__ b0,
                                                            const float* __restrict__ g1,
                                                            const float* __restrict__ b1, void* y, int M, int D,
                                                            float eps, int rev, unsigned int* sat,
                                                            const float* __restrict__ cls,
                                                            const float* __restrict__ pos0, int ntok) {
  enter_precision_mode<OUT>();
  const int lane = threadIdx.x & 63;
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  if (rev) row = M - 1 - row;
  float* xr = x + (size_t)row * D;
  float4 v[MAXV];
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int d = (i * 64 + lane) * 4;
    if (d < D) {
      if (cls != nullptr && row % ntok == 0) {  // CLS row: class_embedding + position_embedding[0] (HF :212-217)
        const float4 c = *(const float4*)(cls + d), p = *(const float4*)(pos0 + d);
        v[i] = make_float4(c.x + p.x, c.y + p.y, c.z + p.z, c.w + p.w);
      } else {
        v[i] = *(const float4*)(xr + d);
      }
    }
  }
  ln_row_apply(v, g0, b0, D, eps, lane);       // pre_layrnorm: the residual stream, written back in fp32
  ln_row_store<MCM_PREC_F32>(v, xr, D, lane, amax);
  ln_row_apply(v, g1, b1, D, eps, lane);       // layer 0's layer_norm1 of exactly those fp32 values
  if constexpr (OUT != MCM_PREC_F32) ln_row_store<OUT, 0, X2>(v, (uint16_t*)y + (size_t)row * D * (X2 ? 2 : 1), D, lane, amax);
  else ln_row_store<OUT>(v, (float*)y + (size_t)row * D, D, lane, amax);
  sat_report<OUT>(amax, sat);
}

// ---- LayerNorm fold (gemm.hip): the LayerNorm between a residual GEMM and its consumer, without its own pass -------
// Weight side, once per weight: with z = gamma o x the consumer computes sum_k z_k W[n,k]; LayerNorm(x) W^T + b =
// ((z W^T)[n] - mean c[n]) rstd + b'[n] with c[n] = sum_k gamma_k W[n,k], b'[n] = b[n] + sum_k beta_k W[n,k].
// W is the OPERAND copy (the values the MFMAs multiply), so the mean term cancels exactly what the GEMM summed.
// One wave per output column n; fp32 sums of K <= 4096 products.
template <int PREC>
__global__ __launch_bounds__(256) void fold_prep_kernel(const uint16_t* __restrict__ w, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ bias,
                                                        float* __restrict__ c, float* __restrict__ bfold, int N, int K) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const uint16_t* wr = w + (size_t)n * K;
  float sc = 0.f, sb = 0.f;
  for (int k = lane; k < K; k += 64) {
    float wv;
    if constexpr (PREC == MCM_PREC_F16) wv = (float)__builtin_bit_cast(_Float16, wr[k]);
    else wv = bf2f(wr[k]);
    sc = fmaf(gamma[k], wv, sc);
    sb = fmaf(beta[k], wv, sb);
  }
  sc = wave_sum(sc);
  sb = wave_sum(sb);
  if (lane == 0) {
    c[n] = sc;
    bfold[n] = (bias ? bias[n] : 0.f) + sb;
  }
}

// Producer side for problems whose residual GEMM has no fused fold epilogue (the tile kernel: small batches): one wave
// per row, 256 columns per pass — the 16 lanes of a DPP row hold one 64-column slot, 4 columns per lane, exactly the
// arrangement of the ping-pong epilogue after its LDS bounce, and the same slot_moments / multiply / pack: same bits.
template <int PREC>
__global__ __launch_bounds__(256) void fold_rows_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        uint16_t* __restrict__ z, float2* __restrict__ part, int M,
                                                        int D, unsigned int* sat) {
  enter_precision_mode<PREC>();
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float amax = 0.f;
  for (int c0 = 0; c0 < D; c0 += 256) {  // D is a multiple of 256 (launch_fold_rows)
    const int col = c0 + lane * 4;
    const f32x4_t v = *(const f32x4_t*)(x + (size_t)row * D + col);
    const f32x4_t gm = *(const f32x4_t*)(gamma + col);
    const f32x4_t zz = fold_scale(v, gm);
    sat_track<PREC>(amax, zz[0], zz[1]);
    sat_track<PREC>(amax, zz[2], zz[3]);
    *(uint2*)(z + (size_t)row * D + col) = make_uint2(pack2<PREC>(zz[0], zz[1]), pack2<PREC>(zz[2], zz[3]));
    float sm, sq;
    slot_moments(v, sm, sq);
    if ((lane & 15) == 0) part[(size_t)(col >> 6) * M + row] = make_float2(sm, sq);
  }
  sat_report<PREC>(amax, sat);
}

// Row side, once per folded LayerNorm: the producer left, per row and 64-column slot, the slot's sum and its sum of
// squares about the slot's own mean; combined here (Chan et al.) into the row's mean and centred variance - the same
// two quantities the LayerNorm kernel computes, without E[x^2] - mean^2 cancellation.  part: [slots][M], rs: [M].
__global__ __launch_bounds__(256) void fold_stats_kernel(const float2* __restrict__ part, int slots, int M, int D,
                                                         float eps, float2* __restrict__ rs) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= M) return;
  float sum = 0.f;
  for (int j = 0; j < slots; ++j) sum += part[(size_t)j * M + row].x;
  const float mean = sum / (float)D;
  float m2 = 0.f;
  for (int j = 0; j < slots; ++j) {
    const float2 p = part[(size_t)j * M + row];
    const float dm = p.x * (1.0f / 64.0f) - mean;
    m2 += p.y + 64.0f * dm * dm;
  }
  const float rstd = 1.0f / sqrtf(m2 / (float)D + eps);
  rs[row] = make_float2(rstd, mean * rstd);
}

}  // namespace

hipError_t launch_layernorm_pre(int prec, float* x, const float* g0, const float* b0, const float* g1,
                                const float* b1, void* y, int M, int D, float eps, hipStream_t s,
                                bool reverse, unsigned int* sat, const float* cls, const float* pos0, int ntok, bool split) {
  if (M <= 0 || D <= 0 || D % 4 || D > 64 * 4 * MAXV) return hipErrorInvalidValue;
  const dim3 grid((M + 3) / 4), block(256);
  const int rev = reverse ? 1 : 0;
  if (split) {  // split rows: fp16 only, whole 64-column blocks
    if (prec != MCM_PREC_F16 || D % 64) return hipErrorInvalidValue;
    hipLaunchKernelGGL((layernorm_pre_kernel<MCM_PREC_F16, true>), grid, block, 0, s, x, g0, b0, g1, b1, y, M, D, eps, rev, sat, cls, pos0, ntok > 0 ? ntok : 1);
    return hipGetLastError();
  }
  if (prec == MCM_PREC_BF16)
    hipLaunchKernelGGL(layernorm_pre_kernel<MCM_PREC_BF16>, grid, block, 0, s, x, g0, b0, g1, b1, y, M, D, eps, rev, sat, cls, pos0, ntok > 0 ? ntok : 1);
  else if (prec == MCM_PREC_F16)
    hipLaunchKernelGGL(layernorm_pre_kernel<MCM_PREC_F16>, grid, block, 0, s, x, g0, b0, g1, b1, y, M, D, eps, rev, sat, cls, pos0, ntok > 0 ? ntok : 1);
  else
    hipLaunchKernelGGL(layernorm_pre_kernel<MCM_PREC_F32>, grid, block, 0, s, x, g0, b0, g1, b1, y, M, D, eps, rev, sat, cls, pos0, ntok > 0 ? ntok : 1);
  return hipGetLastError();
}

hipError_t launch_layernorm(int prec, const float* x, const float* g, const float* b, void* y,
                            int M, int D, float eps, bool out_f32, hipStream_t s, size_t x_stride,
                            size_t y_stride, bool reverse, unsigned int* sat, bool split) {
  if (M <= 0 || D <= 0 || D % 4 || D > 64 * 4 * MAXV) return hipErrorInvalidValue;
  const size_t xs = x_stride ? x_stride : (size_t)D, ys = y_stride ? y_stride : (size_t)D * (split ? 2 : 1);
  if (xs % 4 || ys % 4) return hipErrorInvalidValue;
  if (split) {  // split rows (y_stride counts the split row's elements): fp16 only, whole 64-column blocks
    if (prec != MCM_PREC_F16 || out_f32 || D % 64) return hipErrorInvalidValue;
    hipLaunchKernelGGL((layernorm_kernel<MCM_PREC_F16, true>), dim3((M + 3) / 4), dim3(256), 0, s, x, g, b, y, M, D, eps, xs, ys,
                       reverse ? 1 : 0, 1, sat);
    return hipGetLastError();
  }
  const dim3 grid((M + 3) / 4), block(256);
  const int rev = reverse ? 1 : 0;
  // x is streamed with the non-temporal hint: the 310-MB residual read would otherwise push the
  // 155 MB of LayerNorm output — the next GEMM's X operand — out of L2 / Infinity Cache
  // (measured: GEMM time -4 %, +3.3 % end to end).
  constexpr int nt = 1;
  if (prec == MCM_PREC_BF16 && !out_f32)
    hipLaunchKernelGGL(layernorm_kernel<MCM_PREC_BF16>, grid, block, 0, s, x, g, b, y, M, D, eps, xs, ys, rev, nt, sat);
  else if (prec == MCM_PREC_F16 && !out_f32)
    hipLaunchKernelGGL(layernorm_kernel<MCM_PREC_F16>, grid, block, 0, s, x, g, b, y, M, D, eps, xs, ys, rev, nt, sat);
  else
    hipLaunchKernelGGL(layernorm_kernel<MCM_PREC_F32>, grid, block, 0, s, x, g, b, y, M, D, eps, xs, ys, rev, nt, sat);
  return hipGetLastError();
}

hipError_t launch_fold_prep(int prec, const void* w, const float* gamma, const float* beta, const float* bias,
                            float* c, float* bfold, int N, int K, hipStream_t s) {
  if (prec == MCM_PREC_F32 || N <= 0 || K <= 0) return hipErrorInvalidValue;
  const dim3 grid((N + 3) / 4), block(256);
  if (prec == MCM_PREC_F16)
    hipLaunchKernelGGL(fold_prep_kernel<MCM_PREC_F16>, grid, block, 0, s, (const uint16_t*)w, gamma, beta, bias, c, bfold, N, K);
  else
    hipLaunchKernelGGL(fold_prep_kernel<MCM_PREC_BF16>, grid, block, 0, s, (const uint16_t*)w, gamma, beta, bias, c, bfold, N, K);
  return hipGetLastError();
}

hipError_t launch_fold_stats(const float2* part, int slots, int M, int D, float eps, float2* rs, hipStream_t s) {
  if (slots <= 0 || M <= 0 || D != slots * 64) return hipErrorInvalidValue;
  hipLaunchKernelGGL(fold_stats_kernel, dim3((M + 255) / 256), dim3(256), 0, s, part, slots, M, D, eps, rs);
  return hipGetLastError();
}

hipError_t launch_fold_rows(int prec, const float* x, const float* gamma, void* z, float2* part, int M, int D,
                            hipStream_t s, unsigned int* sat) {
  if (prec == MCM_PREC_F32 || M <= 0 || D <= 0 || D % 256) return hipErrorInvalidValue;
  const dim3 grid((M + 3) / 4), block(256);
  if (prec == MCM_PREC_F16)
    hipLaunchKernelGGL(fold_rows_kernel<MCM_PREC_F16>, grid, block, 0, s, x, gamma, (uint16_t*)z, part, M, D, sat);
  else
    hipLaunchKernelGGL(fold_rows_kernel<MCM_PREC_BF16>, grid, block, 0, s, x, gamma, (uint16_t*)z, part, M, D, sat);
  return hipGetLastError();
}
