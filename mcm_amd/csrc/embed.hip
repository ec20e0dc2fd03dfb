// embed.hip — the HBM-bound data-movement kernels around the towers' GEMMs.
//
//   patchify      CLIPVisionEmbeddings' Conv2d(3,D,k=s=P,bias=False) operand gather
//                 (HF modeling_clip.py:148-154,209-210): NCHW fp32 pixels → patch matrix
//                 [B*np, Kpad], k = (c,py,px) = the [D,3,P,P] weight flattening, in the GEMM
//                 operand dtype, zero-padded to Kpad (L/14: 588 → 640).
//   (row 0 of every image — class_embedding + position_embedding[0], :212-217 — is produced by
//   layernorm.hip's layernorm_pre_kernel, inside the pre_layrnorm pass)
//   text_embed    token_embedding[id] + position_embedding[s]                  (:251-254)
//   cvt_weight    fp32 [rows, cols] → operand dtype [rows, cols_pad], zero padded
//   pool_project  pooled row → LayerNorm → projection (no bias) → L2 normalise, all fp32:
//                 CLS pool + post_layernorm + visual_projection (:650-651,:751) and
//                 EOS pool + final_layer_norm + text_projection (:559-581,:713), fused
//                 with the reference's `x /= x.norm()` (utils/detection_util.py:226,231)
#include "common.hpp"

namespace {

// generic form (any patch size, e.g. 14): one output element per thread iteration
// X2 (fp16): the patch matrix as a split image [B*np, 2 kpad] (GemmArgs::xsplit) — the split-activation arm's operand
template <int OUT, bool X2 = false>
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ px, void* out,
                                                       int B, int S, int P, int kpad) {
  const int g = S / P, np = g * g, kreal = 3 * P * P;
  const size_t total = (size_t)B * np * kpad;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total;
       i += (size_t)gridDim.x * 256) {
    const int k = (int)(i % kpad);
    const size_t m = i / kpad;
    float v = 0.f;
    if (k < kreal) {
      const int b = (int)(m / np), p = (int)(m % np);
      const int gy = p / g, gx = p % g;
      const int c = k / (P * P), rem = k % (P * P), py = rem / P, pxx = rem % P;
      v = px[(((size_t)b * 3 + c) * S + gy * P + py) * S + gx * P + pxx];
    }
    if constexpr (X2) {
      const _Float16 hi = (_Float16)v;
      _Float16* dst = (_Float16*)out + m * 2 * kpad + split_col(k);
      dst[0] = hi;
      dst[64] = (_Float16)(v - (float)hi);
    } else
    if constexpr (OUT == MCM_PREC_BF16) ((uint16_t*)out)[i] = f2bf(v);
    else if constexpr (OUT == MCM_PREC_F16) ((_Float16*)out)[i] = (_Float16)v;
    else ((float*)out)[i] = v;
  }
}

// vector form for P % 8 == 0 and kpad == 3*P*P (B/16, B/32): a thread moves 8 consecutive
// pixels of one patch row (32 B in, 16 B bf16 / 32 B fp32 out); indices are computed once per
// 8 elements and consecutive threads walk a patch row, then the next row of the same patch
template <int OUT>
__global__ __launch_bounds__(256) void patchify8_kernel(const float* __restrict__ px, void* out,
                                                        int B, int S, int P, int kpad) {
  const int g = S / P, np = g * g, k8 = kpad / 8, p8 = P / 8;
  const size_t total = (size_t)B * np * k8;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total;
       i += (size_t)gridDim.x * 256) {
    const int kq = (int)(i % k8);            // 8-element group inside the patch row vector
    const size_t m = i / k8;
    const int b = (int)(m / np), p = (int)(m % np);
    const int gy = p / g, gx = p - gy * g;
    const int rowi = kq / p8, xq = kq - rowi * p8;   // rowi = c*P + py
    const int c = rowi / P, py = rowi - c * P;
    const float* src = px + (((size_t)b * 3 + c) * S + gy * P + py) * S + gx * P + xq * 8;
    const float4 a = *(const float4*)src, d = *(const float4*)(src + 4);
    if constexpr (OUT != MCM_PREC_F32) {
      *(uint4*)((uint16_t*)out + i * 8) =
          make_uint4(pack2<OUT>(a.x, a.y), pack2<OUT>(a.z, a.w), pack2<OUT>(d.x, d.y), pack2<OUT>(d.z, d.w));
    } else {
      *(float4*)((float*)out + i * 8) = a;
      *(float4*)((float*)out + i * 8 + 4) = d;
    }
  }
}

// uint8 ingest: NHWC uint8 [B,S,S,3] -> patch matrix, fusing ToTensor (/255) and
// Normalize ((x-mean)/std, reference utils/train_eval_util.py:27-33) into the operand gather, so
// the H2D feed is 150 KB/image instead of 602 KB.  One thread = one pixel = its 3 channel values
// scattered to the 3 channel planes of the patch row vector k = (c, py, px).
template <int OUT, bool X2 = false>
__global__ __launch_bounds__(256) void patchify_u8_kernel(const uint8_t* __restrict__ px, void* out,
                                                          int B, int S, int P, int kpad, float m0,
                                                          float m1, float m2, float s0, float s1,
                                                          float s2) {
  const int g = S / P, np = g * g;
  const size_t total = (size_t)B * S * S;
  const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int x = (int)(i % S), y = (int)((i / S) % S), b = (int)(i / ((size_t)S * S));
    const int gy = y / P, py = y - gy * P, gx = x / P, pxx = x - gx * P;
    const size_t row = ((size_t)b * np + gy * g + gx) * kpad;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      // same operation order as torchvision: ToTensor = u8 / 255, Normalize = (t - mean) / std
      const float v = ((float)px[i * 3 + c] / 255.0f - mean[c]) / stdv[c];
      const size_t o = row + (size_t)(c * P + py) * P + pxx;
      if constexpr (X2) {
        const _Float16 hi = (_Float16)v;
        _Float16* dst = (_Float16*)out + 2 * row + split_col((c * P + py) * P + pxx);
        dst[0] = hi;
        dst[64] = (_Float16)(v - (float)hi);
      } else
      if constexpr (OUT == MCM_PREC_BF16) ((uint16_t*)out)[o] = f2bf(v);
      else if constexpr (OUT == MCM_PREC_F16) ((_Float16*)out)[o] = (_Float16)v;
      else ((float*)out)[o] = v;
    }
  }
}

// prompt-ensemble bank: feats [K*T, P] unit rows, class-major (row k*T + t) ->
// bank[k] = normalise(mean_t feats[k*T + t]).  One wave per class.
__global__ __launch_bounds__(256) void bank_reduce_kernel(const float* __restrict__ feats, int K,
                                                          int T, int P, float* __restrict__ bank) {
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (k >= K) return;
  float sq = 0.f;
  for (int d = lane; d < P; d += 64) {
    float a = 0.f;
    for (int t = 0; t < T; ++t) a += feats[((size_t)k * T + t) * P + d];
    a /= (float)T;
    bank[(size_t)k * P + d] = a;
    sq += a * a;
  }
  const float rn = 1.0f / sqrtf(wave_sum(sq));
  for (int d = lane; d < P; d += 64) bank[(size_t)k * P + d] *= rn;
}

__global__ __launch_bounds__(256) void text_embed_kernel(const int32_t* __restrict__ ids,
                                                         const float* __restrict__ tok,
                                                         const float* __restrict__ pos, float* x,
                                                         int K, int S, int D) {
  const int row = blockIdx.x;  // k*S + s
  const int s = row % S;
  const int id = ids[row];
  for (int d = threadIdx.x * 4; d < D; d += 1024) {
    const float4 a = *(const float4*)(tok + (size_t)id * D + d);
    const float4 p = *(const float4*)(pos + (size_t)s * D + d);
    *(float4*)(x + (size_t)row * D + d) = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
  }
}

template <int OUT>
__global__ __launch_bounds__(256) void cvt_weight_kernel(const float* __restrict__ src, void* dst,
                                                         int rows, int cols, int cols_pad) {
  const size_t total = (size_t)rows * cols_pad;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total;
       i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % cols_pad);
    const size_t r = i / cols_pad;
    const float v = c < cols ? src[r * cols + c] : 0.f;
    if constexpr (OUT == MCM_PREC_BF16) ((uint16_t*)dst)[i] = f2bf(v);
    else if constexpr (OUT == MCM_PREC_F16) ((_Float16*)dst)[i] = (_Float16)v;
    else ((float*)dst)[i] = v;
  }
}

// Split weights (GemmArgs::ksplit): an fp32 weight as the sum of two 16-bit operands, hi = round(w) and
// lo = round(w - hi) (w - hi is exact in fp32: hi is w's own leading bits), stored K-step-interleaved — 64 hi
// elements then the 64 lo elements of the same columns — so that the GEMM kernels meet both with one staging of the
// X K-step.  hi + lo carries 22 significand bits of w (fp16; lo is subnormal below 2^-14 and then exact to 2^-24
// absolute): the weight operand is exact for every purpose of a 16-bit-activation GEMM.
template <int OUT>
__global__ __launch_bounds__(256) void cvt_weight_split_kernel(const float* __restrict__ src, void* dst,
                                                               int rows, int cols, int cols_pad) {
  const size_t total = (size_t)rows * cols_pad;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total;
       i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % cols_pad);
    const size_t r = i / cols_pad;
    const float v = c < cols ? src[r * cols + c] : 0.f;
    const size_t o = r * 2 * cols_pad + (size_t)(c >> 6) * 128 + (c & 63);
    if constexpr (OUT == MCM_PREC_BF16) {
      const uint16_t hi = f2bf(v);
      ((uint16_t*)dst)[o] = hi;
      ((uint16_t*)dst)[o + 64] = f2bf(v - bf2f(hi));
    } else {
      // a weight beyond the fp16 range (|w| > 65504; count_inexact reports it) would round to inf and leave lo = -inf, a NaN
      // in every product: both halves saturate instead (hi = +-65504, lo = what is left, saturated the same way)
      const float vc = fminf(fmaxf(v, -65504.0f), 65504.0f);
      const _Float16 hi = (_Float16)vc;
      ((_Float16*)dst)[o] = hi;
      ((_Float16*)dst)[o + 64] = (_Float16)fminf(fmaxf(v - (float)hi, -65504.0f), 65504.0f);
    }
  }
}

// elements that are not exactly representable in the 16-bit operand dtype (mcm_weights_operand_exact)
template <int OUT>
__global__ __launch_bounds__(256) void count_inexact_kernel(const float* __restrict__ src, size_t n,
                                                            unsigned long long* count) {
  unsigned int mine = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float v = src[i];
    float back;
    if constexpr (OUT == MCM_PREC_BF16) back = bf2f(f2bf(v));
    else back = (float)(_Float16)v;
    mine += (back != v && v == v) ? 1u : 0u;  // (an out-of-range value comes back as inf: counted)
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(count, (unsigned long long)mine);
}

// one workgroup (16 waves) per pooled row; D <= 1024, P <= 1024.  The projection is a chain
// of dependent load -> fma -> cross-lane reductions per output feature, so it is spread over
// 16 waves (32 features each at P=512) rather than 4.
constexpr int NWP = 16;
__global__ __launch_bounds__(NWP * 64) void pool_project_kernel(
    const float* __restrict__ x, const int32_t* __restrict__ row_idx, int row_stride, int D,
    const float* __restrict__ g, const float* __restrict__ b, float eps,
    const float* __restrict__ proj, int P, float* __restrict__ out, int normalize) {
  __shared__ float y[1024];
  __shared__ float o[1024];
  __shared__ float red[2 * NWP];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t row = row_idx ? (size_t)row_idx[n] : (size_t)n * row_stride;
  const float* xr = x + row * D;
  // LayerNorm (two-pass, fp32); D <= 1024 = one element per thread
  const float xv = tid < D ? xr[tid] : 0.f;
  float s = wave_sum(xv);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int i = 0; i < NWP; ++i) tot += red[i];
  const float mean = tot / (float)D;
  const float c = tid < D ? xv - mean : 0.f;
  float q = wave_sum(c * c);
  if (lane == 0) red[NWP + wave] = q;
  __syncthreads();
  tot = 0.f;
#pragma unroll
  for (int i = 0; i < NWP; ++i) tot += red[NWP + i];
  const float rstd = 1.0f / sqrtf(tot / (float)D + eps);
  if (tid < D) y[tid] = c * rstd * g[tid] + b[tid];
  __syncthreads();
  // projection: wave per output feature, lanes split D
  float sq = 0.f;  // lane 0 of each wave accumulates squares of its outputs
  for (int p = wave; p < P; p += NWP) {
    const float* w = proj + (size_t)p * D;
    float a = 0.f;
    for (int d = lane * 4; d < D; d += 256) {
      const float4 wv = *(const float4*)(w + d);
      a = fmaf(wv.x, y[d], a);
      a = fmaf(wv.y, y[d + 1], a);
      a = fmaf(wv.z, y[d + 2], a);
      a = fmaf(wv.w, y[d + 3], a);
    }
    a = wave_sum(a);
    if (lane == 0) {
      o[p] = a;
      sq += a * a;
    }
  }
  __syncthreads();  // red[] reuse
  if (lane == 0) red[wave] = sq;
  __syncthreads();
  tot = 0.f;
#pragma unroll
  for (int i = 0; i < NWP; ++i) tot += red[i];
  const float rn = normalize ? 1.0f / sqrtf(tot) : 1.0f;  // raw = what HF get_image_features returns
  for (int p = tid; p < P; p += NWP * 64) out[(size_t)n * P + p] = o[p] * rn;
}

inline int grid_for(size_t total) {
  size_t g = (total + 255) / 256;
  return (int)(g > 4096 ? 4096 : (g ? g : 1));
}

}  // namespace

hipError_t launch_patchify(int prec, const float* pixels, void* patches, int B, int image,
                           int patch, int kpad, hipStream_t s, bool split) {
  const int g = image / patch;
  if (split) {
    if (prec != MCM_PREC_F16 || kpad % 64) return hipErrorInvalidValue;
    const size_t n = (size_t)B * g * g * kpad;
    hipLaunchKernelGGL((patchify_kernel<MCM_PREC_F16, true>), dim3(grid_for(n)), dim3(256), 0, s, pixels, patches, B, image, patch, kpad);
    return hipGetLastError();
  }
  const bool vec = patch % 8 == 0 && kpad == 3 * patch * patch && image % 4 == 0;
  const size_t total = (size_t)B * g * g * (vec ? kpad / 8 : kpad);
  const dim3 grid(grid_for(total)), block(256);
#define MCM_LAUNCH_BY_PREC(KERNEL, ...)                                                         \
  do {                                                                                           \
    if (prec == MCM_PREC_BF16) hipLaunchKernelGGL(KERNEL<MCM_PREC_BF16>, grid, block, 0, s, __VA_ARGS__); \
    else if (prec == MCM_PREC_F16) hipLaunchKernelGGL(KERNEL<MCM_PREC_F16>, grid, block, 0, s, __VA_ARGS__); \
    else hipLaunchKernelGGL(KERNEL<MCM_PREC_F32>, grid, block, 0, s, __VA_ARGS__);               \
  } while (0)
  if (vec) MCM_LAUNCH_BY_PREC(patchify8_kernel, pixels, patches, B, image, patch, kpad);
  else MCM_LAUNCH_BY_PREC(patchify_kernel, pixels, patches, B, image, patch, kpad);
  return hipGetLastError();
}

hipError_t launch_patchify_u8(int prec, const uint8_t* pixels, void* patches, int B, int image,
                              int patch, int kpad, const float* mean, const float* stdv,
                              hipStream_t s, bool split) {
  const size_t total = (size_t)B * image * image;
  const dim3 grid(grid_for(total)), block(256);
  if (split && (prec != MCM_PREC_F16 || kpad % 64)) return hipErrorInvalidValue;
  if (kpad != 3 * patch * patch) {  // padded K (L/14): the pad columns must be zero
    hipError_t e = hipMemsetAsync(patches, 0, (size_t)B * (image / patch) * (image / patch) * kpad *
                                                   prec_esize(prec) * (split ? 2 : 1), s);
    if (e != hipSuccess) return e;
  }
  if (split) {
    hipLaunchKernelGGL((patchify_u8_kernel<MCM_PREC_F16, true>), grid, block, 0, s, pixels, patches, B, image, patch, kpad,
                       mean[0], mean[1], mean[2], stdv[0], stdv[1], stdv[2]);
    return hipGetLastError();
  }
  MCM_LAUNCH_BY_PREC(patchify_u8_kernel, pixels, patches, B, image, patch, kpad, mean[0], mean[1],
                     mean[2], stdv[0], stdv[1], stdv[2]);
  return hipGetLastError();
}

hipError_t launch_bank_reduce(const float* feats, int K, int T, int P, float* bank, hipStream_t s) {
  if (K <= 0 || T <= 0 || P <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(bank_reduce_kernel, dim3((K + 3) / 4), dim3(256), 0, s, feats, K, T, P, bank);
  return hipGetLastError();
}

hipError_t launch_text_embed(const int32_t* ids, const float* tok, const float* pos, float* x,
                             int K, int S, int D, hipStream_t s) {
  if (D % 4) return hipErrorInvalidValue;
  hipLaunchKernelGGL(text_embed_kernel, dim3(K * S), dim3(256), 0, s, ids, tok, pos, x, K, S, D);
  return hipGetLastError();
}

hipError_t launch_cvt_weight(int prec, const float* src, void* dst, int rows, int cols,
                             int cols_pad, hipStream_t s) {
  const size_t total = (size_t)rows * cols_pad;
  const dim3 grid(grid_for(total)), block(256);
  MCM_LAUNCH_BY_PREC(cvt_weight_kernel, src, dst, rows, cols, cols_pad);
  return hipGetLastError();
}

hipError_t launch_cvt_weight_split(int prec, const float* src, void* dst, int rows, int cols,
                                   int cols_pad, hipStream_t s) {
  if (prec == MCM_PREC_F32 || cols_pad % 64) return hipErrorInvalidValue;
  const size_t total = (size_t)rows * cols_pad;
  const dim3 grid(grid_for(total)), block(256);
  if (prec == MCM_PREC_BF16) hipLaunchKernelGGL(cvt_weight_split_kernel<MCM_PREC_BF16>, grid, block, 0, s, src, dst, rows, cols, cols_pad);
  else hipLaunchKernelGGL(cvt_weight_split_kernel<MCM_PREC_F16>, grid, block, 0, s, src, dst, rows, cols, cols_pad);
  return hipGetLastError();
}

hipError_t launch_count_inexact(int prec, const float* src, size_t n, unsigned long long* count, hipStream_t s) {
  if (prec == MCM_PREC_F32 || n == 0) return hipSuccess;
  const dim3 grid(grid_for(n)), block(256);
  if (prec == MCM_PREC_BF16) hipLaunchKernelGGL(count_inexact_kernel<MCM_PREC_BF16>, grid, block, 0, s, src, n, count);
  else hipLaunchKernelGGL(count_inexact_kernel<MCM_PREC_F16>, grid, block, 0, s, src, n, count);
  return hipGetLastError();
}

hipError_t launch_pool_project(const float* x, const int32_t* row_idx, int row_stride, int n,
                               int D, const float* g, const float* b, float eps,
                               const float* proj, int P, float* out, hipStream_t s, bool normalize) {
  if (n <= 0 || D > 1024 || D % 4 || P > 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(pool_project_kernel, dim3(n), dim3(NWP * 64), 0, s, x, row_idx, row_stride, D, g, b,
                     eps, proj, P, out, normalize ? 1 : 0);
  return hipGetLastError();
}
