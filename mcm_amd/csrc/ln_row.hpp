// ln_row.hpp — the arithmetic of one LayerNorm row on one 64-lane wave, shared by every place that normalises a
// row: layernorm_kernel / layernorm_pre_kernel (layernorm.hip) and the LayerNorm tail of the residual ping-pong GEMM
// (gemm.hip, "LayerNorm in the tail").  One definition with floating-point contraction switched off and the fused
// multiply-adds written out (the forms hipcc contracted the round-1 LayerNorm kernel to), so the same row gives the same
// bits whichever kernel — whichever batch size — it went through.  The steps are separate functions so
// that the GEMM tail can run them on 8 rows side by side (8 independent cross-lane reductions in flight instead of one
// latency chain); per row the operations and their order are the same.
//
// HF modeling_clip.py:358,360 (layer_norm1/2), :605 pre_layrnorm, :608 post_layernorm, :507 final_layer_norm; eps 1e-5
// (configuration_clip.py).  fp32 statistics, exact two-pass mean / centred variance with the row in registers.
#pragma once
#include "common.hpp"

constexpr int LN_MAXV = 4;  // float4 per lane: D <= 64 * 4 * 4 = 1024

// A lane holds v[i] = columns (i * 64 + lane) * 4 .. + 3 of the row.  NVU > 0: the caller guarantees D == NVU * 256, so
// "column < D" is the compile-time test i < NVU (no per-lane branches); NVU == 0: any D that is a multiple of 4.
template <int NVU>
__device__ __forceinline__ bool ln_has(int i, int lane, int D) {
  if constexpr (NVU > 0) return i < NVU;
  else return (i * 64 + lane) * 4 < D;
}
template <int NVU = 0>
__device__ __forceinline__ float ln_part_sum(const float4 (&v)[LN_MAXV], int D, int lane) {
#pragma clang fp contract(off)
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i)
    if (ln_has<NVU>(i, lane, D)) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  return s;
}
__device__ __forceinline__ float ln_mean(float wave_total, int D) {
#pragma clang fp contract(off)
  return wave_total / (float)D;
}
// centres the row in place and returns this lane's part of the sum of squares
template <int NVU = 0>
__device__ __forceinline__ float ln_center_sq(float4 (&v)[LN_MAXV], float mean, int D, int lane) {
#pragma clang fp contract(off)
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i)
    if (ln_has<NVU>(i, lane, D)) {
      v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
      q += __builtin_fmaf(v[i].x, v[i].x, v[i].y * v[i].y) + __builtin_fmaf(v[i].z, v[i].z, v[i].w * v[i].w);
    }
  return q;
}
__device__ __forceinline__ float ln_rstd(float wave_total_sq, int D, float eps) {
#pragma clang fp contract(off)
  return 1.0f / sqrtf(wave_total_sq / (float)D + eps);
}
template <int NVU = 0>
__device__ __forceinline__ void ln_scale(float4 (&v)[LN_MAXV], float rstd, const float* __restrict__ g,
                                         const float* __restrict__ b, int D, int lane) {
#pragma clang fp contract(off)
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int d = (i * 64 + lane) * 4;
    if (ln_has<NVU>(i, lane, D)) {
      const float4 gv = *(const float4*)(g + d);
      const float4 bv = *(const float4*)(b + d);
      v[i].x = __builtin_fmaf(v[i].x * rstd, gv.x, bv.x);
      v[i].y = __builtin_fmaf(v[i].y * rstd, gv.y, bv.y);
      v[i].z = __builtin_fmaf(v[i].z * rstd, gv.z, bv.z);
      v[i].w = __builtin_fmaf(v[i].w * rstd, gv.w, bv.w);
    }
  }
}
// on return v holds LayerNorm(row) * g + b in fp32
template <int NVU = 0>
__device__ __forceinline__ void ln_row_apply(float4 (&v)[LN_MAXV], const float* __restrict__ g, const float* __restrict__ b,
                                             int D, float eps, int lane) {
  const float mean = ln_mean(wave_sum(ln_part_sum<NVU>(v, D, lane)), D);
  const float rstd = ln_rstd(wave_sum(ln_center_sq<NVU>(v, mean, D, lane)), D, eps);
  ln_scale<NVU>(v, rstd, g, b, D, lane);
}

// the row in the operand dtype OUT (fp32: as it is) to yrow; X2 (16-bit OUT): as a split image of 2 D elements, per 64
// columns hi[64] then lo[64] (GemmArgs::xsplit)
template <int OUT, int NVU = 0, bool X2 = false>
__device__ __forceinline__ void ln_row_store(const float4 (&v)[LN_MAXV], void* yrow, int D, int lane, float& amax) {
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int d = (i * 64 + lane) * 4;
    if (ln_has<NVU>(i, lane, D)) {
      if constexpr (OUT != MCM_PREC_F32 && X2) {
        uint2 hi, lo;
        split2<OUT>(v[i].x, v[i].y, hi.x, lo.x);
        split2<OUT>(v[i].z, v[i].w, hi.y, lo.y);
        sat_track<OUT>(amax, v[i].x, v[i].y);
        sat_track<OUT>(amax, v[i].z, v[i].w);
        uint16_t* dst = (uint16_t*)yrow + split_col(d);
        *(uint2*)dst = hi;
        *(uint2*)(dst + 64) = lo;
      } else if constexpr (OUT != MCM_PREC_F32) {
        uint2 pk;
        pk.x = pack2<OUT>(v[i].x, v[i].y);
        pk.y = pack2<OUT>(v[i].z, v[i].w);
        sat_track<OUT>(amax, v[i].x, v[i].y);
        sat_track<OUT>(amax, v[i].z, v[i].w);
        *(uint2*)((uint16_t*)yrow + d) = pk;
      } else {
        *(float4*)((float*)yrow + d) = v[i];
      }
    }
  }
}

// wave_sum of R independent values: the same butterfly per value (v += shfl_xor(v, 32), 16, ... 1), R exchanges in flight
template <int R>
__device__ __forceinline__ void wave_sum_n(float (&x)[R]) {
#pragma clang fp contract(off)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float t[R];
#pragma unroll
    for (int r = 0; r < R; ++r) t[r] = __shfl_xor(x[r], o, 64);
#pragma unroll
    for (int r = 0; r < R; ++r) x[r] += t[r];
  }
}
