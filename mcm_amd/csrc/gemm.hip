// gemm.hip — Y[M,N] = X[M,K] · W[N,K]^T (+bias, fused epilogues) on gfx950 MFMA.
//
// Replaces every nn.Linear of the CLIP towers (HF modeling_clip.py:309-311 q/k/v as one
// concatenated [3D,D] weight, :333 out_proj, :347-349 fc1/fc2) and the patch-embedding
// conv (:148-154, non-overlapping patches = a GEMM over gathered rows).  X and W are both
// K-contiguous ("B^T input"), so A- and B-fragments are read from LDS the same way.
//
// Structure (v1): 128x128 output tile, 4 waves as 2x2, each wave 64x64 = 4x4 MFMA
// fragments of 16x16; one K-step = 128 bytes of K per row (64 bf16 or 32 fp32), so the
// LDS image, the staging code and the fragment reads are identical for both precisions:
//   bf16: 2 x v_mfma_f32_16x16x32_bf16 per fragment pair and K-step
//   fp32: 8 x v_mfma_f32_16x16x4_f32   (exact fp32 = fmaf chain; the parity arm)
// Global→LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction), two
// stages, next K-step in flight under the MFMAs, one barrier per K-step.
//
// LDS layout (per operand tile, 128 rows x 128 B): rows are paired into 256-B bank rows
// and the 16-B chunk index is XORed with the pair index, so the four 16-lane groups of
// a ds_read_b128 fragment read (16 rows x one chunk) hit 16 distinct 16-B slots:
//   off(r, c) = (r>>1)*256 + ((((r&1)<<3) | ((c ^ (r>>1)) & 7)) << 4)
// LDS-DMA writes lane-linear, so the permutation is applied to each lane's *global*
// source address (guide rule 21: linear dest + inverse-swizzled source + swizzled read).
//
// MFMA operand order is swapped (W fragment as A, X fragment as B) so each lane ends up
// with 4 consecutive output columns of one row: 16-B fp32 / 8-B bf16 epilogue accesses.
#include "common.hpp"

namespace {

constexpr int BM = 128, BN = 128;
constexpr int ROWB = 128;               // bytes of K per row per K-step
constexpr int TILE_BYTES = BM * ROWB;   // 16 KiB per operand per stage
constexpr int STAGE_BYTES = 2 * TILE_BYTES;
constexpr int NSTAGE = 2;

template <int PREC, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ES = prec_esize(PREC);

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

  // XCD-aware bijective remap: hardware places block b on XCD b%8; give each XCD a
  // contiguous run of logical tiles (same X row panel, W walks through its L2).
  const int nbn = (a.N + BN - 1) / BN;
  const int nbm = (a.M + BM - 1) / BM;
  const int nwg = nbn * nbm;
  int lid;
  {
    const int bid = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int m0 = (lid / nbn) * BM;
  const int n0 = (lid % nbn) * BN;

  // ---- staging addresses: 4 LDS-DMA pieces per operand per wave per K-step
  const char* gx[4];
  const char* gw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int blk = i * 4 + wave;             // 1-KiB block of the 16-KiB tile
    const int p = blk * 4 + (lane >> 4);      // row pair 0..63
    const int s = lane & 15;                  // 16-B slot inside the 256-B pair row
    const int row = 2 * p + (s >> 3);
    const int chunk = (s & 7) ^ (p & 7);
    const int mr = min(m0 + row, a.M - 1);
    const int nr = min(n0 + row, a.N - 1);
    gx[i] = (const char*)a.x + ((size_t)mr * a.ldx) * ES + chunk * 16;
    gw[i] = (const char*)a.w + ((size_t)nr * a.K) * ES + chunk * 16;
  }
  auto stage = [&](int st, int kt) {
    char* base = smem + st * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int blk = i * 4 + wave;
      __builtin_amdgcn_global_load_lds((gptr_t)(gx[i] + (size_t)kt * ROWB),
                                       (lptr_t)(base + blk * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(gw[i] + (size_t)kt * ROWB),
                                       (lptr_t)(base + TILE_BYTES + blk * 1024), 16, 0, 0);
    }
  };

  // ---- fragment read offsets (lane part; + f*2048 per 16-row fragment)
  const int wr = wave >> 1, wc = wave & 1;
  const int fr = lane & 15, g = lane >> 4;
  int foff[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
    foff[kk] = (fr >> 1) * 256 + ((((fr & 1) << 3) | (((kk * 4 + g) ^ (fr >> 1)) & 7)) << 4);
  const int xbase = wr * 64 * ROWB;                // rows wr*64.. of the X tile
  const int wbase = TILE_BYTES + wc * 64 * ROWB;   // rows wc*64.. of the W tile

  f32x4_t acc[4][4];  // [fj = n fragment][fi = m fragment]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int nk = (a.K * ES) / ROWB;
  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
    const char* sb = smem + (kt & 1) * STAGE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      uint4 xf[4], wf[4];
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        xf[f] = *(const uint4*)(sb + xbase + f * 2048 + foff[kk]);
        wf[f] = *(const uint4*)(sb + wbase + f * 2048 + foff[kk]);
      }
#pragma unroll
      for (int fj = 0; fj < 4; ++fj)
#pragma unroll
        for (int fi = 0; fi < 4; ++fi) {
          if constexpr (PREC == MCM_PREC_BF16) {
            acc[fj][fi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                __builtin_bit_cast(bf16x8_t, wf[fj]), __builtin_bit_cast(bf16x8_t, xf[fi]),
                acc[fj][fi], 0, 0, 0);
          } else {
            const f32x4_t wv = __builtin_bit_cast(f32x4_t, wf[fj]);
            const f32x4_t xv = __builtin_bit_cast(f32x4_t, xf[fi]);
#pragma unroll
            for (int t = 0; t < 4; ++t)
              acc[fj][fi] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[t], xv[t], acc[fj][fi], 0, 0, 0);
          }
        }
    }
  }

  // ---- epilogue: lane holds Y[m][n..n+3], m = m0+wr*64+fi*16+fr, n = n0+wc*64+fj*16+g*4
#pragma unroll
  for (int fi = 0; fi < 4; ++fi) {
    const int m = m0 + wr * 64 + fi * 16 + fr;
    if (m >= a.M) continue;
    size_t orow;
    const float* prow = nullptr;
    if constexpr (EPI == EPI_PATCH) {
      const int b = m / a.np, p = m - b * a.np;
      orow = (size_t)(b * (a.np + 1) + 1 + p) * a.ldo;
      prow = a.pos + (size_t)(1 + p) * a.N;
    } else {
      orow = (size_t)m * a.ldo;
    }
#pragma unroll
    for (int fj = 0; fj < 4; ++fj) {
      const int n = n0 + wc * 64 + fj * 16 + g * 4;
      if (n >= a.N) continue;
      f32x4_t v = acc[fj][fi];
      if (a.bias) {
        const f32x4_t bv = *(const f32x4_t*)(a.bias + n);
        v += bv;
      }
      if constexpr (EPI == EPI_GELU) {
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = quick_gelu(v[t]);
      }
      if constexpr (EPI == EPI_RESID) {
        f32x4_t* dst = (f32x4_t*)(a.resid + orow + n);
        *dst = *dst + v;
      } else if constexpr (EPI == EPI_PATCH) {
        v += *(const f32x4_t*)(prow + n);
        *(f32x4_t*)((float*)a.out + orow + n) = v;
      } else if constexpr (PREC == MCM_PREC_BF16) {
        uint2 pk;
        pk.x = pack_bf2(v[0], v[1]);
        pk.y = pack_bf2(v[2], v[3]);
        *(uint2*)((uint16_t*)a.out + orow + n) = pk;
      } else {
        *(f32x4_t*)((float*)a.out + orow + n) = v;
      }
    }
  }
}

template <int PREC, int EPI>
hipError_t launch_one(const GemmArgs& a, hipStream_t s) {
  static bool attr_set = false;
  constexpr int lds = NSTAGE * STAGE_BYTES;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_kernel<PREC, EPI>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int nbn = (a.N + BN - 1) / BN, nbm = (a.M + BM - 1) / BM;
  hipLaunchKernelGGL((gemm_kernel<PREC, EPI>), dim3(nbn * nbm), dim3(256), lds, s, a);
  return hipGetLastError();
}

template <int PREC>
hipError_t launch_prec(int epi, const GemmArgs& a, hipStream_t s) {
  switch (epi) {
    case EPI_STORE: return launch_one<PREC, EPI_STORE>(a, s);
    case EPI_GELU: return launch_one<PREC, EPI_GELU>(a, s);
    case EPI_RESID: return launch_one<PREC, EPI_RESID>(a, s);
    case EPI_PATCH: return launch_one<PREC, EPI_PATCH>(a, s);
  }
  return hipErrorInvalidValue;
}

}  // namespace

hipError_t launch_gemm(int prec, int epi, const GemmArgs& a, hipStream_t s) {
  const int es = prec_esize(prec);
  if (a.M <= 0 || a.N <= 0 || a.K <= 0 || (a.K * es) % ROWB || a.N % 16 || (a.ldx * es) % 16 ||
      a.ldo % 4)
    return hipErrorInvalidValue;
  return prec == MCM_PREC_BF16 ? launch_prec<MCM_PREC_BF16>(epi, a, s)
                               : launch_prec<MCM_PREC_F32>(epi, a, s);
}
