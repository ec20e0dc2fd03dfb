// gemm_arms.hpp — the laboratory half of gemm.hip: every GEMM kernel arm that was built, measured and NOT shipped, kept
// compilable and tested (tests/test_gpu_kernels.py runs each against the oracle and bit-for-bit against the shipped
// kernels).  Included by gemm.hip INSIDE its anonymous namespace, and only in the A/B builds: -DMCM_HARNESS
// (libmcm_hip_harness.so, tools/gemm_bench), -DMCM_LN_FOLD / -DMCM_LN_TAIL (make fold / make tail).  libmcm_hip.so does
// not contain a line of this file.  What each arm measured: EXPERIMENTS.md.
//
//   arms::gemm_persist_kernel     persistent 256x128, 3 LDS stages (variants 1 / 2): bound by the L1 -> LDS DMA path
//   arms::gemm_pp_kernel          the shipped ping-pong kernel with its A/B flags: BAL (balanced DMA, variant 7), STAG
//                                 (staggered epilogues, 8), FOLD (LayerNorm fold producer / consumer epilogues), LNT
//                                 (LayerNorm in the tail of the residual GEMMs), the in-loop ablation bits, phase timing
//   arms::gemm_pp32_kernel        the ping-pong loop on 32x32x16 MFMAs (variant 6): same cycles, more power, lower clock
//   arms::gemm_tile_kernel        the 128x128 tile kernel with the LayerNorm-fold consumer epilogue
//   arms::wave_epilogue(_lds)     the epilogues with the fold consumer form and the dbg store bits
// The copies of shipped code in here (tile kernel, ping-pong kernel, the two epilogues) are the round-3 text with every
// flag still in it; the shipped kernels in gemm.hip are the same text with the flags resolved to `false`.
#pragma once

// Ablation bits (GemmArgs::dbg: 1 no refill, 2 no MFMA, 4 no epilogue, 8 folded stores, 16 no 16-bit stores, 64 streamed
// residual rows, 128 cycle stamps, >> 8 de-phasing)
#define DBG(bit) (a.dbg & (bit))
// Ablation bits INSIDE the ping-pong K-loop (1 no LDS-DMA, 2 no MFMA, 32 X panels with the nt hint) cost the loop
// registers and branches even when they are off (-2 ... -10 % on the harness kernel, measured), so they exist only
// in a dedicated build of tools/gemm_bench (-DMCM_HARNESS -DMCM_GEMM_ABLATE), not in libmcm_hip_harness.so.
#if defined(MCM_HARNESS) && defined(MCM_GEMM_ABLATE)
#define ABL(bit) (a.dbg & (bit))
#else
#define ABL(bit) false
#endif

namespace arms {

// epilogue of a (MF*16)x64 wave tile at (mw, nw)
// FOLD: consumer side of the LayerNorm fold (see wave_epilogue_lds); bv then holds b', c and the row statistics are
// read here.  Same arithmetic (fold_apply) as the ping-pong kernel's form: a score does not depend on the kernel.
template <int PREC, int EPI, int MF, bool FOLD = false>
__device__ __forceinline__ void wave_epilogue(const GemmArgs& a, const f32x4_t (&acc)[4][MF],
                                              const f32x4_t (&bv)[4], int mw, int nw, int fr, int g, float& amax) {
  const int n = nw + g * 16;
  if (n >= a.N) return;
  f32x4_t cv[4];
  if constexpr (FOLD) {
#pragma unroll
    for (int fj = 0; fj < 4; ++fj) cv[fj] = *(const f32x4_t*)(a.fold_c + min(n, a.N - 16) + fj * 4);
  }
#pragma unroll
  for (int fi = 0; fi < MF; ++fi) {
    const int m = mw + fi * 16 + fr;
    if (m >= a.M) continue;
    f32x4_t v[4];
    float2 rs = make_float2(1.f, 0.f);
    if constexpr (FOLD) rs = a.fold_rs[m];
#pragma unroll
    for (int fj = 0; fj < 4; ++fj) {
      if constexpr (FOLD) {
#pragma unroll
        for (int t = 0; t < 4; ++t) v[fj][t] = fold_apply(acc[fj][fi][t], rs.x, rs.y, cv[fj][t], bv[fj][t]);
      } else {
        v[fj] = acc[fj][fi] + bv[fj];
      }
      if constexpr (EPI == EPI_GELU) {  // same form in every kernel variant: results must not
#pragma unroll                          // depend on which variant the size heuristic picks
        for (int t = 0; t < 4; ++t)
          v[fj][t] = PREC != MCM_PREC_F32 ? quick_gelu_fast(v[fj][t]) : quick_gelu(v[fj][t]);
      }
    }
    if constexpr (EPI == EPI_RESID) {
      f32x4_t* dst = (f32x4_t*)(a.resid + (size_t)m * a.ldo + n);
      f32x4_t r[4];
#pragma unroll
      for (int fj = 0; fj < 4; ++fj) r[fj] = dst[fj];
#pragma unroll
      for (int fj = 0; fj < 4; ++fj) dst[fj] = r[fj] + v[fj];
    } else if constexpr (EPI == EPI_PATCH) {
      const int b = m / a.np, p = m - b * a.np;
      f32x4_t* dst = (f32x4_t*)((float*)a.out + (size_t)(b * (a.np + 1) + 1 + p) * a.ldo + n);
      const f32x4_t* pr = (const f32x4_t*)(a.pos + (size_t)(1 + p) * a.N + n);
#pragma unroll
      for (int fj = 0; fj < 4; ++fj) dst[fj] = v[fj] + pr[fj];
    } else if constexpr (PREC != MCM_PREC_F32) {
#pragma unroll
      for (int fj = 0; fj < 4; ++fj) {
        sat_track<PREC>(amax, v[fj][0], v[fj][1]);
        sat_track<PREC>(amax, v[fj][2], v[fj][3]);
      }
      uint4* dst = (uint4*)((uint16_t*)a.out + out16_off(a, m, n));
      dst[0] = make_uint4(pack2<PREC>(v[0][0], v[0][1]), pack2<PREC>(v[0][2], v[0][3]),
                          pack2<PREC>(v[1][0], v[1][1]), pack2<PREC>(v[1][2], v[1][3]));
      dst[1] = make_uint4(pack2<PREC>(v[2][0], v[2][1]), pack2<PREC>(v[2][2], v[2][3]),
                          pack2<PREC>(v[3][0], v[3][1]), pack2<PREC>(v[3][2], v[3][3]));
    } else {
      f32x4_t* dst = (f32x4_t*)((float*)a.out + (size_t)m * a.ldo + n);
#pragma unroll
      for (int fj = 0; fj < 4; ++fj) dst[fj] = v[fj];
    }
  }
}

// INTERIOR (16-bit outputs only; the ping-pong kernel): the caller guarantees a full tile and uniform mw / nw;
// rows are then addressed as a uniform base plus one 32-bit lane offset, without bounds checks.
// LayerNorm fold, consumer side (FOLD; ping-pong kernel, 16-bit outputs): the A operand was z = gamma o x instead of
// LayerNorm(x), so the row's normalisation is applied here: out = (acc - mean c_n) rstd + b'_n with c = W gamma and
// b' = b + W beta (both prepared once per weight, launch_fold_prep) and (rstd, mean rstd) per row from the producer's
// moments (launch_fold_stats).  `bv` holds b' and fo.cv holds c for the lane's 16 columns; a lane of the wave keeps
// (rstd, mean rstd) of rows lane and 64 + lane of the wave's 128 and the unit's row is fetched by ds_bpermute.
struct FoldRegs {
  f32x4_t cv[4];
  float rstd[2], mrstd[2];
};
template <int PREC, int EPI, int MF, bool INTERIOR = false, bool FOLD = false>
__device__ __forceinline__ void wave_epilogue_lds(const GemmArgs& a, const f32x4_t (&acc)[4][MF],
                                                  const f32x4_t (&bv)[4], int mw, int nw, int lane,
                                                  char* scratch, float& amax, const FoldRegs* fo = nullptr) {
  const int fr = lane & 15, g = lane >> 4;
  if constexpr (PREC != MCM_PREC_F32 && EPI <= EPI_GELU) {
    // 16-row units ping-pong between the two 2-KiB halves of the window: unit u is converted and
    // written while unit u-1 is read back and stored, so the LDS round trip and the store issue
    // (a 1-KiB store blocks its wave like an LDS-DMA piece does) overlap the next unit's VALU work.
    const int rrow = lane >> 3, c8 = lane & 7;  // read-back: 8 lanes per 128-B row
    const int n = nw + c8 * 8;
    auto write_unit = [&](int u) {
      f32x4_t v[4];
      float rstd = 1.f, mr = 0.f;
      if constexpr (FOLD) {  // row u*16 + fr of the wave's 128: held by lane (u & 3) * 16 + fr, slot u >> 2
        const int src = (((u & 3) << 4) | fr) << 2;
        rstd = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, fo->rstd[u >> 2])));
        mr = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, fo->mrstd[u >> 2])));
      }
#pragma unroll
      for (int fj = 0; fj < 4; ++fj) {
        if constexpr (FOLD) {
#pragma unroll
          for (int t = 0; t < 4; ++t) v[fj][t] = fold_apply(acc[fj][u][t], rstd, mr, fo->cv[fj][t], bv[fj][t]);
        } else {
          v[fj] = acc[fj][u] + bv[fj];
        }
        if constexpr (EPI == EPI_GELU) {
#pragma unroll
          for (int t = 0; t < 4; ++t) v[fj][t] = quick_gelu_fast(v[fj][t]);
        }
        sat_track<PREC>(amax, v[fj][0], v[fj][1]);
        sat_track<PREC>(amax, v[fj][2], v[fj][3]);
      }
      char* w = scratch + (u & 1) * 2048 + fr * 128;
      const int sw = fr & 7;
      *(uint4*)(w + (((g * 2) ^ sw) << 4)) =
          make_uint4(pack2<PREC>(v[0][0], v[0][1]), pack2<PREC>(v[0][2], v[0][3]),
                     pack2<PREC>(v[1][0], v[1][1]), pack2<PREC>(v[1][2], v[1][3]));
      *(uint4*)(w + (((g * 2 + 1) ^ sw) << 4)) =
          make_uint4(pack2<PREC>(v[2][0], v[2][1]), pack2<PREC>(v[2][2], v[2][3]),
                     pack2<PREC>(v[3][0], v[3][1]), pack2<PREC>(v[3][2], v[3][3]));
    };
    auto read_unit = [&](int u, uint4 (&r)[2]) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int row = t * 8 + rrow;
        r[t] = *(const uint4*)(scratch + (u & 1) * 2048 + row * 128 + ((c8 ^ (row & 7)) << 4));
      }
    };
    // head-major outputs (a.hm): the wave's 64 columns are one head's block, whose rows are 128 B apart
    const int ldo_e = MCM_HM(a.hm) ? 64 : a.ldo;
    const int lane_off = rrow * ldo_e + c8 * 8;  // elements
    // INTERIOR: scalar base + one 32-bit lane offset; the base walks down the tile 8 rows per store (two scalar adds
    // per store instead of a 64-bit multiply-add chain and a vector 64-bit add)
    const char* sp = (const char*)a.out + out16_off(a, mw, nw) * 2;
    const size_t sp_step = (size_t)ldo_e * 16;  // bytes per 8 rows
    auto store_unit = [&](int u, const uint4 (&r)[2]) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if constexpr (INTERIOR) {
          if (!DBG(16)) store16_stream_s(sp, (uint32_t)lane_off * 2u, r[t]);
          sp += sp_step;
        } else {
          const int m = mw + u * 16 + t * 8 + rrow;
          if (m < a.M && n < a.N && !DBG(16)) store16_stream((uint16_t*)a.out + out16_off(a, m, n), r[t]);
        }
      }
    };
    write_unit(0);
#pragma unroll
    for (int u = 1; u < MF; ++u) {
      uint4 r[2];
      read_unit(u - 1, r);
      write_unit(u);
      store_unit(u - 1, r);
    }
    {
      uint4 r[2];
      read_unit(MF - 1, r);
      store_unit(MF - 1, r);
    }
  } else {
    // fp32 rows: chunk = 16 rows x 256 B, 16 lanes per row
    const int rrow = lane >> 4, c16 = lane & 15;
    const int n = nw + c16 * 4;
    const bool ncol = n < a.N;
    f32x4_t rnext[4];
    auto row_ptr = [&](int c, int t, bool& ok) -> float* {
      const int m = mw + c * 16 + t * 4 + rrow;
      ok = ncol && m < a.M;
      if constexpr (EPI == EPI_PATCH) {
        const int mm = min(m, a.M - 1), b = mm / a.np, p = mm - b * a.np;
        return (float*)a.out + (size_t)(b * (a.np + 1) + 1 + p) * a.ldo + n;
      } else if constexpr (EPI == EPI_RESID) {
        return a.resid + (size_t)m * a.ldo + n;
      } else {
        return (float*)a.out + (size_t)m * a.ldo + n;
      }
    };
    auto prefetch = [&](int c) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        bool ok;
        const float* ptr;
        if constexpr (EPI == EPI_PATCH) {
          const int m = min(mw + c * 16 + t * 4 + rrow, a.M - 1);
          const int p = m - (m / a.np) * a.np;
          ok = ncol;
          ptr = a.pos + (size_t)(1 + p) * a.N + n;
        } else {
          ptr = row_ptr(c, t, ok);
        }
        rnext[t] = ok ? *(const f32x4_t*)ptr : (f32x4_t){0.f, 0.f, 0.f, 0.f};
      }
    };
    constexpr bool ADD = (EPI == EPI_RESID || EPI == EPI_PATCH);
    if constexpr (ADD) prefetch(0);
#pragma unroll
    for (int c = 0; c < MF; ++c) {
      f32x4_t r[4];
      if constexpr (ADD) {
#pragma unroll
        for (int t = 0; t < 4; ++t) r[t] = rnext[t];
        if (c + 1 < MF) prefetch(c + 1);
      }
      const int sw = fr;  // row = fr inside the chunk
#pragma unroll
      for (int fj = 0; fj < 4; ++fj) {
        f32x4_t v = acc[fj][c] + bv[fj];
        if constexpr (EPI == EPI_GELU) {
#pragma unroll
          for (int t = 0; t < 4; ++t) v[t] = quick_gelu(v[t]);
        }
        *(f32x4_t*)(scratch + fr * 256 + (((g * 4 + fj) ^ sw) << 4)) = v;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int row = t * 4 + rrow;
        f32x4_t v = *(const f32x4_t*)(scratch + row * 256 + ((c16 ^ row) << 4));
        if constexpr (ADD) v += r[t];
        bool ok;
        float* dst = row_ptr(c, t, ok);
        if (ok) *(f32x4_t*)dst = v;  // fp32 / residual rows: streaming them measured no gain
      }
    }
  }
}

template <int PREC, int EPI, bool FOLD = false>
__global__ __launch_bounds__(256, 2) void gemm_tile_kernel(const GemmArgs a) {
  using namespace tile;
  enter_precision_mode<PREC>();
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
    const int nr = min(n0 + (row & 64) + perm_n(row & 63), a.N - 1);
    gx[i] = (const char*)a.x + ((size_t)mr * a.ldx) * ES + chunk * 16;
    gw[i] = (const char*)a.w + ((size_t)nr * a.K) * ES + chunk * 16;
  }
  const uint32_t lds0 = lds_addr(smem);
  auto stage = [&](int st, int kt) {
    const uint32_t base = lds0 + st * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int blk = i * 4 + wave;
      glds16(gx[i] + (size_t)(kt >> a.ksplit) * ROWB, __builtin_amdgcn_readfirstlane(base + blk * 1024));
      glds16(gw[i] + (size_t)kt * ROWB,
             __builtin_amdgcn_readfirstlane(base + TILE_BYTES + blk * 1024));
    }
  };

  const int wr = wave >> 1, wc = wave & 1;
  const int fr = lane & 15, g = lane >> 4;
  const int foff[2] = {frag_off(fr, g, 0), frag_off(fr, g, 1)};
  const int xbase = wr * 64 * ROWB;
  const int wbase = TILE_BYTES + wc * 64 * ROWB;

  f32x4_t acc[4][4];
  zero_acc(acc);
  const int nk = (a.K * ES) / ROWB;
  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
    const char* sb = smem + (kt & 1) * STAGE_BYTES;
    wave_kstep<PREC, 4>(sb + xbase, sb + wbase, foff, acc);
  }
  f32x4_t bv[4];
  load_bias(a, n0 + wc * 64 + g * 16, bv);
  float amax = 0.f;
  wave_epilogue<PREC, EPI, 4, FOLD>(a, acc, bv, m0 + wr * 64, n0 + wc * 64, fr, g, amax);
  sat_report<PREC>(amax, a.sat);
}

// =========================================================================================
// persistent 256x128 kernel, 3-stage LDS-DMA pipeline running across tile boundaries
// =========================================================================================
namespace persist {
constexpr int BM = 256, BN = 128;
constexpr int A_BYTES = BM * ROWB;           // 32 KiB
constexpr int W_BYTES = BN * ROWB;           // 16 KiB
constexpr int STAGE_BYTES = A_BYTES + W_BYTES;
constexpr int NSTAGE = 3;
constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES;  // 144 KiB
constexpr int LOADS_PER_STAGE = 6;               // LDS-DMA instructions per wave per stage
}  // namespace persist

template <int PREC, int EPI, bool COUNT_STORES>
__global__ __launch_bounds__(512, 2) void gemm_persist_kernel(const GemmArgs a) {
  using namespace persist;
  enter_precision_mode<PREC>();
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ES = prec_esize(PREC);
  // store instructions per wave per full tile: 4 rows x (2 x 16 B bf16 | 4 x 16 B fp32)
  constexpr int STORES_PER_EPI = (PREC != MCM_PREC_F32 && EPI <= EPI_GELU) ? 8 : 16;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // 0..7

  // ---- tile schedule: M-tiles are striped over the 8 XCDs (mt = mtl*8 + xcd); the G/8
  // workgroups of an XCD walk that XCD's (mtl, nt) list n-fastest, so concurrently running
  // CUs of one XCD share X row panels through their L2.
  const int nbn = (a.N + BN - 1) / BN;
  const int nbm = (a.M + BM - 1) / BM;
  const int G8 = gridDim.x >> 3;
  const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
  const int nmt_x = (nbm - xcd + 7) >> 3;
  const int ntl_x = nmt_x * nbn;
  const int ntl = jx < ntl_x ? (ntl_x - jx + G8 - 1) / G8 : 0;
  if (ntl == 0) return;
  const int nk = (a.K * ES) / ROWB;
  const int total = ntl * nk;

  // ---- LDS-DMA source geometry of this lane (constant): piece i covers tile rows
  // i*64 + r0, 16-B chunk `chunk` (swizzled)
  const int r0 = wave * 8 + (lane >> 4) * 2 + ((lane & 15) >> 3);
  const int chunk = (lane & 7) ^ (((wave & 1) << 2) | (lane >> 4));
  const char* gx[4];
  const char* gw[2];
  int ji = 0, kti = 0;  // issue cursor: tile index (of this workgroup) and K-step
  auto set_issue_tile = [&](int i) {
    int mtl, nt;
    tile_of(jx + i * G8, nmt_x, nbn, a.gn, mtl, nt);
    const int m0 = (mtl * 8 + xcd) * BM, n0 = nt * BN;
#pragma unroll
    for (int p = 0; p < 4; ++p)
      gx[p] = (const char*)a.x + ((size_t)min(m0 + p * 64 + r0, a.M - 1) * a.ldx) * ES + chunk * 16;
#pragma unroll
    for (int p = 0; p < 2; ++p)
      gw[p] = (const char*)a.w + ((size_t)min(n0 + p * 64 + perm_n(r0), a.N - 1) * a.K) * ES + chunk * 16;
  };
  const uint32_t lds0 = lds_addr(smem);
  auto issue = [&](int st) {
    const uint32_t base = lds0 + st * STAGE_BYTES;
    const size_t ko = (size_t)kti * ROWB;
#pragma unroll
    for (int p = 0; p < 4; ++p)
      glds16(gx[p] + ko, __builtin_amdgcn_readfirstlane(base + (p * 8 + wave) * 1024));
#pragma unroll
    for (int p = 0; p < 2; ++p)
      glds16(gw[p] + ko, __builtin_amdgcn_readfirstlane(base + A_BYTES + (p * 8 + wave) * 1024));
    if (++kti == nk) {
      kti = 0;
      if (++ji < ntl) set_issue_tile(ji);
    }
  };

  const int wr = wave >> 1, wc = wave & 1;
  const int fr = lane & 15, g = lane >> 4;
  const int foff[2] = {frag_off(fr, g, 0), frag_off(fr, g, 1)};
  const int xbase = wr * 64 * ROWB;
  const int wbase = A_BYTES + wc * 64 * ROWB;

  f32x4_t acc[4][4];
  zero_acc(acc);
  float amax = 0.f;

  set_issue_tile(0);
  int issued = 0;
  for (; issued < 2 && issued < total; ++issued) issue(issued);
  int st = 0;          // LDS stage of step s
  int ist = 2;         // LDS stage the next issue goes to
  int jc = 0, ktc = 0; // compute cursor
  int since_epi = 1000;
  int cm0, cn0;        // origin of the tile being computed
  {
    int mtl, nt;
    tile_of(jx, nmt_x, nbn, a.gn, mtl, nt);
    cm0 = (mtl * 8 + xcd) * BM;
    cn0 = nt * BN;
  }
  f32x4_t bv[4];
#pragma unroll
  for (int fj = 0; fj < 4; ++fj) bv[fj] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const bool counted = nk >= 3;  // short K: every wait is vmcnt(0)
  for (int s = 0; s < total; ++s) {
    // Stage s must have landed; everything issued after it may stay in flight.  VMEM issue
    // order around a tile boundary (tile ends at step e):
    //   step e  : [DMA stage e+2] ........ [epilogue stores, E per wave]
    //   step e+1: [bias loads, 4] [DMA stage e+3]
    //   step e+2: [DMA stage e+4]
    // so the ops younger than the awaited stage are 6+E at e+1, 10+E at e+2, else 6.
    if (counted && issued > s + 1) {
      if (COUNT_STORES && since_epi == 1) wait_vmcnt<LOADS_PER_STAGE + STORES_PER_EPI>();
      else if (COUNT_STORES && since_epi == 2) wait_vmcnt<LOADS_PER_STAGE + STORES_PER_EPI + 4>();
      else wait_vmcnt<LOADS_PER_STAGE>();
    } else {
      wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (ktc == 0 && a.bias) load_bias_async(a, cn0 + wc * 64 + g * 16, bv);
    if (issued < total) {  // refill the stage that step s-1 just finished reading
      if (!ABL(1)) issue(ist);
      ist = ist == NSTAGE - 1 ? 0 : ist + 1;
      ++issued;
    }
    const char* sb = smem + st * STAGE_BYTES;
    if (!ABL(2)) wave_kstep<PREC, 4>(sb + xbase, sb + wbase, foff, acc);
    st = st == NSTAGE - 1 ? 0 : st + 1;
    ++since_epi;
    if (++ktc == nk) {
      if (!counted) wait_vmcnt<0>();  // bias was issued in this very tile's first step
#pragma unroll
      for (int fj = 0; fj < 4; ++fj) asm volatile("" : "+v"(bv[fj]));
      if (!DBG(4)) wave_epilogue<PREC, EPI, 4>(a, acc, bv, cm0 + wr * 64, cn0 + wc * 64, fr, g, amax);
      zero_acc(acc);
      // only a full tile issues exactly STORES_PER_EPI stores per wave; ragged tiles fall back
      // to waiting for the stores as well
      since_epi = (cm0 + BM <= a.M && cn0 + BN <= a.N) ? 0 : 1000;
      ktc = 0;
      if (++jc < ntl) {
        int mtl, nt;
        tile_of(jx + jc * G8, nmt_x, nbn, a.gn, mtl, nt);
        cm0 = (mtl * 8 + xcd) * BM;
        cn0 = nt * BN;
      }
    }
  }
  sat_report<PREC>(amax, a.sat);
}

__device__ __forceinline__ void gload16_nt(f32x4_t& dst, const void* sbase, uint32_t voff) {
  asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void gstore16_nt(const void* sbase, uint32_t voff, const f32x4_t& v) {
  asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}

// LayerNorm fold, producer side (EPI_RESID in the ping-pong kernel): wave_epilogue_f32_interior<EPI_RESID> plus, for
// every new residual row segment, (1) z = gamma o x in the operand dtype to fold_z (what the next GEMM multiplies) and
// (2) the segment's moments — its sum and its sum of squares about its own mean, 64 columns per wave — to
// fold_part[column / 64][row].  After the LDS bounce the 16 lanes of a DPP row hold the 64 columns of one row, so a
// moment is 4 DPP adds; the 16 (chunk, row-group) results of 64 rows are parked in the lane whose index equals their
// number and leave as ONE 512-byte store.  All global accesses from asm, counted like the plain form: at the wait of
// chunk c the queue holds [loads c] [stores c-1: 4 x + 4 z (+ 1 moments)] [loads c+1].
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gstore8(const void* sbase, uint32_t voff, const u32x2_t& v) {
  asm volatile("global_store_dwordx2 %0, %1, %2\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}
template <int PREC, int MF>
__device__ __forceinline__ void wave_epilogue_resid_fold(const GemmArgs& a, const f32x4_t (&acc)[4][MF],
                                                         int mw, int nw, int lane, char* scratch, float& amax) {
  static_assert(MF == 8, "the wait counts below are written out for 8 chunks");
  const int fr = lane & 15, g = lane >> 4;
  const int rrow = lane >> 4, c16 = lane & 15;
  const uint32_t voff = (uint32_t)(rrow * a.ldo + c16 * 4) * 4u;  // bytes, fp32 rows
  const uint32_t zoff = voff >> 1;                                 // bytes, 16-bit rows of the same stride
  const char* base = (const char*)a.resid + ((size_t)mw * a.ldo + nw) * 4;
  const char* zbase = (const char*)a.fold_z + ((size_t)mw * a.ldo + nw) * 2;
  auto rowbase = [&](int c, int t) { return base + (size_t)(c * 16 + t * 4) * a.ldo * 4; };
  auto zrowbase = [&](int c, int t) { return zbase + (size_t)(c * 16 + t * 4) * a.ldo * 2; };
  const char* pbase = (const char*)(a.fold_part + (size_t)(nw >> 6) * a.M + mw);
  const uint32_t poff = (uint32_t)(((c16 >> 2) * 16 + (c16 & 3) * 4 + rrow) * 8);
  // gamma and the bias of the lane's 4 columns AFTER the bounce (4 + 4 registers, loaded here: nothing of this
  // epilogue is live across the K loop); (acc + b) + resid as in every other form
  auto xload = [&](f32x4_t& dst, const char* rb) {
    if (DBG(64)) return gload16_nt(dst, rb, voff);
    gload16(dst, rb, voff);
  };
  auto xstore = [&](const char* rb, const f32x4_t& val) {
    if (DBG(64)) return gstore16_nt(rb, voff, val);
    gstore16(rb, voff, val);
  };
  f32x4_t gam, bia;
  gload16(gam, a.fold_g + nw, (uint32_t)c16 * 16u);
  gload16(bia, a.bias + nw, (uint32_t)c16 * 16u);
  f32x4_t buf[2][4];
#pragma unroll
  for (int t = 0; t < 4; ++t) xload(buf[0][t], rowbase(0, t));
  float ms = 0.f, mq = 0.f;
#pragma unroll
  for (int c = 0; c < MF; ++c) {
    if (c + 1 < MF) {
#pragma unroll
      for (int t = 0; t < 4; ++t) xload(buf[(c + 1) & 1][t], rowbase(c + 1, t));
    }
#pragma unroll
    for (int fj = 0; fj < 4; ++fj) *(f32x4_t*)(scratch + fr * 256 + (((g * 4 + fj) ^ fr) << 4)) = acc[fj][c];
    f32x4_t v[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int row = t * 4 + rrow;
      v[t] = *(const f32x4_t*)(scratch + row * 256 + ((c16 ^ row) << 4));
    }
    if (c == 0) {
      wait_vmcnt_pin<4>(buf[0]);
      asm volatile("" : "+v"(gam), "+v"(bia));
    } else if (c + 1 == MF) {
      wait_vmcnt_pin<8>(buf[c & 1]);
    } else if (c == 4) {
      wait_vmcnt_pin<13>(buf[c & 1]);  // stores of chunk 3 include the first moments store
    } else {
      wait_vmcnt_pin<12>(buf[c & 1]);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) v[t] = (v[t] + bia) + buf[c & 1][t];
#pragma unroll
    for (int t = 0; t < 4; ++t) xstore(rowbase(c, t), v[t]);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const f32x4_t z = fold_scale(v[t], gam);
      sat_track<PREC>(amax, z[0], z[1]);
      sat_track<PREC>(amax, z[2], z[3]);
      if constexpr (PREC == MCM_PREC_F16) asm volatile("" : "+v"(amax));  // here, not 128 live values later
      const u32x2_t zz = {pack2<PREC>(z[0], z[1]), pack2<PREC>(z[2], z[3])};
      gstore8(zrowbase(c, t), zoff, zz);
      float sm, sq;
      slot_moments(v[t], sm, sq);
      const bool mine = c16 == (c & 3) * 4 + t;
      ms = mine ? sm : ms;
      mq = mine ? sq : mq;
    }
    if ((c & 3) == 3) {
      const u32x2_t pm = {__builtin_bit_cast(uint32_t, ms), __builtin_bit_cast(uint32_t, mq)};
      gstore8(pbase + (size_t)(c >> 2) * 64 * 8, poff, pm);
    }
  }
}

// the same with the non-temporal hint (harness A/B, dbg bit 32: X pieces streamed so that W stays L2-resident)
__device__ __forceinline__ void glds16s_nt(const void* sbase, uint32_t voff, uint32_t lds_base) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_base)
      : "memory");
}

// harness-only phase timing of the ping-pong kernel: s_memtime deltas accumulated in scalar registers (stamps
// only where the wave has to drain lgkmcnt anyway), split into mid-tile steps and steps that carry an epilogue;
// written out once at the end (a.pos = uint32 buffer [block][wave][2][5]: 4 sums + step count)
#ifdef MCM_GEMM_TRACE
#define PPT_INIT()                                  \
  uint32_t ppt_sum[2][5] = {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}}; \
  uint64_t ppt_prev = __builtin_amdgcn_s_memtime(); \
  int ppt_set = 0
#define PPT(k)                                                    \
  do {                                                            \
    if (DBG(128)) {                                               \
      if ((k) == 0) ppt_set = pend ? 1 : 0;                       \
      const uint64_t t = __builtin_amdgcn_s_memtime();            \
      ppt_sum[ppt_set][k] += (uint32_t)(t - ppt_prev);            \
      ppt_prev = t;                                               \
      if ((k) == 3) ppt_sum[ppt_set][4] += 1;                     \
    }                                                             \
  } while (0)
#define PPT_DUMP()                                                                                      \
  do {                                                                                                  \
    if (DBG(128) && lane == 0) {                                                                        \
      uint32_t* o = (uint32_t*)a.pos + ((size_t)blockIdx.x * 8 + wave) * 10;                            \
      for (int i = 0; i < 2; ++i)                                                                       \
        for (int j = 0; j < 5; ++j) o[i * 5 + j] = ppt_sum[i][j];                                       \
    }                                                                                                   \
  } while (0)
#else
#define PPT_INIT()
#define PPT(k)
#define PPT_DUMP()
#endif

// =========================================================================================
// LayerNorm in the tail (LNT; EPI_RESID, 16-bit operand modes).  A residual GEMM (out-proj, fc2) is always followed by
// the LayerNorm of the rows it has just updated, and its persistent grid always ends ragged: 1 182 tiles on 256
// workgroups are 4.6 rounds, so in the last round 38 % of the CUs have nothing left to do for a whole tile time.
// With LNT the kernel does not end there: a wave that has run out of tiles draws tickets — 32 rows each, in the order
// the row tiles were walked — waits until every tile of the ticket's row tile has been PUBLISHED, and normalises those
// rows (ln_row.hpp: the LayerNorm kernel's arithmetic, bit for bit) into the next GEMM's operand buffer.  The separate
// LayerNorm launch, its 310-MB read in a low-occupancy-free interval of its own, and the ragged tail disappear together.
//
// Coherence.  Everything a ticket touches lives in ONE XCD's L2: row tiles are dealt to XCDs (row tile = mt * 8 + xcd,
// xcd = blockIdx & 7 — all workgroups with the same blockIdx & 7 share an XCD; mcm_api.hip verifies that on the device
// before it ever sets ln_y), a row tile's counters sit in that XCD's region of ln_state, and only waves of that XCD
// normalise its rows.  Publication: a wave's epilogue stores are complete when the vmcnt(0) that ends its next compute
// phase (or follows the last epilogue) has passed; then lane 0 adds 1 to the row tile's counter with an L2 atomic.
// A counter reaches N/256 x 8 (tiles x waves) when the whole 256 x N block is in L2.  The consumer polls with a
// returning L2 atomic (add 0), drops its CU's L1 (buffer_inv sc1) and reads the rows.  All atomics are inline asm
// without scope bits: performed in the XCD's own L2, invisible to hipcc's waitcnt pass.
// Deadlock-free: every workgroup of the grid is resident (one per CU) and a tile's producers never wait for anybody.
// ln_state (per XCD region of ln_rs words): [ln_cap8] counters, tickets drawn, workgroups finished, timeouts.  The
// last workgroup of an XCD to finish zeroes the region's first three parts: the state is all zero between launches
// (no memset launch, no launch parity: a captured graph replays it unchanged).
// =========================================================================================
__device__ __forceinline__ void l2_atomic_add(const void* sbase, uint32_t voff, uint32_t val) {
  asm volatile("global_atomic_add %0, %1, %2" ::"v"(voff), "v"(val), "s"(sbase) : "memory");
}
__device__ __forceinline__ uint32_t l2_atomic_add_ret(const void* sbase, uint32_t voff, uint32_t val) {
  uint32_t r;
  asm volatile("global_atomic_add %0, %1, %2, %3 sc0\n\ts_waitcnt vmcnt(0)"
               : "=&v"(r)
               : "v"(voff), "v"(val), "s"(sbase)
               : "memory");
  return r;
}
// one lane of the wave performs the atomic; every lane gets the value
__device__ __forceinline__ uint32_t wave_l2_add_ret(const void* sbase, uint32_t off, uint32_t val, int lane) {
  uint32_t r = 0;
  if (lane == 0) r = l2_atomic_add_ret(sbase, off, val);
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
}
template <int PREC, int NVU>
__device__ __forceinline__ void ln_tail_nv(const GemmArgs& a, int xcd, int nmt_x, int lane, float& amax) {
  const unsigned int* reg = a.ln_state + (size_t)xcd * a.ln_rs;  // this XCD's region
  const uint32_t need = (uint32_t)(a.N / 256) * 8u;
  const uint32_t ntick = (uint32_t)nmt_x * 8u;
  constexpr int D = NVU * 256;
  for (;;) {
    const uint32_t t = wave_l2_add_ret(reg, (uint32_t)a.ln_cap8 * 4u, 1u, lane);
    if (t >= ntick) break;
    const int mtl = (int)(t >> 3), sub = (int)(t & 7);
    const int mt = a.rev ? nmt_x - 1 - mtl : mtl;
    int spins = 0;
    while (wave_l2_add_ret(reg, (uint32_t)mt * 4u, 0u, lane) < need) {
      __builtin_amdgcn_s_sleep(32);
      if (++spins > (1 << 20)) {  // never in a correct run: count it and go on (wrong rows beat a hung box)
        if (lane == 0) l2_atomic_add(reg, (uint32_t)(a.ln_cap8 + 2) * 4u, 1u);
        break;
      }
    }
    asm volatile("buffer_inv sc1" ::: "memory");
    const size_t row0 = ((size_t)mt * 8 + xcd) * 256 + (size_t)sub * 32;
#pragma unroll 1
    for (int r0 = 0; r0 < 32; r0 += 8) {  // 8 rows side by side: 24 - 32 loads in flight, 8 reductions per exchange
      float4 v[8][LN_MAXV];
      float acc[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float* xr = a.resid + (row0 + r0 + r) * (size_t)D;
#pragma unroll
        for (int i = 0; i < NVU; ++i) {
          typedef float f4_t __attribute__((ext_vector_type(4)));
          const f4_t q = __builtin_nontemporal_load((const f4_t*)(xr + (i * 64 + lane) * 4));
          v[r][i] = make_float4(q.x, q.y, q.z, q.w);
        }
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) acc[r] = ln_part_sum<NVU>(v[r], D, lane);
      wave_sum_n<8>(acc);
#pragma unroll
      for (int r = 0; r < 8; ++r) acc[r] = ln_center_sq<NVU>(v[r], ln_mean(acc[r], D), D, lane);
      wave_sum_n<8>(acc);
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        ln_scale<NVU>(v[r], ln_rstd(acc[r], D, a.ln_eps), a.ln_g, a.ln_b, D, lane);
        ln_row_store<PREC, NVU>(v[r], (uint16_t*)a.ln_y + (row0 + r0 + r) * (size_t)D, D, lane, amax);
      }
    }
  }
}
template <int PREC>
__device__ __forceinline__ void ln_tail(const GemmArgs& a, int xcd, int nmt_x, int lane, float& amax) {
  if (a.N == 768) ln_tail_nv<PREC, 3>(a, xcd, nmt_x, lane, amax);  // the widths of the CLIP vision towers
  else ln_tail_nv<PREC, 4>(a, xcd, nmt_x, lane, amax);             // (launch_one admits 768 and 1024 only)
}
// end of an LNT kernel: the last workgroup of this XCD to get here zeroes the XCD's counters, tickets and finish count
__device__ __forceinline__ void ln_finish(const GemmArgs& a, int xcd, char* smem) {
  const unsigned int* reg = a.ln_state + (size_t)xcd * a.ln_rs;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int* flag = (int*)smem;
  if (threadIdx.x == 0) *flag = l2_atomic_add_ret(reg, (uint32_t)(a.ln_cap8 + 1) * 4u, 1u) == (gridDim.x >> 3) - 1;
  __syncthreads();
  if (*flag) {
    const uint32_t zero = 0u;
    for (int i = threadIdx.x; i < a.ln_cap8 + 2; i += blockDim.x)
      asm volatile("global_store_dword %0, %1, %2" ::"v"((uint32_t)i * 4u), "v"(zero), "s"(reg) : "memory");
  }
}

// =========================================================================================
// LayerNorm by the row panel's cluster (LNC; round 6; EPI_RESID, 16-bit operand modes, N = 768 / 1024).
// VERDICT r5 item 2 asked for the one design that removes LayerNorm's own re-read of the residual stream: a FULL-ROW
// (BN = N) residual-GEMM epilogue that emits the LayerNorm output from registers.  A 256 x 768 register tile does not
// exist on this part (384 KB of accumulators per CU; a 64- or 128-row full-row tile re-stages all of W per 64 / 128
// rows and no longer double-buffers: EXPERIMENTS.md R6.2), but the full-row TILE does — spread over the N / 256
// workgroups that hold the tiles of one 256-row panel.  They sit on ONE XCD (row tile = mt * 8 + xcd) and are walked
// n-fastest, i.e. by consecutive workgroups of the same round, so they finish within a few microseconds of each other.
// The K loop is untouched; the epilogue of a wave (128 rows x 64 columns) becomes
//   A  as the plain residual form: bounce, (acc + bias) + resid, store x — and the new values STAY in the accumulator
//      registers (bounced layout); per row the 64-column moments (sum, centred sum of squares: slot_moments, the fold's)
//      go to fold_part[column / 64][row];
//   B  publish: vmcnt(0), one L2 atomic on the counter of (row panel, upper / lower 128 rows) — the two halves are
//      waves 0-3 / 4-7 of every workgroup, whose epilogues run in DIFFERENT phases of the ping-pong schedule, so a
//      half waits only for the same half of its partner workgroups (12 waves at N = 768) — then poll that counter
//      (bounded: a lost partner costs wrong rows and a counted timeout, not a hung GPU), buffer_inv sc1;
//   C  lane L merges the N / 64 slot moments of rows 2L, 2L + 1 slot by slot (Chan et al.) into (mean, rstd) and parks
//      them in the wave's LDS window;
//   D  LayerNorm from registers: ((v - mean) rstd) gamma + beta (ln_row.hpp's ln_scale arithmetic), packed, stored.
// Nothing re-reads x; 23 LayerNorm launches disappear.  Deadlock-free: every workgroup of the grid is resident, a wave
// publishes BEFORE it waits, workgroups take their tiles in list order, and a wave's partners hold tiles at most two list
// positions away — by induction over the list the earliest waiting tile's partners are never blocked before they publish.
// Statistics: slot-wise two-pass moments combined by Chan's formula — as accurate as the LayerNorm kernel's two-pass
// row statistics, not bit-identical to them (the summation order differs): an A/B arm, compared at fp32 round-off.
// ln_state per XCD: [ln_cap8] counters (index row-panel * 2 + half), +1 finished workgroups, +2 timeouts.
// =========================================================================================
// Kernel arguments that only the LNC epilogue reads, fetched from the kernarg segment INSIDE the epilogue (scalar loads
// from inline asm): as ordinary uses of `a` hipcc loads them at kernel entry and keeps 20 more scalar registers live across
// the K loop, whose spills to VGPR lanes push the loop's fragment offsets to scratch — and a scratch reload in the compute
// phase comes with a vmcnt(0) that drains the LDS-DMA stream (ISA-audited: tests/test_isa_audit.py).
template <typename T, int OFF>
__device__ __forceinline__ T karg() {
  static_assert(sizeof(T) == 4 || sizeof(T) == 8, "a scalar or a pointer");
  const void* kp = (const void*)__builtin_amdgcn_kernarg_segment_ptr();
  if constexpr (sizeof(T) == 8) {
    uint64_t v;
    asm volatile("s_load_dwordx2 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(kp), "n"(OFF) : "memory");
    return __builtin_bit_cast(T, v);
  } else {
    uint32_t v;
    asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(kp), "n"(OFF) : "memory");
    return __builtin_bit_cast(T, v);
  }
}
#define LNC_ARG(field) karg<decltype(GemmArgs::field), (int)offsetof(GemmArgs, field)>()
struct LncArgs {   // the epilogue's own copy of what it needs of GemmArgs
  const float *bias, *ln_g, *ln_b;
  float* resid;
  void* ln_y;
  float2* fold_part;
  float ln_eps;
  int M, N, ldo, ln_cap8;
};
__device__ __forceinline__ LncArgs lnc_args() {
  LncArgs r;
  r.bias = LNC_ARG(bias); r.ln_g = LNC_ARG(ln_g); r.ln_b = LNC_ARG(ln_b); r.resid = LNC_ARG(resid); r.ln_y = LNC_ARG(ln_y);
  r.fold_part = LNC_ARG(fold_part); r.ln_eps = LNC_ARG(ln_eps); r.M = LNC_ARG(M); r.N = LNC_ARG(N); r.ldo = LNC_ARG(ldo);
  r.ln_cap8 = LNC_ARG(ln_cap8);
  return r;
}
template <int PREC, int NS>
__device__ __forceinline__ void lnc_row_stats(const LncArgs& a, int mw, int lane, char* scratch) {
  // rows mw + 2 lane, + 1: the (sum, m2) pairs of the N / 64 slots, 16 bytes per lane and slot, merged slot by slot in slot
  // order (Chan et al.: n, mean, M2 of the union of two sets) — four slots in flight at a time: the 128 accumulator registers
  // are live, a row's 16 slot records at once would spill
  const char* pb = (const char*)(a.fold_part + mw);
  float n = 0.f, m0 = 0.f, q0 = 0.f, m1 = 0.f, q1 = 0.f;
#pragma unroll
  for (int j0 = 0; j0 < NS; j0 += 4) {
    f32x4_t p[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) gload16(p[j], pb + (size_t)(j0 + j) * a.M * 8, (uint32_t)lane * 16u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      asm volatile("" : "+v"(p[j]));
      const float w = 64.0f / (n + 64.0f);          // weight of the new slot in the union
      const float d0 = p[j][0] * (1.0f / 64.0f) - m0, d1 = p[j][2] * (1.0f / 64.0f) - m1;
      q0 += p[j][1] + d0 * d0 * (n * w);
      q1 += p[j][3] + d1 * d1 * (n * w);
      m0 += d0 * w;
      m1 += d1 * w;
      n += 64.0f;
    }
  }
  const f32x4_t st = {m0, 1.0f / sqrtf(q0 / n + a.ln_eps), m1, 1.0f / sqrtf(q1 / n + a.ln_eps)};
  *(f32x4_t*)(scratch + lane * 16) = st;   // row r of the wave's 128: (mean, rstd) at scratch + 8 r
}
template <int PREC, int MF>
__device__ __forceinline__ void wave_epilogue_resid_lnc(f32x4_t (&acc)[4][MF], int mw, int nw, int lane,
                                                        char* scratch, float& amax, const unsigned int* reg, uint32_t ctr_off,
                                                        uint32_t need) {
  static_assert(MF == 8, "the wait counts below are written out for 8 chunks");
  const LncArgs a = lnc_args();
  const int fr = lane & 15, g = lane >> 4;
  const int rrow = lane >> 4, c16 = lane & 15;
  const uint32_t voff = (uint32_t)(rrow * a.ldo + c16 * 4) * 4u;  // bytes, fp32 rows
  // rows are visited in order, 4 at a time: a load pointer and a store pointer walk down the tile by one scalar add each
  // (per-row-group bases computed up front cost 64 scalar registers and push loop state into VGPR lanes: gemm.hip)
  const size_t step = (size_t)a.ldo * 16;  // bytes per 4 fp32 rows
  const char* lp = (const char*)a.resid + ((size_t)mw * a.ldo + nw) * 4;
  const char* sp = lp;
  const char* pbase = (const char*)(a.fold_part + (size_t)(nw >> 6) * a.M + mw);
  const uint32_t poff = (uint32_t)(((c16 >> 2) * 16 + (c16 & 3) * 4 + rrow) * 8);
  // ---- A: the plain residual form; the new rows stay in `acc` (bounced layout: acc[t][c] = rows c*16 + t*4 + rrow,
  // columns c16*4 .. +3), their slot moments leave for fold_part.  Queue at the wait of chunk c, oldest first:
  // [loads c] [stores c-1: 4 x (+ 1 moments after chunks 3 and 7)] [loads c+1]
  f32x4_t bia;
  gload16(bia, a.bias + nw, (uint32_t)c16 * 16u);
  f32x4_t buf[2][4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    gload16(buf[0][t], lp, voff);
    lp += step;
  }
  float ms = 0.f, mq = 0.f;
#pragma unroll
  for (int c = 0; c < MF; ++c) {
    if (c + 1 < MF) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        gload16(buf[(c + 1) & 1][t], lp, voff);
        lp += step;
      }
    }
#pragma unroll
    for (int fj = 0; fj < 4; ++fj) *(f32x4_t*)(scratch + fr * 256 + (((g * 4 + fj) ^ fr) << 4)) = acc[fj][c];
    f32x4_t v[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int row = t * 4 + rrow;
      v[t] = *(const f32x4_t*)(scratch + row * 256 + ((c16 ^ row) << 4));
    }
    if (c == 0) {
      wait_vmcnt_pin<4>(buf[0]);
      asm volatile("" : "+v"(bia));
    } else if (c + 1 == MF) {
      wait_vmcnt_pin<4>(buf[c & 1]);
    } else if (c == 4) {
      wait_vmcnt_pin<9>(buf[c & 1]);   // stores of chunk 3 include the first moments store
    } else {
      wait_vmcnt_pin<8>(buf[c & 1]);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) v[t] = (v[t] + bia) + buf[c & 1][t];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      gstore16(sp, voff, v[t]);
      sp += step;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float sm, sq;
      slot_moments(v[t], sm, sq);
      const bool mine = c16 == (c & 3) * 4 + t;
      ms = mine ? sm : ms;
      mq = mine ? sq : mq;
      acc[t][c] = v[t];
    }
    if ((c & 3) == 3) {
      const u32x2_t pm = {__builtin_bit_cast(uint32_t, ms), __builtin_bit_cast(uint32_t, mq)};
      gstore8(pbase + (size_t)(c >> 2) * 64 * 8, poff, pm);
    }
  }
  // ---- B: publish this wave's moments, wait for the rest of the row panel's half
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) l2_atomic_add(reg, ctr_off, 1u);
  {
    int spins = 0;
    while (wave_l2_add_ret(reg, ctr_off, 0u, lane) < need) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1 << 19)) {  // never in a correct run: count it and go on (wrong rows beat a hung box)
        if (lane == 0) l2_atomic_add(reg, (uint32_t)(a.ln_cap8 + 2) * 4u, 1u);
        break;
      }
    }
  }
  asm volatile("buffer_inv sc1" ::: "memory");
  // ---- C: (mean, rstd) of the wave's 128 rows into its LDS window; gamma / beta of the lane's 4 columns.  Every lane-derived
  // address from here on comes from a fresh opaque copy of the lane id: computed early (hipcc would) they are live across
  // pass A, where 128 accumulators + 48 row registers leave no room
  int l2 = lane;
  asm volatile("" : "+v"(l2));
  const int rrow2 = l2 >> 4, c162 = l2 & 15;
  f32x4_t gam, bet;
  gload16(gam, a.ln_g + nw, (uint32_t)c162 * 16u);
  gload16(bet, a.ln_b + nw, (uint32_t)c162 * 16u);
  if (a.N == 768) lnc_row_stats<PREC, 12>(a, mw, l2, scratch);
  else lnc_row_stats<PREC, 16>(a, mw, l2, scratch);   // (route admits 768 and 1024 only)
  asm volatile("" : "+v"(gam), "+v"(bet));               // (covered by lnc_row_stats' vmcnt(0): they are older)

  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  // ---- D: LayerNorm of the rows in registers (ln_row.hpp ln_center_sq / ln_scale arithmetic), packed, stored
  const char* yp = (const char*)a.ln_y + ((size_t)mw * a.ldo + nw) * 2;
  const size_t ystep = (size_t)a.ldo * 8;   // bytes per 4 16-bit rows
  const uint32_t yoff = (uint32_t)(rrow2 * a.ldo + c162 * 4) * 2u;  // bytes, 16-bit rows
#pragma unroll
  for (int c = 0; c < MF; ++c) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      typedef float f32x2_t __attribute__((ext_vector_type(2)));
      const f32x2_t st = *(const f32x2_t*)(scratch + (c * 16 + t * 4 + rrow2) * 8);
      f32x4_t y;
      {
#pragma clang fp contract(off)
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = __builtin_fmaf((acc[t][c][e] - st[0]) * st[1], gam[e], bet[e]);
      }
      sat_track<PREC>(amax, y[0], y[1]);
      sat_track<PREC>(amax, y[2], y[3]);
      if constexpr (PREC == MCM_PREC_F16) asm volatile("" : "+v"(amax));
      const u32x2_t yy = {pack2<PREC>(y[0], y[1]), pack2<PREC>(y[2], y[3])};
      gstore8(yp, yoff, yy);
      yp += ystep;
    }
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");   // (vmcnt is a 6-bit counter: never more than 28 stores in flight)
    __builtin_amdgcn_sched_barrier(0);                   // (chunk by chunk: hoisting all 32 LDS reads of (mean, rstd) costs 64 registers)
  }
}

// BAL (balanced DMA): waves 0-3 stage their X half and W rows 0-127, waves 4-7 their X half and W rows 128-255 —
// 8 + 8 pieces per step instead of 12 + 4.  The W pieces of waves 4-7 are issued FIRST in their memory phase and
// waited for at its END (vmcnt <= their 4 X pieces), one barrier before waves 0-3 read them; the stage they go to
// was last read (W fragments, by these very waves) a whole step earlier, so no ring of three is needed.
// STAG (staggered epilogues): waves 0-3 run the epilogue of a finished tile BEFORE the barrier that ends the phase
// in which waves 4-7 still compute that tile's last K-step, waves 4-7 theirs one phase later, under the first compute
// phase of waves 0-3 on the next tile: each group's stores and conversions run beside the other group's MFMAs.
// FOLD (LayerNorm fold, 16-bit modes): EPI_RESID runs the producer epilogue (wave_epilogue_resid_fold), EPI_STORE /
// EPI_GELU the consumer form of wave_epilogue_lds; a wave then carries 6 registers of row / column data across its last
// compute phase instead of 16 bias registers.
// LNT: LayerNorm in the tail (above)
template <int PREC, int EPI, bool BAL = false, bool STAG = false, bool FOLD = false, bool LNT = false, bool LNC = false>
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(const GemmArgs a) {
  static_assert(!LNT || (EPI == EPI_RESID && PREC != MCM_PREC_F32 && !FOLD && !BAL && !STAG), "LNT: plain residual form");
  static_assert(!LNC || (EPI == EPI_RESID && PREC != MCM_PREC_F32 && !FOLD && !BAL && !STAG && !LNT), "LNC: plain residual form");
  using namespace p256;
  enter_precision_mode<PREC>();
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ES = prec_esize(PREC);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // 0..7
  const int grp = wave >> 2, w4 = wave & 3;

  const int nbn = a.N / BN, nbm = a.M / BM;
  const int G8 = gridDim.x >> 3;
  const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
  const int nmt_x = (nbm - xcd + 7) >> 3;
  const int ntl_x = nmt_x * nbn;
  const int ntl = jx < ntl_x ? (ntl_x - jx + G8 - 1) / G8 : 0;
  if (ntl == 0) {  // no tile for this workgroup (small problems): it still takes LayerNorm tickets
    if constexpr (LNT) {
      float am = 0.f;
      ln_tail<PREC>(a, xcd, nmt_x, lane, am);
      sat_report<PREC>(am, a.sat);
      ln_finish(a, xcd, smem);
    }
    if constexpr (LNC) ln_finish(a, xcd, smem);   // (counted among the XCD's finished workgroups: the last one zeroes the counters)
    return;
  }
  const int nk = (a.K * ES) / ROWB;
  const int total = ntl * nk;

  const int dmt = G8 / nbn, dnt = G8 - dmt * nbn;
  // GROUPED walk (A/B, mcm_debug_gemm_group_n / gemm_set_group_n): the N tiles of this XCD's list are walked in groups of
  // a.gn — all row tiles against N tiles [0, gn), then [gn, 2 gn), ... — so only gn / nbn of W is live in the XCD's L2 at a
  // time (fc1: 4 of 12 tiles = 1.6 of 4.7 MB), at the price of reading every X panel nbn / gn times.  Unlike cutting the
  // GEMM into column-block launches (mcm_debug_nsplit) it adds no ragged round.  Tile ordinal -> (mtl, nt) by tile_of.
  const bool grouped = a.gn > 0 && a.gn < nbn;
  struct Cursor { int mtl, nt, q; };
  auto cursor_next = [&](Cursor& c) {
    if (grouped) {
      c.q += G8;
      tile_of(c.q, nmt_x, nbn, a.gn, c.mtl, c.nt);
      return;
    }
    c.mtl += dmt;
    c.nt += dnt;
    if (c.nt >= nbn) { c.nt -= nbn; ++c.mtl; }
  };
  auto mt_of = [&](int mtl) { return a.rev ? nmt_x - 1 - mtl : mtl; };
  const size_t sx = (size_t)a.ldx * ES, sw = (size_t)a.K * ES;  // row strides in bytes

  // ---- LDS-DMA side.  Waves 0-3 (rows 0-127 of the tile) stage their half of the X panel and all of W,
  // 12 pieces per wave and step; waves 4-7 stage the other X half, which only they read, 4 pieces.  Every
  // piece is issued in the memory phase of step s for step s+1 and waited for at the end of the issuing
  // wave's compute phase, one phase before its first reader.
  // Per-lane constants (DMA source offsets, fragment offsets) are NOT kept across the loop: 128 accumulators
  // + 64 fragments leave hipcc no room, and a spilled loop invariant comes back through a scratch load whose
  // vmcnt(0) drains the DMA stream.  They are rebuilt from an opaque copy of the lane id where needed (~12 VALU).
  struct LaneK { uint32_t voff_x, voff_w; int fo0, fo1; };
  auto lane_consts = [&]() {
    int l = lane;
    asm volatile("" : "+v"(l));
    const int rr = (l >> 4) * 2 + ((l & 15) >> 3);
    const int chunk = (l & 7) ^ (((w4 & 1) << 2) | (l >> 4));
    LaneK c;
    c.voff_x = (uint32_t)(rr * (uint32_t)sx + chunk * 16);
    c.voff_w = (uint32_t)(perm_n(w4 * 8 + rr) * (uint32_t)sw + chunk * 16);
    c.fo0 = frag_off(l & 15, l >> 4, 0);
    c.fo1 = frag_off(l & 15, l >> 4, 1);
    return c;
  };
  Cursor ci{jx / nbn, jx % nbn, jx};
  if (grouped) tile_of(jx, nmt_x, nbn, a.gn, ci.mtl, ci.nt);
  int ji = 0, kti = 0;
  const char *tx, *tw;  // uniform: first byte of this wave's rows of the tile being staged
  auto set_issue_tile = [&]() {
    const int m0 = (mt_of(ci.mtl) * 8 + xcd) * BM, n0 = ci.nt * BN;
    tx = (const char*)a.x + (size_t)(m0 + grp * 128 + w4 * 8) * sx;
    tw = (const char*)a.w + (size_t)n0 * sw;
  };
  const uint32_t lds0 = lds_addr(smem);
  auto piece = [&](const LaneK& lk, int st, int i) {  // i: 0-3 X pieces, 4-11 W pieces (waves 0-3 only)
    const uint32_t base = lds0 + st * STAGE_BYTES + w4 * 1024;
    const size_t ko = (size_t)kti * ROWB;
    if (ABL(1)) return;  // ablation build: no LDS-DMA
    if (i < 4) {
      const size_t kox = (size_t)(kti >> a.ksplit) * ROWB;  // split weights: X K-step s / 2 meets W' K-steps s (hi), s + 1 (lo)
#if defined(MCM_HARNESS) && defined(MCM_GEMM_ABLATE)
      if (ABL(32)) {
        glds16s_nt(tx + kox + (size_t)(i * 32) * sx, lk.voff_x, base + (grp * 16 + i * 4) * 1024);
        return;
      }
#endif
      glds16s(tx + kox + (size_t)(i * 32) * sx, lk.voff_x, base + (grp * 16 + i * 4) * 1024);
    } else {
      const int q = BAL ? grp * 4 + (i - 4) : i - 4;
      glds16s(tw + ko + (size_t)((q >> 1) * 64 + (q & 1) * 8) * sw, lk.voff_w, base + A_BYTES + q * 4096);
    }
  };
  constexpr int NP0 = BAL ? 8 : 12;  // pieces per step of waves 0-3
  auto issue_done = [&]() {
    if (++kti == nk) {
      kti = 0;
      if (++ji < ntl) {
        cursor_next(ci);
        set_issue_tile();
      }
    }
  };

  // ---- MFMA side
  const int wr = wave >> 2, wc = wave & 3;  // 2 x 4 waves, wave tile 128 x 64
  const int xbase = wr * 128 * ROWB;
  const int wbase = A_BYTES + wc * 64 * ROWB;
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
  // Fragments of a step: the W fragments of both K halves and the X fragments of the first half are read in
  // the memory phase (16 reads, 64 registers); the X fragments of the second half replace those of the first
  // one by one during the compute phase (each after its last use) — the rows they come from are staged by this
  // very wave group, so nobody overwrites them before the group's own next memory phase.
  u32x4_t xf[8], wf[2][4];
  auto readf = [&](const LaneK& lk, int st, int i) {  // memory-phase read i of 16
    const char* sb = smem + st * STAGE_BYTES;
    if (i < 4) wf[0][i] = *(const u32x4_t*)(sb + wbase + i * 2048 + lk.fo0);
    else if (i < 12) xf[i - 4] = *(const u32x4_t*)(sb + xbase + (i - 4) * 2048 + lk.fo0);
    else wf[1][i - 12] = *(const u32x4_t*)(sb + wbase + (i - 12) * 2048 + lk.fo1);
  };
  auto pin_frags = [&]() {  // the fragments are in registers here, not wherever hipcc would sink the reads to
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int f = 0; f < 4; ++f) asm volatile("" : "+v"(wf[0][f]));
#pragma unroll
    for (int f = 0; f < 4; ++f) asm volatile("" : "+v"(wf[1][f]));
#pragma unroll
    for (int f = 0; f < 8; ++f) asm volatile("" : "+v"(xf[f]));
  };
  f32x4_t acc[4][8];
  zero_acc<8>(acc);
  float amax = 0.f;  // fp16 saturation watch (sat_track / sat_report, common.hpp)
  auto mfma_pair = [&](const u32x4_t& wv4, const u32x4_t& xv4, f32x4_t& c) {
    if constexpr (PREC != MCM_PREC_F32) {
      c = mfma16<PREC>(__builtin_bit_cast(uint4, wv4), __builtin_bit_cast(uint4, xv4), c);
    } else {
      const f32x4_t wv = __builtin_bit_cast(f32x4_t, wv4);
      const f32x4_t xv = __builtin_bit_cast(f32x4_t, xv4);
#pragma unroll
      for (int t = 0; t < 4; ++t) c = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[t], xv[t], c, 0, 0, 0);
    }
  };
  auto compute = [&](int fo1, int st) {
    const char* sb = smem + st * STAGE_BYTES;
#pragma unroll
    for (int fi = 0; fi < 8; ++fi) {
#pragma unroll
      for (int fj = 0; fj < 4; ++fj) mfma_pair(wf[0][fj], xf[fi], acc[fj][fi]);
      xf[fi] = *(const u32x4_t*)(sb + xbase + fi * 2048 + fo1);
    }
#pragma unroll
    for (int fi = 0; fi < 8; ++fi)
#pragma unroll
      for (int fj = 0; fj < 4; ++fj) mfma_pair(wf[1][fj], xf[fi], acc[fj][fi]);
  };

  Cursor cc = ci;  // (the issue cursor has not moved yet)
  int em0 = 0, en0 = 0;  // tile whose epilogue is pending
  constexpr bool FOLD_OUT = FOLD && EPI <= EPI_GELU;  // consumer side of the LayerNorm fold
  static_assert(!FOLD || (PREC != MCM_PREC_F32 && EPI != EPI_PATCH), "LayerNorm fold: 16-bit operand modes");
  f32x4_t bv[4];  // bias of the pending tile: asm loads issued at the top of its last compute phase
#pragma unroll
  for (int fj = 0; fj < 4; ++fj) bv[fj] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  // consumer fold: b'[n] and c[n] of column n0 + wc*64 + lane, (rstd, mean rstd) of rows m0 + wr*128 + lane and + 64
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  float fold_b = 0.f, fold_c = 0.f;
  f32x2_t fold_r0 = {0.f, 0.f}, fold_r1 = {0.f, 0.f};
  auto epilogue = [&]() {
    // every lane-derived address of the epilogue is recomputed from an opaque copy of the lane id: hoisted out
    // of the K loop they would occupy ~20 registers that the loop (128 accumulators + 64 fragments) does not have
    int le = lane;
    asm volatile("" : "+v"(le));
    if constexpr (FOLD_OUT) {
      asm volatile("" : "+v"(fold_b), "+v"(fold_c), "+v"(fold_r0), "+v"(fold_r1));
      if (!DBG(4)) {
        char* win = smem + 2 * STAGE_BYTES + wave * 4096;
        FoldRegs fo;
        f32x4_t bx[4];
        const int gl = le >> 4;
#pragma unroll
        for (int fj = 0; fj < 4; ++fj)
#pragma unroll
          for (int t = 0; t < 4; ++t) {  // the lane's 16 columns: gl*16 + fj*4 + t of the wave's 64
            const int src = (gl * 16 + fj * 4 + t) << 2;
            bx[fj][t] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, fold_b)));
            fo.cv[fj][t] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, fold_c)));
          }
        fo.rstd[0] = fold_r0[0]; fo.mrstd[0] = fold_r0[1];
        fo.rstd[1] = fold_r1[0]; fo.mrstd[1] = fold_r1[1];
        wave_epilogue_lds<PREC, EPI, 8, true, true>(a, acc, bx, em0 + wr * 128, en0 + wc * 64, le, win, amax, &fo);
      }
    } else {
      if constexpr (!FOLD && EPI != EPI_RESID) {  // (the residual forms load their bias inside the epilogue: a pin
                                                   // would keep 16 zeros live across the K loop)
#pragma unroll
        for (int fj = 0; fj < 4; ++fj) asm volatile("" : "+v"(bv[fj]));
      }
      if (!DBG(4)) {
        char* win = smem + 2 * STAGE_BYTES + wave * 4096;
        if constexpr (PREC != MCM_PREC_F32 && EPI <= EPI_GELU)
          wave_epilogue_lds<PREC, EPI, 8, true>(a, acc, bv, em0 + wr * 128, en0 + wc * 64, le, win, amax);
        else if constexpr (FOLD)
          wave_epilogue_resid_fold<PREC, 8>(a, acc, em0 + wr * 128, en0 + wc * 64, le, win, amax);
        else if constexpr (LNC)
          wave_epilogue_resid_lnc<PREC, 8>(acc, em0 + wr * 128, en0 + wc * 64, le, win, amax,
                                           LNC_ARG(ln_state) + (size_t)xcd * LNC_ARG(ln_rs),
                                           (uint32_t)((((em0 / BM) >> 3) * 2 + wr) * 4), (uint32_t)(nbn * 4));
        else
          wave_epilogue_f32_interior<EPI, 8>(a, acc, bv, em0 + wr * 128, en0 + wc * 64, le, win);
      }
    }
    zero_acc<8>(acc);
  };
  if (a.dbg >> 8) {  // harness: de-phase the workgroups of an XCD, (dbg >> 8) x 1024 cycles per step of jx & 3
    const uint64_t until = __builtin_amdgcn_s_memtime() + (uint64_t)((jx >> 3) & 3) * (uint64_t)(a.dbg >> 8) * 1024u;
    while (__builtin_amdgcn_s_memtime() < until) __builtin_amdgcn_s_sleep(8);
  }
  set_issue_tile();
  {
    const LaneK lk = lane_consts();
#pragma unroll
    for (int i = 0; i < NP0; ++i)
      if (BAL || i < 4 || !grp) piece(lk, 0, i);
  }
  issue_done();
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (grp) {  // waves 4-7 run one phase behind
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  int ktc = 0;
  bool pend = false;
  int pub = -1;  // LNT: row tile (index within this XCD) whose epilogue this wave has issued but not yet published
  auto phase_barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  PPT_INIT();
  for (int s = 0; s < total; ++s) {
    // ---- memory phase of step s: the step's fragments into registers, interleaved with the DMA issues for
    // step s+1 (a DMA issue blocks the wave on the TA, a ds_read on the LDS queue: alternating them lets the two
    // queues drain side by side).  The first instruction is a ds_read on purpose: hipcc puts a vmcnt(0) in
    // front of the first fragment read after an epilogue with compiler-visible stores, which must not have
    // this phase's DMA to wait for.  After the very last step the issue re-reads the last tile (never used).
    // At a tile boundary (`pend`) waves 0-3 issue, pass the barrier, run the epilogue and only then read the
    // fragments; waves 4-7 run the epilogue first: both epilogues fall into the same phase and no fragment
    // is live across them.
    const int sr = s & 1, si = sr ^ 1;
    const bool split = pend && !grp;
    int fo1;
    if (pend) {
      if (!grp) {
        const LaneK lk = lane_consts();
#pragma unroll
        for (int i = 0; i < NP0; ++i) piece(lk, si, i);
        issue_done();
        if constexpr (!STAG) phase_barrier();
      }
      epilogue();
      if constexpr (LNT) pub = (em0 / BM) >> 3;
      if (!grp) {
        const LaneK lk = lane_consts();
        fo1 = lk.fo1;
#pragma unroll
        for (int i = 0; i < 16; ++i) readf(lk, sr, i);
      }
    }
    if (!split) {
      const LaneK lk = lane_consts();
      fo1 = lk.fo1;
      if (!grp) {
#pragma unroll
        for (int i = 0; i < NP0; ++i) {
          if (i < 8) {
            readf(lk, sr, 2 * i);
            readf(lk, sr, 2 * i + 1);
          }
          piece(lk, si, i);
        }
      } else {
        if constexpr (BAL) {  // W first: it has to land within this phase
#pragma unroll
          for (int i = 4; i < 8; ++i) {
            readf(lk, sr, 2 * (i - 4));
            readf(lk, sr, 2 * (i - 4) + 1);
            piece(lk, si, i);
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            readf(lk, sr, 8 + 2 * i);
            readf(lk, sr, 8 + 2 * i + 1);
            piece(lk, si, i);
          }
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) readf(lk, sr, 4 * i + j);
            piece(lk, si, i);
          }
        }
      }
      issue_done();
    }
    pin_frags();
    if constexpr (BAL) {
      if (grp) wait_vmcnt<4>();  // everything older than this phase's 4 X pieces: the W pieces waves 0-3 read next
    }
    PPT(0);
    if (!split || STAG) phase_barrier();
    PPT(1);
    // ---- compute phase of step s
    if constexpr (FOLD_OUT) {
      if (ktc == nk - 1) {  // 4 asm loads, covered by the wait that ends this phase
        int le = lane;
        asm volatile("" : "+v"(le));
        const int pn = cc.nt * BN + wc * 64, pm = (mt_of(cc.mtl) * 8 + xcd) * BM + wr * 128;
        asm volatile("global_load_dword %0, %1, %2" : "=v"(fold_b) : "v"(le * 4), "s"(a.bias + pn) : "memory");
        asm volatile("global_load_dword %0, %1, %2" : "=v"(fold_c) : "v"(le * 4), "s"(a.fold_c + pn) : "memory");
        asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(fold_r0) : "v"(le * 8), "s"(a.fold_rs + pm) : "memory");
        asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(fold_r1) : "v"(le * 8), "s"(a.fold_rs + pm + 64) : "memory");
      }
    } else if (FOLD || EPI == EPI_RESID) {  // residual forms: the bias (and gamma) are loaded inside the epilogue
    } else if (ktc == nk - 1 && a.bias) {
      int le = lane;
      asm volatile("" : "+v"(le));
      load_bias_async(a, cc.nt * BN + wc * 64 + (le >> 4) * 16, bv);  // covered by the wait that ends this phase
    }
    __builtin_amdgcn_s_setprio(1);
    if (!ABL(2)) compute(fo1, sr);  // ablation build: no MFMAs (and none of the compute phase's fragment reads)
    __builtin_amdgcn_s_setprio(0);
    PPT(2);
    wait_vmcnt<0>();  // this wave's pieces of step s+1, issued a phase ago
    if constexpr (LNT) {  // ... and, in the first phase after an epilogue, its stores: the tile is published
      if (pub >= 0) {
        if (lane == 0) l2_atomic_add(a.ln_state + (size_t)xcd * a.ln_rs, (uint32_t)pub * 4u, 1u);
        pub = -1;
      }
    }
    phase_barrier();
    PPT(3);
    pend = false;
    if (++ktc == nk) {
      ktc = 0;
      pend = true;
      em0 = (mt_of(cc.mtl) * 8 + xcd) * BM;
      en0 = cc.nt * BN;
      cursor_next(cc);
    }
  }
  if (!grp) {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  if (pend) epilogue();
  if constexpr (LNT) {
    if (pend) pub = (em0 / BM) >> 3;
    wait_vmcnt<0>();
    if (pub >= 0 && lane == 0) l2_atomic_add(a.ln_state + (size_t)xcd * a.ln_rs, (uint32_t)pub * 4u, 1u);
    ln_tail<PREC>(a, xcd, nmt_x, lane, amax);
  }
  if constexpr (EPI <= EPI_GELU || FOLD || LNT || LNC) sat_report<PREC>(amax, a.sat);
  if constexpr (LNT || LNC) ln_finish(a, xcd, smem);
  PPT_DUMP();
}

// =========================================================================================
// ping-pong kernel on v_mfma_f32_32x32x16_{f16,bf16} (16-bit operand modes only).  Same tile (256x256),
// wave tiles (128x64), LDS image, DMA schedule and phase structure as gemm_pp_kernel; the compute phase
// issues 32 MFMAs of 32 cycles instead of 64 of 16.  Why it pays here and did not in the barrier-locked
// kernel: in the compute phase the MFMAs run back to back from registers, and a
// 16x16x32 issues every ~19 cycles instead of 16 (the per-instruction issue overhead is paid per MFMA),
// a 32x32x16 every ~32-33 instead of 32; it also reads its A/B operands from the register file half as
// often per FLOP, which is energy on a part that sits at its power limit.
//
// Fragment maps (guide section 3): A = W block (32 tile columns x 16 k), B = X block (32 rows x 16 k),
// lane l supplies row (l & 31), 16-B chunk (2 ks + (l >> 5)) of the 128-B K-step row — the existing
// pair/XOR LDS image serves 32-row fragments conflict-free as well (the four 16-lane groups of a
// ds_read_b128 touch 8 distinct row pairs).  D: lane (j = l & 31, h = l >> 5), register r holds
// (X row j, W-block row 4h + 8(r >> 2) + (r & 3)); W rows are staged in the order perm_n32 so that this
// is tile column h*16 + r: a lane owns 16 consecutive columns of its row in each of the two 32-column
// blocks of the wave tile.
// =========================================================================================
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__device__ __forceinline__ int perm_n32(int i) {  // LDS row i (0..31) of a 32-row W block holds this block column
  return ((i >> 2) & 1) * 16 + (i >> 3) * 4 + (i & 3);
}
template <int PREC>
__device__ __forceinline__ f32x16_t mfma32(uint4 a, uint4 b, f32x16_t c) {  // 32x32x16, fp32 acc
  if constexpr (PREC == MCM_PREC_F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4_t quad(const f32x16_t& v, int q) {
  return (f32x4_t){v[q * 4 + 0], v[q * 4 + 1], v[q * 4 + 2], v[q * 4 + 3]};
}

// 16-bit epilogue of a full 128x64 wave tile held as acc[xb][wb] (32x32 blocks).  Unit = one block
// (32 rows x 32 columns = 2 KiB of 16-bit), units in wb-major order ping-ponging between the two halves
// of the wave's 4-KiB window like wave_epilogue_lds; a lane writes its 32 B of row j, the read-back puts
// four lanes on a 64-B row, so one store instruction writes 16 rows x 64 B.  bv[wb][q]: bias of columns
// nw + wb*32 + h*16 + q*4 .. +3; bv[1] is loaded here (asm, counted) and first used by unit 4.
template <int PREC, int EPI>
__device__ __forceinline__ void wave_epilogue16_b32(const GemmArgs& a, const f32x16_t (&acc)[4][2], f32x4_t (&bv0)[4],
                                                    int mw, int nw, int lane, char* scratch, float& amax) {
  const int j = lane & 31, h = lane >> 5;
  f32x4_t bv1[4];
  if (a.bias) {
    const float* p = a.bias + nw + 32 + h * 16;
#pragma unroll
    for (int q = 0; q < 4; ++q) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bv1[q]) : "v"(p + q * 4) : "memory");
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) bv1[q] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }
  const int rrow = lane >> 2, c4 = lane & 3;  // read-back: 4 lanes per 64-B row
  auto write_unit = [&](int u, const f32x4_t (&bv)[4]) {
    const int wb = u >> 2, xb = u & 3;
    f32x4_t v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      v[q] = quad(acc[xb][wb], q) + bv[q];
      if constexpr (EPI == EPI_GELU) {
#pragma unroll
        for (int t = 0; t < 4; ++t) v[q][t] = quick_gelu_fast(v[q][t]);
      }
      sat_track<PREC>(amax, v[q][0], v[q][1]);
      sat_track<PREC>(amax, v[q][2], v[q][3]);
    }
    char* w = scratch + (u & 1) * 2048 + j * 64;
    const int sw = (j >> 1) & 3;
    *(uint4*)(w + (((h * 2) ^ sw) << 4)) = make_uint4(pack2<PREC>(v[0][0], v[0][1]), pack2<PREC>(v[0][2], v[0][3]),
                                                      pack2<PREC>(v[1][0], v[1][1]), pack2<PREC>(v[1][2], v[1][3]));
    *(uint4*)(w + (((h * 2 + 1) ^ sw) << 4)) = make_uint4(pack2<PREC>(v[2][0], v[2][1]), pack2<PREC>(v[2][2], v[2][3]),
                                                          pack2<PREC>(v[3][0], v[3][1]), pack2<PREC>(v[3][2], v[3][3]));
  };
  auto read_unit = [&](int u, uint4 (&r)[2]) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int row = t * 16 + rrow;
      r[t] = *(const uint4*)(scratch + (u & 1) * 2048 + row * 64 + ((c4 ^ ((row >> 1) & 3)) << 4));
    }
  };
  const int lane_off = rrow * a.ldo + c4 * 8;  // elements
  auto store_unit = [&](int u, const uint4 (&r)[2]) {
    const int wb = u >> 2, xb = u & 3;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      uint16_t* rowbase = (uint16_t*)a.out + (size_t)(mw + xb * 32 + t * 16) * a.ldo + nw + wb * 32;
      if (!DBG(16)) store16_stream(rowbase + lane_off, r[t]);
    }
  };
  write_unit(0, bv0);
#pragma unroll
  for (int u = 1; u < 8; ++u) {
    uint4 r[2];
    read_unit(u - 1, r);
    if (u == 4) {
      // VMEM queue behind the four bv1 loads: the stores of units 0..2 (2 each)
      asm volatile("s_waitcnt vmcnt(6)" : "+v"(bv1[0]), "+v"(bv1[1]), "+v"(bv1[2]), "+v"(bv1[3])::"memory");
    }
    if (u < 4) write_unit(u, bv0);
    else write_unit(u, bv1);
    store_unit(u - 1, r);
  }
  {
    uint4 r[2];
    read_unit(7, r);
    store_unit(7, r);
  }
}

// fp32-row epilogue (residual read-modify-write) of the same wave tile.  Unit = one 32x32 block = 32 rows x
// 128 B = the whole 4-KiB window; eight lanes read a row back, so every global access instruction moves
// 8 rows x 128 B.  The bias (of the four columns a lane owns AFTER the bounce) is added after the bounce:
// (acc + b) + resid, the order of every other GEMM kernel here.  All global accesses are asm, counted:
// queue at the wait of unit u, oldest first: [loads u] [stores u-1] [loads u+1]  =>  vmcnt <= 8.
__device__ __forceinline__ void wave_epilogue_resid_b32(const GemmArgs& a, const f32x16_t (&acc)[4][2],
                                                        const f32x4_t (&bvf)[2], int mw, int nw, int lane, char* scratch) {
  const int j = lane & 31, h = lane >> 5;
  const int rrow = lane >> 3, c8 = lane & 7;
  const uint32_t voff = (uint32_t)(rrow * a.ldo + c8 * 4) * 4u;  // bytes
  const char* base = (const char*)a.resid + ((size_t)mw * a.ldo + nw) * 4;
  auto rowbase = [&](int u, int t) {
    const int wb = u >> 2, xb = u & 3;
    return base + ((size_t)(xb * 32 + t * 8) * a.ldo + wb * 32) * 4;
  };
  f32x4_t buf[2][4];
#pragma unroll
  for (int t = 0; t < 4; ++t) gload16(buf[0][t], rowbase(0, t), voff);
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int wb = u >> 2, xb = u & 3;
    if (u + 1 < 8) {
#pragma unroll
      for (int t = 0; t < 4; ++t) gload16(buf[(u + 1) & 1][t], rowbase(u + 1, t), voff);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) *(f32x4_t*)(scratch + j * 128 + (((h * 4 + q) ^ (j & 7)) << 4)) = quad(acc[xb][wb], q);
    f32x4_t v[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int row = t * 8 + rrow;
      v[t] = *(const f32x4_t*)(scratch + row * 128 + ((c8 ^ (row & 7)) << 4)) + bvf[wb];
    }
    if (u == 0 || u + 1 == 8) wait_vmcnt_pin<4>(buf[u & 1]);
    else wait_vmcnt_pin<8>(buf[u & 1]);
#pragma unroll
    for (int t = 0; t < 4; ++t) v[t] += buf[u & 1][t];
#pragma unroll
    for (int t = 0; t < 4; ++t) gstore16(rowbase(u, t), voff, v[t]);
  }
}

template <int PREC, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_pp32_kernel(const GemmArgs a) {
  using namespace p256;
  static_assert(PREC != MCM_PREC_F32 && EPI != EPI_PATCH, "16-bit operand modes, interior tiles");
  enter_precision_mode<PREC>();
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ES = 2;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // 0..7
  const int grp = wave >> 2, w4 = wave & 3;

  const int nbn = a.N / BN, nbm = a.M / BM;
  const int G8 = gridDim.x >> 3;
  const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
  const int nmt_x = (nbm - xcd + 7) >> 3;
  const int ntl_x = nmt_x * nbn;
  const int ntl = jx < ntl_x ? (ntl_x - jx + G8 - 1) / G8 : 0;
  if (ntl == 0) return;
  const int nk = (a.K * ES) / ROWB;
  const int total = ntl * nk;

  const int dmt = G8 / nbn, dnt = G8 - dmt * nbn;
  struct Cursor { int mtl, nt; };
  auto cursor_next = [&](Cursor& c) {
    c.mtl += dmt;
    c.nt += dnt;
    if (c.nt >= nbn) { c.nt -= nbn; ++c.mtl; }
  };
  auto mt_of = [&](int mtl) { return a.rev ? nmt_x - 1 - mtl : mtl; };
  const size_t sx = (size_t)a.ldx * ES, sw = (size_t)a.K * ES;  // row strides in bytes

  // ---- LDS-DMA side: exactly gemm_pp_kernel's (12 pieces per step from waves 0-3, 4 from waves 4-7), only the
  // order of the W rows inside a 32-row block differs (perm_n32)
  struct LaneK { uint32_t voff_x, voff_w; int fo0; };
  auto lane_consts = [&]() {
    int l = lane;
    asm volatile("" : "+v"(l));
    const int rr = (l >> 4) * 2 + ((l & 15) >> 3);
    const int chunk = (l & 7) ^ (((w4 & 1) << 2) | (l >> 4));
    LaneK c;
    c.voff_x = (uint32_t)(rr * (uint32_t)sx + chunk * 16);
    c.voff_w = (uint32_t)(perm_n32(w4 * 8 + rr) * (uint32_t)sw + chunk * 16);
    const int j = l & 31, h = l >> 5;
    c.fo0 = (j >> 1) * 256 + ((((j & 1) << 3) | ((h ^ (j >> 1)) & 7)) << 4);  // K16 sub-step ks: fo0 ^ (ks << 5)
    return c;
  };
  Cursor ci{jx / nbn, jx % nbn};
  int ji = 0, kti = 0;
  const char *tx, *tw;
  auto set_issue_tile = [&]() {
    const int m0 = (mt_of(ci.mtl) * 8 + xcd) * BM, n0 = ci.nt * BN;
    tx = (const char*)a.x + (size_t)(m0 + grp * 128 + w4 * 8) * sx;
    tw = (const char*)a.w + (size_t)n0 * sw;
  };
  const uint32_t lds0 = lds_addr(smem);
  auto piece = [&](const LaneK& lk, int st, int i) {  // i: 0-3 X pieces, 4-11 W pieces (waves 0-3 only)
    const uint32_t base = lds0 + st * STAGE_BYTES + w4 * 1024;
    const size_t ko = (size_t)kti * ROWB;
    if (i < 4) {
      glds16s(tx + ko + (size_t)(i * 32) * sx, lk.voff_x, base + (grp * 16 + i * 4) * 1024);
    } else {
      const int q = i - 4;  // LDS rows q*32 + w4*8 + rr of the W panel = tile columns q*32 + perm_n32(w4*8 + rr)
      glds16s(tw + ko + (size_t)(q * 32) * sw, lk.voff_w, base + A_BYTES + q * 4096);
    }
  };
  auto issue_done = [&]() {
    if (++kti == nk) {
      kti = 0;
      if (++ji < ntl) {
        cursor_next(ci);
        set_issue_tile();
      }
    }
  };

  // ---- MFMA side: wave tile 128 x 64 = 4 X blocks x 2 W blocks of 32x32, K-step = 4 sub-steps of 16
  const int wr = wave >> 2, wc = wave & 3;
  const int xbase = wr * 128 * ROWB;
  const int wbase = A_BYTES + wc * 64 * ROWB;
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
  // memory phase: all 8 W fragments and the X fragments of sub-steps 0 and 1 (64 registers); the X fragments of
  // sub-steps 2 and 3 replace them during the compute phase, each after its last use
  u32x4_t xf[2][4], wf[4][2];
  auto readf = [&](const LaneK& lk, int st, int i) {  // memory-phase read i of 16, in order of first use
    const char* sb = smem + st * STAGE_BYTES;
    if (i < 2) wf[0][i] = *(const u32x4_t*)(sb + wbase + i * 4096 + lk.fo0);
    else if (i < 6) xf[0][i - 2] = *(const u32x4_t*)(sb + xbase + (i - 2) * 4096 + lk.fo0);
    else if (i < 8) wf[1][i - 6] = *(const u32x4_t*)(sb + wbase + (i - 6) * 4096 + (lk.fo0 ^ 32));
    else if (i < 12) xf[1][i - 8] = *(const u32x4_t*)(sb + xbase + (i - 8) * 4096 + (lk.fo0 ^ 32));
    else wf[2 + ((i - 12) >> 1)][(i - 12) & 1] =
        *(const u32x4_t*)(sb + wbase + ((i - 12) & 1) * 4096 + (lk.fo0 ^ ((2 + ((i - 12) >> 1)) << 5)));
  };
  auto pin_frags = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int f = 0; f < 2; ++f) asm volatile("" : "+v"(wf[k][f]));
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int f = 0; f < 4; ++f) asm volatile("" : "+v"(xf[k][f]));
  };
  f32x16_t acc[4][2];
  float amax = 0.f;
  auto zero = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int w = 0; w < 2; ++w)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][w][r] = 0.f;
  };
  zero();
  auto compute = [&](int fo0, int st) {
    const char* sb = smem + st * STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int xb = 0; xb < 4; ++xb) {
#pragma unroll
        for (int wb = 0; wb < 2; ++wb)
          acc[xb][wb] = mfma32<PREC>(__builtin_bit_cast(uint4, wf[ks][wb]), __builtin_bit_cast(uint4, xf[ks][xb]), acc[xb][wb]);
        xf[ks][xb] = *(const u32x4_t*)(sb + xbase + xb * 4096 + (fo0 ^ ((ks + 2) << 5)));
      }
#pragma unroll
    for (int ks = 2; ks < 4; ++ks)
#pragma unroll
      for (int xb = 0; xb < 4; ++xb)
#pragma unroll
        for (int wb = 0; wb < 2; ++wb)
          acc[xb][wb] = mfma32<PREC>(__builtin_bit_cast(uint4, wf[ks][wb]), __builtin_bit_cast(uint4, xf[ks & 1][xb]), acc[xb][wb]);
  };

  Cursor cc{jx / nbn, jx % nbn};
  int em0 = 0, en0 = 0;  // tile whose epilogue is pending
  // bias registers of the pending tile, asm loads issued at the top of its last compute phase: 16-bit outputs
  // need the 16 columns of the lane's first block (the second block's are loaded inside the epilogue), the
  // fp32-row form the 4 + 4 columns the lane owns after the LDS bounce
  constexpr bool RESID = (EPI == EPI_RESID);
  f32x4_t bv[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) bv[q] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  auto bias_issue = [&](int n0) {
    int le = lane;
    asm volatile("" : "+v"(le));
    if constexpr (RESID) {
      const float* p = a.bias + n0 + wc * 64 + (le & 7) * 4;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bv[0]) : "v"(p) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bv[1]) : "v"(p + 32) : "memory");
    } else {
      const float* p = a.bias + n0 + wc * 64 + (le >> 5) * 16;
#pragma unroll
      for (int q = 0; q < 4; ++q) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bv[q]) : "v"(p + q * 4) : "memory");
    }
  };
  auto epilogue = [&]() {
    int le = lane;
    asm volatile("" : "+v"(le));
#pragma unroll
    for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(bv[q]));
    if (!DBG(4)) {
      char* win = smem + 2 * STAGE_BYTES + wave * 4096;
      if constexpr (RESID) {
        const f32x4_t bvf[2] = {bv[0], bv[1]};
        wave_epilogue_resid_b32(a, acc, bvf, em0 + wr * 128, en0 + wc * 64, le, win);
      } else {
        wave_epilogue16_b32<PREC, EPI>(a, acc, bv, em0 + wr * 128, en0 + wc * 64, le, win, amax);
      }
    }
    zero();
  };
  set_issue_tile();
  {
    const LaneK lk = lane_consts();
#pragma unroll
    for (int i = 0; i < 12; ++i)
      if (i < 4 || !grp) piece(lk, 0, i);
  }
  issue_done();
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (grp) {  // waves 4-7 run one phase behind
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  int ktc = 0;
  bool pend = false;
  auto phase_barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int s = 0; s < total; ++s) {
    const int sr = s & 1, si = sr ^ 1;
    const bool split = pend && !grp;
    int fo0;
    if (pend) {
      if (!grp) {
        const LaneK lk = lane_consts();
#pragma unroll
        for (int i = 0; i < 12; ++i) piece(lk, si, i);
        issue_done();
        phase_barrier();
      }
      epilogue();
      if (!grp) {
        const LaneK lk = lane_consts();
        fo0 = lk.fo0;
#pragma unroll
        for (int i = 0; i < 16; ++i) readf(lk, sr, i);
      }
    }
    if (!split) {
      const LaneK lk = lane_consts();
      fo0 = lk.fo0;
      if (!grp) {
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          if (i < 8) {
            readf(lk, sr, 2 * i);
            readf(lk, sr, 2 * i + 1);
          }
          piece(lk, si, i);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) readf(lk, sr, 4 * i + jj);
          piece(lk, si, i);
        }
      }
      issue_done();
    }
    pin_frags();
    if (!split) phase_barrier();
    // ---- compute phase of step s
    if (ktc == nk - 1 && a.bias) bias_issue(cc.nt * BN);  // covered by the wait that ends this phase
    __builtin_amdgcn_s_setprio(1);
    compute(fo0, sr);
    __builtin_amdgcn_s_setprio(0);
    wait_vmcnt<0>();  // this wave's pieces of step s+1, issued a phase ago
    phase_barrier();
    pend = false;
    if (++ktc == nk) {
      ktc = 0;
      pend = true;
      em0 = (mt_of(cc.mtl) * 8 + xcd) * BM;
      en0 = cc.nt * BN;
      cursor_next(cc);
    }
  }
  if (!grp) {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  if (pend) epilogue();
  if constexpr (!RESID) sat_report<PREC>(amax, a.sat);
}



// ---- launchers of the arms ----------------------------------------------------------------------------------------
template <int PREC, int EPI>
hipError_t launch_tile_fold(const GemmArgs& a, hipStream_t s) {
  static PerDeviceFlag attr_set;
  if (!attr_set.get()) {
    hipError_t e = hipFuncSetAttribute((const void*)arms::gemm_tile_kernel<PREC, EPI, true>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, tile::LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_set.set();
  }
  const int nbn = (a.N + tile::BN - 1) / tile::BN, nbm = (a.M + tile::BM - 1) / tile::BM;
  hipLaunchKernelGGL((arms::gemm_tile_kernel<PREC, EPI, true>), dim3(nbn * nbm), dim3(256), tile::LDS_BYTES, s, a);
  return hipGetLastError();
}
template <int PREC, int EPI, bool CS>
hipError_t launch_persist(const GemmArgs& a, hipStream_t s) {
  static PerDeviceFlag attr_set;
  if (!attr_set.get()) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_persist_kernel<PREC, EPI, CS>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, persist::LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_set.set();
  }
  hipLaunchKernelGGL((gemm_persist_kernel<PREC, EPI, CS>), dim3(persistent_grid()), dim3(512), persist::LDS_BYTES, s, a);
  return hipGetLastError();
}
template <int PREC, int EPI, bool BAL = false, bool STAG = false, bool FOLD = false, bool LNT = false, bool LNC = false>
hipError_t launch_pp(const GemmArgs& a, hipStream_t s) {
  static PerDeviceFlag attr_set;
  if (!attr_set.get()) {
    hipError_t e = hipFuncSetAttribute((const void*)arms::gemm_pp_kernel<PREC, EPI, BAL, STAG, FOLD, LNT, LNC>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, p256::LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_set.set();
  }
  hipLaunchKernelGGL((arms::gemm_pp_kernel<PREC, EPI, BAL, STAG, FOLD, LNT, LNC>), dim3(persistent_grid()), dim3(512), p256::LDS_BYTES, s, a);
  return hipGetLastError();
}
template <int PREC, int EPI>
hipError_t launch_pp32(const GemmArgs& a, hipStream_t s) {
  static PerDeviceFlag attr_set;
  if (!attr_set.get()) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_pp32_kernel<PREC, EPI>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, p256::LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_set.set();
  }
  hipLaunchKernelGGL((gemm_pp32_kernel<PREC, EPI>), dim3(persistent_grid()), dim3(512), p256::LDS_BYTES, s, a);
  return hipGetLastError();
}

// Routing of a launch to an arm.  Returns true when the launch is the arms' business (`*err` then holds its status),
// false when the shipped kernels of gemm.hip take it.  v = size_policy(M, N) (with a forced variant already applied).
template <int PREC, int EPI>
bool route(int& v, const GemmArgs& a, hipStream_t s, hipError_t* err) {
  const bool whole = a.M % p256::BM == 0 && a.N % p256::BN == 0;
  const bool fold = !a.lnc && (a.fold_z != nullptr || a.fold_rs != nullptr);
  *err = hipErrorInvalidValue;
  if (fold) {  // LayerNorm fold: measured slower than the LayerNorm launches (EXPERIMENTS.md)
    if constexpr (EPI != EPI_PATCH && PREC != MCM_PREC_F32) {
      const bool sides = EPI == EPI_RESID ? (a.fold_z && a.fold_g && a.fold_part && a.bias && !a.fold_rs)
                                          : (a.fold_rs && a.fold_c && a.bias && !a.fold_z);
      if (v == 5 && sides && whole && a.ldo == a.N) {
        *err = launch_pp<PREC, EPI, false, false, true>(a, s);
      } else if constexpr (EPI <= EPI_GELU) {
        if (v == 0 && sides) *err = launch_tile_fold<PREC, EPI>(a, s);
      }
    }
    return true;
  }
  if (a.ln_y && a.lnc) {  // LayerNorm by the row panel's cluster (LNC): counters for two halves per row panel
    if constexpr (EPI == EPI_RESID && PREC != MCM_PREC_F32) {
      if (v == 5 && whole && (a.N == 768 || a.N == 1024) && a.ldo == a.N && a.ln_g && a.ln_b && a.ln_state && a.fold_part && a.bias &&
          a.ln_cap8 >= 2 * ((a.M / p256::BM + 7) / 8) && persistent_grid() % 8 == 0)
        *err = launch_pp<PREC, EPI, false, false, false, false, true>(a, s);
    }
    return true;
  }
  if (a.ln_y) {  // LayerNorm in the tail: the ping-pong kernel's residual form only (the caller asked gemm_ln_tail_ok)
    if constexpr (EPI == EPI_RESID && PREC != MCM_PREC_F32) {
      if (v == 5 && whole && (a.N == 768 || a.N == 1024) && a.ldo == a.N && a.ln_g && a.ln_b && a.ln_state &&
          a.ln_cap8 * 8 >= a.M / p256::BM && persistent_grid() % 8 == 0)
        *err = launch_pp<PREC, EPI, false, false, false, true>(a, s);
    }
    return true;
  }
#ifdef MCM_HARNESS
  if (v == 1) { *err = launch_persist<PREC, EPI, false>(a, s); return true; }
  if (v == 2) { *err = launch_persist<PREC, EPI, true>(a, s); return true; }
  if (v == 7 || v == 8 || v == 6 || v == 9) {  // arms of the ping-pong kernel: whole tiles only, else as 5
    if constexpr (EPI != EPI_PATCH) {
      if (whole) {
        if (v == 7) { *err = launch_pp<PREC, EPI, true>(a, s); return true; }          // balanced DMA
        if (v == 8) { *err = launch_pp<PREC, EPI, false, true>(a, s); return true; }   // staggered epilogues
        if (v == 9) { *err = launch_pp<PREC, EPI>(a, s); return true; }                // the flagged text with every flag off
        if constexpr (PREC != MCM_PREC_F32) {
          if (v == 6 && !a.hm) { *err = launch_pp32<PREC, EPI>(a, s); return true; }   // 32x32x16 MFMAs
        }
      }
    }
    v = 5;
  }
#endif
  return false;
}

}  // namespace arms
