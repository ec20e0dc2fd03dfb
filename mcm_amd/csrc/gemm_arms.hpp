// gemm_arms.hpp — the laboratory half of gemm.hip: every GEMM kernel arm that was built, measured and NOT shipped, kept
// compilable and tested (tests/test_gpu_kernels.py runs each against the oracle and bit-for-bit against the shipped
// kernels).  Included by gemm.hip INSIDE its anonymous namespace, and only in the A/B builds: -DMCM_HARNESS
// (libmcm_hip_harness.so, tools/gemm_bench), -DMCM_LN_FOLD / -DMCM_LN_TAIL (make fold / make tail).  libmcm_hip.so does
// not contain a line of this file.  What each arm measured: EXPERIMENTS.md.
//
//   arms::gemm_pp_kernel          the shipped ping-pong kernel with its A/B flags: FOLD (LayerNorm fold producer / consumer
//                                 epilogues), LNT (LayerNorm in the tail of the residual GEMMs), LNC (round 6: LayerNorm by
//                                 the row panel's cluster of workgroups — the full-row epilogue), the grouped tile walk
//                                 (variant 9), the in-loop ablation bits, phase timing
// Removed in round 6 (measured negative in rounds 2 - 4; EXPERIMENTS.md "Removed arms", git history): the persistent 256x128
// 3-stage kernel (variants 1 / 2), the ping-pong loop on 32x32x16 MFMAs (6), balanced DMA (7), staggered epilogues (8).
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
// exist on this part (768 KB of accumulators against a 512-KB register file; a 128-row full-row tile needs 192 accumulator registers; the 64-row
// one was built and measured, ROW64 below / EXPERIMENTS.md R6.7: -21 %), but the full-row TILE does — spread over the N / 256
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
  int M, N, ldo, ln_cap8, spin;
};
__device__ __forceinline__ LncArgs lnc_args() {
  LncArgs r;
  r.bias = LNC_ARG(bias); r.ln_g = LNC_ARG(ln_g); r.ln_b = LNC_ARG(ln_b); r.resid = LNC_ARG(resid); r.ln_y = LNC_ARG(ln_y);
  r.fold_part = LNC_ARG(fold_part); r.ln_eps = LNC_ARG(ln_eps); r.M = LNC_ARG(M); r.N = LNC_ARG(N); r.ldo = LNC_ARG(ldo);
  r.ln_cap8 = LNC_ARG(ln_cap8); r.spin = LNC_ARG(lnc_spin);
  return r;
}
// one slot's (sum, centred sum of squares) joins the running (n, mean, M2) of a row (Chan et al.) — written out with contraction
// off: the in-kernel statistics (lnc_row_stats) and the clean-up kernel's (lnc_cleanup_kernel) must agree to the bit, whichever
// of the two normalises a segment
__device__ __forceinline__ void chan_join(float n, float& m, float& q, float sum, float m2) {
#pragma clang fp contract(off)
  const float w = 64.0f / (n + 64.0f);          // weight of the new slot in the union
  const float d = sum * (1.0f / 64.0f) - m;
  q = q + (m2 + (d * d) * (n * w));
  m = m + d * w;
}
__device__ __forceinline__ float chan_rstd(float q, float n, float eps) {
#pragma clang fp contract(off)
  return 1.0f / sqrtf(q / n + eps);
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
      chan_join(n, m0, q0, p[j][0], p[j][1]);
      chan_join(n, m1, q1, p[j][2], p[j][3]);
      n += 64.0f;
    }
  }
  const f32x4_t st = {m0, chan_rstd(q0, n, a.ln_eps), m1, chan_rstd(q1, n, a.ln_eps)};
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
  if (a.spin < 0) {  // the first form (R6.3): wait for the partners however long it takes
    int spins = 0;
    while (wave_l2_add_ret(reg, ctr_off, 0u, lane) < need) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1 << 19)) {  // never in a correct run: count it and go on (wrong rows beat a hung box)
        if (lane == 0) l2_atomic_add(reg, (uint32_t)(a.ln_cap8 + 2) * 4u, 1u);
        break;
      }
    }
  } else {
    // DEFER form (R6.4): poll at most a.spin times; a wave whose partners are not there yet (a panel whose tiles straddle two
    // rounds of the grid: 32 workgroups per XCD, 3 tiles per panel — or plain skew) does not wait: it leaves its 128 x 64
    // segment to lnc_cleanup_kernel, which finds the segment's bit missing in the panel half's mask, and goes on to its next
    // tile.  x and the moments are in memory already (A, B); only the LayerNorm output of the segment is owed.
    int spins = 0;
    bool ready;
    for (;;) {
      ready = wave_l2_add_ret(reg, ctr_off, 0u, lane) >= need;
      if (ready || spins >= a.spin) break;
      ++spins;
      __builtin_amdgcn_s_sleep(2);
    }
    if (!ready) return;
    if (lane == 0)
      asm volatile("global_atomic_or %0, %1, %2" ::"v"(ctr_off + (uint32_t)(a.ln_cap8 + 3) * 4u), "v"(1u << (nw >> 6)), "s"(reg) : "memory");
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

// FOLD (LayerNorm fold, 16-bit modes): EPI_RESID runs the producer epilogue (wave_epilogue_resid_fold), EPI_STORE /
// EPI_GELU the consumer form of wave_epilogue_lds; a wave then carries 6 registers of row / column data across its last
// compute phase instead of 16 bias registers.
// LNT: LayerNorm in the tail (above)
template <int PREC, int EPI, bool FOLD = false, bool LNT = false, bool LNC = false>
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(const GemmArgs a) {
  static_assert(!LNT || (EPI == EPI_RESID && PREC != MCM_PREC_F32 && !FOLD), "LNT: plain residual form");
  static_assert(!LNC || (EPI == EPI_RESID && PREC != MCM_PREC_F32 && !FOLD && !LNT), "LNC: plain residual form");
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
      const int q = i - 4;
      glds16s(tw + ko + (size_t)((q >> 1) * 64 + (q & 1) * 8) * sw, lk.voff_w, base + A_BYTES + q * 4096);
    }
  };
  constexpr int NP0 = 12;  // pieces per step of waves 0-3
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
        phase_barrier();
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
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int j = 0; j < 4; ++j) readf(lk, sr, 4 * i + j);
          piece(lk, si, i);
        }
      }
      issue_done();
    }
    pin_frags();
    PPT(0);
    if (!split) phase_barrier();
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
template <int PREC, int EPI, bool FOLD = false, bool LNT = false, bool LNC = false>
hipError_t launch_pp(const GemmArgs& a, hipStream_t s) {
  static PerDeviceFlag attr_set;
  if (!attr_set.get()) {
    hipError_t e = hipFuncSetAttribute((const void*)arms::gemm_pp_kernel<PREC, EPI, FOLD, LNT, LNC>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, p256::LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_set.set();
  }
  hipLaunchKernelGGL((arms::gemm_pp_kernel<PREC, EPI, FOLD, LNT, LNC>), dim3(persistent_grid()), dim3(512), p256::LDS_BYTES, s, a);
  return hipGetLastError();
}
// LNC defer form: one 4-wave workgroup per (XCD, row panel, half).  Reads the half's mask (bit = 64-column slot whose LayerNorm
// was written from registers), zeroes it, and — almost always — exits.  Otherwise: (mean, rstd) of the 128 rows from the slot
// moments (chan_join / chan_rstd: the in-kernel arithmetic), then the missing segments, dealt to the four waves: x from memory
// (the values the accumulators held), ((v - mean) rstd) gamma + beta with contraction off as in pass D, packed, stored.
// ln_state word 2 * ln_cap8 + 3 of the XCD's region counts the segments done here (sticky: mcm_debug_ln_cluster_deferred).
template <int PREC, int NS>
__global__ __launch_bounds__(256) void lnc_cleanup_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                          const float* __restrict__ b, void* __restrict__ y,
                                                          const float2* __restrict__ part, int M, float eps,
                                                          unsigned int* state, int ln_rs, int ln_cap8, unsigned int* sat) {
  enter_precision_mode<PREC>();
  constexpr int D = NS * 64;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int m0 = ((idx >> 1) * 8 + xcd) * 256 + (idx & 1) * 128;
  if (m0 >= M) return;
  unsigned int* reg = state + (size_t)xcd * ln_rs;
  __shared__ unsigned int smask;
  __shared__ float2 st[128];
  if (threadIdx.x == 0) {
    smask = reg[ln_cap8 + 3 + idx];
    reg[ln_cap8 + 3 + idx] = 0u;
  }
  __syncthreads();
  const unsigned int miss = ~smask & ((1u << NS) - 1u);
  if (!miss) return;
  if (threadIdx.x < 128) {
    const int r = threadIdx.x;
    float n = 0.f, m = 0.f, q = 0.f;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      const float2 pj = part[(size_t)j * M + m0 + r];
      chan_join(n, m, q, pj.x, pj.y);
      n += 64.0f;
    }
    st[r] = make_float2(m, chan_rstd(q, n, eps));
  }
  if (threadIdx.x == 0) atomicAdd(reg + 2 * ln_cap8 + 3, (unsigned int)__popc(miss));
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rr = lane >> 4, c4 = (lane & 15) * 4;
  float amax = 0.f;
  int k = 0;
  for (int sg = 0; sg < NS; ++sg) {
    if (!((miss >> sg) & 1u)) continue;
    if ((k++ & 3) != wave) continue;
    const int c0 = sg * 64 + c4;
    const float4 gv = *(const float4*)(g + c0), bv = *(const float4*)(b + c0);
#pragma unroll 4
    for (int i = 0; i < 32; ++i) {
      const int r = i * 4 + rr;
      const float4 v = *(const float4*)(x + (size_t)(m0 + r) * D + c0);
      const float2 s2 = st[r];
      float4 o;
      {
#pragma clang fp contract(off)
        o.x = __builtin_fmaf((v.x - s2.x) * s2.y, gv.x, bv.x);
        o.y = __builtin_fmaf((v.y - s2.x) * s2.y, gv.y, bv.y);
        o.z = __builtin_fmaf((v.z - s2.x) * s2.y, gv.z, bv.z);
        o.w = __builtin_fmaf((v.w - s2.x) * s2.y, gv.w, bv.w);
      }
      sat_track<PREC>(amax, o.x, o.y);
      sat_track<PREC>(amax, o.z, o.w);
      uint2 pk;
      pk.x = pack2<PREC>(o.x, o.y);
      pk.y = pack2<PREC>(o.z, o.w);
      *(uint2*)((uint16_t*)y + (size_t)(m0 + r) * D + c0) = pk;
    }
  }
  sat_report<PREC>(amax, sat);
}
template <int PREC>
hipError_t launch_lnc_cleanup_p(const float* x, const float* g, const float* b, void* y, const float2* part, int M, int D, float eps,
                                unsigned int* ln_state, int ln_rs, int ln_cap8, hipStream_t s, unsigned int* sat) {
  const int idxs = 2 * ((M / 256 + 7) / 8);  // (panel, half) pairs per XCD region
  if (M % 256 || idxs > ln_cap8 || ln_rs < 2 * ln_cap8 + 4) return hipErrorInvalidValue;
  if (D == 768) hipLaunchKernelGGL((lnc_cleanup_kernel<PREC, 12>), dim3(8 * idxs), dim3(256), 0, s, x, g, b, y, part, M, eps, ln_state, ln_rs, ln_cap8, sat);
  else if (D == 1024) hipLaunchKernelGGL((lnc_cleanup_kernel<PREC, 16>), dim3(8 * idxs), dim3(256), 0, s, x, g, b, y, part, M, eps, ln_state, ln_rs, ln_cap8, sat);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

// =========================================================================================
// ROW64 (round 6, EXPERIMENTS.md R6.7): the LITERAL full-row tile of VERDICT r5 item 2 — BM = 64 rows x BN = N (768 / 1024)
// in ONE workgroup, so that the residual GEMM's epilogue owns whole rows: (acc + bias) + resid -> x (written once), exact
// two-pass row statistics over the workgroup's 8 waves, LayerNorm written from the accumulator registers.  No LayerNorm
// launch, no re-read of x, no cross-workgroup traffic at all (LNC's cost).  Built to be MEASURED (R6.7: -21.5 %, TA-bound K loop);
// 8 waves side by side, wave w = columns [w N/8, (w+1) N/8) of all 64 rows: 4 x FW accumulator tiles (96 / 128 registers).
// K-step = 32 elements (64 B per row): W is wave-PRIVATE (only wave w reads W rows [w N/8, ...)), double-buffered per wave by
// LDS-DMA and waited for with the wave's own vmcnt — no barrier; X (64 rows, shared) is staged XG K-steps at a time by waves
// 0 - 3, one barrier per XG K-steps.  LDS piece = 16 rows x 64 B; the 16-byte chunk c of row r sits at slot c ^ (r >= 8 ? 3 : 0)
// (applied to the lane's global source address, the DMA destination is lane-linear).  That XOR is made for ds_read_b128's REAL
// lane groups — {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS): with lane (fr, g) reading chunk g of
// row fr, a group holds rows {0-3, 12-15} at chunk g0 and rows {4-11} at chunk g0 ^ 1, and the four rows that share a 64-byte
// column of the 256-byte bank row (r, r+4, r+8, r+12) land on slots g0, g0 ^ 1, g0 ^ 2, g0 ^ 3.  (The first cut XOR-ed by
// (r >> 2) & 3 — conflict-free for contiguous 16-lane groups, 2-way for the real ones: SQ_LDS_BANK_CONFLICT 48 %, R6.7.)
// Persistent over 64-row tiles.
// Statistics: per-wave partial sums in a fixed order, exchanged through LDS — two-pass like the LayerNorm kernel, another
// summation order: scores equal the launched path's to fp32 round-off (as LNC), not bit for bit.
// =========================================================================================
namespace row64 {
constexpr int BM = 64, KS = 64;   // rows per tile; bytes of K per row per K-step
template <int FW, int NST> constexpr int xg() { return NST == 3 ? 2 : FW <= 6 ? 4 : 2; }    // K-steps per X group
template <int FW, int NST> constexpr int xbuf() { return BM * KS * xg<FW, NST>(); }         // bytes of one X group
template <int FW> constexpr int wstage() { return FW * 1024; }                             // bytes of one wave's W stage
// NST = 3 (W two K-steps ahead; N = 768 only): 16 + 144 KiB = all of the CU's LDS, the statistics scratch then aliases X group 0
template <int FW, int NST> constexpr int lds_bytes() { return 2 * xbuf<FW, NST>() + 8 * NST * wstage<FW>() + (NST == 3 ? 0 : 2 * 8 * BM * 4); }
}  // namespace row64

template <int PREC, int FW, int NST>
__global__ __launch_bounds__(512, 2) void gemm_row64_ln_kernel(const GemmArgs a) {
  using namespace row64;
  static_assert(PREC != MCM_PREC_F32, "16-bit operand modes");
  static_assert(NST == 2 || NST == 3, "W stages per wave");
  enter_precision_mode<PREC>();
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int XG = xg<FW, NST>(), XBUF = xbuf<FW, NST>(), WST = wstage<FW>(), NW = FW * 16, N = 8 * NW;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, g = lane >> 4;
  const int nk = a.K * 2 / KS;           // K-steps (a.K 16-bit elements per row)
  const int ntiles = a.M / BM;
  const uint32_t lds0 = lds_addr(smem);
  const uint32_t wlds = lds0 + 2 * XBUF + wave * NST * WST;
  float* scratch = (float*)(NST == 3 ? smem : smem + 2 * XBUF + 8 * NST * WST);   // [2][8 waves][64 rows]; NST = 3: over X group 0
  // DMA source of this lane inside a piece: row r = lane >> 2, slot lane & 3 holds chunk (lane & 3) ^ ((r >> 2) & 3)
  const int pr = lane >> 2, pc = (lane & 3) ^ ((pr & 8) ? 3 : 0);
  const size_t sx = (size_t)a.ldx * 2, sw = (size_t)a.K * 2;
  // W pieces: rows of the [N, K] operand (16 half cache lines per piece), or — a.wblk — the blocked image, where piece
  // (K-step kt, 16-row block b) is the 1 KiB at ((kt * N / 16) + b) * 1024, already in LDS order: eight whole cache lines
  const char* wsrc = a.wblk ? (const char*)a.w + (size_t)(wave * FW) * 1024 + lane * 16
                            : (const char*)a.w + (size_t)(wave * NW + pr) * sw + pc * 16;
  const size_t wpf = a.wblk ? (size_t)1024 : (size_t)16 * sw;            // bytes from a wave's fragment f to f + 1
  const size_t wpk = a.wblk ? (size_t)(N / 16) * 1024 : (size_t)KS;      // ... from K-step kt to kt + 1
  const int foff = fr * 64 + ((g ^ ((fr & 8) ? 3 : 0)) << 4);
  float amax = 0.f;

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int m0 = tile * BM;
    const char* xsrc = (const char*)a.x + (size_t)(m0 + pr) * sx + pc * 16;
    auto issue_w = [&](int kt) {
      const uint32_t base = wlds + (kt % NST) * WST;
#pragma unroll
      for (int f = 0; f < FW; ++f) glds16(wsrc + f * wpf + kt * wpk, base + f * 1024);
    };
    auto issue_x = [&](int gi) {   // waves 0 - 3: row block `wave` of the XG K-steps of group gi
      const uint32_t base = lds0 + (gi & 1) * XBUF;
#pragma unroll
      for (int j = 0; j < XG; ++j)
        glds16(xsrc + (size_t)(wave * 16) * sx + (size_t)(gi * XG + j) * KS, base + (j * 4 + wave) * 1024);
    };
    f32x4_t acc[4][FW];
#pragma unroll
    for (int fx = 0; fx < 4; ++fx)
#pragma unroll
      for (int fw = 0; fw < FW; ++fw) acc[fx][fw] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    if (wave < 4) issue_x(0);
    issue_w(0);
    if constexpr (NST == 3) issue_w(1);   // (nk >= 2 always: K >= 64)
    for (int kt = 0; kt < nk; ++kt) {
      const int j = kt % XG, gi = kt / XG;
      // W stays NST - 1 K-steps ahead; the wait lets exactly the pieces of the steps after kt stay in flight (X pieces are issued
      // at group starts, XG steps before their first reader: always among the completed)
      if (kt + NST - 1 < nk) {
        issue_w(kt + NST - 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 1) * FW) : "memory");
      } else if (NST == 3 && kt + 1 < nk) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FW) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (j == 0) {
        __builtin_amdgcn_s_barrier();   // X group gi is in LDS for everyone; everyone is done with group gi - 1
        asm volatile("" ::: "memory");
        if ((gi + 1) * XG < nk && wave < 4) issue_x(gi + 1);
      }
      const char* xb = smem + (gi & 1) * XBUF + j * 4096 + foff;
      const char* wb = smem + 2 * XBUF + wave * NST * WST + (kt % NST) * WST + foff;
      uint4 xf[4], wf[FW];
#pragma unroll
      for (int fx = 0; fx < 4; ++fx) xf[fx] = *(const uint4*)(xb + fx * 1024);
#pragma unroll
      for (int fw = 0; fw < FW; ++fw) wf[fw] = *(const uint4*)(wb + fw * 1024);
#pragma unroll
      for (int fw = 0; fw < FW; ++fw)
#pragma unroll
        for (int fx = 0; fx < 4; ++fx) acc[fx][fw] = mfma16<PREC>(wf[fw], xf[fx], acc[fx][fw]);
    }
    // ---- epilogue: lane (fr, g) holds rows m0 + fx*16 + fr, columns wave*NW + fw*16 + g*4 .. +3
    if constexpr (NST == 3) __syncthreads();   // (the scratch lies over X group 0: every wave has read its last fragments)
    const int n0 = wave * NW + g * 4;
    float rs[4] = {0.f, 0.f, 0.f, 0.f};
    {
#pragma clang fp contract(off)
#pragma unroll
      for (int fw = 0; fw < FW; ++fw) {
        const f32x4_t bia = *(const f32x4_t*)(a.bias + n0 + fw * 16);
#pragma unroll
        for (int fx = 0; fx < 4; ++fx) {
          float* xr = a.resid + (size_t)(m0 + fx * 16 + fr) * a.ldo + n0 + fw * 16;
          const f32x4_t v = (acc[fx][fw] + bia) + *(const f32x4_t*)xr;
          *(f32x4_t*)xr = v;
          acc[fx][fw] = v;
          rs[fx] += (v[0] + v[1]) + (v[2] + v[3]);
        }
      }
    }
    auto exchange = [&](float (&part)[4], int buf) {   // wave partials -> row totals (fixed order over the waves)
#pragma clang fp contract(off)
#pragma unroll
      for (int fx = 0; fx < 4; ++fx) {
        part[fx] += __shfl_xor(part[fx], 16, 64);
        part[fx] += __shfl_xor(part[fx], 32, 64);
      }
      float* sc = scratch + buf * 8 * BM;
      if (g == 0) {
#pragma unroll
        for (int fx = 0; fx < 4; ++fx) sc[wave * BM + fx * 16 + fr] = part[fx];
      }
      __syncthreads();
#pragma unroll
      for (int fx = 0; fx < 4; ++fx) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) t += sc[w * BM + fx * 16 + fr];
        part[fx] = t;
      }
    };
    exchange(rs, 0);
    float mean[4], q[4];
    {
#pragma clang fp contract(off)
#pragma unroll
      for (int fx = 0; fx < 4; ++fx) {
        mean[fx] = rs[fx] / (float)N;
        q[fx] = 0.f;
#pragma unroll
        for (int fw = 0; fw < FW; ++fw) {
          const f32x4_t d = acc[fx][fw] - mean[fx];
          acc[fx][fw] = d;
          q[fx] += __builtin_fmaf(d[0], d[0], d[1] * d[1]) + __builtin_fmaf(d[2], d[2], d[3] * d[3]);
        }
      }
    }
    exchange(q, 1);
    {
#pragma clang fp contract(off)
      float rstd[4];
#pragma unroll
      for (int fx = 0; fx < 4; ++fx) rstd[fx] = 1.0f / sqrtf(q[fx] / (float)N + a.ln_eps);
#pragma unroll
      for (int fw = 0; fw < FW; ++fw) {
        const f32x4_t gam = *(const f32x4_t*)(a.ln_g + n0 + fw * 16), bet = *(const f32x4_t*)(a.ln_b + n0 + fw * 16);
#pragma unroll
        for (int fx = 0; fx < 4; ++fx) {
          f32x4_t y;
#pragma unroll
          for (int e = 0; e < 4; ++e) y[e] = __builtin_fmaf(acc[fx][fw][e] * rstd[fx], gam[e], bet[e]);
          sat_track<PREC>(amax, y[0], y[1]);
          sat_track<PREC>(amax, y[2], y[3]);
          uint2 pk;
          pk.x = pack2<PREC>(y[0], y[1]);
          pk.y = pack2<PREC>(y[2], y[3]);
          *(uint2*)((uint16_t*)a.ln_y + (size_t)(m0 + fx * 16 + fr) * N + n0 + fw * 16) = pk;
        }
      }
    }
    __syncthreads();   // the scratch and the X buffers are free for the next tile
  }
  sat_report<PREC>(amax, a.sat);
}
// [N, K] 16-bit operand -> the blocked piece image (one thread per 16-byte chunk)
__global__ void row64_block_w_kernel(const uint4* __restrict__ w, uint4* __restrict__ out, int N, int K) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // destination chunk index
  const size_t nchunks = (size_t)N * K / 8;
  if (i >= nchunks) return;
  const int lane = (int)(i & 63);
  const size_t piece = i >> 6;
  const int nb = N / 16, b = (int)(piece % nb), kt = (int)(piece / nb);
  const int r = lane >> 2, c = (lane & 3) ^ ((r & 8) ? 3 : 0);
  out[i] = w[((size_t)(b * 16 + r) * K + (size_t)kt * 32) / 8 + c];
}
template <int PREC, int FW, int NST>
hipError_t launch_row64_ln_f(const GemmArgs& a, hipStream_t s) {
  static PerDeviceFlag attr_set;
  if (!attr_set.get()) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_row64_ln_kernel<PREC, FW, NST>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       row64::lds_bytes<FW, NST>());
    if (e != hipSuccess) return e;
    attr_set.set();
  }
  const int ntiles = a.M / row64::BM, grid = ntiles < persistent_grid() ? ntiles : persistent_grid();
  constexpr int lds = row64::lds_bytes<FW, NST>();
  hipLaunchKernelGGL((gemm_row64_ln_kernel<PREC, FW, NST>), dim3(grid), dim3(512), lds, s, a);
  return hipGetLastError();
}
// stages: 2 or 3 W stages per wave (3: N = 768 only — 160 KiB of LDS; other widths run 2)
template <int PREC>
hipError_t launch_row64_ln_p(const GemmArgs& a, hipStream_t s, int stages) {
  if (a.M <= 0 || a.M % row64::BM || a.K % 128 || a.ldo != a.N || a.xsplit || a.ksplit || !a.bias || !a.resid || !a.ln_g || !a.ln_b ||
      !a.ln_y)
    return hipErrorInvalidValue;
  if (a.N == 768) return stages == 3 ? launch_row64_ln_f<PREC, 6, 3>(a, s) : launch_row64_ln_f<PREC, 6, 2>(a, s);
  if (a.N == 1024) return launch_row64_ln_f<PREC, 8, 2>(a, s);
  return hipErrorInvalidValue;
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
        *err = launch_pp<PREC, EPI, true>(a, s);
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
        *err = launch_pp<PREC, EPI, false, false, true>(a, s);
    }
    return true;
  }
  if (a.ln_y) {  // LayerNorm in the tail: the ping-pong kernel's residual form only (the caller asked gemm_ln_tail_ok)
    if constexpr (EPI == EPI_RESID && PREC != MCM_PREC_F32) {
      if (v == 5 && whole && (a.N == 768 || a.N == 1024) && a.ldo == a.N && a.ln_g && a.ln_b && a.ln_state &&
          a.ln_cap8 * 8 >= a.M / p256::BM && persistent_grid() % 8 == 0)
        *err = launch_pp<PREC, EPI, false, true>(a, s);
    }
    return true;
  }
#ifdef MCM_HARNESS
  if (v == 9) {  // the flagged text of the ping-pong kernel with every flag off (honours mcm_debug_gemm_group_n): whole tiles, else as 5
    if constexpr (EPI != EPI_PATCH) {
      if (whole) { *err = launch_pp<PREC, EPI>(a, s); return true; }
    }
    v = 5;
  }
#endif
  return false;
}

}  // namespace arms
