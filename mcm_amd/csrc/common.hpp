// common.hpp — shared types of the gfx950 MCM kernels (internal; the public boundary is
// include/mcm.h).  CDNA4 only: 64-lane wavefronts, MFMA, 160 KiB LDS, no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mcm.h"

// hipFuncSetAttribute acts on the CURRENT device's copy of a kernel: a launcher's "already raised the dynamic-LDS limit"
// flag is kept per device, so that a process holding handles on two devices sets it on both.
struct PerDeviceFlag {
  bool done[64] = {};
  static int dev() { int d = 0; return hipGetDevice(&d) == hipSuccess && d >= 0 && d < 64 ? d : -1; }
  bool get() const { const int d = dev(); return d >= 0 && done[d]; }
  void set() { const int d = dev(); if (d >= 0) done[d] = true; }
};

// CUs of the CURRENT device, cached per device (a process may hold handles on several devices, and a partitioned or
// CU-masked lease reports fewer than 256): the grid of every persistent kernel and the size policies that depend on it
// (a process-wide static used the first device's count everywhere).  0 on error.
inline int device_cu_count() {
  static int cus[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) return 0;
  if (dev < 64 && cus[dev] > 0) return cus[dev];
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  if (dev < 64) cus[dev] = n;
  return n;
}

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// fp32 → bf16, round-to-nearest-even (NaN kept quiet)
__device__ __forceinline__ uint16_t f2bf(float f) {
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf2f(uint16_t h) {
  return __builtin_bit_cast(float, (uint32_t)h << 16);
}
// two fp32 → packed bf16x2 (lo in bits 0-15), round-to-nearest-even, in ONE instruction: the vector
// conversion selects gfx950's v_cvt_pk_bf16_f32 (the software f2bf above costs ~6 VALU per element and
// was a large share of every bf16 epilogue).  Deliberately NOT inline asm: hipcc inserts no wait states
// between an asm statement's VALU write and an MFMA that reads the value as an operand.
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
  typedef float f2_t __attribute__((ext_vector_type(2)));
  const f2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf2_t));
}

// LDS-DMA, 16 B per lane: lane i's 16 bytes at `gsrc` land at LDS byte address lds_base + 16*i
// (lds_base wave-uniform).  Inline asm on purpose: hipcc drains vmcnt(0) before ANY ds_read
// that follows a __builtin_amdgcn_global_load_lds (it cannot prove the LDS read does not
// alias the DMA destination), which serialises a multi-stage pipeline.  An asm DMA is
// invisible to that pass; the kernel counts it itself: s_waitcnt vmcnt(N) → s_barrier →
// ds_read (guide §5.7).  M0 is compiler-reserved, so it is saved and restored here.
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_base) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_base)
      : "memory");
}
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

// two fp32 → packed fp16x2, round-to-nearest-even: one v_cvt_pk_f16_f32 on gfx950 (the vector
// conversion selects it; two scalar casts cost v_cvt_f16_f32 x2 + v_pack_b32_f16)
__device__ __forceinline__ uint32_t pack_h2(float lo, float hi) {
  typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
  typedef float f2_t __attribute__((ext_vector_type(2)));
  const f2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, h2_t));
}
// fp16 overflow guard.  MODE.FP16_OVFL (bit 23) makes every f32 -> f16 conversion of the wave saturate
// finite out-of-range values to +-65504 instead of +-inf (inf and NaN inputs stay what they are).
// Measured on gfx950 (tools/isa_probe.hip, removed in round 6: git history): v_cvt_pk_f16_f32 honours it — 65520, 7e4, 1e6, 3.4e38 all
// give 65504.  One scalar instruction per wave, nothing per element: an activation that leaves the
// fp16 range (a QuickGELU output or a q/k/v value above 65504) costs that element's precision, not an
// inf that turns the whole row into NaN at the next LayerNorm.  Set once at kernel entry by every kernel
// that writes fp16.
template <int PREC>
__device__ __forceinline__ void enter_precision_mode() {
  if constexpr (PREC == MCM_PREC_F16) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
}

// fp16 saturation watch.  FP16_OVFL makes an out-of-range activation cost that element's precision instead
// of turning a row into NaN — silently.  Every kernel that packs fp32 values into fp16 activations therefore
// keeps, per lane, the running max |value| it packed (one v_max3_f32 with |.| modifiers per PAIR, fp16 mode
// only) and reports at its end: if any lane of a wave packed a value that saturated (|v| >= 65520, the first
// value that rounds above 65504), one lane adds 1 to the handle's sticky device counter (mcm_saturation_count).
// The atomic is inline asm like every other global access of the GEMM epilogues (invisible to hipcc's waitcnt
// pass) and is the last VMEM instruction of its wave.
template <int PREC>
__device__ __forceinline__ void sat_track(float& amax, float a, float b) {
#ifndef MCM_NO_SAT_TRACK  // (defined only for an overhead A/B build: make CXXFLAGS+=-DMCM_NO_SAT_TRACK)
  if constexpr (PREC == MCM_PREC_F16) amax = fmaxf(fmaxf(__builtin_fabsf(a), __builtin_fabsf(b)), amax);
#endif
}
template <int PREC>
__device__ __forceinline__ void sat_report(float amax, unsigned int* counter) {
  if constexpr (PREC == MCM_PREC_F16) {
    if (counter != nullptr && __builtin_amdgcn_ballot_w64(amax >= 65520.0f) != 0) {
      if ((threadIdx.x & 63) == 0) {
        const unsigned int one = 1u;
        asm volatile("global_atomic_add %0, %1, off" ::"v"(counter), "v"(one) : "memory");
      }
    }
  }
}

// 16-bit operand modes: PREC selects the element format of MFMA operands and 16-bit outputs
template <int PREC>
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  if constexpr (PREC == MCM_PREC_F16) return pack_h2(lo, hi);
  else return pack_bf2(lo, hi);
}
template <int PREC>
__device__ __forceinline__ f32x4_t mfma16(uint4 a, uint4 b, f32x4_t c) {  // 16x16x32, fp32 acc
  if constexpr (PREC == MCM_PREC_F16)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a),
                                                  __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                   __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// QuickGELU: x * sigmoid(1.702 x)  (transformers activations.py:117-123)
__device__ __forceinline__ float quick_gelu(float x) { return x / (1.0f + __expf(-1.702f * x)); }
// bf16-output form: v_exp_f32 + v_rcp_f32 (≈1 ulp each) instead of the IEEE division expansion
__device__ __forceinline__ float quick_gelu_fast(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * x));
}

// wave64 butterfly reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- LayerNorm fold: moments of a 64-column segment held by the 16 lanes of a DPP row (4 columns per lane) ----------
// Used by the residual epilogue of the ping-pong GEMM and by fold_rows_kernel (layernorm.hip), which must produce the
// SAME bits for the same row (a score must not depend on which kernel the batch size selected): every operation is an
// explicitly rounded one, in a fixed order, so contraction cannot differ between the two call sites.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float x) {  // x of the lane the DPP control selects (within a row of 16)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {  // sum over the 16 lanes of a DPP row, in every lane
#pragma clang fp contract(off)
  v = v + dpp_move<0xB1>(v);   // quad_perm [1,0,3,2]
  v = v + dpp_move<0x4E>(v);   // quad_perm [2,3,0,1]
  v = v + dpp_move<0x141>(v);  // row_half_mirror
  v = v + dpp_move<0x140>(v);  // row_mirror
  return v;
}
// sum of the segment and its sum of squares about the segment's own mean (every lane of the row gets both)
__device__ __forceinline__ void slot_moments(const f32x4_t& v, float& sum, float& m2) {
#pragma clang fp contract(off)  // (hipcc's default would fuse differently at different call sites)
  sum = row16_sum((v[0] + v[1]) + (v[2] + v[3]));
  const float mean = sum * (1.0f / 64.0f);
  const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
  m2 = row16_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
}
// z = gamma o x, the fold's A operand, before it is packed to the operand dtype
__device__ __forceinline__ f32x4_t fold_scale(const f32x4_t& v, const f32x4_t& gamma) {
#pragma clang fp contract(off)
  return v * gamma;
}
// consumer side of the fold: (acc - mean c) rstd + b' as two fused multiply-adds, the same in every GEMM kernel
__device__ __forceinline__ float fold_apply(float acc, float rstd, float mrstd, float c, float b) {
  return __builtin_fmaf(acc, rstd, __builtin_fmaf(-mrstd, c, b));
}

// split pair of a 16-bit operand dtype (split weights / split activations): hi = round(v), lo = round(v - hi); v - hi is
// exact in fp32 (hi is v's own leading bits)
template <int PREC>
__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
  hi = pack2<PREC>(a, b);
  float ha, hb;
  if constexpr (PREC == MCM_PREC_F16) {
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    const h2_t h = __builtin_bit_cast(h2_t, hi);
    ha = (float)h[0];
    hb = (float)h[1];
  } else {
    ha = __builtin_bit_cast(float, hi << 16);
    hb = __builtin_bit_cast(float, hi & 0xffff0000u);
  }
  lo = pack2<PREC>(a - ha, b - hb);
}
// element offset of logical column n inside a split row (per 64 columns: hi[64] then lo[64]); the lo element is 64 further
__host__ __device__ constexpr size_t split_col(int n) { return (size_t)(n >> 6) * 128 + (n & 63); }

// element size in bytes of the MFMA operand dtype of a precision mode
__host__ __device__ constexpr int prec_esize(int prec) { return prec == MCM_PREC_F32 ? 4 : 2; }

// ---- host launchers (one per kernel family; defined in the .hip files) ----------------

enum GemmEpi : int {
  EPI_STORE = 0,   // out[M,N] (operand dtype) = acc + bias
  EPI_GELU = 1,    // out = QuickGELU(acc + bias)
  EPI_RESID = 2,   // resid[M,N] (fp32) += acc + bias
  EPI_PATCH = 3,   // x[b*(np+1)+1+p, :] (fp32) = acc + pos[1+p, :]   (m = b*np + p)
  // split-activation arm (fp16 only; GemmArgs::xsplit): the store epilogues with the output written as a SPLIT image,
  // the next GEMM's X operand — out[M, 2N]: per 64 columns hi[64] = round(v) then lo[64] = round(v - hi)
  EPI_STORE_X2 = 4,
  EPI_GELU_X2 = 5,  // QuickGELU in its exact form (the fp32 arm's), then split
};
__host__ __device__ constexpr bool epi_x2(int e) { return e == EPI_STORE_X2 || e == EPI_GELU_X2; }
__host__ __device__ constexpr bool epi_store16(int e) { return e <= EPI_GELU || epi_x2(e); }  // 16-bit [M, N] / [M, 2N] outputs
__host__ __device__ constexpr bool epi_gelu(int e) { return e == EPI_GELU || e == EPI_GELU_X2; }

struct GemmArgs {
  const void* x;      // [M, K] operand dtype, row stride ldx elements
  const void* w;      // [N, K] operand dtype, row stride K
  const float* bias;  // [N] or nullptr
  void* out;          // EPI_STORE / EPI_GELU: [M, N] operand dtype; EPI_PATCH: fp32 x
  float* resid;       // EPI_RESID: [M, N] fp32
  const float* pos;   // EPI_PATCH: position embedding [np+1, N]
  int M, N, K;
  int ldx;            // elements
  int ldo;            // elements (row stride of out / resid)
  int np;             // EPI_PATCH: patches per image
  const float* px;    // EPI_PATCH: NCHW fp32 pixels [B,3,img,img] to gather the A operand from (nullptr: read `x`, the patch
  int img, patch;     //   matrix patchify wrote); image and patch size.  See gemm_patch_takes_pixels / gemm_p256_kernel
  int gn;             // persistent kernel: N-tiles per L2 group (0 = default)
  int rev;            // persistent kernel: walk the M tiles from the last to the first
  int dbg;            // ablation bits, read only in -DMCM_HARNESS builds: 1 no refill, 2 no MFMA, 4 no epilogue
  unsigned int* sat;  // fp16 saturation counter of the handle (nullptr: not reported), see sat_report
  // Split weights (16-bit modes; gemm.hip "Split weights"): 1 = `w` is the [N, K] image cvt_weight_split writes for a
  // logical [N, K/2] weight — per 128-byte K-step the fp16/bf16 rounding W_hi of the fp32 weight followed by the
  // rounding W_lo of the remainder — and `x` is the logical [M, K/2] operand (ldx counts its elements): K-steps 2s and
  // 2s+1 of W meet K-step s of X, so acc = X W_hi^T + X W_lo^T in one fp32 accumulator chain.  0 = plain operands.
  int ksplit;
  // Split activations (fp16 mode; the re-scoring arm): 1 = `x` is the SPLIT image of a logical [M, K]
  // activation — [M, 2K], per 128-byte K-step X_hi[64] = round(x) followed by X_lo[64] = round(x - X_hi), written by the
  // producing kernel (LayerNorm, attention, the *_X2 GEMM epilogues, patchify) — and ldx counts the image's elements.
  // K-steps 2s (hi) and 2s+1 (lo) of X meet K-step s of W: acc = X_hi W^T + X_lo W^T, ~22 significand bits of the
  // activation against an exact fp16 weight, in the unchanged K loop.  With ksplit as well: X (hi, lo) x W (hi, lo), four
  // passes per logical K-step.  The kernels' K index: t = kt >> (ksplit + xsplit); X step (t << xsplit) | (kt & xsplit),
  // W step (t << ksplit) | ((kt >> xsplit) & ksplit).
  int xsplit;
  // LayerNorm fold (16-bit modes, ping-pong kernel, gemm.hip "LayerNorm fold"): the LayerNorm between a residual GEMM
  // and the GEMM that consumes its output is not launched; both sides are set or null together per GEMM.
  void* fold_z;           // EPI_RESID (producer): [M, N] operand dtype, z = gamma o (new residual row)
  const float* fold_g;    //   gamma [N] of the LayerNorm that follows
  float2* fold_part;      //   per-row partial moments [N / 64][M]: (sum, sum of squares about the 64-column mean)
  const float2* fold_rs;  // EPI_STORE / EPI_GELU (consumer): per-row (rstd, mean * rstd) [M]
  const float* fold_c;    //   c [N] = W gamma; `bias` then holds b + W beta
  // 16-bit outputs, head-major (0 = row-major with stride ldo): column block n >> 6 — the 64 dims of one head of q, k
  // or v — is a contiguous [hm rows][64] array, element (m, n) at ((n >> 6) * hm + m) * 64 + (n & 63).  The QKV
  // projection writes this form so that an attention workgroup's K / V / Q rows are 25 KiB of consecutive bytes
  // instead of 197 segments of 128 B at a 4.6-KB stride.  An A/B arm: the shipped library passes 0.
  int hm;
  // LayerNorm in the tail (EPI_RESID, ping-pong kernel, 16-bit modes; gemm.hip "LayerNorm in the tail"): the LayerNorm
  // that follows this residual GEMM is computed by the GEMM kernel's own waves once they run out of tiles
  const float* ln_g;       // gamma [N]; ln_y == nullptr: not fused, the caller launches the LayerNorm
  const float* ln_b;       // beta [N]
  void* ln_y;              // LayerNorm output [M, N] in the operand dtype, row stride N
  float ln_eps;
  unsigned int* ln_state;  // 8 regions (one per XCD) of ln_rs words: [ln_cap8] row-tile counters, tickets, finished
  int ln_rs, ln_cap8;      //   workgroups, timeouts; all zero between launches (the kernel leaves it so)
  // LayerNorm by the row panel's CLUSTER (round 6 arm, gemm_arms.hpp "LNC"; needs ln_g / ln_b / ln_y / ln_state and fold_part):
  // 1 = the N / 256 workgroups that hold the tiles of one 256-row panel act as ONE full-row tile — each keeps its new residual
  // values in the accumulator registers, publishes its 64-column row moments (fold_part), waits for the panel's other
  // workgroups through a counter in ln_state, combines the N / 64 slot moments of its rows and writes the LayerNorm output
  // from registers: no LayerNorm launch and no re-read of the residual stream.  0 = off (the shipped library never sets it).
  int lnc;
  // LNC, defer form (R6.4): polls of the partners' counter a wave spends before it leaves its segment's LayerNorm to
  // launch_lnc_cleanup (which the caller must then launch behind the GEMM); < 0 = the first form: wait (bounded, counted)
  int lnc_spin;
  // ROW64 arm: 1 = `w` is the blocked image launch_row64_block_w writes ([K-step of 32][16-row block][1 KiB piece])
  int wblk;
};
// head-major rows of a 16-bit output as the kernels see them: the shipped library never sets GemmArgs::hm (an A/B arm of
// the harness library), so outside -DMCM_HARNESS builds the layout tests fold away at compile time
#ifdef MCM_HARNESS
#define MCM_HM(hm) (hm)
#else
#define MCM_HM(hm) 0
#endif
// element offset of (m, n) in a 16-bit GEMM output
__device__ __forceinline__ size_t out16_off(const GemmArgs& a, int m, int n) {
  return MCM_HM(a.hm) ? ((size_t)(n >> 6) * a.hm + m) * 64 + (n & 63) : (size_t)m * a.ldo + n;
}
hipError_t launch_gemm(int prec, int epi, const GemmArgs& a, hipStream_t s);
// which LayerNorm-fold form launch_gemm has for this problem: 1 = ping-pong kernel (producer and consumer epilogues),
// 2 = tile kernel (consumer epilogue only; its producer is launch_fold_rows after the plain residual GEMM), 0 = none
int gemm_fold_kind(int epi, int M, int N);
bool gemm_patch_takes_pixels(int prec, int M, int N, int kpad, int patch, int image);
bool gemm_ln_tail_ok(int prec, int M, int N);
int gemm_persistent_grid();
#ifdef MCM_HARNESS
void gemm_set_persistent_grid(int n);  // A/B: n workgroups (a multiple of 8) instead of one per CU; 0 restores
#endif
// LayerNorm fold, weight side: c[n] = sum_k gamma[k] W[n,k], bfold[n] = bias[n] + sum_k beta[k] W[n,k]  (W: operand dtype)
hipError_t launch_fold_prep(int prec, const void* w, const float* gamma, const float* beta, const float* bias,
                            float* c, float* bfold, int N, int K, hipStream_t s);
// LayerNorm fold, producer side without a fused epilogue (problems the tile kernel takes): z = gamma o x in the operand
// dtype and the row moments, bit-identical to what the ping-pong kernel's residual epilogue writes
hipError_t launch_fold_rows(int prec, const float* x, const float* gamma, void* z, float2* part, int M, int D,
                            hipStream_t s, unsigned int* sat = nullptr);
// LayerNorm fold, row side: partial moments [slots][M] -> (rstd, mean * rstd) [M]
hipError_t launch_fold_stats(const float2* part, int slots, int M, int D, float eps, float2* rs, hipStream_t s);
#ifdef MCM_HARNESS  // tools/gemm_bench.hip and libmcm_hip_harness.so only
// LNC defer form: the segments (128 rows x 64 columns) whose LayerNorm the residual GEMM's waves left behind — read from the
// masks in ln_state, normalised from x and the published slot moments with the in-kernel arithmetic (same bits), masks zeroed
hipError_t launch_lnc_cleanup(int prec, const float* x, const float* g, const float* b, void* y, const float2* part, int M, int D,
                              float eps, unsigned int* ln_state, int ln_rs, int ln_cap8, hipStream_t s, unsigned int* sat);
// ROW64 arm (gemm_arms.hpp, R6.7): out-proj / fc2 as 64-row FULL-ROW tiles whose epilogue writes x and the LayerNorm output
// (GemmArgs: x, w, bias, resid, M % 64 == 0, N in {768, 1024} = ldo, K, ldx, ln_g, ln_b, ln_y, ln_eps, sat)
hipError_t launch_gemm_row64_ln(int prec, const GemmArgs& a, hipStream_t s, int stages);   // stages: W stages per wave, 2 or 3
hipError_t launch_row64_block_w(const void* w, void* blocked, int N, int K, hipStream_t s);   // 16-bit [N, K] -> the ROW64 piece image
void gemm_set_group_n(int gn);
void gemm_set_dbg(int d);
void gemm_set_variant(int v);  // -1 auto (the shipped policy), 0 ... 8: see gemm.hip
void attention_set_variant(int v);  // 1 = the shipped policy, 0 = round-1 kernel, 10 / 11 deals, 21 / 36 persistent / 8-wave form forced
void attention_set_spin_budget(unsigned int polls);
#endif

// x_stride / y_stride: row strides in elements (0 = D, contiguous rows)
hipError_t launch_layernorm(int prec, const float* x, const float* g, const float* b, void* y,
                            int M, int D, float eps, bool out_f32, hipStream_t s,
                            size_t x_stride = 0, size_t y_stride = 0, bool reverse = false,
                            unsigned int* sat = nullptr, bool split = false);

// pre_layrnorm (in place, fp32) + layer 0's layer_norm1 (operand dtype of `prec`, to y) in one pass; with cls != null
// row 0 of every ntok-row image is taken as cls + pos0 (class_embedding + position_embedding[0]) instead of read
hipError_t launch_layernorm_pre(int prec, float* x, const float* g0, const float* b0, const float* g1,
                                const float* b1, void* y, int M, int D, float eps, hipStream_t s,
                                bool reverse = false, unsigned int* sat = nullptr, const float* cls = nullptr,
                                const float* pos0 = nullptr, int ntok = 0, bool split = false);

// qrows: number of leading query rows per sequence to compute (0 / L = all)
// hm: 0 = qkv is [rows][3 D] row-major; > 0 = head-major as GemmArgs::hm writes it ([3 heads][hm rows][64], 16-bit
// modes only; `qkv` is then the base of the whole array and the launch covers sequences from row 0)
// split: qkv [rows][6 D] and out [rows][2 D] are split images (fp16, not causal; attention.hip attn_tr_kernel<X2>)
hipError_t launch_attention(int prec, const void* qkv, void* out, int nseq, int L, int heads,
                            bool causal, int qrows, hipStream_t s, bool reverse = false, int hm = 0, bool split = false,
                            unsigned int* fault = nullptr);   // fault: the handle's host-mapped kernel-fault word (attn_ps_kernel)

// split: the patch matrix as a split image [B*np, 2 kpad] (fp16; GemmArgs::xsplit)
hipError_t launch_patchify(int prec, const float* pixels, void* patches, int B, int image,
                           int patch, int kpad, hipStream_t s, bool split = false);
hipError_t launch_patchify_u8(int prec, const uint8_t* pixels, void* patches, int B, int image,
                              int patch, int kpad, const float* mean, const float* stdv,
                              hipStream_t s, bool split = false);
hipError_t launch_bank_reduce(const float* feats, int K, int T, int P, float* bank, hipStream_t s);
hipError_t launch_text_embed(const int32_t* ids, const float* tok, const float* pos, float* x,
                             int K, int S, int D, hipStream_t s);
hipError_t launch_cvt_weight(int prec, const float* src, void* dst, int rows, int cols,
                             int cols_pad, hipStream_t s);
// split form (16-bit modes): dst [rows, 2 * cols_pad], per 64-element K-step hi[64] then lo[64] with
// hi = round(w), lo = round(w - hi) in the operand dtype (GemmArgs::ksplit); cols_pad % 64 == 0
hipError_t launch_cvt_weight_split(int prec, const float* src, void* dst, int rows, int cols,
                                   int cols_pad, hipStream_t s);
// *count (device, uint64) += elements of src [n] that do not round-trip through the 16-bit operand dtype of `prec`
hipError_t launch_count_inexact(int prec, const float* src, size_t n, unsigned long long* count, hipStream_t s);

// pooled row → LayerNorm → projection (no bias) → L2 normalise; out fp32 [n, P]
hipError_t launch_pool_project(const float* x, const int32_t* row_idx, int row_stride, int n,
                               int D, const float* g, const float* b, float eps,
                               const float* proj, int P, float* out, hipStream_t s,
                               bool normalize = true);
// score.hip: Mahalanobis baseline (reference utils/detection_util.py:146-207)
hipError_t launch_maha_prepare(const float* means, const float* prec, int C, int P, double* w, double* c,
                               hipStream_t s);
hipError_t launch_maha_score(const float* feats, int B, const float* prec, const double* w, const double* c,
                             int C, int P, float* scores, hipStream_t s);

hipError_t launch_score(const float* img, int B, const float* text, int K, int P, float T,
                        int kind, float* scores, hipStream_t s);

// metrics.hip: AUROC / AUPR / FPR@recall of two device score vectors; results land in the first
// 4 doubles of `workspace` (>= measures_workspace_bytes(n_pos + n_neg)), *out_dev points at them
size_t measures_workspace_bytes(long n);
hipError_t launch_measures(const float* pos, long n_pos, const float* neg, long n_neg, int negate,
                           double level, void* workspace, double** out_dev, hipStream_t s);

// fixed-edge histogram (numpy.histogram semantics), counts[nb] int64 on the device; nb <= 8192
hipError_t launch_histogram(const float* x, long n, const float* edges, int nb, unsigned long long* counts,
                            hipStream_t s);

// preprocess.hip: Resize(S) + CenterCrop(S) of uint8 RGB images; geometry (resized size, crop
// origin) is computed by the caller, one PrepImage per image
struct PrepImage {
  const uint8_t* src;  // [H, W, 3] uint8 RGB on the device
  int32_t H, W;        // source size
  int32_t nh, nw;      // size after Resize(S)
  int32_t top, left;   // CenterCrop origin inside the resized image
};
// jpeg.hip: one decoded-to-coefficients image as the kernels see it
struct JpegImageDev {
  int32_t width, height, ncomp;
  int32_t H, V;              // luma sampling factors (chroma is 1 x 1)
  int32_t wb[3], hb[3];      // blocks per row / column of each component
  int64_t coef_off[3];       // bytes into the coefficient buffer: int16 [hb][wb][64], natural order
  int64_t plane_off[3];      // bytes into the sample-plane workspace: uint8 [hb * 8][wb * 8]
  int64_t rgb_off;           // bytes into the output: uint8 [height][width][3]
};
hipError_t launch_jpeg_reconstruct(const JpegImageDev* meta_dev, const uint16_t* quant_dev, const void* coef_dev,
                                   uint8_t* planes_dev, uint8_t* rgb_dev, int n, int max_blocks, int max_pixels, hipStream_t s);
int prep_max_taps();
// coef_dev: prep_coef_bytes(max_batch, S) of workspace for the per-image coefficient tables of the LDS form (nullptr: the
// fused form everywhere); fused_only (A/B, harness builds): every workgroup takes the rounds-2/3 form
size_t prep_coef_bytes(int max_batch, int S);
hipError_t launch_resize_crop(const PrepImage* meta_dev, int32_t* coef_dev, int B, int S, uint8_t* dst, hipStream_t s,
                              bool fused_only = false);
