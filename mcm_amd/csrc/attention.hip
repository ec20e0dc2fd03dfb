// attention.hip — fused multi-head SDPA for short, fixed sequences (197 / 257 / 50 vision
// tokens, <=77 text tokens), head_dim 64, scale 0.125, fp32 softmax.
//
// Replaces CLIPAttention's SDPA (HF modeling_clip.py:259-277 eager definition, :313-331;
// causal for the text tower :543-556).  One workgroup per (sequence, head): the whole K
// and V of the head live in LDS (52 KiB at 197 keys), so nothing is re-read and the
// [L,L] score matrix never exists in HBM.
//
// 16-bit path (attn_tr_kernel, MFMA 16x16x32): each wave owns 16-query blocks.
//   S^T = K·Q^T   (K fragment as A, Q fragment as B)  → a lane holds, for ONE query
//                 (lane&15), 4 consecutive keys per 16-key tile: the softmax row lives in
//                 registers, the max needs only 2 cross-lane steps (xor 16, 32);
//   P (16-bit) stays in those registers and is fed straight back as the B operand of
//   O^T = V^T·P^T; V stays row-major in LDS and the V^T fragments come from the gfx950
//                 transpose read (ds_read_b64_tr_b16); the softmax denominator is one more
//                 MFMA per key step (all-ones A operand).
// fp32 path: the vision tower of the parity arm on fp32 MFMAs (attn_f32_mfma_kernel, round 5), the causal text tower on the plain
//   fp32 VALU kernel; both with exact expf.
// The round-1 kernel (attn_bf16_kernel: V transposed while staged through registers) exists in the
// harness build only (-DMCM_HARNESS), as the A/B arm of tests/test_gpu_kernels.py.
// Measurements: DESIGN.md section 4.2.
#include "common.hpp"

namespace {

__device__ __forceinline__ int ktile_off(int r, int c) {  // same image as the GEMM tile
  const int p = r >> 1;
  return p * 256 + ((((r & 1) << 3) | ((c ^ p) & 7)) << 4);
}

#ifdef MCM_HARNESS  // round-1 kernel: A/B arm only
__host__ __device__ constexpr int vt_stride(int LP) {  // bytes; ≡ 16 (mod 256): ds_read_b64
  return ((LP * 2 - 16 + 255) / 256) * 256 + 16;       // of 16 rows x 2 groups is conflict-free
}

template <int PREC, int LP, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_bf16_kernel(const uint16_t* __restrict__ qkv,
                                                           uint16_t* __restrict__ out, int L,
                                                           int heads, int qrows, int rev) {
  enter_precision_mode<PREC>();
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NT = LP / 16;       // key tiles
  constexpr int NU = LP / 32;       // key tile pairs (PV k-steps)
  constexpr int VS = vt_stride(LP);
  constexpr int MAXQB = (LP / 16 + 3) / 4;  // q-blocks per wave
  char* Ks = smem;                  // [LP rows][128 B], swizzled
  char* Vt = smem + LP * 128;       // [64 d][VS bytes]: V transposed, keys contiguous; the
                                    // key-pair index is XORed with (d>>3)<<2 (conflict-free
                                    // transposing writes, pairs of keys stay adjacent)

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bid = rev ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;
  const int seq = bid / heads, h = bid - seq * heads;
  const int D = heads * 64;
  const size_t rs = (size_t)3 * D;  // qkv row stride (elements)
  const uint16_t* base = qkv + (size_t)seq * L * rs + h * 64;
  const int fr = lane & 15, g = lane >> 4;

  // ---- all global reads are issued up front so their latency is paid once per workgroup:
  // Q fragments of every q-block this wave owns, then K (LDS-DMA), then V
  const int nqb = (qrows + 15) / 16;
  uint4 qf[MAXQB][2];
#pragma unroll
  for (int i = 0; i < MAXQB; ++i) {
    const int qr = min((wave + 4 * i) * 16 + fr, L - 1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
      qf[i][kk] = (wave + 4 * i < nqb) ? *(const uint4*)(base + (size_t)qr * rs + (kk * 4 + g) * 8)
                                       : make_uint4(0, 0, 0, 0);
  }
  for (int blk = wave; blk < LP / 8; blk += 4) {
    const int p = blk * 4 + (lane >> 4), s = lane & 15;
    const int row = min(2 * p + (s >> 3), L - 1);
    const int chunk = (s & 7) ^ (p & 7);
    __builtin_amdgcn_global_load_lds((gptr_t)(base + (size_t)row * rs + D + chunk * 8),
                                     (lptr_t)(Ks + blk * 1024), 16, 0, 0);
  }
  // V transposed: a thread takes (key pair x 8 dims) items and writes 8 dwords per item.  All of
  // a thread's V loads are issued before the first LDS write so the global latency is paid
  // once, not once per item.
  constexpr int NIT = ((LP / 2) * 8 + 255) / 256;
  uint4 va[NIT], vb[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int item = threadIdx.x + it * 256;
    const int dc = item & 7, kp = item >> 3;
    va[it] = make_uint4(0, 0, 0, 0);
    vb[it] = make_uint4(0, 0, 0, 0);
    if (item < (LP / 2) * 8) {
      if (2 * kp < L) va[it] = *(const uint4*)(base + (size_t)(2 * kp) * rs + 2 * D + dc * 8);
      if (2 * kp + 1 < L) vb[it] = *(const uint4*)(base + (size_t)(2 * kp + 1) * rs + 2 * D + dc * 8);
    }
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int item = threadIdx.x + it * 256;
    if (item >= (LP / 2) * 8) break;
    const int dc = item & 7, kp = item >> 3;
    const uint32_t a[4] = {va[it].x, va[it].y, va[it].z, va[it].w};
    const uint32_t b[4] = {vb[it].x, vb[it].y, vb[it].z, vb[it].w};
    const int kpos = (kp ^ (dc << 2)) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t lo = (a[j] & 0xffffu) | (b[j] << 16);
      const uint32_t hi = (a[j] >> 16) | (b[j] & 0xffff0000u);
      *(uint32_t*)(Vt + (dc * 8 + 2 * j) * VS + kpos) = lo;
      *(uint32_t*)(Vt + (dc * 8 + 2 * j + 1) * VS + kpos) = hi;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int koff[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) koff[kk] = ktile_off(fr, kk * 4 + g);
  constexpr float SC = 0.125f * 1.4426950408889634f;  // scale * log2(e)

#pragma unroll
  for (int i = 0; i < MAXQB; ++i) {
    const int qb = wave + 4 * i;
    if (qb >= nqb) break;
    const int q = qb * 16 + fr;
    const uint4 q0 = qf[i][0], q1 = qf[i][1];
    // S^T tiles (fragment reads software-pipelined one tile ahead; the sched_barrier stops
    // hipcc from hoisting all NT*2 ds_read_b128 up front, which spills at NT = 14/18)
    f32x4_t s[NT];
    uint4 kn0 = *(const uint4*)(Ks + koff[0]), kn1 = *(const uint4*)(Ks + koff[1]);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const uint4 k0 = kn0, k1 = kn1;
      if (t + 1 < NT) {
        kn0 = *(const uint4*)(Ks + (t + 1) * 2048 + koff[0]);
        kn1 = *(const uint4*)(Ks + (t + 1) * 2048 + koff[1]);
      }
      s[t] = mfma16<PREC>(k0, q0, (f32x4_t){0.f, 0.f, 0.f, 0.f});
      s[t] = mfma16<PREC>(k1, q1, s[t]);
      __builtin_amdgcn_sched_barrier(0);
    }
    // mask + row max (row = this lane's query; keys spread over regs and the 4 g-groups).
    // Only tiles that can contain an invalid key are masked (uniform test per tile); the
    // 0.125*log2(e) scale is folded into the exponent's fma.
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const bool full = (t * 16 + 15 < L) && (!CAUSAL || t * 16 + 15 <= qb * 16);
      if (!full) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = t * 16 + g * 4 + r;
          const bool ok = key < L && (!CAUSAL || key <= q);
          s[t][r] = ok ? s[t][r] : -INFINITY;
        }
      }
      m = fmaxf(fmaxf(fmaxf(fmaxf(m, s[t][0]), s[t][1]), s[t][2]), s[t][3]);  // two v_max3_f32
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const float msc = m * SC;
    // P = exp2(s*SC - m*SC) → bf16 in MFMA operand order; fp32 row sum
    float lsum = 0.f;
    uint4 pf[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      uint32_t w[4];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int t = 2 * u + half;
        float e[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(fmaf(s[t][r], SC, -msc));
        lsum += (e[0] + e[1]) + (e[2] + e[3]);
        w[half * 2 + 0] = pack2<PREC>(e[0], e[1]);
        w[half * 2 + 1] = pack2<PREC>(e[2], e[3]);
      }
      pf[u] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    lsum += __shfl_xor(lsum, 16, 64);
    lsum += __shfl_xor(lsum, 32, 64);
    const float rl = 1.0f / lsum;
    // O^T = V^T · P^T
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      f32x4_t o = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      const char* vrow = Vt + (dt * 16 + fr) * VS;
      const int vx = (dt * 2 + (fr >> 3)) << 2;  // this row's key-pair XOR
      auto vaddr = [&](int t) { return vrow + (((t * 8 + g * 2) ^ vx) << 2); };
      uint2 ln = *(const uint2*)vaddr(0), hn = *(const uint2*)vaddr(1);
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const uint2 lo = ln, hi = hn;
        if (u + 1 < NU) {
          ln = *(const uint2*)vaddr(2 * u + 2);
          hn = *(const uint2*)vaddr(2 * u + 3);
        }
        o = mfma16<PREC>(make_uint4(lo.x, lo.y, hi.x, hi.y), pf[u], o);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (q < L) {
        uint2 pk;
        pk.x = pack2<PREC>(o[0] * rl, o[1] * rl);
        pk.y = pack2<PREC>(o[2] * rl, o[3] * rl);
        *(uint2*)(out + ((size_t)seq * L + q) * D + h * 64 + dt * 16 + g * 4) = pk;
      }
    }
  }
}

#endif  // MCM_HARNESS

// =========================================================================================
// Round-2 kernel: K AND V by LDS-DMA (no register staging, no transposing LDS writes), V^T
// fragments by the gfx950 transpose read, row sums on the matrix pipe, three workgroups per CU.
//
// What changed against attn_bf16_kernel above and why (rocprofv3 of round 1: 180 us per launch at
// B/16 batch 512, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 30 %, VALU-issue bound):
//   * V stays row-major in LDS ([key][64 d], 128-B rows, the 32-B column segment XORed with
//     (key>>1)&3) and is filled by LDS-DMA exactly like K.  The PV A-operand (V^T, 4 keys x 16 dims
//     per 16-lane group) comes from ds_read_b64_tr_b16: lane i of a group passes the address of 4
//     contiguous dims of key i/4 and receives dim i of the 4 keys (tools/isa_probe.hip, removed in round 6: git history).  Gone: 8
//     global V loads, ~64 ds_write_b32 and the XOR address arithmetic per thread, and the bank
//     conflicts of the transposing writes.  The read address is lane_base[dt] + tile * 2048: one VGPR
//     per 16-dim block, everything else immediate offsets.
//   * the softmax denominator is one more MFMA per key step (A = all ones): 52 v_add per 16-query
//     block leave the VALU, which is the pipe this kernel is bound by; it also makes the denominator
//     the sum of the ROUNDED probabilities, i.e. the weights that multiply V sum to exactly 1.
//   * keys are padded to a multiple of 16, not 32 (an odd tile count ends with one 16x16x16 MFMA), so
//     197 tokens need 2 x 208 x 128 B = 52 KiB and three workgroups fit a CU (was 62 KiB, two): more
//     loads in flight per CU for a kernel whose floor is HBM (620 MB per launch).
// =========================================================================================
typedef short v4s_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s_t* lds_v4s_t;
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint2 tr_read16(const char* p) {  // ds_read_b64_tr_b16
  return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t)p));
}

// Two things learned on the device while bringing this kernel up (round 2):
//  * every value that feeds an MFMA must come from an instruction hipcc can see.  The bf16 pack used to
//    be an inline-asm v_cvt_pk_bf16_f32; hipcc pads no wait states between an asm statement's VALU write
//    and an MFMA reading it as an operand (guide §5.7 item 2), and here the packed probabilities go
//    straight into the row-sum and PV MFMAs: wrong rows for fixed lane groups, bf16 only.  pack_bf2
//    (common.hpp) is now __builtin_convertvector, which selects the same instruction with the hazard
//    handled by the compiler.
//  * an odd tile count does NOT end with a v_mfma_f32_16x16x16 in the same accumulator chain as the
//    16x16x32 steps: every instantiation that mixed the two shapes in one chain (NT = 3, 5, 7, 13, 17)
//    returned wrong 16-dim blocks whose position moved with register allocation, instantiations with one
//    shape (NT = 1, 4) were exact.  The 16-key tail is a 32-key step whose upper half is zero on both
//    operands (one extra 16-cycle MFMA per 16-dim block).
// mfma_keep additionally keeps A and B live past the instruction so the result is never allocated on top
// of an operand (no instruction emitted; not the cause of either fault, kept as cheap insurance).
template <int PREC>
__device__ __forceinline__ f32x4_t mfma_keep(uint4 a, uint4 b, f32x4_t c) {
  f32x4_t d = mfma16<PREC>(a, b, c);
  asm volatile("" : "+v"(d) : "v"(__builtin_bit_cast(u32x4_t, a)), "v"(__builtin_bit_cast(u32x4_t, b)));
  return d;
}

// (The s_setprio arms, the two-pass 80-register form and the phase probes of rounds 2 - 5 were measured and removed in round 6:
// EXPERIMENTS.md "Removed arms", git history before round 6.)
// X2 (fp16; the split-activation arm): qkv and out are SPLIT images — [rows][6 D] in, [rows][2 D] out,
// per 64 columns (= one head of q, k or v) hi[64] then lo[64] — and every product runs as three fp16 MFMAs on the hi / lo
// pairs: S = K_lo Q_hi + K_hi Q_lo + K_hi Q_hi, O = V_lo P_hi + V_hi P_lo + V_hi P_hi (the lo x lo terms are below fp32
// round-off), P itself split AFTER a scale of 2^12 (exp2 of s - max + 12: probabilities of 1 / 197 would otherwise put
// their lo half into fp16's subnormals; the scale cancels in O / rowsum).  ~22 significand bits on both operands of both
// GEMMs, the same softmax arithmetic as the 16-bit kernel.  K and V hi AND lo stay in LDS (4 images, 104 KiB at 197 tokens):
// one workgroup per CU — the arm re-scores a few hundred images per data set.
// NFULL (round 5): the number of leading key tiles the launcher GUARANTEES to hold valid keys only (bidirectional form: every
// tile but the last when the sequence needs exactly NT tiles).  The key-validity mask of those tiles folds away at compile time —
// with a run-time L hipcc predicates it instead of branching, 3 VALU instructions per score (v_cmp, v_cndmask, an index v_or) on a
// kernel whose compute phase is bound by VALU issue: 158 of the 565 issue slots of a 16-query block at B/16.  Same bits.
template <int PREC, int NT, bool CAUSAL, int NW, int OCC, bool X2 = false, int NFULL = 0>
__global__ __launch_bounds__(NW * 64, OCC) void attn_tr_kernel(const uint16_t* __restrict__ qkv,
                                                                         uint16_t* __restrict__ out, int L,
                                                                         int heads, int qrows, int rev, int hm) {
  static_assert(!X2 || PREC == MCM_PREC_F16, "split activations: fp16");
  enter_precision_mode<PREC>();
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int LP = NT * 16;           // padded keys
  constexpr int MAXQB = (NT + NW - 1) / NW;   // q-blocks per wave (NW waves per workgroup)
  constexpr uint32_t ONE2 = PREC == MCM_PREC_F16 ? 0x3c003c00u : 0x3f803f80u;  // two 1.0 operands
  char* Ks = smem;               // [LP][128 B], GEMM-style pair/XOR image
  char* Vs = smem + LP * 128;    // [LP][128 B], 32-B segment s stored at s ^ ((key >> 1) & 3)
  constexpr int LO = 2 * LP * 128;  // X2: the lo images of K and V follow the hi images, same layouts

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bid = (rev & 1) ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;
  int seq = bid / heads, h = bid - seq * heads;
  if (rev & 2) {  // XCD-aware deal (sequences a multiple of 8): workgroup b runs on XCD b % 8, and the `heads`
                  // workgroups of a sequence are consecutive on ONE XCD — the 128-B q / k / v segments of a qkv row
                  // are neighbours in memory, so a row is fetched by one L2, at about the same time
    const int j = bid >> 3, sj = j / heads;
    seq = sj * 8 + (bid & 7);
    h = j - sj * heads;
  }
  // Which wave takes which q-blocks.  The q-blocks are dealt round-robin to the waves (13 blocks on 8 waves: 2-2-2-2-2-1-1-1),
  // and waves w and w + 4 share a SIMD, so SIMD 0 always carried 4 blocks and the others 3 — and the kernel is bound by
  // the VALU / transcendental issue of its softmax (exp2 runs at quarter rate), i.e. by the busiest SIMD.  With rev & 4
  // the deal is rotated by the workgroup's index (bits 3-4: workgroups of one XCD share bid & 7), so the 2.5 workgroups
  // resident on a CU load different SIMDs most: 3.25 blocks per SIMD on average instead of 4 on one.  Same bits (a
  // q-block's arithmetic does not depend on the wave that runs it).
  const int wq = (rev & 4) ? (wave + NW - ((bid >> 3) & 3) % NW) % NW : wave;
  const int D = heads * 64;
  // qkv layout.  Row-major (hm = 0): [rows][3 D], a head's q / k / v are 128-B segments of 4.6-KB rows.  Head-major
  // (hm = rows of the array, what the QKV projection writes in the model): [3 heads][hm][64] — this workgroup's Q, K
  // and V are three runs of L x 128 consecutive bytes.  rs: row stride, KO / VO: from a q row to the k / v row (elements)
  const size_t rs = X2 ? (size_t)6 * D : MCM_HM(hm) ? (size_t)64 : (size_t)3 * D;
  const size_t KO = X2 ? (size_t)2 * D : MCM_HM(hm) ? (size_t)heads * hm * 64 : (size_t)D, VO = 2 * KO;
  const uint16_t* base = X2 ? qkv + (size_t)seq * L * rs + h * 128
                            : MCM_HM(hm) ? qkv + ((size_t)h * hm + (size_t)seq * L) * 64 : qkv + (size_t)seq * L * rs + h * 64;
  const int fr = lane & 15, g = lane >> 4;

  // ---- every global read is issued up front: Q fragments of this wave's q-blocks, K, V
  const int nqb = (qrows + 15) / 16;
  // Q of the wave's FIRST q-block is requested here, with K and V; Q of a further block is requested as soon as the previous
  // block's Q K^T is done (it lands under that block's softmax and P V): the Q fragments of one block are live at a time, which
  // is what lets the B/16 instantiation fit 80 registers — three workgroups per CU instead of two (EXPERIMENTS.md R5.8)
  uint4 qcur[2], qlcur[X2 ? 2 : 1];
  auto load_q = [&](int i) {
    const int qr = min((wq + NW * i) * 16 + fr, L - 1);
    const bool have = wq + NW * i < nqb;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      qcur[kk] = have ? *(const uint4*)(base + (size_t)qr * rs + (kk * 4 + g) * 8) : make_uint4(0, 0, 0, 0);
      if constexpr (X2) qlcur[kk] = have ? *(const uint4*)(base + (size_t)qr * rs + 64 + (kk * 4 + g) * 8) : make_uint4(0, 0, 0, 0);
    }
  };
  load_q(0);
  for (int blk = wave; blk < LP / 8; blk += NW) {  // 1-KiB pieces: 8 key rows each
#pragma unroll
    for (int part = 0; part < (X2 ? 2 : 1); ++part) {  // hi image, (X2) lo image
      {
        const int p = blk * 4 + (lane >> 4), s = lane & 15;
        const int row = min(2 * p + (s >> 3), L - 1);
        const int chunk = (s & 7) ^ (p & 7);
        __builtin_amdgcn_global_load_lds((gptr_t)(base + (size_t)row * rs + KO + part * 64 + chunk * 8),
                                         (lptr_t)(Ks + part * LO + blk * 1024), 16, 0, 0);
      }
      {
        const int row = blk * 8 + (lane >> 3), pc = lane & 7;
        const int lc = ((((pc >> 1) ^ (row >> 1)) & 3) << 1) | (pc & 1);  // logical 16-B chunk of this slot
        __builtin_amdgcn_global_load_lds((gptr_t)(base + (size_t)min(row, L - 1) * rs + VO + part * 64 + lc * 8),
                                         (lptr_t)(Vs + part * LO + blk * 1024), 16, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int koff[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) koff[kk] = ktile_off(fr, kk * 4 + g);
  // V^T fragment of key tile t, dims [16 dt, 16 dt + 16): this lane reads 4 dims of key 16 t + 4 g + fr/4.
  // (key >> 1) & 3 = (2 g + (fr >> 3)) & 3 for every t, so the swizzle is a per-lane constant.
  const char* vlane[4];
  {
    const int sw = (2 * g + (fr >> 3)) & 3;
    const char* vb = Vs + (4 * g + (fr >> 2)) * 128 + (fr & 3) * 8;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) vlane[dt] = vb + ((dt ^ sw) << 5);
  }
  constexpr float SC = 0.125f * 1.4426950408889634f;  // scale * log2(e)

#pragma unroll
  for (int i = 0; i < MAXQB; ++i) {
    const int qb = wq + NW * i;
    if (qb >= nqb) break;
    const int q = qb * 16 + fr;
    const uint4 q0 = qcur[0], q1 = qcur[1];
    uint4 ql0 = make_uint4(0, 0, 0, 0), ql1 = ql0;
    if constexpr (X2) { ql0 = qlcur[0]; ql1 = qlcur[1]; }
    const f32x4_t zero = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    f32x4_t lacc = zero;
    f32x4_t o[4] = {zero, zero, zero, zero};
    constexpr int NS = (NT + 1) / 2;
    // mask of key tile t for this lane's query (only tiles that can hold an invalid key are touched)
    auto mask_tile = [&](f32x4_t& st, int t) {
      if (!CAUSAL && t < NFULL) return;   // (t is a constant after unrolling)
      const bool full = (t * 16 + 15 < L) && (!CAUSAL || t * 16 + 15 <= qb * 16);
      if (!full) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = t * 16 + g * 4 + r;
          const bool ok = key < L && (!CAUSAL || key <= q);
          st[r] = ok ? st[r] : -INFINITY;
        }
      }
    };
    f32x4_t s[NT];
    uint4 kn0 = *(const uint4*)(Ks + koff[0]), kn1 = *(const uint4*)(Ks + koff[1]);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const uint4 k0 = kn0, k1 = kn1;
      if (t + 1 < NT) {
        kn0 = *(const uint4*)(Ks + (t + 1) * 2048 + koff[0]);
        kn1 = *(const uint4*)(Ks + (t + 1) * 2048 + koff[1]);
      }
      if constexpr (X2) {  // the two cross terms first (small), then the leading one: one fp32 accumulator chain
        const uint4 kl0 = *(const uint4*)(Ks + LO + t * 2048 + koff[0]), kl1 = *(const uint4*)(Ks + LO + t * 2048 + koff[1]);
        s[t] = mfma_keep<PREC>(kl0, q0, (f32x4_t){0.f, 0.f, 0.f, 0.f});
        s[t] = mfma_keep<PREC>(kl1, q1, s[t]);
        s[t] = mfma_keep<PREC>(k0, ql0, s[t]);
        s[t] = mfma_keep<PREC>(k1, ql1, s[t]);
        s[t] = mfma_keep<PREC>(k0, q0, s[t]);
        s[t] = mfma_keep<PREC>(k1, q1, s[t]);
      } else {
        s[t] = mfma_keep<PREC>(k0, q0, (f32x4_t){0.f, 0.f, 0.f, 0.f});
        s[t] = mfma_keep<PREC>(k1, q1, s[t]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (i + 1 < MAXQB) load_q(i + 1);   // the next block's Q: under this block's softmax and P V
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      mask_tile(s[t], t);
      m = fmaxf(fmaxf(fmaxf(fmaxf(m, s[t][0]), s[t][1]), s[t][2]), s[t][3]);
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const float msc = X2 ? m * SC - 12.0f : m * SC;  // X2: P scaled by 2^12 (see the kernel's header); cancels in O / rowsum
    // P = exp2(s*SC - m*SC) in MFMA operand order; the row sum is taken on the matrix pipe below
    uint2 pt[NT];
    uint2 pl[X2 ? NT : 1];  // X2: the lo halves of P
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float e[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(fmaf(s[t][r], SC, -msc));
      if constexpr (X2) {
        split2<PREC>(e[0], e[1], pt[t].x, pl[t].x);
        split2<PREC>(e[2], e[3], pt[t].y, pl[t].y);
      } else {
        pt[t] = make_uint2(pack2<PREC>(e[0], e[1]), pack2<PREC>(e[2], e[3]));
      }
    }
    // key step u: tiles 2u, 2u+1 (the last step of an odd tile count has a zero upper half).  Kept a macro:
    // through a lambda the (never taken) pt[NT] index of the last step sent the whole array to scratch.
#define MCM_PSTEP(u)                                                                             \
  ((2 * (u) + 1 < NT) ? make_uint4(pt[2 * (u)].x, pt[2 * (u)].y, pt[(2 * (u) + 1 < NT) ? 2 * (u) + 1 : 0].x, \
                                   pt[(2 * (u) + 1 < NT) ? 2 * (u) + 1 : 0].y)                  \
                      : make_uint4(pt[2 * (u)].x, pt[2 * (u)].y, 0u, 0u))
    // O^T = V^T · P^T for the four 16-dim blocks and the row sum (all-ones A operand), key step by key step: five
    // independent accumulator chains in flight, so no MFMA waits for the one just issued (dim-block-outer order
    // made each of the 7 steps of a chain wait out the previous step's latency)
#define MCM_PLSTEP(u)                                                                            \
  ((2 * (u) + 1 < NT) ? make_uint4(pl[2 * (u)].x, pl[2 * (u)].y, pl[(2 * (u) + 1 < NT) ? 2 * (u) + 1 : 0].x, \
                                   pl[(2 * (u) + 1 < NT) ? 2 * (u) + 1 : 0].y)                  \
                      : make_uint4(pl[2 * (u)].x, pl[2 * (u)].y, 0u, 0u))
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const uint4 pu = MCM_PSTEP(u);
      uint4 plu = make_uint4(0u, 0u, 0u, 0u);
      if constexpr (X2) {
        plu = MCM_PLSTEP(u);
        lacc = mfma_keep<PREC>(make_uint4(ONE2, ONE2, ONE2, ONE2), plu, lacc);
      }
      lacc = mfma_keep<PREC>(make_uint4(ONE2, ONE2, ONE2, ONE2), pu, lacc);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const char* vp = vlane[dt];
        const uint2 lo = tr_read16(vp + (2 * u) * 2048);
        const uint2 hi = (2 * u + 1 < NT) ? tr_read16(vp + (2 * u + 1) * 2048) : make_uint2(0u, 0u);
        if constexpr (X2) {  // V_lo P_hi and V_hi P_lo first, then the leading term
          const uint2 llo = tr_read16(vp + LO + (2 * u) * 2048);
          const uint2 lhi = (2 * u + 1 < NT) ? tr_read16(vp + LO + (2 * u + 1) * 2048) : make_uint2(0u, 0u);
          o[dt] = mfma_keep<PREC>(make_uint4(llo.x, llo.y, lhi.x, lhi.y), pu, o[dt]);
          o[dt] = mfma_keep<PREC>(make_uint4(lo.x, lo.y, hi.x, hi.y), plu, o[dt]);
        }
        o[dt] = mfma_keep<PREC>(make_uint4(lo.x, lo.y, hi.x, hi.y), pu, o[dt]);
      }
    }
#undef MCM_PLSTEP
#undef MCM_PSTEP
    const float rl = 1.0f / lacc[0];
    // A lane holds 4 dims (8 B) of each of the four 16-dim blocks.  Lanes g and g^1 (16 lanes apart) trade one
    // block of each pair by v_permlane16_swap, after which a lane owns 8 consecutive dims (16 B) of ONE block:
    // two 16-byte stores per lane and q-block instead of four 8-byte ones, 64 contiguous bytes per row and store.
    uint32_t pk[4][2];
    uint32_t pkl[X2 ? 4 : 1][2];  // X2: the lo halves of the output
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      if constexpr (X2) {
        split2<PREC>(o[dt][0] * rl, o[dt][1] * rl, pk[dt][0], pkl[dt][0]);
        split2<PREC>(o[dt][2] * rl, o[dt][3] * rl, pk[dt][1], pkl[dt][1]);
      } else {
        pk[dt][0] = pack2<PREC>(o[dt][0] * rl, o[dt][1] * rl);
        pk[dt][1] = pack2<PREC>(o[dt][2] * rl, o[dt][3] * rl);
      }
    }
    uint4 wide[2];
    uint4 widel[X2 ? 2 : 1];
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      // vdst = block 2pr, src = block 2pr+1: the odd rows of vdst and the even rows of src change places
      const auto w0 = __builtin_amdgcn_permlane16_swap(pk[2 * pr][0], pk[2 * pr + 1][0], false, false);
      const auto w1 = __builtin_amdgcn_permlane16_swap(pk[2 * pr][1], pk[2 * pr + 1][1], false, false);
      wide[pr] = make_uint4(w0[0], w1[0], w0[1], w1[1]);
      if constexpr (X2) {
        const auto l0 = __builtin_amdgcn_permlane16_swap(pkl[2 * pr][0], pkl[2 * pr + 1][0], false, false);
        const auto l1 = __builtin_amdgcn_permlane16_swap(pkl[2 * pr][1], pkl[2 * pr + 1][1], false, false);
        widel[pr] = make_uint4(l0[0], l1[0], l0[1], l1[1]);
      }
    }
    if (q < L) {
      uint16_t* orow = X2 ? out + ((size_t)seq * L + q) * 2 * D + h * 128 : out + ((size_t)seq * L + q) * D + h * 64;
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {  // even g: block 2pr, dims g*4 .. g*4+7; odd g: block 2pr+1, dims (g-1)*4 ..
        *(uint4*)(orow + (2 * pr + (g & 1)) * 16 + (g & 2) * 4) = wide[pr];
        if constexpr (X2) *(uint4*)(orow + 64 + (2 * pr + (g & 1)) * 16 + (g & 2) * 4) = widel[pr];
      }
    }
  }
}

// ---- persistent, specialised form of the 16-bit kernel (round 5, EXPERIMENTS.md R5.9) -------------------------------------
// The phase probes say: the memory phases of attn_tr_kernel alone take 88 us per launch (7 TB/s), its arithmetic alone 94 us, and
// the kernel 145 us (standalone, B/16 batch 512) — two 8-wave workgroups per CU, each loading THEN computing, overlap only by
// chance.  Here the two run side by side by construction: ONE 16-wave workgroup per CU that lives for the whole launch,
//   * NLD loader waves (the last ones): nothing but LDS-DMA of the K and V images of job n + 2 (a job = one (sequence, head)) into
//     a ring of three 52-KiB buffers, `ready[loader]` = number of jobs whose pieces have landed (s_waitcnt vmcnt + one LDS store);
//     a loader wave keeps at most WIN + 1 pieces in flight;
//   * 16 - NLD compute waves: 16-query blocks handed out by an LDS counter in job order (so a SIMD that carries a loader wave
//     simply takes fewer blocks), each waits for `ready > job`, runs the q-block of attn_tr_kernel unchanged (same arithmetic in the
//     same order: bit-identical results), and counts itself into `done[buffer]` after its last LDS read — which is what the loaders
//     wait for before they overwrite a buffer.
// No workgroup barrier after the first one; no wave waits for a load it issued itself.
// EVERY WAIT IS BOUNDED (round 6): a wave that has polled `spin_budget` times without its condition coming true — a protocol slip,
// a lost wave — raises the workgroup's abort word, stores 1 to the handle's fault word (host-mapped: the next API call on the
// handle returns MCM_EHIP, mcm_kernel_faults reads it) and returns; every other wave of the workgroup sees the abort word in its
// own polls and returns too.  The workgroup's output rows are then garbage, but the launch ENDS (the budget, 2^22 polls of >= 64
// cycles, is three orders of magnitude above the kernel's whole run time) instead of holding the GPU until a watchdog fires.
// (Measured and removed in round 6 — EXPERIMENTS.md "Removed arms": 1 / 2 loader waves, windows 0 / 1 / 2 / 8 / 16, K and V as
// separate ring entries, the no-load / no-arithmetic / nobody-waits probes.)
constexpr int PS_WIN = 4;                      // a loader wave keeps at most PS_WIN + 1 pieces in flight
constexpr unsigned int PS_SPIN_BUDGET = 1u << 22;
template <int PREC, int NT, int NFULL, int NLD>
__global__ __launch_bounds__(1024) void attn_ps_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out, int L,
                                                        int heads, int qrows, int njobs, int rev, unsigned int spin_budget,
                                                        unsigned int* __restrict__ fault) {
  enter_precision_mode<PREC>();
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int LP = NT * 16, BUF = 2 * LP * 128, NCW = 16 - NLD;
  constexpr int PIECES = 2 * (LP / 8), PPL = PIECES / NLD;   // 1-KiB pieces of a job's K + V images, per loader wave
  static_assert(PIECES % NLD == 0 && PPL < 60, "pieces per loader wave: whole, and within vmcnt's range");
  constexpr uint32_t ONE2 = PREC == MCM_PREC_F16 ? 0x3c003c00u : 0x3f803f80u;
  // flags (LDS, after the three buffers): [0..3] jobs landed per loader wave, [4..6] q-blocks done with the job, per buffer,
  // cumulative, [7] next block, [11] abort (a wait ran out of its budget somewhere in this workgroup)
  volatile uint32_t* flags = (volatile uint32_t*)(smem + 3 * BUF);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int D = heads * 64;
  const size_t rs = (size_t)3 * D;
  const int nj = (njobs - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // jobs blockIdx.x + n gridDim.x, n < nj
  const int nqb = (qrows + 15) / 16;
  // rev: the jobs in descending order (the model alternates the walk direction from kernel to kernel, DESIGN.md 4.4d: the rows the
  // QKV GEMM wrote last are the ones still in the Infinity Cache)
  auto jobid = [&](int n) { const int j = (int)blockIdx.x + n * (int)gridDim.x; return rev ? njobs - 1 - j : j; };
  if (threadIdx.x < 12) flags[threadIdx.x] = (threadIdx.x < 4 && (int)threadIdx.x >= NLD) ? 0x7fffffffu : 0u;
  __syncthreads();
  auto give_up = [&]() {   // this wave's wait ran out: stop the workgroup, tell the host
    flags[11] = 1u;
    if (lane == 0 && fault) __hip_atomic_store(fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  };

  if (wave >= NCW) {  // ---------------- loader wave
    const int lw = wave - NCW;
    __builtin_amdgcn_s_setprio(3);
    for (int n = 0; n < nj; ++n) {
      const int job = jobid(n);
      const int seq = job / heads, h = job - seq * heads;
      const uint16_t* base = qkv + (size_t)seq * L * rs + h * 64;
      const int b = n % 3;
      const uint32_t need = (uint32_t)(nqb * (n / 3));   // q-blocks of this slot's earlier jobs
      if (flags[4 + b] < need) {
        // the slot is still being read: everything this wave has in flight belongs to jobs < n — say so before waiting
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        flags[lw] = (uint32_t)n;
        unsigned int polls = 0;
        while (flags[4 + b] < need) {
          if (flags[11] != 0u) return;
          if (++polls > spin_budget) { give_up(); return; }
          __builtin_amdgcn_s_sleep(2);
        }
      }
      const uint32_t kb = lds_addr(smem + b * BUF), vb = kb + LP * 128;
#pragma unroll(PPL > 13 ? 2 : PPL)
      for (int i = 0; i < PPL; ++i) {
        const int piece = lw * PPL + i;
        if (piece < LP / 8) {   // K piece: 8 key rows in the GEMM-style pair / XOR image
          const int blk = piece;
          const int p = blk * 4 + (lane >> 4), sl = lane & 15;
          const int row = min(2 * p + (sl >> 3), L - 1);
          const int chunk = (sl & 7) ^ (p & 7);
          glds16(base + (size_t)row * rs + D + chunk * 8, kb + blk * 1024);
        } else {                // V piece: 8 key rows, 32-B segments XORed with (key >> 1) & 3
          const int blk = piece - LP / 8;
          const int row = blk * 8 + (lane >> 3), pc = lane & 7;
          const int lc = ((((pc >> 1) ^ (row >> 1)) & 3) << 1) | (pc & 1);
          glds16(base + (size_t)min(row, L - 1) * rs + 2 * D + lc * 8, vb + blk * 1024);
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PS_WIN) : "memory");
      }
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPL) : "memory");   // every piece older than this job's has landed
      flags[lw] = (uint32_t)n;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    flags[lw] = (uint32_t)nj;
    return;
  }

  // ---------------- compute wave
  const int fr = lane & 15, g = lane >> 4;
  const int total = nj * nqb;
  auto fetch = [&]() {  // the next q-block of this workgroup, in job order
    uint32_t t = 0;
    if (lane == 0) t = atomicAdd((uint32_t*)(smem + 3 * BUF) + 7, 1u);
    return (int)__builtin_amdgcn_readfirstlane(t);
  };
  uint4 qcur[2];
  auto load_q = [&](int T) {
    const int n = T / nqb, qb = T - n * nqb;
    const int job = jobid(n);
    const int seq = job / heads, h = job - seq * heads;
    const uint16_t* base = qkv + (size_t)seq * L * rs + h * 64;
    const int qr = min(qb * 16 + fr, L - 1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) qcur[kk] = *(const uint4*)(base + (size_t)qr * rs + (kk * 4 + g) * 8);
  };
  int koff[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) koff[kk] = ktile_off(fr, kk * 4 + g);
  int voff[4];
  {
    const int sw = (2 * g + (fr >> 3)) & 3;
    const int vb = LP * 128 + (4 * g + (fr >> 2)) * 128 + (fr & 3) * 8;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) voff[dt] = vb + ((dt ^ sw) << 5);
  }
  constexpr float SC = 0.125f * 1.4426950408889634f;
  int T = fetch();
  if (T < total) load_q(T);
  while (T < total) {
    const int n = T / nqb, qb = T - n * nqb;
    const int job = jobid(n);
    const int seq = job / heads, h = job - seq * heads;
    const int b = n % 3;
    const int q = qb * 16 + fr;
    {  // job n has landed (every loader wave says so)
      unsigned int polls = 0;
      for (;;) {
        const uint32_t r0 = flags[0], r1 = flags[1], r2 = flags[2], r3 = flags[3];
        if ((int)min(min(r0, r1), min(r2, r3)) > n) break;
        if (flags[11] != 0u) return;
        if (++polls > spin_budget) { give_up(); return; }
        __builtin_amdgcn_s_sleep(1);
      }
      asm volatile("" ::: "memory");
    }
    const char* Ks = smem + b * BUF;
    const uint4 q0 = qcur[0], q1 = qcur[1];
    const f32x4_t zero = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    f32x4_t lacc = zero;
    f32x4_t o[4] = {zero, zero, zero, zero};
    constexpr int NS = (NT + 1) / 2;
    f32x4_t s[NT];
    uint4 kn0 = *(const uint4*)(Ks + koff[0]), kn1 = *(const uint4*)(Ks + koff[1]);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const uint4 k0 = kn0, k1 = kn1;
      if (t + 1 < NT) {
        kn0 = *(const uint4*)(Ks + (t + 1) * 2048 + koff[0]);
        kn1 = *(const uint4*)(Ks + (t + 1) * 2048 + koff[1]);
      }
      s[t] = mfma_keep<PREC>(k0, q0, (f32x4_t){0.f, 0.f, 0.f, 0.f});
      s[t] = mfma_keep<PREC>(k1, q1, s[t]);
      __builtin_amdgcn_sched_barrier(0);
    }
    const int Tn = fetch();          // the next block and its Q: under this block's softmax and P V
    if (Tn < total) load_q(Tn);
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (t >= NFULL && t * 16 + 15 >= L) {
#pragma unroll
        for (int r = 0; r < 4; ++r) s[t][r] = (t * 16 + g * 4 + r < L) ? s[t][r] : -INFINITY;
      }
      m = fmaxf(fmaxf(fmaxf(fmaxf(m, s[t][0]), s[t][1]), s[t][2]), s[t][3]);
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const float msc = m * SC;
    uint2 pt[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float e[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(fmaf(s[t][r], SC, -msc));
      pt[t] = make_uint2(pack2<PREC>(e[0], e[1]), pack2<PREC>(e[2], e[3]));
    }
#define MCM_PSTEP(u)                                                                             \
  ((2 * (u) + 1 < NT) ? make_uint4(pt[2 * (u)].x, pt[2 * (u)].y, pt[(2 * (u) + 1 < NT) ? 2 * (u) + 1 : 0].x, \
                                   pt[(2 * (u) + 1 < NT) ? 2 * (u) + 1 : 0].y)                  \
                      : make_uint4(pt[2 * (u)].x, pt[2 * (u)].y, 0u, 0u))
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const uint4 pu = MCM_PSTEP(u);
      lacc = mfma_keep<PREC>(make_uint4(ONE2, ONE2, ONE2, ONE2), pu, lacc);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const char* vp = Ks + voff[dt];
        const uint2 lo = tr_read16(vp + (2 * u) * 2048);
        const uint2 hi = (2 * u + 1 < NT) ? tr_read16(vp + (2 * u + 1) * 2048) : make_uint2(0u, 0u);
        o[dt] = mfma_keep<PREC>(make_uint4(lo.x, lo.y, hi.x, hi.y), pu, o[dt]);
      }
    }
#undef MCM_PSTEP
    // this wave's last LDS read of the buffer has been issued (LDS executes a wave's instructions in order): count the block
    asm volatile("" ::: "memory");
    if (lane == 0) atomicAdd((uint32_t*)(smem + 3 * BUF) + 4 + b, 1u);
    asm volatile("" ::: "memory");
    const float rl = 1.0f / lacc[0];
    uint32_t pk[4][2];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      pk[dt][0] = pack2<PREC>(o[dt][0] * rl, o[dt][1] * rl);
      pk[dt][1] = pack2<PREC>(o[dt][2] * rl, o[dt][3] * rl);
    }
    uint4 wide[2];
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      const auto w0 = __builtin_amdgcn_permlane16_swap(pk[2 * pr][0], pk[2 * pr + 1][0], false, false);
      const auto w1 = __builtin_amdgcn_permlane16_swap(pk[2 * pr][1], pk[2 * pr + 1][1], false, false);
      wide[pr] = make_uint4(w0[0], w1[0], w0[1], w1[1]);
    }
    if (q < L) {
      uint16_t* orow = out + ((size_t)seq * L + q) * D + h * 64;
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) *(uint4*)(orow + (2 * pr + (g & 1)) * 16 + (g & 2) * 4) = wide[pr];
    }
    T = Tn;
  }
}

#ifdef MCM_HARNESS
unsigned int g_ps_spin_budget = PS_SPIN_BUDGET;   // mcm_debug_attn_spin_budget: 0 = every wait gives up at once (the fault path's test)
#endif

template <int PREC, int NT, int NFULL, int NLD>
hipError_t launch_ps(const void* qkv, void* out, int nseq, int L, int heads, int qrows, hipStream_t s, int rev, unsigned int* fault) {
  constexpr int lds = 3 * 2 * NT * 16 * 128 + 64;
  static PerDeviceFlag attr_set;
  const int cus = device_cu_count();
  if (cus <= 0) return hipErrorInvalidDevice;
  if (!attr_set.get()) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_ps_kernel<PREC, NT, NFULL, NLD>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    attr_set.set();
  }
  const int njobs = nseq * heads;
#ifdef MCM_HARNESS
  const unsigned int budget = g_ps_spin_budget;
#else
  const unsigned int budget = PS_SPIN_BUDGET;
#endif
  hipLaunchKernelGGL((attn_ps_kernel<PREC, NT, NFULL, NLD>), dim3(min(cus, njobs)), dim3(1024), lds, s, (const uint16_t*)qkv,
                     (uint16_t*)out, L, heads, qrows, njobs, rev & 1, budget, fault);
  return hipGetLastError();
}

// ---- fp32 parity arm -------------------------------------------------------------------
template <bool CAUSAL>
__global__ __launch_bounds__(256) void attn_f32_kernel(const float* __restrict__ qkv,
                                                       float* __restrict__ out, int L, int heads,
                                                       int qrows) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Ks = (float*)smem;            // [L][65]
  float* Vs = Ks + (size_t)L * 65;     // [L][64]
  float* qs = Vs + (size_t)L * 64;     // [4][64]
  float* ps = qs + 4 * 64;             // [4][L]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int seq = blockIdx.x / heads, h = blockIdx.x - seq * heads;
  const int D = heads * 64;
  const size_t rs = (size_t)3 * D;
  const float* base = qkv + (size_t)seq * L * rs + h * 64;
  for (int i = threadIdx.x; i < L * 64; i += 256) {
    const int j = i >> 6, d = i & 63;
    Ks[j * 65 + d] = base[(size_t)j * rs + D + d];
    Vs[j * 64 + d] = base[(size_t)j * rs + 2 * D + d];
  }
  __syncthreads();
  float* myq = qs + wave * 64;
  float* myp = ps + wave * L;
  for (int q = wave; q < qrows; q += 4) {
    myq[lane] = base[(size_t)q * rs + lane];
    __builtin_amdgcn_wave_barrier();
    const int jmax = CAUSAL ? q + 1 : L;
    float m = -INFINITY;
    for (int j = lane; j < jmax; j += 64) {
      float a = 0.f;
#pragma unroll 16
      for (int d = 0; d < 64; ++d) a = fmaf(myq[d], Ks[j * 65 + d], a);
      a *= 0.125f;
      myp[j] = a;
      m = fmaxf(m, a);
    }
    m = wave_max(m);
    float z = 0.f;
    for (int j = lane; j < jmax; j += 64) {
      const float e = expf(myp[j] - m);
      myp[j] = e;
      z += e;
    }
    z = wave_sum(z);
    __builtin_amdgcn_wave_barrier();
    float o = 0.f;
    for (int j = 0; j < jmax; ++j) o = fmaf(myp[j], Vs[j * 64 + lane], o);
    out[((size_t)seq * L + q) * D + h * 64 + lane] = o / z;
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- fp32 parity arm on the matrix pipe (round 5) ---------------------------------------------------------------------
// The bidirectional (vision-tower) form of the fp32 attention on v_mfma_f32_16x16x4_f32: exact fp32 products, fp32
// accumulation — the arithmetic of the VALU kernel above (q.k summed in another order; softmax with the same expf; tests hold
// both to the oracle at 1e-4 / 1e-5), 14 x its speed: the VALU kernel was 75 of the exact-fp32 arm's 207 ms per 512 images.
//   S^T = K Q^T:  A = K tile (lane (fr, g): key fr, dims 16 g + j over the 16 MFMAs j of a tile), B = Q^T (query fr, the same
//                 dims) — the matrix pipe's four k-slots carry dims {j, 16 + j, 32 + j, 48 + j}, so a lane's 16 operand values
//                 are 16 CONSECUTIVE floats of its row (four ds_read_b128 / four global 16-byte loads);
//                 result: lane (fr, g) holds keys 4 g + r (r = 0..3) of the tile for query fr — the softmax row in registers.
//   O^T = V^T P^T: MFMA r of a tile takes keys {4 g + r}: B = the lane's own P value r, A = V[key 4 g + r][dim block + fr].
// K and V of the head sit in LDS as fp32 rows of 68 floats (272 B: the 16 rows of a b128 fragment read fall on distinct bank
// groups).  One workgroup per (sequence, head), 8 waves, q-blocks dealt round-robin; the CLS-only last layer asks for qrows = 1.
template <int NT, int NW>
__global__ __launch_bounds__(NW * 64, 1) void attn_f32_mfma_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                   int L, int heads, int qrows) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int LP = NT * 16, RS = 68;
  float* Ks = (float*)smem;        // [LP][68]
  float* Vs = Ks + LP * RS;        // [LP][68]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int seq = blockIdx.x / heads, h = blockIdx.x - seq * heads;
  const int D = heads * 64;
  const size_t rs = (size_t)3 * D;
  const float* base = qkv + (size_t)seq * L * rs + h * 64;
  const int fr = lane & 15, g = lane >> 4;
  for (int i = threadIdx.x; i < LP * 16; i += NW * 64) {   // 16 float4 per row
    const int row = i >> 4, c = (i & 15) * 4;
    f32x4_t kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
    if (row < L) {
      kv = *(const f32x4_t*)(base + (size_t)row * rs + D + c);
      vv = *(const f32x4_t*)(base + (size_t)row * rs + 2 * D + c);
    }
    *(f32x4_t*)(Ks + row * RS + c) = kv;
    *(f32x4_t*)(Vs + row * RS + c) = vv;
  }
  __syncthreads();
  const int nqb = (qrows + 15) / 16;
  for (int qb = wave; qb < nqb; qb += NW) {
    const int q = qb * 16 + fr;
    float qv[16];
    {
      const float* qp = base + (size_t)min(q, L - 1) * rs + 16 * g;
#pragma unroll
      for (int j4 = 0; j4 < 4; ++j4) {
        const f32x4_t t = *(const f32x4_t*)(qp + 4 * j4);
        qv[4 * j4] = t[0]; qv[4 * j4 + 1] = t[1]; qv[4 * j4 + 2] = t[2]; qv[4 * j4 + 3] = t[3];
      }
    }
    f32x4_t s[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float* kp = Ks + (t * 16 + fr) * RS + 16 * g;
      f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j4 = 0; j4 < 4; ++j4) {
        const f32x4_t kk = *(const f32x4_t*)(kp + 4 * j4);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kk[e], qv[4 * j4 + e], acc, 0, 0, 0);
      }
      s[t] = acc;
      __builtin_amdgcn_sched_barrier(0);  // (without it hipcc hoists every tile's fragment reads: 256 VGPRs + scratch)
    }
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool ok = t * 16 + g * 4 + r < L;
        s[t][r] = ok ? s[t][r] * 0.125f : -INFINITY;
        m = fmaxf(m, s[t][r]);
      }
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float z = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = expf(s[t][r] - m);   // (-inf - m: 0)
        s[t][r] = e;
        z += e;
      }
    }
    z += __shfl_xor(z, 16, 64);
    z += __shfl_xor(z, 32, 64);
    f32x4_t o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* vp = Vs + (t * 16 + 4 * g + r) * RS + fr;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vp[dt * 16], s[t][r], o[dt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (q < L && q < qrows) {
      const float rz = 1.0f / z;
      float* orow = out + ((size_t)seq * L + q) * D + h * 64 + 4 * g;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) *(f32x4_t*)(orow + dt * 16) = o[dt] * rz;
    }
  }
}

template <int NT, int NW>
hipError_t launch_f32_mfma(const void* qkv, void* out, int nseq, int L, int heads, int qrows, hipStream_t s) {
  constexpr int lds = NT * 16 * 68 * 2 * (int)sizeof(float);
  static_assert(lds <= 160 * 1024, "K and V of one head must fit the LDS");
  static PerDeviceFlag attr_set;
  if (!attr_set.get()) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_f32_mfma_kernel<NT, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    attr_set.set();
  }
  hipLaunchKernelGGL((attn_f32_mfma_kernel<NT, NW>), dim3(nseq * heads), dim3(NW * 64), lds, s, (const float*)qkv, (float*)out, L,
                     heads, qrows);
  return hipGetLastError();
}

#ifdef MCM_HARNESS
template <int PREC, int LP>
hipError_t launch_bf16(const void* qkv, void* out, int nseq, int L, int heads, bool causal,
                       int qrows, hipStream_t s, int rev) {
  constexpr int lds = LP * 128 + 64 * vt_stride(LP);
  static PerDeviceFlag attr_set;
  if (!attr_set.get()) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_bf16_kernel<PREC, LP, false>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)attn_bf16_kernel<PREC, LP, true>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    attr_set.set();
  }
  if (causal)
    hipLaunchKernelGGL((attn_bf16_kernel<PREC, LP, true>), dim3(nseq * heads), dim3(256), lds, s,
                       (const uint16_t*)qkv, (uint16_t*)out, L, heads, qrows, rev);
  else
    hipLaunchKernelGGL((attn_bf16_kernel<PREC, LP, false>), dim3(nseq * heads), dim3(256), lds, s,
                       (const uint16_t*)qkv, (uint16_t*)out, L, heads, qrows, rev);
  return hipGetLastError();
}

#endif

#ifdef MCM_HARNESS
int g_attn_variant = 1;  // 1 = the shipped policy, 0 = attn_bf16_kernel (round 1), 10 / 11 = XCD-aware / rotated deals of
                         // attn_tr_kernel, 21 = the persistent form at every size, 36 = attn_tr_kernel at every size
#endif

template <int PREC, int NT, int NW, int OCC, bool X2 = false, int NFULL = 0>
hipError_t launch_tr(const void* qkv, void* out, int nseq, int L, int heads, bool causal, int qrows,
                     hipStream_t s, int rev, int hm) {
  constexpr int lds = NT * 16 * 128 * (X2 ? 4 : 2);
  static PerDeviceFlag attr_set;
  if (!attr_set.get()) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_tr_kernel<PREC, NT, false, NW, OCC, X2, NFULL>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if constexpr (!X2) {
      if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)attn_tr_kernel<PREC, NT, true, NW, OCC, X2>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    }
    if (e != hipSuccess) return e;
    attr_set.set();
  }
  if constexpr (X2) {  // (the vision tower only: no causal form)
    if (causal) return hipErrorInvalidValue;
    hipLaunchKernelGGL((attn_tr_kernel<PREC, NT, false, NW, OCC, true, NFULL>), dim3(nseq * heads), dim3(NW * 64), lds, s,
                       (const uint16_t*)qkv, (uint16_t*)out, L, heads, qrows, rev, hm);
    return hipGetLastError();
  } else {
  if (causal)
    hipLaunchKernelGGL((attn_tr_kernel<PREC, NT, true, NW, OCC>), dim3(nseq * heads), dim3(NW * 64), lds, s,
                       (const uint16_t*)qkv, (uint16_t*)out, L, heads, qrows, rev, hm);
  else
    hipLaunchKernelGGL((attn_tr_kernel<PREC, NT, false, NW, OCC, false, NFULL>), dim3(nseq * heads), dim3(NW * 64), lds, s,
                       (const uint16_t*)qkv, (uint16_t*)out, L, heads, qrows, rev, hm);
  return hipGetLastError();
  }
}

// split-activation arm (fp16): key tiles of the three checkpoint geometries (B/32: 50 tokens = 4 tiles; B/16: 197 = 13;
// L/14: 257 = 17) plus the small test towers
hipError_t launch_tr_x2(const void* qkv, void* out, int nseq, int L, int heads, int qrows, hipStream_t s, int rev) {
  const int nt = (L + 15) / 16;
#define MCM_TRX(N, W) \
  if (nt <= N) return launch_tr<MCM_PREC_F16, N, W, 1, true>(qkv, out, nseq, L, heads, false, qrows, s, rev, 0)
  if (nt == 13) return launch_tr<MCM_PREC_F16, 13, 8, 1, true, 12>(qkv, out, nseq, L, heads, false, qrows, s, rev, 0);
  MCM_TRX(2, 4); MCM_TRX(4, 4); MCM_TRX(8, 4); MCM_TRX(13, 8); MCM_TRX(17, 8); MCM_TRX(18, 8);
#undef MCM_TRX
  return hipErrorInvalidValue;
}



// Waves per workgroup.  The q-blocks of a sequence are dealt round-robin to the waves, so the slowest wave has
// ceil(q-blocks / waves) of them.  Measured at B/16 batch 512 / L/14 batch 256 (tools/attn_probe.py — removed in round 6 —, same
// box, old kernel 215 / 295 us): 4 waves (4-3-3-3 blocks) 164 - 167 / 183 - 186 us, 5 - 6 waves 171 - 178, 7 waves
// 154 - 168, **8 waves (2-2-2-2-2-1-1-1) 149 - 153 / 156 - 158 us**; occupancy bounds 1 ... 4 waves per SIMD make no
// difference at 8 waves (90 / 116 VGPRs either way), 6 and more cost spills.  Results are bit-identical
// for every wave count (a q-block's arithmetic does not depend on which wave runs it).
template <int PREC>
hipError_t launch_tr_by_tiles(const void* qkv, void* out, int nseq, int L, int heads, bool causal, int qrows,
                              hipStream_t s, int rev, int hm, unsigned int* fault) {
  const int nt = (L + 15) / 16;
#ifdef MCM_HARNESS
  if (g_attn_variant == 10 && nseq % 8 == 0) rev |= 2;  // XCD-aware deal of the (sequence, head) workgroups
  if (g_attn_variant == 11) rev |= 4;                   // q-blocks dealt to the waves rotated per workgroup (SIMD balance)
#endif
  // B/16's 13 key tiles, every query row, from 16 jobs per CU on (batch 342 at 12 heads): the persistent form — loader waves and
  // compute waves side by side (-3.5 ... -5.5 % per launch at batch 512 / 768, break-even at 256, slower below: tools/attn_sweep.py)
  bool persistent = nt == 13 && !causal && !hm && qrows == L && (int64_t)nseq * heads >= 16 * (int64_t)device_cu_count();
#ifdef MCM_HARNESS
  if (g_attn_variant == 36) persistent = false;                      // A/B: the 8-wave kernel at every size
  if (g_attn_variant == 21) persistent = nt == 13 && !causal && !hm;  // A/B: the persistent form at every size and query count
#endif
  if (persistent) return launch_ps<PREC, 13, 12, 4>(qkv, out, nseq, L, heads, qrows, s, rev, fault);
  // the three checkpoint geometries need exactly 4 / 13 / 17 key tiles (50 / 197 / 257 tokens): every tile but the last is full
#define MCM_TR_EXACT(N, W, O) \
  if (nt == N && !causal) return launch_tr<PREC, N, W, O, false, N - 1>(qkv, out, nseq, L, heads, false, qrows, s, rev, hm)
  MCM_TR_EXACT(4, 4, 3); MCM_TR_EXACT(13, 8, 3); MCM_TR_EXACT(17, 8, 3);
#undef MCM_TR_EXACT
#define MCM_TR(N, W, O) \
  if (nt <= N) return launch_tr<PREC, N, W, O>(qkv, out, nseq, L, heads, causal, qrows, s, rev, hm)
  MCM_TR(1, 4, 3); MCM_TR(2, 4, 3); MCM_TR(3, 4, 3); MCM_TR(4, 4, 3); MCM_TR(5, 4, 3); MCM_TR(6, 4, 3);
  MCM_TR(8, 4, 3); MCM_TR(10, 8, 3); MCM_TR(13, 8, 3); MCM_TR(17, 8, 3); MCM_TR(18, 8, 3);
#undef MCM_TR
  return hipErrorInvalidValue;
}


}  // namespace

#ifdef MCM_HARNESS
void attention_set_variant(int v) { g_attn_variant = v; }
void attention_set_spin_budget(unsigned int polls) { g_ps_spin_budget = polls; }
#endif

hipError_t launch_attention(int prec, const void* qkv, void* out, int nseq, int L, int heads,
                            bool causal, int qrows, hipStream_t s, bool reverse, int hm, bool split, unsigned int* fault) {
  if (nseq <= 0 || L <= 0 || heads <= 0) return hipErrorInvalidValue;
  if (qrows <= 0 || qrows > L) qrows = L;
  if (split) {  // split images in and out (GemmArgs::xsplit): fp16, bidirectional, row-major
    if (prec != MCM_PREC_F16 || causal || hm) return hipErrorInvalidValue;
    return launch_tr_x2(qkv, out, nseq, L, heads, qrows, s, reverse ? 1 : 0);
  }
  if (hm && (prec == MCM_PREC_F32 || (int64_t)hm < (int64_t)nseq * L)) return hipErrorInvalidValue;
#ifndef MCM_HARNESS
  if (hm) return hipErrorInvalidValue;  // head-major qkv: harness library only
#endif
#ifdef MCM_HARNESS
  if (hm && g_attn_variant == 0) return hipErrorInvalidValue;  // the round-1 kernel reads row-major qkv only
#endif
#ifdef MCM_HARNESS
  if (prec != MCM_PREC_F32 && g_attn_variant == 0) {
#define MCM_ATTN_BY_LP(P)                                                                      \
  do {                                                                                          \
    if (L <= 32) return launch_bf16<P, 32>(qkv, out, nseq, L, heads, causal, qrows, s, reverse ? 1 : 0);         \
    if (L <= 64) return launch_bf16<P, 64>(qkv, out, nseq, L, heads, causal, qrows, s, reverse ? 1 : 0);         \
    if (L <= 96) return launch_bf16<P, 96>(qkv, out, nseq, L, heads, causal, qrows, s, reverse ? 1 : 0);         \
    if (L <= 128) return launch_bf16<P, 128>(qkv, out, nseq, L, heads, causal, qrows, s, reverse ? 1 : 0);       \
    if (L <= 224) return launch_bf16<P, 224>(qkv, out, nseq, L, heads, causal, qrows, s, reverse ? 1 : 0);       \
    if (L <= 288) return launch_bf16<P, 288>(qkv, out, nseq, L, heads, causal, qrows, s, reverse ? 1 : 0);       \
    return hipErrorInvalidValue;                                                                \
  } while (0)
    if (prec == MCM_PREC_F16) MCM_ATTN_BY_LP(MCM_PREC_F16);
    MCM_ATTN_BY_LP(MCM_PREC_BF16);
#undef MCM_ATTN_BY_LP
  }
#endif
  if (prec == MCM_PREC_F16)
    return launch_tr_by_tiles<MCM_PREC_F16>(qkv, out, nseq, L, heads, causal, qrows, s, reverse ? 1 : 0, hm, fault);
  if (prec == MCM_PREC_BF16)
    return launch_tr_by_tiles<MCM_PREC_BF16>(qkv, out, nseq, L, heads, causal, qrows, s, reverse ? 1 : 0, hm, fault);
  if (!causal) {  // the vision tower's form: fp32 MFMAs (attn_f32_mfma_kernel); the causal text tower keeps the VALU kernel
    const int nt = (L + 15) / 16;
#define MCM_F32A(N, W) \
  if (nt <= N) return launch_f32_mfma<N, W>(qkv, out, nseq, L, heads, qrows, s)
    MCM_F32A(2, 4); MCM_F32A(4, 4); MCM_F32A(8, 4); MCM_F32A(13, 8); MCM_F32A(18, 8);
#undef MCM_F32A
  }
  const int lds = (L * 65 + L * 64 + 4 * 64 + 4 * L) * (int)sizeof(float);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  static PerDeviceFlag attr_set;
  if (!attr_set.get()) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_f32_kernel<false>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)attn_f32_kernel<true>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set.set();
  }
  if (causal)
    hipLaunchKernelGGL(attn_f32_kernel<true>, dim3(nseq * heads), dim3(256), lds, s,
                       (const float*)qkv, (float*)out, L, heads, qrows);
  else
    hipLaunchKernelGGL(attn_f32_kernel<false>, dim3(nseq * heads), dim3(256), lds, s,
                       (const float*)qkv, (float*)out, L, heads, qrows);
  return hipGetLastError();
}
