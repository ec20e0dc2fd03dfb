// mcm_api.hip — the C ABI of libmcm_hip.so (include/mcm.h): parameter store, workspace,
// the two CLIP towers as sequences of kernel launches on the caller's stream, the fused
// scoring tail, per-kernel HIP-event timing, and operator-level entry points.
//
// Tower structure follows HF modeling_clip.py (the library the reference delegates to):
//   vision  CLIPVisionEmbeddings.forward :202-218 → pre_layrnorm :642 → CLIPEncoderLayer
//           ×L :362-383 → CLS pool + post_layernorm :650-651 → visual_projection :751
//   text    CLIPTextEmbeddings :232-256 → causal CLIPEncoderLayer ×L → final_layer_norm
//           :559 → EOS pool :561-581 → text_projection :713
// and the reference's own tail utils/detection_util.py:226,231-248.
// The residual stream is fp32 in HBM; vision GEMM operands are fp16 or bf16 (cfg.precision; exact fp32 in the
// parity mode), the text tower always runs the exact-fp32 kernels.
#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "common.hpp"

namespace {

thread_local std::string g_create_err;

struct Param {
  std::vector<int64_t> shape;
  int64_t numel = 0;
  float* dev = nullptr;
  bool set = false;
};

struct LayerW {
  void *wqkv = nullptr, *wo = nullptr, *w1 = nullptr, *w2 = nullptr;  // operand dtype
  float *bqkv = nullptr;                                               // [3D] packed
  const float *bo, *b1, *b2, *ln1w, *ln1b, *ln2w, *ln2b;               // fp32 masters
  // LayerNorm fold (vision tower, 16-bit modes): c = W gamma and b' = b + W beta of the two GEMMs that follow a LayerNorm
  float *cqkv = nullptr, *bqkvf = nullptr, *c1 = nullptr, *b1f = nullptr;
  // ROW64 arm, blocked W (harness; built on first use): wo / w2 as [K-step of 32][16-row block][1 KiB piece, chunks XOR-ed] — the
  // LDS image of a piece, contiguous in memory, so that every LDS-DMA piece is eight whole cache lines (gemm_arms.hpp ROW64)
  void *wo_blk = nullptr, *w2_blk = nullptr;
};

struct Tower {
  int D = 0, heads = 0, layers = 0, ff = 0;
  int prec = MCM_PREC_BF16;  // operand mode of this tower's GEMMs / attention / LayerNorm output
  std::vector<LayerW> L;
  bool split = false;        // GEMM weights held as W_hi + W_lo (GemmArgs::ksplit; include/mcm.h MCM_WEIGHTS_*)
};

struct EvPair {
  hipEvent_t a, b;
  int kc, kc2;  // kc2: a sub-class of kc that also gets this launch (MCM_KC_GEMM_*), -1 = none
};

}  // namespace

struct mcm_handle {
  mcm_config cfg;
  std::map<std::string, Param> params;
  bool finalized = false;
  Tower vis, txt;
  void* wpatch = nullptr;  // [v_width, kpad] operand dtype ([v_width, 2 kpad] split)
  uint64_t w_inexact = 0;  // vision GEMM-weight elements that are not operand-dtype numbers (mcm_finalize_weights)
  int kpad = 0, np = 0, ntok = 0;
  // workspace
  float* x = nullptr;
  void *ln = nullptr, *qkv = nullptr, *att = nullptr, *hbuf = nullptr, *patches = nullptr;
  float* feat = nullptr;          // [max_batch, proj_dim] scratch for mcm_score
  int32_t *ids_dev = nullptr, *rowidx_dev = nullptr;
  int32_t *ids_pin = nullptr, *rowidx_pin = nullptr;
  // mcm_resize_crop_u8 geometry: a ring of PREP_RING pinned staging buffers (max_batch entries each) so that a call never
  // has to drain the stream before it may write the next batch's geometry; prep_ev[k] = "the copy out of slot k is done"
  static constexpr int PREP_RING = 4;
  PrepImage *prep_pin = nullptr, *prep_dev = nullptr;
  int32_t* prep_coef = nullptr;  // PREP_RING x prep_coef_bytes: the resize kernel's coefficient tables (allocated by the first call)
  // mcm_jpeg_reconstruct: per-image records + quantisation tables through the same kind of ring (allocated by the first
  // call), and the sample-plane workspace (grown on demand; one per handle: the kernels of consecutive calls are stream-ordered)
  char* jpg_pin = nullptr;
  char* jpg_dev = nullptr;
  bool jpg_ready = false;
  hipEvent_t jpg_ev[PREP_RING] = {nullptr, nullptr, nullptr, nullptr};
  unsigned jpg_next = 0;
  uint8_t* jpg_planes = nullptr;
  size_t jpg_planes_bytes = 0;
  hipEvent_t prep_ev[PREP_RING] = {nullptr, nullptr, nullptr, nullptr};
  unsigned prep_next = 0;
  int64_t max_rows = 0;
  int x2_batch = 0;    // largest batch of the split-activation arm (= max_batch on fp16 handles, whose activation buffers are sized for it; 0: not fp16)
  size_t hbuf_bytes = 0;
  std::vector<void*> owned;       // every hipMalloc'd pointer
  // profiling
  bool prof = false;
  std::vector<EvPair> ev_pool;
  size_t ev_used = 0;
  double ms_acc[MCM_KC_COUNT] = {0};
  int64_t launches[MCM_KC_COUNT] = {0};
  double flops[MCM_KC_COUNT] = {0};
  bool flip = false;  // walk direction of the next kernel (next_dir)
  unsigned int* sat_dev = nullptr;  // sticky fp16 saturation counter (common.hpp sat_report)
  // sticky kernel-fault word: pinned host memory mapped into the device; a persistent kernel whose wait ran out of its budget
  // stores 1 there (attention.hip attn_ps_kernel) and every later compute call on the handle returns MCM_EHIP (check_ready)
  unsigned int* fault_pin = nullptr;
  unsigned int* fault_dev = nullptr;
  bool sat_on = true;
  float2 *fold_part = nullptr, *fold_rs = nullptr;  // LayerNorm fold: row moments [v_width / 64][rows], (rstd, mean rstd) [rows]
  // LayerNorm in the tail of the residual GEMMs (gemm.hip): per-XCD regions of counters, all zero between launches;
  // nullptr when the device did not pass the workgroup -> XCD check (xcd_round_robin) or the widths do not qualify
  unsigned int* ln_state = nullptr;
  int ln_rs = 0, ln_cap8 = 0;
  std::string err;
};

namespace {

int fail(mcm_handle* h, int code, const std::string& msg) {
  if (h) h->err = msg;
  else g_create_err = msg;
  return code;
}

#define HIP_TRY(h, expr)                                                                 \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess)                                                                \
      return fail(h, MCM_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e));       \
  } while (0)

void add_param(mcm_handle* h, const std::string& name, std::vector<int64_t> shape) {
  Param p;
  p.shape = shape;
  p.numel = 1;
  for (auto d : shape) p.numel *= d;
  h->params[name] = p;
}

void add_layer_params(mcm_handle* h, const std::string& pre, int D, int ff) {
  for (const char* pr : {"q_proj", "k_proj", "v_proj", "out_proj"}) {
    add_param(h, pre + ".self_attn." + pr + ".weight", {D, D});
    add_param(h, pre + ".self_attn." + pr + ".bias", {D});
  }
  for (const char* ln : {"layer_norm1", "layer_norm2"}) {
    add_param(h, pre + "." + ln + ".weight", {D});
    add_param(h, pre + "." + ln + ".bias", {D});
  }
  add_param(h, pre + ".mlp.fc1.weight", {ff, D});
  add_param(h, pre + ".mlp.fc1.bias", {ff});
  add_param(h, pre + ".mlp.fc2.weight", {D, ff});
  add_param(h, pre + ".mlp.fc2.bias", {D});
}

int dev_alloc(mcm_handle* h, void** out, size_t bytes) {
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
  if (e != hipSuccess)
    return fail(h, MCM_ENOMEM, "hipMalloc(" + std::to_string(bytes) + "): " + hipGetErrorString(e));
  h->owned.push_back(p);
  *out = p;
  return MCM_OK;
}

const float* W(mcm_handle* h, const std::string& name) { return h->params.at(name).dev; }

// ---- profiling wrappers ---------------------------------------------------------------
struct Scope {
  mcm_handle* h;
  hipStream_t s;
  int kc;
  EvPair* ev = nullptr;
  Scope(mcm_handle* h_, hipStream_t s_, int kc_, double fl, int kc2 = -1) : h(h_), s(s_), kc(kc_) {
    if (!h->prof) return;
    if (h->ev_used == h->ev_pool.size()) {
      EvPair e;
      (void)hipEventCreate(&e.a);
      (void)hipEventCreate(&e.b);
      h->ev_pool.push_back(e);
    }
    ev = &h->ev_pool[h->ev_used++];
    ev->kc = kc;
    ev->kc2 = kc2;
    h->launches[kc] += 1;
    h->flops[kc] += fl;
    if (kc2 >= 0) {
      h->launches[kc2] += 1;
      h->flops[kc2] += fl;
    }
    (void)hipEventRecord(ev->a, s);
  }
  ~Scope() {
    if (ev) (void)hipEventRecord(ev->b, s);
  }
};

// Consecutive kernels of a tower walk their rows in opposite directions: a producer's last rows are
// the ones still in L2 / Infinity Cache, so its consumer starts there (results do not depend on the
// order).  `flip` alternates per launch.  Measured: LayerNorm 2.19 -> 1.99 ms per step (it reads the
// residual stream the previous GEMM just wrote), +0.9 % end to end.
bool next_dir(mcm_handle* h) {
  h->flip = !h->flip;
  return h->flip;
}
#ifdef MCM_HARNESS
int g_patch_fold = 1;  // A/B (mcm_debug_patch_fold): 0 = patchify + plain patch GEMM (rounds 1 - 3), 1 = pixel-gathering patch GEMM
#else
constexpr int g_patch_fold = 1;
#endif
#ifdef MCM_HARNESS
int g_resize_fused_only = 0;  // A/B (mcm_debug_resize_fused_only): 1 = the resize kernel's rounds-2/3 form everywhere
#else
constexpr int g_resize_fused_only = 0;
#endif
#ifdef MCM_HARNESS
int g_nsplit = 1;  // A/B (mcm_debug_nsplit): the wide store GEMMs (QKV, fc1) as n launches over column blocks of N / n
#endif
hipError_t gemm(mcm_handle* h, hipStream_t s, int prec, int epi, const GemmArgs& a_in) {
  GemmArgs a = a_in;
#ifdef MCM_HARNESS
  if (g_nsplit > 1 && epi <= EPI_GELU && a.N >= 2048 && a.N % (256 * g_nsplit) == 0 && !a.fold_rs && !a.hm && a.M > 4096) {
    // W re-fetch experiment: the XCD's L2 (4 MiB) cannot hold all of W (fc1: 4.7 MB) beside the X
    // panels in flight, so W streams through it once per tile round; with the columns cut in n blocks only N / n of W
    // is live per launch (X is then read n times).  Same bits (a column's K-sum does not depend on its neighbours).
    const int n = g_nsplit, nb = a.N / n;
    const size_t wrow = (size_t)a.K * prec_esize(prec) * (a.ksplit ? 2 : 1);
    hipError_t e = hipSuccess;
    g_nsplit = 1;
    for (int i = 0; i < n && e == hipSuccess; ++i) {
      GemmArgs p = a_in;
      p.N = nb;
      p.w = (const char*)a_in.w + (size_t)i * nb * wrow;
      p.bias = a_in.bias ? a_in.bias + (size_t)i * nb : nullptr;
      p.out = (char*)a_in.out + (size_t)i * nb * prec_esize(prec);
      e = gemm(h, s, prec, epi, p);
    }
    g_nsplit = n;
    return e;
  }
#endif
  a.rev = next_dir(h) ? 1 : 0;
  a.sat = h->sat_on ? h->sat_dev : nullptr;
  // the four whole-batch shapes of an encoder layer also get a class of their own (out-proj is the HBM-bound one:
  // bench.py roofline_hbm_kernels); small launches (CLS-only last layer, short prompt banks) count in MCM_KC_GEMM only
  int shape = -1;
  if (a.M > 4096)
    shape = (epi == EPI_STORE || epi == EPI_STORE_X2) ? MCM_KC_GEMM_QKV : epi_gelu(epi) ? MCM_KC_GEMM_FC1
          : epi == EPI_RESID ? (a.N == a.K ? MCM_KC_GEMM_OUTPROJ : MCM_KC_GEMM_FC2) : -1;
  Scope sc(h, s, MCM_KC_GEMM, 2.0 * a.M * (double)a.N * a.K, shape);  // algorithmic FLOP: the logical K, split or not
  // Row padding into the workspace: the ping-pong kernel takes problems made of whole 256-row tiles only, so a
  // dense activation GEMM whose M is not a multiple of 256 is run on M rounded up.  The extra rows exist (every
  // activation buffer is allocated in whole 256-row tiles and zeroed once), every output row depends on its own
  // input row only, and the pad rows' results are never read: same bits for the real rows, and batches that are not
  // multiples of 256 images (any batch at ViT-L/14's 257 tokens but 256 k; ragged last batches) get the fast kernel.
  const bool from_row0 = a.x == h->ln || a.x == h->att || a.x == h->hbuf;  // not a chunk that starts mid-buffer
  if (epi != EPI_PATCH && from_row0 && a.M % 256 != 0 && a.N % 256 == 0 && a.ldx == a.K && a.ldo == a.N) {
    const int64_t mp = ((int64_t)a.M + 255) / 256 * 256;
    if (mp <= h->max_rows) a.M = (int)mp;
  }
  if (a.ksplit) a.K *= 2;  // the callers describe the logical problem; the split image has 2 K columns per row
  if (a.xsplit) a.ldx *= 2;           // ... and so have the rows of a split X
  if (epi_x2(epi)) a.ldo *= 2;        // ... and of a split output
  return launch_gemm(prec, epi, a, s);
}
hipError_t lnorm(mcm_handle* h, hipStream_t s, int prec, const float* x, const float* g, const float* b,
                 void* y, int M, int D, bool out_f32, bool split = false) {
  Scope sc(h, s, MCM_KC_LAYERNORM, 8.0 * M * D);
  if (split)
    return launch_layernorm(prec, x, g, b, y, M, D, h->cfg.ln_eps, false, s, 0, 0, next_dir(h),
                            h->sat_on ? h->sat_dev : nullptr, true);
#ifdef MCM_HARNESS
  // timing experiment (results are garbage): what a tower without its big LayerNorm launches would cost.
  // MCM_ABL_SKIP_LN=1: nothing in their place; =2: a write of the 16-bit output's size (the extra epilogue store
  // of a LayerNorm folded into the neighbouring GEMMs)
  static const char* skip = getenv("MCM_ABL_SKIP_LN");
  if (skip && M > 4096 && !out_f32) {
    if (skip[0] == '2') return hipMemsetAsync(h->qkv, 0, (size_t)M * D * prec_esize(prec), s);  // dead at both LN sites
    return hipSuccess;
  }
#endif
  return launch_layernorm(prec, x, g, b, y, M, D, h->cfg.ln_eps, out_f32, s, 0, 0, next_dir(h),
                          h->sat_on ? h->sat_dev : nullptr);
}
// seq0: first sequence of the launch (a chunk of the batch starts there)
hipError_t attn(mcm_handle* h, hipStream_t s, int prec, int nseq, int L, int heads, bool causal,
                int qrows = 0, int seq0 = 0, int hm = 0, bool split = false) {
  const int q = qrows > 0 ? qrows : L;
  Scope sc(h, s, MCM_KC_ATTENTION, 4.0 * nseq * heads * (double)q * L * 64 * (causal ? 0.5 : 1.0));
  if (split) return launch_attention(prec, h->qkv, h->att, nseq, L, heads, causal, qrows, s, next_dir(h), 0, true);
  const size_t es = prec_esize(prec), D = (size_t)heads * 64, row0 = (size_t)seq0 * L;
  return launch_attention(prec, (const char*)h->qkv + row0 * 3 * D * es, (char*)h->att + row0 * D * es, nseq, L,
                          heads, causal, qrows, s, next_dir(h), hm, false, h->fault_dev);
}
// A/B arm (harness: mcm_debug_qkv_head_major): qkv of the 16-bit towers head-major ([3 heads][rows][64],
// GemmArgs::hm) between the QKV projection and attention.  Bit-identical; attention 1.63 -> 1.56 ms per step, the QKV
// projection's stores +0.04 ... 0.08 ms: no net gain, the shipped library keeps [rows][3 D].
#ifdef MCM_HARNESS
int g_qkv_head_major = 0;
#else
constexpr int g_qkv_head_major = 0;
#endif
#ifdef MCM_HARNESS
int g_qkv_chunks = 1;  // A/B: QKV projection + attention per chunk of the batch (qkv of a chunk stays in the Infinity Cache)
int g_ln_fold = 0;     // A/B: 1 = LayerNorm fold (mcm_debug_ln_fold); 0 = every LayerNorm as its own launch (shipped)
#else
constexpr int g_qkv_chunks = 1;
#ifdef MCM_LN_FOLD  // A/B build of the shipped library with the fold on (make fold: libmcm_hip_fold.so, tools/bench_with_lib.py)
constexpr int g_ln_fold = 1;
#else
constexpr int g_ln_fold = 0;
#endif
#endif
// rows of a dense activation GEMM as gemm() runs it (whole 256-row tiles when the workspace has them)
int64_t padded_rows(const mcm_handle* h, int M) {
  const int64_t mp = ((int64_t)M + 255) / 256 * 256;
  return mp <= h->max_rows ? mp : M;
}
hipError_t fold_stats(mcm_handle* h, hipStream_t s, int Mp, int D) {
  Scope sc(h, s, MCM_KC_LAYERNORM, 4.0 * Mp * (D / 64));
  return launch_fold_stats(h->fold_part, D / 64, Mp, D, h->cfg.ln_eps, h->fold_rs, s);
}
hipError_t lnorm_strided(mcm_handle* h, hipStream_t s, int prec, const float* x, const float* g,
                         const float* b, void* y, int M, int D, size_t xs, size_t ys, bool split = false) {
  Scope sc(h, s, MCM_KC_LAYERNORM, 8.0 * M * D);
  return launch_layernorm(prec, x, g, b, y, M, D, h->cfg.ln_eps, false, s, xs, ys, false,
                          h->sat_on ? h->sat_dev : nullptr, split);
}

// LayerNorm in the tail of the residual GEMMs (gemm.hip "LayerNorm in the tail"): 1 = the LayerNorm that follows a
// whole-batch out-proj / fc2 of a 16-bit tower is computed by that GEMM's own waves; 0 = every LayerNorm is a launch
// An A/B arm: bit-identical, equal at ViT-B/16 batch 512, +0.5 % at ViT-L/14, -1 ... -7 % on smaller problems
#ifdef MCM_HARNESS
int g_ln_tail = 0;  // mcm_debug_ln_tail
int g_ln_cluster = 0;  // mcm_debug_ln_cluster: LayerNorm by the row panel's cluster of workgroups (gemm_arms.hpp "LNC", round 6)
int g_ln_row = 0;      // mcm_debug_ln_row: out-proj / fc2 + the LayerNorm behind them as 64-row FULL-ROW tiles (gemm_arms.hpp ROW64, R6.7)
int g_lnc_spin = -1;   // mcm_debug_ln_cluster_spin: < 0 the first form (wait for the partners); n >= 0 the defer form: n polls, then
                       // the segment is left to launch_lnc_cleanup behind the GEMM
#elif defined(MCM_LN_TAIL)  // A/B build of the shipped library with the tail on
constexpr int g_ln_tail = 1;
#else
constexpr int g_ln_tail = 0;
#endif
#ifndef MCM_HARNESS
constexpr int g_ln_cluster = 0, g_lnc_spin = -1, g_ln_row = 0;
#endif
// The tail's coherence argument needs every workgroup with the same blockIdx & 7 on the same XCD (one L2).  That is
// how the dispatcher deals workgroups in the default (SPX) mode; it is checked on the device, once per handle, with
// the grid the persistent kernels use: XCC_ID of every workgroup.  Anything else (another partition mode, a masked
// lease) and the tail is simply not used.
__global__ void xcc_probe_kernel(unsigned int* out) {
  if (threadIdx.x == 0) {
    unsigned int id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    out[blockIdx.x] = id & 0xfu;
  }
}
bool xcd_round_robin(mcm_handle* h, int grid) {
  if (grid < 8 || grid % 8 || grid > 1024) return false;
  unsigned int* dev = nullptr;
  if (hipMalloc((void**)&dev, (size_t)grid * sizeof(unsigned int)) != hipSuccess) return false;
  std::vector<unsigned int> ids((size_t)grid, 0xffu);
  bool ok = true;
  for (int rep = 0; rep < 3 && ok; ++rep) {  // (the mapping must not depend on what ran before)
    hipLaunchKernelGGL(xcc_probe_kernel, dim3(grid), dim3(512), 0, 0, dev);
    ok = hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess &&
         hipMemcpy(ids.data(), dev, (size_t)grid * sizeof(unsigned int), hipMemcpyDeviceToHost) == hipSuccess;
    for (int b = 0; ok && b < grid; ++b) ok = ids[(size_t)b] == ids[(size_t)(b & 7)];
    for (int i = 0; ok && i < 8; ++i)
      for (int j = 0; ok && j < i; ++j) ok = ids[(size_t)i] != ids[(size_t)j];
  }
  (void)hipFree(dev);
  return ok;
}

// CLIPEncoderLayer.forward ×layers on x [nseq*L, D] (fp32, in place).
// pooled_row0: the caller consumes only row 0 of every sequence (CLS pooling, HF
// modeling_clip.py:649-651).  The last layer then computes K/V for all tokens but Q, the
// attention output, out_proj, LN2 and the MLP for row 0 only — identical results for the
// consumed rows (every op after attention is row-wise), 1/12 less work for a 12-layer tower.
// ln1_of_layer0_done: the caller has already produced layer 0's layer_norm1 output in h->ln (the vision tower fuses it
// with pre_layrnorm, launch_layernorm_pre)
// fold_ok (vision tower) and the harness switch mcm_debug_ln_fold(1): LayerNorm fold, an A/B arm (measured 1 % slower
// end to end than the LayerNorm launches; the shipped library never takes it).  The LayerNorm between a
// residual GEMM and the GEMM behind it is not launched: the residual epilogue also writes z = gamma o x (into h->ln,
// where the LayerNorm output would have gone) and the row moments, fold_stats turns those into (rstd, mean rstd) per
// row, and the consumer's epilogue normalises (gemm.hip, "LayerNorm fold").  Not for the layer the caller pools row 0 of
// (its LayerNorms see other rows / strides) and not for layer 0's layer_norm1 (fused with pre_layrnorm by the caller).
// x2 (fp16 vision tower; the split-activation arm, mcm_score_x2): every activation that feeds an MFMA — LayerNorm outputs,
// q / k / v, the attention output, the QuickGELU output — is carried as a split image (hi + lo, twice the columns) and every
// GEMM runs GemmArgs::xsplit; the residual stream, LayerNorm statistics and softmax are fp32 as always.
int run_layers(mcm_handle* h, hipStream_t s, const Tower& t, int nseq, int L, bool causal,
               bool pooled_row0, bool ln1_of_layer0_done = false, bool fold_ok = false, bool x2 = false) {
  const int M = nseq * L, D = t.D, P = t.prec;
  const int es = prec_esize(P);
  const int Mp = (int)padded_rows(h, M);
  const int ks = t.split ? 1 : 0;
  const int xs = x2 ? 1 : 0;
  const int epi_store = x2 ? EPI_STORE_X2 : EPI_STORE, epi_act = x2 ? EPI_GELU_X2 : EPI_GELU;
  if (x2) fold_ok = false;
  const size_t wrow = (size_t)D * es * (t.split ? 2 : 1);  // bytes per row of a [*, D] weight image
  // producer form: 1 = fused into the residual GEMM's epilogue (ping-pong kernel), 2 = plain residual GEMM + fold_rows
  // (tile kernel: small batches) - bit-identical; the consumers need a fold epilogue in whichever kernel they take
  const int pkind = gemm_fold_kind(EPI_RESID, Mp, D);
  const bool can_fold = fold_ok && g_ln_fold && g_qkv_chunks == 1 && h->fold_part && P != MCM_PREC_F32 && !t.split &&
                        t.L[0].cqkv != nullptr && pkind != 0 && gemm_fold_kind(EPI_STORE, Mp, 3 * D) != 0 &&
                        gemm_fold_kind(EPI_GELU, Mp, t.ff) != 0;
  auto produce = [&](GemmArgs& g, const float* gamma) -> hipError_t {  // residual GEMM + z / moments / statistics
    if (pkind == 1) { g.fold_z = h->ln; g.fold_g = gamma; g.fold_part = h->fold_part; }
    hipError_t e = gemm(h, s, P, EPI_RESID, g);
    if (e == hipSuccess && pkind == 2) {
      Scope sc(h, s, MCM_KC_LAYERNORM, 6.0 * Mp * D);
      e = launch_fold_rows(P, h->x, gamma, h->ln, h->fold_part, Mp, D, s, h->sat_on ? h->sat_dev : nullptr);
    }
    return e == hipSuccess ? fold_stats(h, s, Mp, D) : e;
  };
  bool ln1_folded = false;  // h->ln holds gamma1 o x and h->fold_rs the row statistics of this layer's layer_norm1
  // LayerNorm in the tail: the residual GEMMs of whole-batch layers also produce the LayerNorm that follows them
  const bool tail_ok0 = g_ln_tail && !x2 && !can_fold && !t.split && h->ln_state && Mp % 256 == 0 && Mp / 256 <= h->ln_cap8 * 8 &&
                        gemm_ln_tail_ok(P, Mp, D);
  // LayerNorm by the row panel's cluster (LNC, harness arm): the same hand-over of gamma / beta / output / counters, plus the
  // slot-moment buffer; the residual epilogue itself writes the LayerNorm output (no idle-wave tickets, no re-read of x)
  const bool cluster_ok = g_ln_cluster && !g_ln_tail && !x2 && !can_fold && !t.split && h->ln_state && h->fold_part && Mp % 256 == 0 &&
                          2 * ((Mp / 256 + 7) / 8) <= h->ln_cap8 && gemm_ln_tail_ok(P, Mp, D);
  auto with_tail = [&](GemmArgs& g, const float* gamma, const float* beta) {
    g.ln_g = gamma; g.ln_b = beta; g.ln_y = h->ln; g.ln_eps = h->cfg.ln_eps;
    g.ln_state = h->ln_state; g.ln_rs = h->ln_rs; g.ln_cap8 = h->ln_cap8;
    if (cluster_ok) { g.lnc = 1; g.fold_part = h->fold_part; g.lnc_spin = g_lnc_spin; }
  };
  // LNC defer form: the clean-up launch behind a residual GEMM that ran with GemmArgs::lnc (a few hundred workgroups that read one
  // mask word and exit, plus the segments the GEMM's waves did not wait for)
  auto lnc_cleanup = [&](const float* gamma, const float* beta) -> hipError_t {
#ifdef MCM_HARNESS
    if (!cluster_ok || g_lnc_spin < 0) return hipSuccess;
    Scope sc(h, s, MCM_KC_LAYERNORM, 8.0 * Mp * D);
    return launch_lnc_cleanup(P, h->x, gamma, beta, h->ln, h->fold_part, Mp, D, h->cfg.ln_eps, h->ln_state, h->ln_rs, h->ln_cap8, s,
                              h->sat_on ? h->sat_dev : nullptr);
#else
    return hipSuccess;
#endif
  };
  // ROW64 arm (harness): the residual GEMM as 64-row full-row tiles whose epilogue writes x and the LayerNorm output
  const bool row_ok = g_ln_row && !g_ln_tail && !g_ln_cluster && !x2 && !can_fold && !t.split && P != MCM_PREC_F32 && Mp % 64 == 0 &&
                      (D == 768 || D == 1024) && t.ff % 128 == 0;
  auto row_gemm = [&](GemmArgs g, const float* gamma, const float* beta, void** blk) -> hipError_t {
#ifdef MCM_HARNESS
    g.ln_g = gamma; g.ln_b = beta; g.ln_y = h->ln; g.ln_eps = h->cfg.ln_eps;
    g.M = Mp; g.sat = h->sat_on ? h->sat_dev : nullptr;
    if (g_ln_row >= 3) {   // blocked W: built once per weight (not inside a graph capture: the first call allocates)
      if (!*blk) {
        if (dev_alloc(h, blk, (size_t)g.N * g.K * 2)) return hipErrorOutOfMemory;
        hipError_t e = launch_row64_block_w(g.w, *blk, g.N, g.K, s);
        if (e != hipSuccess) return e;
      }
      g.w = *blk;
      g.wblk = 1;
    }
    Scope sc(h, s, MCM_KC_GEMM, 2.0 * g.M * (double)g.N * g.K, g.N == g.K ? MCM_KC_GEMM_OUTPROJ : MCM_KC_GEMM_FC2);
    return launch_gemm_row64_ln(P, g, s, (g_ln_row == 2 || g_ln_row == 4) ? 3 : 2);
#else
    (void)g; (void)gamma; (void)beta; (void)blk;
    return hipErrorInvalidValue;
#endif
  };
  const bool tail_ok = tail_ok0 || cluster_ok || row_ok;   // any of the arms: the residual GEMM also produces the LayerNorm behind it
  bool ln1_by_tail = false;  // h->ln already holds this layer's layer_norm1 (written by the previous layer's fc2)
  for (int l = 0; l < t.layers; ++l) {
    const LayerW& w = t.L[l];
    const bool cls = pooled_row0 && l == t.layers - 1 && L > 1;
    if (!(l == 0 && ln1_of_layer0_done) && !ln1_folded && !ln1_by_tail)
      HIP_TRY(h, lnorm(h, s, P, h->x, w.ln1w, w.ln1b, h->ln, M, D, false, x2));
    ln1_by_tail = false;
    if (!cls) {
      const int nch = (!x2 && g_qkv_chunks > 1 && nseq % g_qkv_chunks == 0) ? g_qkv_chunks : 1;
      for (int c = 0; c < nch; ++c) {
        const int sq = nseq / nch, r0 = c * sq * L;
        GemmArgs a{};
        a.x = (const char*)h->ln + (size_t)r0 * D * es; a.w = w.wqkv; a.bias = w.bqkv;
        a.out = (char*)h->qkv + (size_t)r0 * 3 * D * es;
        a.M = sq * L; a.N = 3 * D; a.K = D; a.ldx = D; a.ldo = 3 * D; a.ksplit = ks; a.xsplit = xs;
        if (ln1_folded) { a.bias = w.bqkvf; a.fold_rs = h->fold_rs; a.fold_c = w.cqkv; }
        // whole-batch launches of a 16-bit tower hand q / k / v over head-major (same bytes in h->qkv, other order;
        // the row-0-only layer below and the fp32 towers keep [rows][3 D])
        const int hm = (g_qkv_head_major && !x2 && nch == 1 && P != MCM_PREC_F32 && t.heads * 64 == D) ? Mp : 0;
        a.hm = hm;
        HIP_TRY(h, gemm(h, s, P, epi_store, a));
        HIP_TRY(h, attn(h, s, P, sq, L, t.heads, causal, 0, c * sq, hm, x2));
      }
    } else {
      GemmArgs kv{};  // K and V of every token: weight rows [D, 3D), output columns [D, 3D)
      kv.x = h->ln; kv.w = (const char*)w.wqkv + (size_t)D * wrow; kv.bias = w.bqkv + D;
      kv.out = (char*)h->qkv + (size_t)D * es * (1 + xs);  // (split rows: D logical columns are 2 D elements)
      kv.M = M; kv.N = 2 * D; kv.K = D; kv.ldx = D; kv.ldo = 3 * D; kv.ksplit = ks; kv.xsplit = xs;
      HIP_TRY(h, gemm(h, s, P, epi_store, kv));
      GemmArgs q{};   // Q of row 0 of every sequence (row stride L*D in, L*3D out)
      q.x = h->ln; q.w = w.wqkv; q.bias = w.bqkv; q.out = h->qkv;
      q.M = nseq; q.N = D; q.K = D; q.ldx = L * D; q.ldo = L * 3 * D; q.ksplit = ks; q.xsplit = xs;
      HIP_TRY(h, gemm(h, s, P, epi_store, q));
      HIP_TRY(h, attn(h, s, P, nseq, L, t.heads, causal, 1, 0, 0, x2));
    }
    const int Mr = cls ? nseq : M;            // rows that continue
    const int rs = cls ? L * D : D;           // their stride in x / att
    const bool fold2 = can_fold && !cls;      // layer_norm2 folded into out-proj / fc1
    GemmArgs o{};
    o.x = h->att; o.w = w.wo; o.bias = w.bo; o.resid = h->x;
    o.M = Mr; o.N = D; o.K = D; o.ldx = rs; o.ldo = rs; o.ksplit = ks; o.xsplit = xs;
    if (fold2) {
      HIP_TRY(h, produce(o, w.ln2w));
    } else if (row_ok && !cls) {
      HIP_TRY(h, row_gemm(o, w.ln2w, w.ln2b, const_cast<void**>(&t.L[l].wo_blk)));
    } else if (tail_ok && !cls) {
      with_tail(o, w.ln2w, w.ln2b);  // layer_norm2 by the out-proj kernel's idle waves
      HIP_TRY(h, gemm(h, s, P, EPI_RESID, o));
      HIP_TRY(h, lnc_cleanup(w.ln2w, w.ln2b));
    } else {
      HIP_TRY(h, gemm(h, s, P, EPI_RESID, o));
      if (!cls) HIP_TRY(h, lnorm(h, s, P, h->x, w.ln2w, w.ln2b, h->ln, M, D, false, x2));
      else HIP_TRY(h, lnorm_strided(h, s, P, h->x, w.ln2w, w.ln2b, h->ln, Mr, D, (size_t)rs, (size_t)D * (1 + xs), x2));
    }
    GemmArgs f1{};
    f1.x = h->ln; f1.w = w.w1; f1.bias = w.b1; f1.out = h->hbuf;
    f1.M = Mr; f1.N = t.ff; f1.K = D; f1.ldx = D; f1.ldo = t.ff; f1.ksplit = ks; f1.xsplit = xs;
    if (fold2) { f1.bias = w.b1f; f1.fold_rs = h->fold_rs; f1.fold_c = w.c1; }
    HIP_TRY(h, gemm(h, s, P, epi_act, f1));
    // the next layer's layer_norm1 folded into fc2 / the next QKV projection (not into the row-0-only layer)
    ln1_folded = fold2 && l + 1 < t.layers && !(pooled_row0 && l + 1 == t.layers - 1 && L > 1);
    GemmArgs f2{};
    f2.x = h->hbuf; f2.w = w.w2; f2.bias = w.b2; f2.resid = h->x;
    f2.M = Mr; f2.N = D; f2.K = t.ff; f2.ldx = t.ff; f2.ldo = rs; f2.ksplit = ks; f2.xsplit = xs;
    if (ln1_folded) {
      HIP_TRY(h, produce(f2, t.L[l + 1].ln1w));
    } else {
      if (row_ok && !cls && l + 1 < t.layers) {
        ln1_by_tail = true;
        HIP_TRY(h, row_gemm(f2, t.L[l + 1].ln1w, t.L[l + 1].ln1b, const_cast<void**>(&t.L[l].w2_blk)));
      } else {
        if (tail_ok && !cls && l + 1 < t.layers) {  // the next layer's layer_norm1 (all rows, also in front of a row-0-only layer)
          with_tail(f2, t.L[l + 1].ln1w, t.L[l + 1].ln1b);
          ln1_by_tail = true;
        }
        HIP_TRY(h, gemm(h, s, P, EPI_RESID, f2));
        if (ln1_by_tail) HIP_TRY(h, lnc_cleanup(t.L[l + 1].ln1w, t.L[l + 1].ln1b));
      }
    }
  }
  return MCM_OK;
}

int build_tower(mcm_handle* h, Tower& t, const std::string& tower, hipStream_t s) {
  const int prec = t.prec, D = t.D, ff = t.ff;
  const int es = prec_esize(prec) * (t.split ? 2 : 1);  // bytes per logical weight element
  auto cvt = [&](const float* src, void* dst, int rows, int cols) {
    return t.split ? launch_cvt_weight_split(prec, src, dst, rows, cols, cols, s)
                   : launch_cvt_weight(prec, src, dst, rows, cols, cols, s);
  };
  t.L.resize(t.layers);
  for (int l = 0; l < t.layers; ++l) {
    LayerW& w = t.L[l];
    const std::string pre = tower + ".encoder.layers." + std::to_string(l);
    int rc;
    if ((rc = dev_alloc(h, &w.wqkv, (size_t)3 * D * D * es))) return rc;
    if ((rc = dev_alloc(h, &w.wo, (size_t)D * D * es))) return rc;
    if ((rc = dev_alloc(h, &w.w1, (size_t)ff * D * es))) return rc;
    if ((rc = dev_alloc(h, &w.w2, (size_t)D * ff * es))) return rc;
    if ((rc = dev_alloc(h, (void**)&w.bqkv, (size_t)3 * D * sizeof(float)))) return rc;
    const char* parts[3] = {"q_proj", "k_proj", "v_proj"};
    for (int p = 0; p < 3; ++p) {
      HIP_TRY(h, cvt(W(h, pre + ".self_attn." + parts[p] + ".weight"), (char*)w.wqkv + (size_t)p * D * D * es, D, D));
      HIP_TRY(h, hipMemcpyAsync(w.bqkv + (size_t)p * D, W(h, pre + ".self_attn." + parts[p] + ".bias"),
                                (size_t)D * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    HIP_TRY(h, cvt(W(h, pre + ".self_attn.out_proj.weight"), w.wo, D, D));
    HIP_TRY(h, cvt(W(h, pre + ".mlp.fc1.weight"), w.w1, ff, D));
    HIP_TRY(h, cvt(W(h, pre + ".mlp.fc2.weight"), w.w2, D, ff));
    if (&t == &h->vis && h->fold_part && !t.split) {  // LayerNorm fold: the column vectors of the two LayerNorm consumers
      if ((rc = dev_alloc(h, (void**)&w.cqkv, (size_t)3 * D * sizeof(float)))) return rc;
      if ((rc = dev_alloc(h, (void**)&w.bqkvf, (size_t)3 * D * sizeof(float)))) return rc;
      if ((rc = dev_alloc(h, (void**)&w.c1, (size_t)ff * sizeof(float)))) return rc;
      if ((rc = dev_alloc(h, (void**)&w.b1f, (size_t)ff * sizeof(float)))) return rc;
      HIP_TRY(h, launch_fold_prep(prec, w.wqkv, W(h, pre + ".layer_norm1.weight"), W(h, pre + ".layer_norm1.bias"),
                                  w.bqkv, w.cqkv, w.bqkvf, 3 * D, D, s));
      HIP_TRY(h, launch_fold_prep(prec, w.w1, W(h, pre + ".layer_norm2.weight"), W(h, pre + ".layer_norm2.bias"),
                                  W(h, pre + ".mlp.fc1.bias"), w.c1, w.b1f, ff, D, s));
    }
    w.bo = W(h, pre + ".self_attn.out_proj.bias");
    w.b1 = W(h, pre + ".mlp.fc1.bias");
    w.b2 = W(h, pre + ".mlp.fc2.bias");
    w.ln1w = W(h, pre + ".layer_norm1.weight");
    w.ln1b = W(h, pre + ".layer_norm1.bias");
    w.ln2w = W(h, pre + ".layer_norm2.weight");
    w.ln2b = W(h, pre + ".layer_norm2.bias");
  }
  return MCM_OK;
}

int check_ready(mcm_handle* h) {
  if (!h) return MCM_EINVAL;
  if (!h->finalized) return fail(h, MCM_ENOWEIGHT, "mcm_finalize_weights has not been called");
  if (h->fault_pin && *(volatile unsigned int*)h->fault_pin)
    return fail(h, MCM_EHIP, "a persistent kernel of an earlier call gave up a wait that exceeded its poll budget (attn_ps_kernel): "
                             "that call's results are invalid and the handle is unusable — mcm_kernel_faults");
  return MCM_OK;
}

}  // namespace

extern "C" {

int mcm_abi_version(void) { return MCM_ABI_VERSION; }

const char* mcm_last_error(const mcm_handle* h) { return h ? h->err.c_str() : g_create_err.c_str(); }

int mcm_create(const mcm_config* cfg, mcm_handle** out) {
  if (!cfg || !out) return fail(nullptr, MCM_EINVAL, "null argument");
  if (cfg->abi_version != MCM_ABI_VERSION) return fail(nullptr, MCM_EINVAL, "ABI version mismatch");
  const mcm_config& c = *cfg;
  if (c.precision != MCM_PREC_BF16 && c.precision != MCM_PREC_F32 && c.precision != MCM_PREC_F16)
    return fail(nullptr, MCM_EINVAL, "unknown precision");
  if (c.weight_operands < MCM_WEIGHTS_AUTO || c.weight_operands > MCM_WEIGHTS_SPLIT)
    return fail(nullptr, MCM_EINVAL, "unknown weight_operands");
  if (c.v_heads <= 0 || c.t_heads <= 0 || c.v_width != c.v_heads * 64 || c.t_width != c.t_heads * 64)
    return fail(nullptr, MCM_EINVAL, "head_dim must be 64");
  if (c.patch_size <= 0 || c.image_size % c.patch_size)
    return fail(nullptr, MCM_EINVAL, "image_size must be a multiple of patch_size");
  if (c.v_width % 64 || c.t_width % 64 || c.v_mlp % 64 || c.t_mlp % 64 || c.v_width > 1024 ||
      c.t_width > 1024 || c.proj_dim % 4 || c.proj_dim > 1024)
    return fail(nullptr, MCM_EINVAL, "widths must be multiples of 64 and <= 1024; proj_dim % 4 == 0");
  if (c.max_batch <= 0 || c.max_prompt_tokens <= 0 || c.max_positions <= 0 ||
      c.max_prompt_tokens < c.max_positions)
    return fail(nullptr, MCM_EINVAL, "bad workspace bounds");
  {
    const int g = c.image_size / c.patch_size;
    if (g * g + 1 > 288) return fail(nullptr, MCM_EINVAL, "more than 288 vision tokens");
  }
  hipError_t e = hipSetDevice(c.device);
  if (e != hipSuccess) return fail(nullptr, MCM_EHIP, std::string("hipSetDevice: ") + hipGetErrorString(e));

  mcm_handle* h = new mcm_handle();
  h->cfg = c;
  const int g = c.image_size / c.patch_size;
  h->np = g * g;
  h->ntok = h->np + 1;
  const int es = prec_esize(c.precision);
  const int kreal = 3 * c.patch_size * c.patch_size;
  const int kalign = 128 / es;
  h->kpad = (kreal + kalign - 1) / kalign * kalign;
  // The text tower always runs in exact fp32 (MCM_PREC_F32 kernels), whatever cfg.precision says: the
  // prompt bank is encoded once per dataset, off the hot loop, and a bank rounded to 16-bit operands
  // is a FIXED perturbation of every cosine of every image — it shifts AUROC / FPR95 systematically
  // instead of averaging out.  cfg.precision selects the vision tower's operand mode.
  h->vis = Tower{c.v_width, c.v_heads, c.v_layers, c.v_mlp, c.precision, {}};
  h->txt = Tower{c.t_width, c.t_heads, c.t_layers, c.t_mlp, MCM_PREC_F32, {}};

  // parameter registry (HF state_dict names)
  add_param(h, "vision_model.embeddings.class_embedding", {c.v_width});
  add_param(h, "vision_model.embeddings.patch_embedding.weight", {c.v_width, 3, c.patch_size, c.patch_size});
  add_param(h, "vision_model.embeddings.position_embedding.weight", {h->ntok, c.v_width});
  add_param(h, "vision_model.pre_layrnorm.weight", {c.v_width});
  add_param(h, "vision_model.pre_layrnorm.bias", {c.v_width});
  for (int l = 0; l < c.v_layers; ++l)
    add_layer_params(h, "vision_model.encoder.layers." + std::to_string(l), c.v_width, c.v_mlp);
  add_param(h, "vision_model.post_layernorm.weight", {c.v_width});
  add_param(h, "vision_model.post_layernorm.bias", {c.v_width});
  add_param(h, "visual_projection.weight", {c.proj_dim, c.v_width});
  add_param(h, "text_model.embeddings.token_embedding.weight", {c.vocab_size, c.t_width});
  add_param(h, "text_model.embeddings.position_embedding.weight", {c.max_positions, c.t_width});
  for (int l = 0; l < c.t_layers; ++l)
    add_layer_params(h, "text_model.encoder.layers." + std::to_string(l), c.t_width, c.t_mlp);
  add_param(h, "text_model.final_layer_norm.weight", {c.t_width});
  add_param(h, "text_model.final_layer_norm.bias", {c.t_width});
  add_param(h, "text_projection.weight", {c.proj_dim, c.t_width});

  int rc = MCM_OK;
  for (auto& kv : h->params) {
    if ((rc = dev_alloc(h, (void**)&kv.second.dev, (size_t)kv.second.numel * sizeof(float)))) break;
  }
  // activation workspace
  // activation rows: whole 256-row GEMM tiles (see gemm(): problems are padded up into these rows)
  const int64_t mv = ((int64_t)c.max_batch * h->ntok + 255) / 256 * 256, mt = ((int64_t)c.max_prompt_tokens + 255) / 256 * 256;
  h->max_rows = mv > mt ? mv : mt;
  // shared by both towers: vision rows in the operand dtype of cfg.precision, text rows in fp32
  // fp16 handles: the split-activation arm (mcm_score_x2) carries every such row as hi + lo — twice the width — and runs at the
  // same batch as the fp16 arm, so these buffers are allocated at twice the bytes (B/16, batch 512: +1.5 GB of the part's 288)
  // cfg.x2_max_batch (ABI 4): 0 = the arm runs at the full batch (the buffers at twice the bytes), n > 0 = at most n images per
  // x2 call (the buffers hold max(max_batch rows, 2 x the x2 rows)), < 0 = no split workspace at all
  if (c.precision == MCM_PREC_F16 && c.x2_max_batch >= 0)
    h->x2_batch = c.x2_max_batch == 0 || c.x2_max_batch > c.max_batch ? c.max_batch : c.x2_max_batch;
  const int64_t mvx = ((int64_t)h->x2_batch * h->ntok + 255) / 256 * 256;   // token rows of the largest x2 batch (0: none)
  auto both = [&](int64_t vcols, int64_t tcols) {
    const size_t a = (size_t)mv * vcols * es, ax = (size_t)mvx * vcols * es * 2, b = (size_t)mt * tcols * sizeof(float);
    return std::max(std::max(a, ax), b);
  };
  const int64_t dmax = c.v_width > c.t_width ? c.v_width : c.t_width;
  const size_t xb = (size_t)h->max_rows * dmax * sizeof(float), lnb = both(c.v_width, c.t_width),
               qkvb = both(3 * (int64_t)c.v_width, 3 * (int64_t)c.t_width), attb = both(c.v_width, c.t_width);
  if (!rc) rc = dev_alloc(h, (void**)&h->x, xb);
  if (!rc) rc = dev_alloc(h, &h->ln, lnb);
  if (!rc) rc = dev_alloc(h, &h->qkv, qkvb);
  if (!rc) rc = dev_alloc(h, &h->att, attb);
  h->hbuf_bytes = both(c.v_mlp, c.t_mlp);
  if (!rc) rc = dev_alloc(h, &h->hbuf, h->hbuf_bytes);
  // zeroed once (a ragged vision batch zeroes its pad rows again: encode_image_impl, "Pad rows")
  if (!rc && (hipMemset(h->x, 0, xb) != hipSuccess || hipMemset(h->ln, 0, lnb) != hipSuccess ||
              hipMemset(h->qkv, 0, qkvb) != hipSuccess || hipMemset(h->att, 0, attb) != hipSuccess ||
              hipMemset(h->hbuf, 0, h->hbuf_bytes) != hipSuccess))
    rc = fail(h, MCM_EHIP, "hipMemset of the activation workspace");
  if (!rc) rc = dev_alloc(h, &h->patches, std::max((size_t)c.max_batch * h->np * h->kpad * es, (size_t)h->x2_batch * h->np * h->kpad * es * 2));
  if (!rc) rc = dev_alloc(h, (void**)&h->feat, (size_t)c.max_batch * c.proj_dim * sizeof(float));
  if (!rc) rc = dev_alloc(h, (void**)&h->ids_dev, (size_t)mt * sizeof(int32_t));
  if (!rc) rc = dev_alloc(h, (void**)&h->rowidx_dev, (size_t)mt * sizeof(int32_t));
  if (!rc) rc = dev_alloc(h, &h->wpatch, (size_t)c.v_width * h->kpad * es * 2);  // (x 2: room for the split image)
  if (!rc && hipHostMalloc((void**)&h->ids_pin, (size_t)mt * sizeof(int32_t)) != hipSuccess)
    rc = fail(h, MCM_ENOMEM, "hipHostMalloc ids");
  if (!rc && hipHostMalloc((void**)&h->rowidx_pin, (size_t)mt * sizeof(int32_t)) != hipSuccess)
    rc = fail(h, MCM_ENOMEM, "hipHostMalloc rowidx");
  if (!rc) rc = dev_alloc(h, (void**)&h->prep_dev, (size_t)mcm_handle::PREP_RING * c.max_batch * sizeof(PrepImage));
#if defined(MCM_HARNESS) || defined(MCM_LN_FOLD)  // LayerNorm fold (A/B arm): row moments and row statistics
  if (!rc && c.precision != MCM_PREC_F32 && c.v_width % 256 == 0 && c.v_mlp % 256 == 0) {
    rc = dev_alloc(h, (void**)&h->fold_part, (size_t)(c.v_width / 64) * mv * sizeof(float2));
    if (!rc) rc = dev_alloc(h, (void**)&h->fold_rs, (size_t)mv * sizeof(float2));
  }
#endif
#if defined(MCM_HARNESS) || defined(MCM_LN_TAIL)  // LayerNorm in the tail (A/B arm): its counters, if the device qualifies
  if (!rc && c.precision != MCM_PREC_F32 && (c.v_width == 768 || c.v_width == 1024) && xcd_round_robin(h, gemm_persistent_grid())) {
    h->ln_cap8 = 2 * (int)((mv / 256 + 7) / 8);  // (x 2: the cluster arm counts the upper and lower half of a row panel separately)
    h->ln_rs = (2 * h->ln_cap8 + 4 + 63) / 64 * 64;  // words per XCD region: whole 256-B blocks, no line shared between XCDs
                                                     // ([cap8] counters, 3 words, [cap8] segment masks + 1 word of the LNC defer form)
    const size_t bytes = (size_t)8 * h->ln_rs * sizeof(unsigned int);
    rc = dev_alloc(h, (void**)&h->ln_state, bytes);
    if (!rc && hipMemset(h->ln_state, 0, bytes) != hipSuccess) rc = fail(h, MCM_EHIP, "hipMemset LayerNorm-tail state");
  }
#endif
  if (!rc) rc = dev_alloc(h, (void**)&h->sat_dev, 16);
  if (!rc && hipMemset(h->sat_dev, 0, 16) != hipSuccess) rc = fail(h, MCM_EHIP, "hipMemset saturation counter");
  if (!rc && hipHostMalloc((void**)&h->fault_pin, 64, hipHostMallocMapped) != hipSuccess) rc = fail(h, MCM_ENOMEM, "hipHostMalloc fault word");
  if (!rc) {
    memset(h->fault_pin, 0, 64);
    if (hipHostGetDevicePointer((void**)&h->fault_dev, h->fault_pin, 0) != hipSuccess) rc = fail(h, MCM_EHIP, "hipHostGetDevicePointer fault word");
  }
  if (!rc && hipHostMalloc((void**)&h->prep_pin, (size_t)mcm_handle::PREP_RING * c.max_batch * sizeof(PrepImage)) != hipSuccess)
    rc = fail(h, MCM_ENOMEM, "hipHostMalloc prep");
  for (int k = 0; !rc && k < mcm_handle::PREP_RING; ++k)
    if (hipEventCreateWithFlags(&h->prep_ev[k], hipEventDisableTiming) != hipSuccess) rc = fail(h, MCM_EHIP, "hipEventCreate prep");
  if (rc) {
    g_create_err = h->err;
    mcm_destroy(h);
    return rc;
  }
  *out = h;
  return MCM_OK;
}

void mcm_destroy(mcm_handle* h) {
  if (!h) return;
  (void)hipDeviceSynchronize();
  for (void* p : h->owned) (void)hipFree(p);
  if (h->ids_pin) (void)hipHostFree(h->ids_pin);
  if (h->rowidx_pin) (void)hipHostFree(h->rowidx_pin);
  if (h->prep_pin) (void)hipHostFree(h->prep_pin);
  if (h->fault_pin) (void)hipHostFree(h->fault_pin);
  if (h->jpg_pin) (void)hipHostFree(h->jpg_pin);
  if (h->jpg_planes) (void)hipFree(h->jpg_planes);
  for (auto& e : h->jpg_ev)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : h->prep_ev)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : h->ev_pool) {
    (void)hipEventDestroy(e.a);
    (void)hipEventDestroy(e.b);
  }
  delete h;
}

namespace {
float half_bits_to_float(uint16_t v) {  // IEEE binary16 -> binary32, exact (host side of mcm_set_weight)
  const uint32_t sign = (uint32_t)(v & 0x8000u) << 16, e = (v >> 10) & 0x1fu, m = v & 0x3ffu;
  uint32_t u;
  if (e == 0) {
    if (m == 0) {
      u = sign;
    } else {  // subnormal: m * 2^-24
      const float f = (float)m * (1.0f / 16777216.0f);
      uint32_t fu;
      memcpy(&fu, &f, 4);
      u = sign | fu;
    }
  } else if (e == 31) {
    u = sign | 0x7f800000u | (m << 13);
  } else {
    u = sign | ((e + 112u) << 23) | (m << 13);
  }
  float out;
  memcpy(&out, &u, 4);
  return out;
}
}  // namespace

int mcm_set_weight(mcm_handle* h, const char* hf_name, const void* host_ptr, int32_t dtype, const int64_t* shape,
                   int32_t ndim) {
  if (!h || !hf_name || !host_ptr || (ndim > 0 && !shape)) return fail(h, MCM_EINVAL, "null argument");
  if (dtype != MCM_DT_F32 && dtype != MCM_DT_F16 && dtype != MCM_DT_BF16) return fail(h, MCM_EINVAL, "unknown dtype");
  auto it = h->params.find(hf_name);
  if (it == h->params.end()) return fail(h, MCM_ENAME, std::string("unknown parameter ") + hf_name);
  Param& p = it->second;
  bool ok = (size_t)ndim == p.shape.size();
  for (int i = 0; ok && i < ndim; ++i) ok = shape[i] == p.shape[i];
  if (!ok) return fail(h, MCM_ESHAPE, std::string("shape mismatch for ") + hf_name);
  const void* src = host_ptr;
  std::vector<float> wide;
  if (dtype != MCM_DT_F32) {  // the fp32 master is the exact widening of a 16-bit checkpoint value
    wide.resize((size_t)p.numel);
    const uint16_t* q = (const uint16_t*)host_ptr;
    for (int64_t i = 0; i < p.numel; ++i) {
      if (dtype == MCM_DT_F16) {
        wide[(size_t)i] = half_bits_to_float(q[i]);
      } else {
        const uint32_t u = (uint32_t)q[i] << 16;
        memcpy(&wide[(size_t)i], &u, 4);
      }
    }
    src = wide.data();
  }
  HIP_TRY(h, hipMemcpy(p.dev, src, (size_t)p.numel * sizeof(float), hipMemcpyHostToDevice));
  p.set = true;
  h->finalized = false;
  return MCM_OK;
}

int mcm_finalize_weights(mcm_handle* h) {
  if (!h) return MCM_EINVAL;
  for (auto& kv : h->params)
    if (!kv.second.set) return fail(h, MCM_ENOWEIGHT, "parameter never set: " + kv.first);
  if (!h->vis.L.empty()) return fail(h, MCM_EINVAL, "weights already finalized; create a new handle to reload");
  const mcm_config& c = h->cfg;
  // Which vision GEMM weights are NOT numbers of the operand dtype?  (include/mcm.h MCM_WEIGHTS_*: a weight that is one
  // is lossless as a single 16-bit operand; one that is not is rounded - unless the split form carries the remainder.)
  h->w_inexact = 0;
  if (c.precision != MCM_PREC_F32) {
    unsigned long long* cnt = nullptr;
    HIP_TRY(h, hipMalloc((void**)&cnt, sizeof(*cnt)));
    hipError_t e = hipMemset(cnt, 0, sizeof(*cnt));
    for (auto& kv : h->params) {
      const std::string& n = kv.first;
      const bool gemm_w = n.rfind("vision_model.", 0) == 0 && kv.second.shape.size() >= 2 &&
                          (n.find("_proj.weight") != std::string::npos || n.find(".mlp.fc") != std::string::npos ||
                           n.find("patch_embedding.weight") != std::string::npos);
      if (gemm_w && e == hipSuccess)
        e = launch_count_inexact(c.precision, kv.second.dev, (size_t)kv.second.numel, cnt, nullptr);
    }
    unsigned long long v = 0;
    if (e == hipSuccess) e = hipMemcpy(&v, cnt, sizeof(v), hipMemcpyDeviceToHost);
    (void)hipFree(cnt);
    if (e != hipSuccess) return fail(h, MCM_EHIP, std::string("counting inexact weights: ") + hipGetErrorString(e));
    h->w_inexact = v;
  }
  h->vis.split = c.precision != MCM_PREC_F32 &&
                 (c.weight_operands == MCM_WEIGHTS_SPLIT || (c.weight_operands == MCM_WEIGHTS_AUTO && h->w_inexact > 0));
  // operand copies are built from the fp32 masters once (pointers are stable afterwards)
  int rc;
  if ((rc = build_tower(h, h->vis, "vision_model", nullptr))) return rc;
  if ((rc = build_tower(h, h->txt, "text_model", nullptr))) return rc;
  const float* wp = W(h, "vision_model.embeddings.patch_embedding.weight");
  const int kreal = 3 * c.patch_size * c.patch_size;
  if (h->vis.split) HIP_TRY(h, launch_cvt_weight_split(c.precision, wp, h->wpatch, c.v_width, kreal, h->kpad, nullptr));
  else HIP_TRY(h, launch_cvt_weight(c.precision, wp, h->wpatch, c.v_width, kreal, h->kpad, nullptr));
  HIP_TRY(h, hipDeviceSynchronize());
  h->finalized = true;
  return MCM_OK;
}

int mcm_weights_operand_exact(mcm_handle* h, uint64_t* inexact_host, int32_t* split_host) {
  int rc = check_ready(h);
  if (rc) return rc;
  if (inexact_host) *inexact_host = h->w_inexact;
  if (split_host) *split_host = h->vis.split ? 1 : 0;
  return MCM_OK;
}

}  // extern "C" (re-opened below)

namespace {
// preprocess constants of the reference (utils/train_eval_util.py:27-28)
const float kClipMean[3] = {0.48145466f, 0.4578275f, 0.40821073f};
const float kClipStd[3] = {0.26862954f, 0.26130258f, 0.27577711f};

// x2: the split-activation arm (include/mcm.h mcm_score_x2) — same weights, same workspace, B <= h->x2_batch
int encode_image_impl(mcm_handle* h, const void* pixels_dev, bool u8, int32_t B, float* out_dev,
                      void* stream, bool normalize = true, bool x2 = false) {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!pixels_dev || !out_dev) return fail(h, MCM_EINVAL, "null pointer");
  if (x2 && h->x2_batch <= 0) return fail(h, MCM_EINVAL, "the split-activation arm needs an fp16 handle created with a split-activation workspace (cfg.x2_max_batch >= 0)");
  if (B <= 0 || B > (x2 ? h->x2_batch : h->cfg.max_batch))
    return fail(h, MCM_ERANGE, x2 ? "batch exceeds mcm_x2_max_batch" : "batch exceeds cfg.max_batch");
  hipStream_t s = (hipStream_t)stream;
  const mcm_config& c = h->cfg;
  const int D = c.v_width;

  // fp32 NCHW pixels: the patch GEMM gathers its A operand from the image itself (gemm_p256_kernel, GemmArgs::px) when the
  // geometry allows (B/16, B/32 at batches the persistent kernel takes); otherwise — uint8 ingest, L/14's padded K, small
  // batches — patchify writes the patch matrix first
  const bool from_px = !u8 && !x2 && g_patch_fold &&
                       gemm_patch_takes_pixels(c.precision, B * h->np, D, h->kpad, c.patch_size, c.image_size);
  if (!from_px) {
    Scope sc(h, s, MCM_KC_PATCHIFY, 0.0);
    if (u8)
      HIP_TRY(h, launch_patchify_u8(c.precision, (const uint8_t*)pixels_dev, h->patches, B, c.image_size,
                                    c.patch_size, h->kpad, kClipMean, kClipStd, s, x2));
    else
      HIP_TRY(h, launch_patchify(c.precision, (const float*)pixels_dev, h->patches, B, c.image_size,
                                 c.patch_size, h->kpad, s, x2));
  }
  GemmArgs a{};
  a.px = from_px ? (const float*)pixels_dev : nullptr; a.img = c.image_size; a.patch = c.patch_size;
  a.x = h->patches; a.w = h->wpatch; a.bias = nullptr; a.out = h->x;
  a.pos = W(h, "vision_model.embeddings.position_embedding.weight");
  a.M = B * h->np; a.N = D; a.K = h->kpad; a.ldx = h->kpad; a.ldo = D; a.np = h->np; a.ksplit = h->vis.split ? 1 : 0;
  a.xsplit = x2 ? 1 : 0;
  HIP_TRY(h, gemm(h, s, c.precision, EPI_PATCH, a));
  {  // Pad rows.  gemm() runs the dense activation GEMMs on whole 256-row tiles; the rows between B * ntok and the next
     // multiple of 256 are computed and never read.  The buffers they live in are shared with the fp32 text tower, so
     // what they hold is arbitrary (fp32 bit patterns read as fp16 decode to inf / NaN, which would trip the saturation
     // watch): a ragged batch zeroes them first, and they then only ever carry bias-only values.
    const int64_t M = (int64_t)B * h->ntok, mp = padded_rows(h, (int)M);
    if (mp > M) {
      const size_t es = (size_t)prec_esize(c.precision) * (x2 ? 2 : 1), pad = (size_t)(mp - M);
      HIP_TRY(h, hipMemsetAsync(h->x + M * D, 0, pad * D * sizeof(float), s));
      HIP_TRY(h, hipMemsetAsync((char*)h->ln + (size_t)M * D * es, 0, pad * D * es, s));
      HIP_TRY(h, hipMemsetAsync((char*)h->att + (size_t)M * D * es, 0, pad * D * es, s));
      HIP_TRY(h, hipMemsetAsync((char*)h->hbuf + (size_t)M * c.v_mlp * es, 0, pad * c.v_mlp * es, s));
    }
  }
  // the CLS row of every image (class_embedding + position_embedding[0], HF :212-217) is produced inside the
  // LayerNorm pass below instead of by a launch of its own
  {  // pre_layrnorm (fp32, in place) and layer 0's layer_norm1 in one pass over x; a 1-layer tower whose only
     // layer is the CLS-only one still works: its layer_norm1 is over all rows either way
    Scope sc(h, s, MCM_KC_LAYERNORM, 16.0 * B * h->ntok * D);
    HIP_TRY(h, launch_layernorm_pre(c.precision, h->x, W(h, "vision_model.pre_layrnorm.weight"),
                                    W(h, "vision_model.pre_layrnorm.bias"), h->vis.L[0].ln1w, h->vis.L[0].ln1b,
                                    h->ln, B * h->ntok, D, c.ln_eps, s, next_dir(h),
                                    h->sat_on ? h->sat_dev : nullptr,
                                    W(h, "vision_model.embeddings.class_embedding"), a.pos, h->ntok, x2));
  }
  if ((rc = run_layers(h, s, h->vis, B, h->ntok, false, true, true, true, x2))) return rc;
  {
    Scope sc(h, s, MCM_KC_POOL_PROJECT, 2.0 * B * D * c.proj_dim);
    HIP_TRY(h, launch_pool_project(h->x, nullptr, h->ntok, B, D,
                                   W(h, "vision_model.post_layernorm.weight"),
                                   W(h, "vision_model.post_layernorm.bias"), c.ln_eps,
                                   W(h, "visual_projection.weight"), c.proj_dim, out_dev, s, normalize));
  }
  return MCM_OK;
}
}  // namespace

extern "C" {

int mcm_encode_image(mcm_handle* h, const float* pixels_dev, int32_t B, float* out_dev, void* stream) {
  return encode_image_impl(h, pixels_dev, false, B, out_dev, stream);
}

int mcm_encode_image_raw(mcm_handle* h, const float* pixels_dev, int32_t B, float* out_dev, void* stream) {
  return encode_image_impl(h, pixels_dev, false, B, out_dev, stream, false);
}

int mcm_encode_image_ex(mcm_handle* h, const void* pixels_dev, int32_t pixel_format, int32_t B,
                        int32_t normalize, float* out_dev, void* stream) {
  if (pixel_format != MCM_PIXELS_F32_NCHW && pixel_format != MCM_PIXELS_U8_NHWC)
    return fail(h, MCM_EINVAL, "unknown pixel_format");
  return encode_image_impl(h, pixels_dev, pixel_format == MCM_PIXELS_U8_NHWC, B, out_dev, stream,
                           normalize != 0);
}

int mcm_x2_max_batch(const mcm_handle* h) { return h ? h->x2_batch : 0; }

int mcm_kernel_faults(const mcm_handle* h) { return h && h->fault_pin ? (int)*(volatile unsigned int*)h->fault_pin : 0; }

int mcm_encode_image_x2(mcm_handle* h, const void* pixels_dev, int32_t pixel_format, int32_t B, int32_t normalize,
                        float* out_dev, void* stream) {
  if (pixel_format != MCM_PIXELS_F32_NCHW && pixel_format != MCM_PIXELS_U8_NHWC)
    return fail(h, MCM_EINVAL, "unknown pixel_format");
  return encode_image_impl(h, pixels_dev, pixel_format == MCM_PIXELS_U8_NHWC, B, out_dev, stream, normalize != 0, true);
}

int mcm_score_x2(mcm_handle* h, const void* pixels_dev, int32_t pixel_format, int32_t B, const float* text_feat_dev,
                 int32_t K, float T, int32_t kind, float* scores_dev, void* stream) {
  int rc = mcm_encode_image_x2(h, pixels_dev, pixel_format, B, 1, h ? h->feat : nullptr, stream);
  if (rc) return rc;
  return mcm_score_features(h, h->feat, B, text_feat_dev, K, T, kind, scores_dev, stream);
}

int mcm_maha_prepare(mcm_handle* h, const float* means_dev, const float* prec_dev, int32_t C,
                     double* w_dev, double* k_dev, void* stream) {
  if (!h) return MCM_EINVAL;
  if (!means_dev || !prec_dev || !w_dev || !k_dev || C <= 0) return fail(h, MCM_EINVAL, "bad argument");
  HIP_TRY(h, launch_maha_prepare(means_dev, prec_dev, C, h->cfg.proj_dim, w_dev, k_dev, (hipStream_t)stream));
  return MCM_OK;
}

int mcm_maha_score_features(mcm_handle* h, const float* feats_dev, int32_t B, const float* prec_dev,
                            const double* w_dev, const double* k_dev, int32_t C, float* scores_dev,
                            void* stream) {
  if (!h) return MCM_EINVAL;
  if (!feats_dev || !prec_dev || !w_dev || !k_dev || !scores_dev || B <= 0 || C <= 0)
    return fail(h, MCM_EINVAL, "bad argument");
  HIP_TRY(h, launch_maha_score(feats_dev, B, prec_dev, w_dev, k_dev, C, h->cfg.proj_dim, scores_dev,
                               (hipStream_t)stream));
  return MCM_OK;
}

int mcm_encode_image_u8(mcm_handle* h, const uint8_t* pixels_dev, int32_t B, float* out_dev,
                        void* stream) {
  return encode_image_impl(h, pixels_dev, true, B, out_dev, stream);
}

int mcm_score_u8(mcm_handle* h, const uint8_t* pixels_dev, int32_t B, const float* text_feat_dev,
                 int32_t K, float T, int32_t kind, float* scores_dev, void* stream) {
  int rc = encode_image_impl(h, pixels_dev, true, B, h ? h->feat : nullptr, stream);
  if (rc) return rc;
  return mcm_score_features(h, h->feat, B, text_feat_dev, K, T, kind, scores_dev, stream);
}

namespace {
// torchvision's Resize(int) and CenterCrop arithmetic (see preprocess.hip): resized size and crop
// origin of an H x W image for a square target S
bool prep_geometry(int32_t H, int32_t W, int32_t S, PrepImage& g) {
  const int32_t shrt = W <= H ? W : H, lng = W <= H ? H : W;
  g.H = H;
  g.W = W;
  if (shrt == S) {
    g.nh = H;
    g.nw = W;
  } else {
    const int32_t nl = (int32_t)((double)S * (double)lng / (double)shrt);  // int(size * long / short)
    g.nw = W <= H ? S : nl;
    g.nh = W <= H ? nl : S;
  }
  if (g.nh < S || g.nw < S) return false;
  auto round_half_even = [](double v) {  // Python 3 round() on the x.0 / x.5 values that occur here
    const double f = floor(v), d = v - f;
    if (d < 0.5) return (int32_t)f;
    if (d > 0.5) return (int32_t)f + 1;
    return ((int64_t)f % 2 == 0) ? (int32_t)f : (int32_t)f + 1;
  };
  g.top = round_half_even((g.nh - S) / 2.0);
  g.left = round_half_even((g.nw - S) / 2.0);
  return true;
}
}  // namespace

int mcm_resize_crop_u8(mcm_handle* h, const uint8_t* const* src_dev_ptrs, const int32_t* heights,
                       const int32_t* widths, int32_t B, uint8_t* dst_dev, void* stream) {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!src_dev_ptrs || !heights || !widths || !dst_dev) return fail(h, MCM_EINVAL, "null pointer");
  if (B <= 0) return fail(h, MCM_EINVAL, "B must be positive");
  if (B > h->cfg.max_batch) return fail(h, MCM_ERANGE, "batch exceeds max_batch");
  const int32_t S = h->cfg.image_size;
  hipStream_t s = (hipStream_t)stream;
  const unsigned slot = h->prep_next++ % mcm_handle::PREP_RING;
  PrepImage* pin = h->prep_pin + (size_t)slot * h->cfg.max_batch;
  PrepImage* dev = h->prep_dev + (size_t)slot * h->cfg.max_batch;
  HIP_TRY(h, hipEventSynchronize(h->prep_ev[slot]));  // the copy out of this slot, PREP_RING calls ago (long done: no stall)
  for (int32_t b = 0; b < B; ++b) {
    PrepImage& g = pin[b];
    if (!src_dev_ptrs[b] || heights[b] <= 0 || widths[b] <= 0) return fail(h, MCM_EINVAL, "bad image");
    g.src = src_dev_ptrs[b];
    if (!prep_geometry(heights[b], widths[b], S, g)) return fail(h, MCM_EINVAL, "image smaller than the crop");
    const int taps_x = g.nw != g.W ? 2 * (int)ceil(fmax((double)g.W / g.nw, 1.0)) + 1 : 1;
    const int taps_y = g.nh != g.H ? 2 * (int)ceil(fmax((double)g.H / g.nh, 1.0)) + 1 : 1;
    if (taps_x > prep_max_taps() || taps_y > prep_max_taps())
      return fail(h, MCM_ERANGE, "downscale factor above 31 is not supported");
  }
  HIP_TRY(h, hipMemcpyAsync(dev, pin, (size_t)B * sizeof(PrepImage), hipMemcpyHostToDevice, s));
  HIP_TRY(h, hipEventRecord(h->prep_ev[slot], s));
  if (!h->prep_coef) {  // first call on this handle (one synchronous allocation; text-only / pre-cropped users never pay it)
    rc = dev_alloc(h, (void**)&h->prep_coef, (size_t)mcm_handle::PREP_RING * prep_coef_bytes(h->cfg.max_batch, S));
    if (rc) return rc;
  }
  int32_t* coef = (int32_t*)((char*)h->prep_coef + (size_t)slot * prep_coef_bytes(h->cfg.max_batch, S));
  HIP_TRY(h, launch_resize_crop(dev, coef, B, S, dst_dev, s, g_resize_fused_only != 0));
  return MCM_OK;
}

int mcm_jpeg_reconstruct(mcm_handle* h, const void* coef_dev, const mcm_jpeg_image* meta, const uint16_t* quant, int32_t n,
                         uint8_t* rgb_dev, const int64_t* rgb_offsets, void* stream) {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!coef_dev || !meta || !quant || !rgb_dev || !rgb_offsets) return fail(h, MCM_EINVAL, "null pointer");
  if (n <= 0) return fail(h, MCM_EINVAL, "n must be positive");
  if (n > h->cfg.max_batch) return fail(h, MCM_ERANGE, "batch exceeds max_batch");
  hipStream_t s = (hipStream_t)stream;
  // a slot of the staging ring: max_batch records, then (16-byte aligned: the kernel reads them as 16-byte rows) the tables
  const size_t qoff = ((size_t)h->cfg.max_batch * sizeof(JpegImageDev) + 15) / 16 * 16;
  const size_t slot_bytes = (qoff + (size_t)h->cfg.max_batch * 3 * 64 * sizeof(uint16_t) + 255) / 256 * 256;
  if ((uintptr_t)coef_dev & 15) return fail(h, MCM_EINVAL, "coef_dev must be 16-byte aligned");
  if (!h->jpg_ready) {  // first call on this handle; `jpg_ready` only once EVERY piece exists (a failed half is retried)
    if (!h->jpg_pin && hipHostMalloc((void**)&h->jpg_pin, mcm_handle::PREP_RING * slot_bytes) != hipSuccess) {
      h->jpg_pin = nullptr;
      return fail(h, MCM_ENOMEM, "hipHostMalloc jpeg");
    }
    if (!h->jpg_dev && (rc = dev_alloc(h, (void**)&h->jpg_dev, mcm_handle::PREP_RING * slot_bytes))) return rc;
    for (int k = 0; k < mcm_handle::PREP_RING; ++k)
      if (!h->jpg_ev[k] && hipEventCreateWithFlags(&h->jpg_ev[k], hipEventDisableTiming) != hipSuccess) {
        h->jpg_ev[k] = nullptr;
        return fail(h, MCM_EHIP, "hipEventCreate jpeg");
      }
    h->jpg_ready = true;
  }
  const unsigned slot = h->jpg_next++ % mcm_handle::PREP_RING;
  char* pin = h->jpg_pin + slot * slot_bytes;
  char* dev = h->jpg_dev + slot * slot_bytes;
  HIP_TRY(h, hipEventSynchronize(h->jpg_ev[slot]));
  JpegImageDev* md = (JpegImageDev*)pin;
  uint16_t* qd = (uint16_t*)(pin + qoff);
  size_t planes = 0;
  int m = 0, max_blocks = 0, max_pixels = 0;
  for (int32_t i = 0; i < n; ++i) {
    const mcm_jpeg_image& im = meta[i];
    if (im.status != 0) continue;  // not taken by the entropy decoder: the caller fills this image's pixels itself
    if (im.width <= 0 || im.height <= 0 || (im.ncomp != 1 && im.ncomp != 3) || rgb_offsets[i] < 0) return fail(h, MCM_EINVAL, "bad jpeg record");
    JpegImageDev& d = md[m];
    d.width = im.width; d.height = im.height; d.ncomp = im.ncomp;
    d.H = im.ncomp == 3 ? im.hs[0] : 1;
    d.V = im.ncomp == 3 ? im.vs[0] : 1;
    if (im.ncomp == 3 && !((d.H == 1 && d.V == 1) || (d.H == 2 && d.V == 1) || (d.H == 2 && d.V == 2))) return fail(h, MCM_EINVAL, "sampling not taken");
    int blocks = 0;
    for (int c = 0; c < 3; ++c) {
      d.wb[c] = c < im.ncomp ? im.wb[c] : 0;
      d.hb[c] = c < im.ncomp ? im.hb[c] : 0;
      d.coef_off[c] = c < im.ncomp ? im.coef_off[c] : 0;
      d.plane_off[c] = (int64_t)planes;
      if (c < im.ncomp) {
        if (im.wb[c] <= 0 || im.hb[c] <= 0 || im.coef_off[c] < 0 || (im.coef_off[c] & 15)) return fail(h, MCM_EINVAL, "bad jpeg plane");
        planes += (size_t)im.wb[c] * im.hb[c] * 64;
        blocks += im.wb[c] * im.hb[c];
      }
    }
    d.rgb_off = rgb_offsets[i];
    memcpy(qd + (size_t)m * 192, quant + (size_t)i * 192, 192 * sizeof(uint16_t));
    max_blocks = blocks > max_blocks ? blocks : max_blocks;
    max_pixels = im.width * im.height > max_pixels ? im.width * im.height : max_pixels;
    ++m;
  }
  if (m == 0) return MCM_OK;
  if (planes > h->jpg_planes_bytes) {  // grow (synchronous, rare: the first batches of a run)
    HIP_TRY(h, hipStreamSynchronize(s));
    if (h->jpg_planes) (void)hipFree(h->jpg_planes);
    h->jpg_planes = nullptr;
    h->jpg_planes_bytes = 0;
    const size_t want = planes + planes / 4 + (1 << 20);
    if (hipMalloc((void**)&h->jpg_planes, want) != hipSuccess) return fail(h, MCM_ENOMEM, "hipMalloc jpeg planes");
    h->jpg_planes_bytes = want;
  }
  HIP_TRY(h, hipMemcpyAsync(dev, pin, (size_t)m * sizeof(JpegImageDev), hipMemcpyHostToDevice, s));
  HIP_TRY(h, hipMemcpyAsync(dev + qoff, pin + qoff, (size_t)m * 192 * sizeof(uint16_t), hipMemcpyHostToDevice, s));
  HIP_TRY(h, hipEventRecord(h->jpg_ev[slot], s));
  HIP_TRY(h, launch_jpeg_reconstruct((const JpegImageDev*)dev, (const uint16_t*)(dev + qoff), coef_dev, h->jpg_planes, rgb_dev, m,
                                     max_blocks, max_pixels, s));
  return MCM_OK;
}

int mcm_reduce_bank(mcm_handle* h, const float* feats_dev, int32_t K, int32_t T, float* bank_dev,
                    void* stream) {
  if (!h) return MCM_EINVAL;
  if (!feats_dev || !bank_dev || K <= 0 || T <= 0) return fail(h, MCM_EINVAL, "bad argument");
  HIP_TRY(h, launch_bank_reduce(feats_dev, K, T, h->cfg.proj_dim, bank_dev, (hipStream_t)stream));
  return MCM_OK;
}

int mcm_measures(mcm_handle* h, const float* pos_dev, int64_t n_pos, const float* neg_dev,
                 int64_t n_neg, int32_t negate, double recall_level, double* out_host, void* stream) {
  if (!h) return MCM_EINVAL;
  if (!pos_dev || !neg_dev || !out_host) return fail(h, MCM_EINVAL, "null pointer");
  if (n_pos <= 0 || n_neg <= 0 || n_pos + n_neg > 0x7fffffffLL)
    return fail(h, MCM_EINVAL, "both score vectors must be non-empty");
  if (!(recall_level >= 0.0 && recall_level <= 1.0)) return fail(h, MCM_EINVAL, "recall_level outside [0,1]");
  hipStream_t s = (hipStream_t)stream;
  // the three per-example count arrays (12 B per score) live in the MLP activation buffer, which is
  // idle between encode calls and ordered against them by the stream (the caller passes the stream its
  // encode / score calls ran on — include/mcm.h); score vectors that do not fit it get a one-off
  // stream-ordered allocation instead of a refusal
  const size_t need = measures_workspace_bytes((long)(n_pos + n_neg));
  void* ws = h->hbuf;
  bool own = false;
  if (need > h->hbuf_bytes) {
    HIP_TRY(h, hipMallocAsync(&ws, need, s));
    own = true;
  }
  double* out_dev = nullptr;
  hipError_t e = launch_measures(pos_dev, (long)n_pos, neg_dev, (long)n_neg, negate, recall_level, ws, &out_dev, s);
  if (e == hipSuccess) e = hipMemcpyAsync(out_host, out_dev, 3 * sizeof(double), hipMemcpyDeviceToHost, s);
  if (own) (void)hipFreeAsync(ws, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) return fail(h, MCM_EHIP, std::string("mcm_measures: ") + hipGetErrorString(e));
  return MCM_OK;
}

int mcm_score_histogram(mcm_handle* h, const float* scores_dev, int64_t n, const float* edges_dev,
                        int32_t n_bins, int64_t* counts_dev, void* stream) {
  if (!h) return MCM_EINVAL;
  if (!scores_dev || !edges_dev || !counts_dev || n < 0 || n_bins <= 0)
    return fail(h, MCM_EINVAL, "bad argument");
  if (n_bins > 8192) return fail(h, MCM_ERANGE, "at most 8192 bins");
  HIP_TRY(h, launch_histogram(scores_dev, (long)n, edges_dev, n_bins, (unsigned long long*)counts_dev,
                              (hipStream_t)stream));
  return MCM_OK;
}

int mcm_encode_text(mcm_handle* h, const int32_t* ids_host, int32_t K, int32_t S, float* out_dev,
                    void* stream) {
  return mcm_encode_text_ex(h, ids_host, K, S, 1, out_dev, stream);
}

int mcm_encode_text_ex(mcm_handle* h, const int32_t* ids_host, int32_t K, int32_t S, int32_t normalize,
                       float* out_dev, void* stream) {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!ids_host || !out_dev) return fail(h, MCM_EINVAL, "null pointer");
  const mcm_config& c = h->cfg;
  if (K <= 0 || S <= 0) return fail(h, MCM_EINVAL, "K and S must be positive");
  if (S > c.max_positions)  // HF modeling_clip.py:241-245 raises ValueError
    return fail(h, MCM_ERANGE, "sequence length exceeds max_position_embeddings");
  for (int64_t i = 0; i < (int64_t)K * S; ++i)
    if (ids_host[i] < 0 || ids_host[i] >= c.vocab_size) return fail(h, MCM_EINVAL, "token id out of range");
  hipStream_t s = (hipStream_t)stream;
  const int D = c.t_width;
  const int chunk = c.max_prompt_tokens / S;  // prompts per pass
  for (int k0 = 0; k0 < K; k0 += chunk) {
    const int kc = (K - k0) < chunk ? (K - k0) : chunk;
    HIP_TRY(h, hipStreamSynchronize(s));  // pinned staging buffers are reused
    for (int k = 0; k < kc; ++k) {
      const int32_t* row = ids_host + (size_t)(k0 + k) * S;
      int best = 0;  // first EOS = argmax id (HF modeling_clip.py:561-581)
      for (int j = 0; j < S; ++j) {
        h->ids_pin[(size_t)k * S + j] = row[j];
        if (row[j] > row[best]) best = j;
      }
      h->rowidx_pin[k] = k * S + best;
    }
    HIP_TRY(h, hipMemcpyAsync(h->ids_dev, h->ids_pin, (size_t)kc * S * sizeof(int32_t),
                              hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(h->rowidx_dev, h->rowidx_pin, (size_t)kc * sizeof(int32_t),
                              hipMemcpyHostToDevice, s));
    {
      Scope sc(h, s, MCM_KC_EMBED, 0.0);
      HIP_TRY(h, launch_text_embed(h->ids_dev, W(h, "text_model.embeddings.token_embedding.weight"),
                                   W(h, "text_model.embeddings.position_embedding.weight"), h->x,
                                   kc, S, D, s));
    }
    if ((rc = run_layers(h, s, h->txt, kc, S, true, false))) return rc;
    {
      Scope sc(h, s, MCM_KC_POOL_PROJECT, 2.0 * kc * D * c.proj_dim);
      HIP_TRY(h, launch_pool_project(h->x, h->rowidx_dev, 0, kc, D,
                                     W(h, "text_model.final_layer_norm.weight"),
                                     W(h, "text_model.final_layer_norm.bias"), c.ln_eps,
                                     W(h, "text_projection.weight"), c.proj_dim,
                                     out_dev + (size_t)k0 * c.proj_dim, s, normalize != 0));
    }
  }
  return MCM_OK;
}

int mcm_score_features(mcm_handle* h, const float* img_feat_dev, int32_t B, const float* text_feat_dev,
                       int32_t K, float T, int32_t kind, float* scores_dev, void* stream) {
  if (!h) return MCM_EINVAL;
  if (!img_feat_dev || !text_feat_dev || !scores_dev) return fail(h, MCM_EINVAL, "null pointer");
  if (B <= 0 || K <= 0 || kind < 0 || kind > MCM_SCORE_VAR || !(T > 0.f))
    return fail(h, MCM_EINVAL, "bad B / K / kind / T");
  hipStream_t s = (hipStream_t)stream;
  Scope sc(h, s, MCM_KC_SCORE, 2.0 * B * (double)K * h->cfg.proj_dim);
  HIP_TRY(h, launch_score(img_feat_dev, B, text_feat_dev, K, h->cfg.proj_dim, T, kind, scores_dev, s));
  return MCM_OK;
}

int mcm_score(mcm_handle* h, const float* pixels_dev, int32_t B, const float* text_feat_dev, int32_t K,
              float T, int32_t kind, float* scores_dev, void* stream) {
  int rc = mcm_encode_image(h, pixels_dev, B, h ? h->feat : nullptr, stream);
  if (rc) return rc;
  return mcm_score_features(h, h->feat, B, text_feat_dev, K, T, kind, scores_dev, stream);
}

int mcm_saturation_check(mcm_handle* h, int32_t on) {
  if (!h) return MCM_EINVAL;
  h->sat_on = on != 0;
  return MCM_OK;
}

int mcm_saturation_count(mcm_handle* h, int32_t reset, uint64_t* count_host, void* stream) {
  if (!h || !count_host) return fail(h, MCM_EINVAL, "null argument");
  hipStream_t s = (hipStream_t)stream;
  unsigned int v = 0;
  HIP_TRY(h, hipMemcpyAsync(&v, h->sat_dev, sizeof(v), hipMemcpyDeviceToHost, s));
  if (reset) HIP_TRY(h, hipMemsetAsync(h->sat_dev, 0, sizeof(v), s));
  HIP_TRY(h, hipStreamSynchronize(s));
  *count_host = v;
  return MCM_OK;
}

int mcm_profile_enable(mcm_handle* h, int32_t on) {
  if (!h) return MCM_EINVAL;
  h->prof = on != 0;
  return MCM_OK;
}

int mcm_profile_read(mcm_handle* h, double* ms_out, int64_t* launches_out, double* flops_out) {
  if (!h) return MCM_EINVAL;
  HIP_TRY(h, hipDeviceSynchronize());
  for (size_t i = 0; i < h->ev_used; ++i) {
    float ms = 0.f;
    HIP_TRY(h, hipEventElapsedTime(&ms, h->ev_pool[i].a, h->ev_pool[i].b));
    h->ms_acc[h->ev_pool[i].kc] += ms;
    if (h->ev_pool[i].kc2 >= 0) h->ms_acc[h->ev_pool[i].kc2] += ms;
  }
  h->ev_used = 0;
  for (int k = 0; k < MCM_KC_COUNT; ++k) {
    if (ms_out) ms_out[k] = h->ms_acc[k];
    if (launches_out) launches_out[k] = h->launches[k];
    if (flops_out) flops_out[k] = h->flops[k];
    h->ms_acc[k] = 0;
    h->launches[k] = 0;
    h->flops[k] = 0;
  }
  return MCM_OK;
}

// ---- operator-level entry points ------------------------------------------------------

int mcm_op_linear(mcm_handle* h, int32_t prec, const void* x_dev, const void* w_dev, const float* bias_dev,
                  void* y_dev, float* resid_dev, int32_t M, int32_t N, int32_t K, int32_t epi,
                  void* stream) {
  return mcm_op_linear_ex(h, prec, x_dev, w_dev, bias_dev, y_dev, resid_dev, M, N, K, epi, 0, stream);
}

int mcm_op_linear_ex(mcm_handle* h, int32_t prec, const void* x_dev, const void* w_dev, const float* bias_dev,
                     void* y_dev, float* resid_dev, int32_t M, int32_t N, int32_t K, int32_t epi, int32_t flags,
                     void* stream) {
  if (!h) return MCM_EINVAL;
  if (epi < EPI_STORE || epi > EPI_RESID) return fail(h, MCM_EINVAL, "bad epilogue");
  const bool split = (flags & MCM_LINEAR_SPLIT_W) != 0, xs = (flags & MCM_LINEAR_SPLIT_X) != 0, os = (flags & MCM_LINEAR_SPLIT_OUT) != 0;
  if ((flags & ~(MCM_LINEAR_SPLIT_W | MCM_LINEAR_SPLIT_X | MCM_LINEAR_SPLIT_OUT)) || (split && (prec == MCM_PREC_F32 || K % 64)))
    return fail(h, MCM_EINVAL, "bad flags (split weights: 16-bit modes, K % 64 == 0)");
  if ((xs || os) && (prec != MCM_PREC_F16 || K % 64 || (os && (epi > EPI_GELU || N % 64))))
    return fail(h, MCM_EINVAL, "bad flags (split activations / outputs: fp16 mode, K % 64 == 0; outputs: epilogues 0 / 1, N % 64 == 0)");
  GemmArgs a{};
  a.x = x_dev; a.w = w_dev; a.bias = bias_dev; a.out = y_dev; a.resid = resid_dev;
  a.M = M; a.N = N; a.K = split ? 2 * K : K; a.ldx = xs ? 2 * K : K; a.ldo = os ? 2 * N : N; a.ksplit = split ? 1 : 0;
  a.xsplit = xs ? 1 : 0;
  if (os) epi = epi == EPI_STORE ? EPI_STORE_X2 : EPI_GELU_X2;
  a.sat = h->sat_on ? h->sat_dev : nullptr;
  HIP_TRY(h, launch_gemm(prec, epi, a, (hipStream_t)stream));
  return MCM_OK;
}

int mcm_op_split_weight(mcm_handle* h, int32_t prec, const float* w_dev, int32_t N, int32_t K, void* out_dev,
                        void* stream) {
  if (!h) return MCM_EINVAL;
  if (!w_dev || !out_dev || N <= 0 || K <= 0 || K % 64 || prec == MCM_PREC_F32)
    return fail(h, MCM_EINVAL, "split weights: 16-bit modes, K % 64 == 0");
  HIP_TRY(h, launch_cvt_weight_split(prec, w_dev, out_dev, N, K, K, (hipStream_t)stream));
  return MCM_OK;
}

int mcm_op_layernorm(mcm_handle* h, int32_t prec, const float* x_dev, const float* gamma_dev,
                     const float* beta_dev, void* y_dev, int32_t M, int32_t D, float eps, int32_t out_f32,
                     void* stream) {
  if (!h) return MCM_EINVAL;
  HIP_TRY(h, launch_layernorm(prec, x_dev, gamma_dev, beta_dev, y_dev, M, D, eps, out_f32 != 0,
                              (hipStream_t)stream, 0, 0, false, h->sat_on ? h->sat_dev : nullptr));
  return MCM_OK;
}

int mcm_op_layernorm_split(mcm_handle* h, const float* x_dev, const float* gamma_dev, const float* beta_dev,
                           void* y_dev, int32_t M, int32_t D, float eps, void* stream) {
  if (!h) return MCM_EINVAL;
  HIP_TRY(h, launch_layernorm(MCM_PREC_F16, x_dev, gamma_dev, beta_dev, y_dev, M, D, eps, false, (hipStream_t)stream, 0, 0,
                              false, h->sat_on ? h->sat_dev : nullptr, true));
  return MCM_OK;
}

int mcm_op_attention_split(mcm_handle* h, const void* qkv_dev, void* out_dev, int32_t nseq, int32_t seq_len,
                           int32_t heads, void* stream) {
  if (!h) return MCM_EINVAL;
  HIP_TRY(h, launch_attention(MCM_PREC_F16, qkv_dev, out_dev, nseq, seq_len, heads, false, 0, (hipStream_t)stream, false, 0, true));
  return MCM_OK;
}

int mcm_op_attention(mcm_handle* h, int32_t prec, const void* qkv_dev, void* out_dev, int32_t nseq,
                     int32_t seq_len, int32_t heads, int32_t causal, void* stream) {
  if (!h) return MCM_EINVAL;
  HIP_TRY(h, launch_attention(prec, qkv_dev, out_dev, nseq, seq_len, heads, causal != 0, 0,
                              (hipStream_t)stream, false, 0, false, h->fault_dev));
  return MCM_OK;
}

#ifdef MCM_HARNESS  // libmcm_hip_harness.so only: process-wide A/B switches for tests and tools
int mcm_debug_attention_variant(int32_t variant) {
  if (variant != 0 && variant != 1 && variant != 10 && variant != 11 && variant != 21 && variant != 36) return MCM_EINVAL;
  attention_set_variant(variant);
  return MCM_OK;
}

// polls a wait of the persistent attention kernel may take before it gives up (attention.hip PS_SPIN_BUDGET = 1 << 22 is what
// the shipped library always passes); 0 = every wait that has to wait gives up at once: the test of the fault path
int mcm_debug_attn_spin_budget(int64_t polls) {
  if (polls < 0 || polls > 0xffffffffLL) return MCM_EINVAL;
  attention_set_spin_budget((unsigned int)polls);
  return MCM_OK;
}

// clears the handle's sticky kernel-fault word (tests of the fault path only: a real fault leaves results invalid)
int mcm_debug_clear_faults(mcm_handle* h) {
  if (!h || !h->fault_pin) return MCM_EINVAL;
  (void)hipDeviceSynchronize();
  *(volatile unsigned int*)h->fault_pin = 0u;
  return MCM_OK;
}

// mcm_op_attention with the two launch parameters only the model sets: the number of query rows (the CLS-only last layer) and
// the walk direction
int mcm_debug_op_attention(mcm_handle* h, int32_t prec, const void* qkv_dev, void* out_dev, int32_t nseq, int32_t seq_len,
                           int32_t heads, int32_t causal, int32_t qrows, int32_t reverse, void* stream) {
  if (!h) return MCM_EINVAL;
  HIP_TRY(h, launch_attention(prec, qkv_dev, out_dev, nseq, seq_len, heads, causal != 0, qrows, (hipStream_t)stream, reverse != 0, 0, false,
                              h->fault_dev));
  return MCM_OK;
}

int mcm_debug_gemm_dbg(int32_t bits) {  // ablation / A-B bits of gemm.hip (GemmArgs::dbg)
  gemm_set_dbg(bits);
  return MCM_OK;
}

int mcm_debug_ln_fold(int32_t on) {  // 0 (shipped behaviour): every LayerNorm as its own launch; 1: folded where the GEMMs qualify
  g_ln_fold = on ? 1 : 0;
  return MCM_OK;
}

int mcm_debug_ln_tail(int32_t on) {  // 1: LayerNorm in the tail of the residual GEMMs; 0 (shipped behaviour): LayerNorm launches
  g_ln_tail = on ? 1 : 0;
  return MCM_OK;
}
int mcm_debug_ln_cluster(int32_t on) {  // 1: LayerNorm by the row panel's cluster of workgroups (gemm_arms.hpp LNC); 0 (shipped behaviour): LayerNorm launches
  g_ln_cluster = on ? 1 : 0;
  return MCM_OK;
}
int mcm_debug_ln_row(int32_t on) {  // 1: out-proj / fc2 + LayerNorm as 64-row full-row tiles (gemm_arms.hpp ROW64); 0 (shipped behaviour): launches
  g_ln_row = (on >= 1 && on <= 4) ? on : 0;   // 2 / 4: three W stages per wave (N = 768); 3 / 4: W in the blocked layout
  return MCM_OK;
}
int mcm_debug_ln_cluster_spin(int32_t polls) {  // < 0: the first LNC form (waits); n >= 0: the defer form with n polls (and the clean-up launch)
  g_lnc_spin = polls < 0 ? -1 : polls;
  return MCM_OK;
}
int mcm_debug_ln_cluster_deferred(mcm_handle* h, uint64_t* count_host) {  // segments the clean-up kernel normalised since mcm_create
  if (!h || !count_host) return MCM_EINVAL;
  *count_host = 0;
  if (!h->ln_state) return MCM_OK;
  HIP_TRY(h, hipDeviceSynchronize());
  for (int x = 0; x < 8; ++x) {
    unsigned int v = 0;
    HIP_TRY(h, hipMemcpy(&v, h->ln_state + (size_t)x * h->ln_rs + 2 * h->ln_cap8 + 3, sizeof(v), hipMemcpyDeviceToHost));
    *count_host += v;
  }
  return MCM_OK;
}
int mcm_debug_ln_tail_timeouts(mcm_handle* h, uint64_t* count_host) {  // tickets that gave up waiting (0 in a correct run)
  if (!h || !count_host) return MCM_EINVAL;
  *count_host = 0;
  if (!h->ln_state) return MCM_OK;
  HIP_TRY(h, hipDeviceSynchronize());
  for (int x = 0; x < 8; ++x) {
    unsigned int v = 0;
    HIP_TRY(h, hipMemcpy(&v, h->ln_state + (size_t)x * h->ln_rs + h->ln_cap8 + 2, sizeof(v), hipMemcpyDeviceToHost));
    *count_host += v;
  }
  return MCM_OK;
}
int mcm_debug_qkv_head_major(int32_t on) {
  g_qkv_head_major = on ? 1 : 0;
  return MCM_OK;
}
int mcm_debug_patch_fold(int32_t on) {  // 1 (shipped): the patch GEMM gathers pixels itself; 0: patchify + plain GEMM
  g_patch_fold = on ? 1 : 0;
  return MCM_OK;
}
int mcm_debug_persistent_grid(int32_t n) {  // 0 (shipped): one workgroup per CU; n: the persistent GEMMs take n CUs (a multiple of 8)
  if (n < 0 || n % 8 != 0) return MCM_EINVAL;
  gemm_set_persistent_grid(n);
  return MCM_OK;
}
int mcm_debug_resize_fused_only(int32_t on) {  // 0 (shipped): LDS form where the window fits; 1: the fused form everywhere
  g_resize_fused_only = on ? 1 : 0;
  return MCM_OK;
}
int mcm_debug_gemm_group_n(int32_t gn) {  // 0 (default): the plain n-fastest walk; g > 0: N tiles walked in groups of g (arms kernel, variant 9)
  if (gn < 0 || gn > 64) return MCM_EINVAL;
  gemm_set_group_n(gn);
  return MCM_OK;
}
int mcm_debug_nsplit(int32_t n) {  // 1 (shipped): one launch per GEMM; 2 / 3 / 4: QKV and fc1 as n column-block launches
  if (n < 1 || n > 4) return MCM_EINVAL;
  g_nsplit = n;
  return MCM_OK;
}
int mcm_debug_qkv_chunks(int32_t n) {
  if (n < 1 || n > 16) return MCM_EINVAL;
  g_qkv_chunks = n;
  return MCM_OK;
}

int mcm_debug_gemm_variant(int32_t variant) {
  if (variant != -1 && variant != 0 && variant != 3 && variant != 4 && variant != 5 && variant != 9 && variant != 11) return MCM_EINVAL;
  gemm_set_variant(variant);
  return MCM_OK;
}

#endif

}  // extern "C"
