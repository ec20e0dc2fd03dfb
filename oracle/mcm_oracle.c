/*
 * mcm_oracle.c — CPU restatement of the reference MCM scoring path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under mcm_amd/ may import, link or call this; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as
 * the checker.  It is pinned against fixtures captured from the reference's own
 * arithmetic (Hugging Face transformers CLIPModel + the reference's get_ood_scores_clip
 * and get_measures, run in the build container): tests/golden/make_golden.py.
 *
 * "HF:" = transformers/models/clip/modeling_clip.py (5.15.0, the un-pinned third-party
 * dependency the reference delegates all model arithmetic to, utils/train_eval_util.py:9,23).
 * "REF:" = /root/reference.
 *
 * Plain C, fp32 storage and fp32 accumulation like the reference's fp32 torch path
 * (row sums for LayerNorm/softmax accumulate in double: the order-independent choice).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/mcm.h"

#define ORC_MAX_PARAMS 1024

typedef struct {
  char name[160];
  float* data;
  int64_t numel;
} orc_param;

typedef struct orc_handle {
  mcm_config cfg;
  orc_param params[ORC_MAX_PARAMS];
  int nparams;
  char err[256];
} orc_handle;

/* ------------------------------------------------------------------ parameter store */

int orc_create(const mcm_config* cfg, orc_handle** out) {
  if (!cfg || !out) return MCM_EINVAL;
  orc_handle* h = (orc_handle*)calloc(1, sizeof(orc_handle));
  if (!h) return MCM_ENOMEM;
  h->cfg = *cfg;
  *out = h;
  return MCM_OK;
}

void orc_destroy(orc_handle* h) {
  if (!h) return;
  for (int i = 0; i < h->nparams; ++i) free(h->params[i].data);
  free(h);
}

const char* orc_last_error(const orc_handle* h) { return h ? h->err : "null handle"; }

int orc_set_weight(orc_handle* h, const char* name, const float* ptr, const int64_t* shape,
                   int32_t ndim) {
  if (!h || !name || !ptr || ndim < 0) return MCM_EINVAL;
  int64_t n = 1;
  for (int i = 0; i < ndim; ++i) n *= shape[i];
  orc_param* p = NULL;
  for (int i = 0; i < h->nparams; ++i)
    if (!strcmp(h->params[i].name, name)) p = &h->params[i];
  if (!p) {
    if (h->nparams == ORC_MAX_PARAMS) return MCM_ENOMEM;
    p = &h->params[h->nparams++];
    snprintf(p->name, sizeof(p->name), "%s", name);
    p->data = NULL;
  }
  free(p->data);
  p->data = (float*)malloc(sizeof(float) * (size_t)n);
  if (!p->data) return MCM_ENOMEM;
  memcpy(p->data, ptr, sizeof(float) * (size_t)n);
  p->numel = n;
  return MCM_OK;
}

static const float* P(orc_handle* h, const char* fmt, const char* tower, int layer) {
  char name[160];
  snprintf(name, sizeof(name), fmt, tower, layer);
  for (int i = 0; i < h->nparams; ++i)
    if (!strcmp(h->params[i].name, name)) return h->params[i].data;
  snprintf(h->err, sizeof(h->err), "missing parameter %s", name);
  return NULL;
}
static const float* P0(orc_handle* h, const char* name) { return P(h, name, "", 0); }

/* ------------------------------------------------------------------ operators */

/* nn.LayerNorm (HF:358,360,605,608; biased variance, eps inside the sqrt). */
void orc_layernorm(const float* x, const float* g, const float* b, float* y, int64_t M,
                   int32_t D, float eps) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < M; ++i) {
    const float* r = x + i * D;
    double s = 0.0;
    for (int d = 0; d < D; ++d) s += r[d];
    const double mean = s / D;
    double v = 0.0;
    for (int d = 0; d < D; ++d) {
      const double c = r[d] - mean;
      v += c * c;
    }
    const float rstd = (float)(1.0 / sqrt(v / D + (double)eps));
    const float mu = (float)mean;
    for (int d = 0; d < D; ++d) y[i * D + d] = (r[d] - mu) * rstd * g[d] + b[d];
  }
}

typedef float v8f __attribute__((vector_size(32), aligned(4)));

static inline float hsum8(v8f v) {
  return ((v[0] + v[4]) + (v[1] + v[5])) + ((v[2] + v[6]) + (v[3] + v[7]));
}

/* nn.Linear: y[M,N] = x[M,K] · w[N,K]^T + bias (HF:309-311,333,347-349; weights are
 * [out,in] row-major).  4x4 register-blocked dot products, 8-wide fp32 lanes. */
void orc_linear(const float* x, const float* w, const float* bias, float* y, int64_t M,
                int32_t N, int32_t K) {
  const int K8 = K & ~7;
#pragma omp parallel for schedule(dynamic, 4)
  for (int64_t i0 = 0; i0 < M; i0 += 4) {
    const int mi = (int)((M - i0) < 4 ? (M - i0) : 4);
    for (int j0 = 0; j0 < N; j0 += 4) {
      const int nj = (N - j0) < 4 ? (N - j0) : 4;
      v8f acc[4][4];
      float tail[4][4];
      for (int a = 0; a < 4; ++a)
        for (int c = 0; c < 4; ++c) {
          acc[a][c] = (v8f){0, 0, 0, 0, 0, 0, 0, 0};
          tail[a][c] = 0.f;
        }
      if (mi == 4 && nj == 4) {
        const float *x0 = x + (i0 + 0) * K, *x1 = x + (i0 + 1) * K, *x2 = x + (i0 + 2) * K,
                    *x3 = x + (i0 + 3) * K;
        const float *w0 = w + (int64_t)(j0 + 0) * K, *w1 = w + (int64_t)(j0 + 1) * K,
                    *w2 = w + (int64_t)(j0 + 2) * K, *w3 = w + (int64_t)(j0 + 3) * K;
        for (int k = 0; k < K8; k += 8) {
          const v8f a0 = *(const v8f*)(x0 + k), a1 = *(const v8f*)(x1 + k),
                    a2 = *(const v8f*)(x2 + k), a3 = *(const v8f*)(x3 + k);
          const v8f b0 = *(const v8f*)(w0 + k), b1 = *(const v8f*)(w1 + k),
                    b2 = *(const v8f*)(w2 + k), b3 = *(const v8f*)(w3 + k);
          acc[0][0] += a0 * b0; acc[0][1] += a0 * b1; acc[0][2] += a0 * b2; acc[0][3] += a0 * b3;
          acc[1][0] += a1 * b0; acc[1][1] += a1 * b1; acc[1][2] += a1 * b2; acc[1][3] += a1 * b3;
          acc[2][0] += a2 * b0; acc[2][1] += a2 * b1; acc[2][2] += a2 * b2; acc[2][3] += a2 * b3;
          acc[3][0] += a3 * b0; acc[3][1] += a3 * b1; acc[3][2] += a3 * b2; acc[3][3] += a3 * b3;
        }
      } else {
        for (int a = 0; a < mi; ++a)
          for (int c = 0; c < nj; ++c) {
            const float* xr = x + (i0 + a) * K;
            const float* wr = w + (int64_t)(j0 + c) * K;
            v8f s = (v8f){0, 0, 0, 0, 0, 0, 0, 0};
            for (int k = 0; k < K8; k += 8) s += *(const v8f*)(xr + k) * *(const v8f*)(wr + k);
            acc[a][c] = s;
          }
      }
      for (int a = 0; a < mi; ++a)
        for (int c = 0; c < nj; ++c) {
          const float* xr = x + (i0 + a) * K;
          const float* wr = w + (int64_t)(j0 + c) * K;
          for (int k = K8; k < K; ++k) tail[a][c] += xr[k] * wr[k];
          y[(i0 + a) * N + j0 + c] = hsum8(acc[a][c]) + tail[a][c] + (bias ? bias[j0 + c] : 0.f);
        }
    }
  }
}

/* quick_gelu: x * sigmoid(1.702 x) (transformers/activations.py:117-123, selected by
 * hidden_act='quick_gelu', HF configuration_clip.py:54,105). */
void orc_quick_gelu(float* x, int64_t n) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) x[i] = x[i] / (1.0f + expf(-1.702f * x[i]));
}

/* Multi-head scaled-dot-product attention (HF:259-277 eager definition; :313-331).
 * qkv: [nseq*L, 3*heads*hd] packed [q|k|v]; out: [nseq*L, heads*hd].  scale = hd^-0.5.
 * causal: position i attends j<=i (text tower, HF:543-556). */
void orc_attention(const float* qkv, float* out, int32_t nseq, int32_t L, int32_t heads,
                   int32_t hd, int32_t causal) {
  const int D = heads * hd;
  const float scale = 1.0f / sqrtf((float)hd);
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
  for (int s = 0; s < nseq; ++s)
    for (int hh = 0; hh < heads; ++hh) {
      float* p = (float*)malloc(sizeof(float) * (size_t)L);
      for (int i = 0; i < L; ++i) {
        const float* q = qkv + ((int64_t)s * L + i) * 3 * D + hh * hd;
        const int jmax = causal ? i + 1 : L;
        float m = -INFINITY;
        for (int j = 0; j < jmax; ++j) {
          const float* k = qkv + ((int64_t)s * L + j) * 3 * D + D + hh * hd;
          float a = 0.f;
          for (int d = 0; d < hd; ++d) a += q[d] * k[d];
          a *= scale;
          p[j] = a;
          if (a > m) m = a;
        }
        double z = 0.0;
        for (int j = 0; j < jmax; ++j) {
          p[j] = expf(p[j] - m);
          z += p[j];
        }
        const float rz = (float)(1.0 / z);
        float* o = out + ((int64_t)s * L + i) * D + hh * hd;
        for (int d = 0; d < hd; ++d) o[d] = 0.f;
        for (int j = 0; j < jmax; ++j) {
          const float* v = qkv + ((int64_t)s * L + j) * 3 * D + 2 * D + hh * hd;
          const float pj = p[j] * rz;
          for (int d = 0; d < hd; ++d) o[d] += pj * v[d];
        }
      }
      free(p);
    }
}

/* x / ||x||_2 per row (REF: utils/detection_util.py:226,231). */
void orc_l2_normalize(float* x, int64_t M, int32_t D) {
  for (int64_t i = 0; i < M; ++i) {
    double s = 0.0;
    for (int d = 0; d < D; ++d) s += (double)x[i * D + d] * x[i * D + d];
    const float r = (float)(1.0 / sqrt(s));
    for (int d = 0; d < D; ++d) x[i * D + d] *= r;
  }
}

/* ------------------------------------------------------------------ encoder layer */

/* CLIPEncoderLayer.forward (HF:362-383): x += attn(LN1(x)); x += mlp(LN2(x)). */
static int encoder_layer(orc_handle* h, const char* tower, int l, float* x, int32_t nseq,
                         int32_t L, int32_t D, int32_t heads, int32_t ff, int32_t causal) {
  const int64_t M = (int64_t)nseq * L;
  const float* ln1w = P(h, "%s.encoder.layers.%d.layer_norm1.weight", tower, l);
  const float* ln1b = P(h, "%s.encoder.layers.%d.layer_norm1.bias", tower, l);
  const float* qw = P(h, "%s.encoder.layers.%d.self_attn.q_proj.weight", tower, l);
  const float* qb = P(h, "%s.encoder.layers.%d.self_attn.q_proj.bias", tower, l);
  const float* kw = P(h, "%s.encoder.layers.%d.self_attn.k_proj.weight", tower, l);
  const float* kb = P(h, "%s.encoder.layers.%d.self_attn.k_proj.bias", tower, l);
  const float* vw = P(h, "%s.encoder.layers.%d.self_attn.v_proj.weight", tower, l);
  const float* vb = P(h, "%s.encoder.layers.%d.self_attn.v_proj.bias", tower, l);
  const float* ow = P(h, "%s.encoder.layers.%d.self_attn.out_proj.weight", tower, l);
  const float* ob = P(h, "%s.encoder.layers.%d.self_attn.out_proj.bias", tower, l);
  const float* ln2w = P(h, "%s.encoder.layers.%d.layer_norm2.weight", tower, l);
  const float* ln2b = P(h, "%s.encoder.layers.%d.layer_norm2.bias", tower, l);
  const float* f1w = P(h, "%s.encoder.layers.%d.mlp.fc1.weight", tower, l);
  const float* f1b = P(h, "%s.encoder.layers.%d.mlp.fc1.bias", tower, l);
  const float* f2w = P(h, "%s.encoder.layers.%d.mlp.fc2.weight", tower, l);
  const float* f2b = P(h, "%s.encoder.layers.%d.mlp.fc2.bias", tower, l);
  if (!ln1w || !ln1b || !qw || !qb || !kw || !kb || !vw || !vb || !ow || !ob || !ln2w ||
      !ln2b || !f1w || !f1b || !f2w || !f2b)
    return MCM_ENOWEIGHT;

  float* t = (float*)malloc(sizeof(float) * (size_t)M * D);
  float* qkv = (float*)malloc(sizeof(float) * (size_t)M * 3 * D);
  float* a = (float*)malloc(sizeof(float) * (size_t)M * D);
  float* u = (float*)malloc(sizeof(float) * (size_t)M * (ff > D ? ff : D));
  if (!t || !qkv || !a || !u) return MCM_ENOMEM;

  orc_layernorm(x, ln1w, ln1b, t, M, D, h->cfg.ln_eps);
  /* separate q/k/v projections (HF:309-311), packed [q|k|v] per row */
  float* tmp = u;
  const float* ws[3] = {qw, kw, vw};
  const float* bs[3] = {qb, kb, vb};
  for (int part = 0; part < 3; ++part) {
    orc_linear(t, ws[part], bs[part], tmp, M, D, D);
    for (int64_t i = 0; i < M; ++i)
      memcpy(qkv + i * 3 * D + part * D, tmp + i * D, sizeof(float) * (size_t)D);
  }
  orc_attention(qkv, a, nseq, L, heads, D / heads, causal);
  orc_linear(a, ow, ob, t, M, D, D);
  for (int64_t i = 0; i < M * D; ++i) x[i] += t[i];

  orc_layernorm(x, ln2w, ln2b, t, M, D, h->cfg.ln_eps);
  orc_linear(t, f1w, f1b, u, M, ff, D);
  orc_quick_gelu(u, M * ff);
  orc_linear(u, f2w, f2b, t, M, D, ff);
  for (int64_t i = 0; i < M * D; ++i) x[i] += t[i];

  free(t); free(qkv); free(a); free(u);
  return MCM_OK;
}

/* ------------------------------------------------------------------ vision tower */

/* CLIPVisionEmbeddings.forward (HF:202-218) + pre_layrnorm (HF:642), then the first
 * `nlayers` encoder layers.  pixels: fp32 NCHW.  hidden: [B, 1+np, D]. */
int orc_vision_hidden(orc_handle* h, const float* pixels, int32_t B, int32_t nlayers,
                      float* hidden) {
  const mcm_config* c = &h->cfg;
  const int S = c->image_size, Pz = c->patch_size, g = S / Pz, np = g * g, D = c->v_width;
  const int Kp = 3 * Pz * Pz, N = np + 1;
  const float* pw = P0(h, "vision_model.embeddings.patch_embedding.weight");
  const float* cls = P0(h, "vision_model.embeddings.class_embedding");
  const float* pos = P0(h, "vision_model.embeddings.position_embedding.weight");
  const float* plw = P0(h, "vision_model.pre_layrnorm.weight");
  const float* plb = P0(h, "vision_model.pre_layrnorm.bias");
  if (!pw || !cls || !pos || !plw || !plb) return MCM_ENOWEIGHT;
  if (nlayers < 0 || nlayers > c->v_layers) return MCM_EINVAL;

  /* Conv2d(3, D, kernel=stride=patch, bias=False) as a GEMM over non-overlapping
   * patches: k = (c, py, px) matches the [D,3,P,P] weight flattening (HF:148-154). */
  float* patches = (float*)malloc(sizeof(float) * (size_t)B * np * Kp);
  float* pe = (float*)malloc(sizeof(float) * (size_t)B * np * D);
  if (!patches || !pe) return MCM_ENOMEM;
  for (int b = 0; b < B; ++b)
    for (int gy = 0; gy < g; ++gy)
      for (int gx = 0; gx < g; ++gx) {
        float* row = patches + ((int64_t)b * np + gy * g + gx) * Kp;
        for (int ch = 0; ch < 3; ++ch)
          for (int py = 0; py < Pz; ++py)
            memcpy(row + (ch * Pz + py) * Pz,
                   pixels + (((int64_t)b * 3 + ch) * S + gy * Pz + py) * S + gx * Pz,
                   sizeof(float) * (size_t)Pz);
      }
  orc_linear(patches, pw, NULL, pe, (int64_t)B * np, D, Kp);
  /* [CLS | patches] + position_embedding (HF:212-217) */
  for (int b = 0; b < B; ++b)
    for (int n = 0; n < N; ++n) {
      float* dst = hidden + ((int64_t)b * N + n) * D;
      const float* src = n == 0 ? cls : pe + ((int64_t)b * np + n - 1) * D;
      for (int d = 0; d < D; ++d) dst[d] = src[d] + pos[(int64_t)n * D + d];
    }
  free(patches); free(pe);
  orc_layernorm(hidden, plw, plb, hidden, (int64_t)B * N, D, c->ln_eps);
  for (int l = 0; l < nlayers; ++l) {
    int rc = encoder_layer(h, "vision_model", l, hidden, B, N, D, c->v_heads, c->v_mlp, 0);
    if (rc) return rc;
  }
  return MCM_OK;
}

/* CLIPModel.get_image_features (HF:719-753): CLS pool → post_layernorm (HF:650-651) →
 * visual_projection (no bias, HF:674,751); optional L2 normalise
 * (REF utils/detection_util.py:226).  out: [B, proj_dim]. */
int orc_encode_image(orc_handle* h, const float* pixels, int32_t B, float* out,
                     int32_t normalize) {
  const mcm_config* c = &h->cfg;
  const int g = c->image_size / c->patch_size, N = g * g + 1, D = c->v_width;
  float* hidden = (float*)malloc(sizeof(float) * (size_t)B * N * D);
  float* pooled = (float*)malloc(sizeof(float) * (size_t)B * D);
  if (!hidden || !pooled) return MCM_ENOMEM;
  int rc = orc_vision_hidden(h, pixels, B, c->v_layers, hidden);
  if (rc) { free(hidden); free(pooled); return rc; }
  const float* lw = P0(h, "vision_model.post_layernorm.weight");
  const float* lb = P0(h, "vision_model.post_layernorm.bias");
  const float* pj = P0(h, "visual_projection.weight");
  if (!lw || !lb || !pj) return MCM_ENOWEIGHT;
  for (int b = 0; b < B; ++b)
    memcpy(pooled + (int64_t)b * D, hidden + (int64_t)b * N * D, sizeof(float) * (size_t)D);
  orc_layernorm(pooled, lw, lb, pooled, B, D, c->ln_eps);
  orc_linear(pooled, pj, NULL, out, B, c->proj_dim, D);
  if (normalize) orc_l2_normalize(out, B, c->proj_dim);
  free(hidden); free(pooled);
  return MCM_OK;
}

/* ------------------------------------------------------------------ text tower */

/* CLIPTextEmbeddings (HF:232-256) + `nlayers` causal encoder layers (HF:543-556).
 * ids: int32 [K,S].  hidden: [K,S,D]. */
int orc_text_hidden(orc_handle* h, const int32_t* ids, int32_t K, int32_t S, int32_t nlayers,
                    float* hidden) {
  const mcm_config* c = &h->cfg;
  const int D = c->t_width;
  const float* te = P0(h, "text_model.embeddings.token_embedding.weight");
  const float* pe = P0(h, "text_model.embeddings.position_embedding.weight");
  if (!te || !pe) return MCM_ENOWEIGHT;
  if (S > c->max_positions || nlayers < 0 || nlayers > c->t_layers) return MCM_EINVAL;
  for (int k = 0; k < K; ++k)
    for (int s = 0; s < S; ++s) {
      const int32_t id = ids[(int64_t)k * S + s];
      if (id < 0 || id >= c->vocab_size) return MCM_EINVAL;
      float* dst = hidden + ((int64_t)k * S + s) * D;
      for (int d = 0; d < D; ++d) dst[d] = te[(int64_t)id * D + d] + pe[(int64_t)s * D + d];
    }
  for (int l = 0; l < nlayers; ++l) {
    int rc = encoder_layer(h, "text_model", l, hidden, K, S, D, c->t_heads, c->t_mlp, 1);
    if (rc) return rc;
  }
  return MCM_OK;
}

/* CLIPModel.get_text_features (HF:683-715): final_layer_norm (HF:559) → pooled row = first
 * EOS = argmax id (HF:561-581; EOS 49407 is the largest id) → text_projection (no bias,
 * HF:675,713); optional L2 normalise (REF utils/detection_util.py:231). */
int orc_encode_text(orc_handle* h, const int32_t* ids, int32_t K, int32_t S, float* out,
                    int32_t normalize) {
  const mcm_config* c = &h->cfg;
  const int D = c->t_width;
  float* hidden = (float*)malloc(sizeof(float) * (size_t)K * S * D);
  float* pooled = (float*)malloc(sizeof(float) * (size_t)K * D);
  if (!hidden || !pooled) return MCM_ENOMEM;
  int rc = orc_text_hidden(h, ids, K, S, c->t_layers, hidden);
  if (rc) { free(hidden); free(pooled); return rc; }
  const float* lw = P0(h, "text_model.final_layer_norm.weight");
  const float* lb = P0(h, "text_model.final_layer_norm.bias");
  const float* pj = P0(h, "text_projection.weight");
  if (!lw || !lb || !pj) return MCM_ENOWEIGHT;
  for (int k = 0; k < K; ++k) {
    int best = 0;
    for (int s = 1; s < S; ++s)
      if (ids[(int64_t)k * S + s] > ids[(int64_t)k * S + best]) best = s;
    memcpy(pooled + (int64_t)k * D, hidden + ((int64_t)k * S + best) * D,
           sizeof(float) * (size_t)D);
  }
  orc_layernorm(pooled, lw, lb, pooled, K, D, c->ln_eps);
  orc_linear(pooled, pj, NULL, out, K, c->proj_dim, D);
  if (normalize) orc_l2_normalize(out, K, c->proj_dim);
  free(hidden); free(pooled);
  return MCM_OK;
}

/* ------------------------------------------------------------------ scoring tail */

/* REF utils/detection_util.py:232-248.  img [B,Pd], text [K,Pd] (both L2-normalised),
 * scores [B] fp32.  sim is fp32; the reductions follow numpy/scipy on the fp32 softmax:
 *   MCM / max-logit: -max;  energy: -T*logsumexp(sim/T);  entropy: scipy.stats.entropy
 *   (pk normalised by its sum, natural log);  var: -np.var (ddof=0). */
int orc_score_features(const float* img, int32_t B, const float* text, int32_t K, int32_t Pd,
                       float T, int32_t kind, float* scores) {
  if (kind < 0 || kind > MCM_SCORE_VAR) return MCM_EINVAL;
#pragma omp parallel for schedule(static)
  for (int b = 0; b < B; ++b) {
    float* s = (float*)malloc(sizeof(float) * (size_t)K);
    float m = -INFINITY;
    for (int k = 0; k < K; ++k) {
      float a = 0.f;
      for (int d = 0; d < Pd; ++d) a += img[(int64_t)b * Pd + d] * text[(int64_t)k * Pd + d];
      s[k] = a;
      if (a > m) m = a;
    }
    if (kind == MCM_SCORE_MAX_LOGIT) {
      scores[b] = -m;
    } else {
      /* softmax(output / T) in fp32 (torch F.softmax: subtract max, exp, divide) */
      double z = 0.0;
      const float mt = m / T;
      for (int k = 0; k < K; ++k) {
        s[k] = expf(s[k] / T - mt);
        z += s[k];
      }
      if (kind == MCM_SCORE_ENERGY) {
        scores[b] = -(T * (mt + (float)log(z)));
      } else {
        const float rz = (float)(1.0 / z);
        for (int k = 0; k < K; ++k) s[k] *= rz;
        if (kind == MCM_SCORE_MCM) {
          float pm = 0.f;
          for (int k = 0; k < K; ++k) pm = s[k] > pm ? s[k] : pm;
          scores[b] = -pm;
        } else if (kind == MCM_SCORE_ENTROPY) {
          double tot = 0.0, e = 0.0;
          for (int k = 0; k < K; ++k) tot += s[k];
          for (int k = 0; k < K; ++k) {
            const double p = s[k] / tot;
            if (p > 0.0) e -= p * log(p);
          }
          scores[b] = (float)e;
        } else { /* var */
          double mean = 0.0, v = 0.0;
          for (int k = 0; k < K; ++k) mean += s[k];
          mean /= K;
          for (int k = 0; k < K; ++k) v += (s[k] - mean) * (s[k] - mean);
          scores[b] = (float)(-(v / K));
        }
      }
    }
    free(s);
  }
  return MCM_OK;
}

/* ---- image preprocessing: Resize(S) + CenterCrop(S) on uint8 RGB ----------
 * Restates what the reference's loader transform does before ToTensor/Normalize
 * (reference utils/train_eval_util.py:27-33: transforms.Resize(224), transforms.CenterCrop(224)
 * on the PIL image ImageFolder hands out).  Both steps live in third-party code that is not
 * under /root/reference:
 *   - torchvision.transforms (not installed in this image; the reference does not pin a version):
 *     Resize(int) scales the SHORT side to S, long side = int(S * long / short), default
 *     interpolation BILINEAR, and returns the image untouched when short == S; CenterCrop takes
 *     top = round((h - S) / 2), left = round((w - S) / 2) with Python's round-half-to-even.
 *   - Pillow (12.2.0 here) Image.resize(..., BILINEAR) = ImagingResample in src/libImaging/
 *     Resample.c: separable, antialiased (support = max(scale, 1)), horizontal pass first, each
 *     pass in 22-bit fixed point with the result rounded back to uint8.
 * Pinned by tests/golden/preprocess.npz, generated with Pillow itself (tests/golden/make_golden.py).
 * Integer arithmetic after the double-precision coefficient set-up: parity is bit-exact. */
#define ORC_PBITS 22
#define ORC_KMAX 160 /* taps per output coordinate: 2*ceil(scale)+1, i.e. scale factors up to 79 */

/* no fused multiply-add here: Pillow's wheels are built for baseline x86-64, and a contracted
 * (xx + 0.5) * scale ... would not round like theirs */
__attribute__((optimize("fp-contract=off")))
static int resample_coeffs(int in_size, int out_size, int xx, int32_t* kk, int* xmin_out) {
  /* Resample.c precompute_coeffs + normalize_coeffs_8bpc for one output coordinate */
  const double scale = (double)in_size / (double)out_size;
  const double fs = scale < 1.0 ? 1.0 : scale;
  const double support = 1.0 * fs; /* bilinear filter support = 1 */
  const double center = (xx + 0.5) * scale;
  const double ss = 1.0 / fs;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > in_size) xmax = in_size;
  const int n = xmax - xmin;
  double k[ORC_KMAX], ww = 0.0;
  if (n > ORC_KMAX) return -1;
  for (int x = 0; x < n; ++x) {
    double v = (x + xmin - center + 0.5) * ss;
    if (v < 0.0) v = -v;
    const double w = v < 1.0 ? 1.0 - v : 0.0;
    k[x] = w;
    ww += w;
  }
  for (int x = 0; x < n; ++x) {
    if (ww != 0.0) k[x] /= ww;
    kk[x] = k[x] < 0 ? (int32_t)(-0.5 + k[x] * (1 << ORC_PBITS)) : (int32_t)(0.5 + k[x] * (1 << ORC_PBITS));
  }
  *xmin_out = xmin;
  return n;
}

static inline uint8_t clip8(int32_t v) {
  v >>= ORC_PBITS;
  return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

static long round_half_even(double v) { /* Python 3 round() on x.0 / x.5 values */
  const double f = floor(v);
  const double d = v - f;
  if (d < 0.5) return (long)f;
  if (d > 0.5) return (long)f + 1;
  return ((long)f % 2 == 0) ? (long)f : (long)f + 1;
}

void orc_resized_size(int32_t H, int32_t W, int32_t S, int32_t* nh, int32_t* nw) {
  const int32_t shrt = W <= H ? W : H, lng = W <= H ? H : W;
  if (shrt == S) { *nh = H; *nw = W; return; }
  const int32_t nl = (int32_t)((double)S * (double)lng / (double)shrt); /* int(size * long / short) */
  if (W <= H) { *nw = S; *nh = nl; } else { *nw = nl; *nh = S; }
}

/* src [H,W,3] uint8 RGB -> dst [S,S,3]; returns 0, or -1 when the image is smaller than the crop
 * after resizing (cannot happen for Resize(S)+CenterCrop(S)) or the scale exceeds ORC_KMAX taps */
int orc_resize_crop_u8(const uint8_t* src, int32_t H, int32_t W, int32_t S, uint8_t* dst) {
  int32_t nh, nw;
  orc_resized_size(H, W, S, &nh, &nw);
  if (nh < S || nw < S) return -1;
  const int top = (int)round_half_even((nh - S) / 2.0), left = (int)round_half_even((nw - S) / 2.0);
  const int rx = nw != W, ry = nh != H; /* Pillow skips a pass whose size does not change */
  int32_t kx[ORC_KMAX], ky[ORC_KMAX];
  for (int yy = 0; yy < S; ++yy) {
    int ymin = top + yy, ny = 1;
    ky[0] = 1 << ORC_PBITS;
    if (ry && (ny = resample_coeffs(H, nh, top + yy, ky, &ymin)) < 0) return -1;
    for (int xx = 0; xx < S; ++xx) {
      int xmin = left + xx, nx = 1;
      kx[0] = 1 << ORC_PBITS;
      if (rx && (nx = resample_coeffs(W, nw, left + xx, kx, &xmin)) < 0) return -1;
      for (int c = 0; c < 3; ++c) {
        int32_t v = 1 << (ORC_PBITS - 1);
        for (int y = 0; y < ny; ++y) {
          const uint8_t* row = src + ((size_t)(ymin + y) * W + xmin) * 3 + c;
          uint8_t hpx;
          if (rx) {
            int32_t hacc = 1 << (ORC_PBITS - 1);
            for (int x = 0; x < nx; ++x) hacc += (int32_t)row[x * 3] * kx[x];
            hpx = clip8(hacc);
          } else {
            hpx = row[0];
          }
          v += (int32_t)hpx * ky[y];
        }
        dst[((size_t)yy * S + xx) * 3 + c] = ry ? clip8(v) : (uint8_t)((v - (1 << (ORC_PBITS - 1))) >> ORC_PBITS);
      }
    }
  }
  return 0;
}

/* ---- Mahalanobis baseline (--score maha) -------------------------------------------------------
 * Restates the scoring loop of get_Mahalanobis_score, reference utils/detection_util.py:191-199:
 * per class, zero_f = features - class_mean; -0.5 * diag(zero_f @ precision @ zero_f^T); max over
 * classes; the function returns the negation.  fp32 like the reference (two matrix products per
 * class, then the diagonal).  feats [B,P], means [C,P], prec [P,P] -> out [B]. */
void orc_maha_scores(const float* feats, int32_t B, const float* means, int32_t C, const float* prec,
                     int32_t P, float* out) {
#pragma omp parallel for schedule(static)
  for (int32_t b = 0; b < B; ++b) {
    float* z = (float*)malloc((size_t)P * sizeof(float));
    float* zp = (float*)malloc((size_t)P * sizeof(float));
    float best = -INFINITY;
    for (int32_t c = 0; c < C; ++c) {
      for (int32_t i = 0; i < P; ++i) z[i] = feats[(size_t)b * P + i] - means[(size_t)c * P + i];
      for (int32_t j = 0; j < P; ++j) zp[j] = 0.f;
      for (int32_t i = 0; i < P; ++i) { /* zero_f @ precision */
        const float zi = z[i];
        const float* pr = prec + (size_t)i * P;
        for (int32_t j = 0; j < P; ++j) zp[j] += zi * pr[j];
      }
      float d = 0.f;
      for (int32_t j = 0; j < P; ++j) d += zp[j] * z[j];
      const float sc = -0.5f * d;
      if (sc > best) best = sc;
    }
    out[b] = -best;
    free(z);
    free(zp);
  }
}
