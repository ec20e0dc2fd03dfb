/*
 * mcm_cpu.c — libmcm_cpu.so: the CPU twin of the C ABI (include/mcm.h), SURVEY.md section 8b ("identical exports from
 * libmcm_cpu.so: stream argument ignored, pointers are host") and BASELINE config 1 ("CPU path, plumbing, no GPU").
 *
 * TEST INFRASTRUCTURE, like everything under oracle/: the entry points forward to the CPU restatement in mcm_oracle.c
 * (compiled into this library).  It lets a host without a GPU exercise a binding written against include/mcm.h — same
 * struct, same names, same argument meaning and error codes — and is what tests/test_cpu_abi_twin.py runs config 1's
 * plumbing through.  Nothing under mcm_amd/ loads it: the product path has no CPU fallback and fails loudly without
 * libmcm_hip.so and a GPU.  `*_dev` pointers of the header are HOST pointers here; `stream` is ignored; every call is
 * synchronous.  Entry points of the hot path only (the set section 8b names, plus the _ex / u8-free variants the Python
 * mirror calls); the device-only extras (profiling, saturation watch, operator hooks, resize, metrics) are not twinned.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/mcm.h"

/* the oracle's own interface (mcm_oracle.c) */
typedef struct orc_handle orc_handle;
int orc_create(const mcm_config* cfg, orc_handle** out);
void orc_destroy(orc_handle* h);
const char* orc_last_error(const orc_handle* h);
int orc_set_weight(orc_handle* h, const char* name, const float* ptr, const int64_t* shape, int32_t ndim);
int orc_encode_image(orc_handle* h, const float* pixels, int32_t B, float* out, int32_t normalize);
int orc_encode_text(orc_handle* h, const int32_t* ids, int32_t K, int32_t S, float* out, int32_t normalize);
int orc_score_features(const float* img, int32_t B, const float* text, int32_t K, int32_t Pd, float T, int32_t kind,
                       float* scores);

struct mcm_handle {
  orc_handle* o;
  mcm_config cfg;
  int finalized;
  char err[256];
};

static char g_create_err[256];

static int fail(mcm_handle* h, int code, const char* msg) {
  snprintf(h ? h->err : g_create_err, 256, "%s", msg);
  return code;
}

int mcm_abi_version(void) { return MCM_ABI_VERSION; }

const char* mcm_last_error(const mcm_handle* h) { return h ? h->err : g_create_err; }

int mcm_create(const mcm_config* cfg, mcm_handle** out) {
  if (!cfg || !out) return fail(NULL, MCM_EINVAL, "null argument");
  if (cfg->abi_version != MCM_ABI_VERSION) return fail(NULL, MCM_EINVAL, "ABI version mismatch");
  if (cfg->max_batch <= 0 || cfg->max_prompt_tokens <= 0) return fail(NULL, MCM_EINVAL, "bad workspace bounds");
  mcm_handle* h = (mcm_handle*)calloc(1, sizeof(mcm_handle));
  if (!h) return fail(NULL, MCM_ENOMEM, "calloc");
  h->cfg = *cfg;
  int rc = orc_create(cfg, &h->o);
  if (rc) {
    free(h);
    return fail(NULL, rc, "orc_create");
  }
  *out = h;
  return MCM_OK;
}

void mcm_destroy(mcm_handle* h) {
  if (!h) return;
  orc_destroy(h->o);
  free(h);
}

static float half_to_float(uint16_t v) {
  const uint32_t sign = (uint32_t)(v & 0x8000u) << 16, e = (v >> 10) & 0x1fu, m = v & 0x3ffu;
  uint32_t u;
  float f;
  if (e == 0) {
    f = (float)m * (1.0f / 16777216.0f);
    memcpy(&u, &f, 4);
    u |= sign;
  } else if (e == 31) {
    u = sign | 0x7f800000u | (m << 13);
  } else {
    u = sign | ((e + 112u) << 23) | (m << 13);
  }
  memcpy(&f, &u, 4);
  return f;
}

int mcm_set_weight(mcm_handle* h, const char* hf_name, const void* host_ptr, int32_t dtype, const int64_t* shape,
                   int32_t ndim) {
  if (!h || !hf_name || !host_ptr || (ndim > 0 && !shape)) return fail(h, MCM_EINVAL, "null argument");
  if (dtype != MCM_DT_F32 && dtype != MCM_DT_F16 && dtype != MCM_DT_BF16) return fail(h, MCM_EINVAL, "unknown dtype");
  int64_t n = 1;
  for (int i = 0; i < ndim; ++i) n *= shape[i];
  const float* src = (const float*)host_ptr;
  float* wide = NULL;
  if (dtype != MCM_DT_F32) {
    wide = (float*)malloc(sizeof(float) * (size_t)n);
    if (!wide) return fail(h, MCM_ENOMEM, "malloc");
    const uint16_t* q = (const uint16_t*)host_ptr;
    for (int64_t i = 0; i < n; ++i) {
      if (dtype == MCM_DT_F16) {
        wide[i] = half_to_float(q[i]);
      } else {
        const uint32_t u = (uint32_t)q[i] << 16;
        memcpy(&wide[i], &u, 4);
      }
    }
    src = wide;
  }
  int rc = orc_set_weight(h->o, hf_name, src, shape, ndim);
  free(wide);
  h->finalized = 0;
  return rc ? fail(h, rc, "orc_set_weight") : MCM_OK;
}

int mcm_finalize_weights(mcm_handle* h) {
  if (!h) return MCM_EINVAL;
  h->finalized = 1; /* a missing parameter is reported by the first encode call that needs it (MCM_ENOWEIGHT) */
  return MCM_OK;
}

int mcm_weights_operand_exact(mcm_handle* h, uint64_t* inexact_host, int32_t* split_host) {
  if (!h) return MCM_EINVAL;
  if (!h->finalized) return fail(h, MCM_ENOWEIGHT, "mcm_finalize_weights has not been called");
  if (inexact_host) *inexact_host = 0; /* fp32 arithmetic throughout: every weight is exact */
  if (split_host) *split_host = 0;
  return MCM_OK;
}

static int ready(mcm_handle* h) {
  if (!h) return MCM_EINVAL;
  if (!h->finalized) return fail(h, MCM_ENOWEIGHT, "mcm_finalize_weights has not been called");
  return MCM_OK;
}

static int from_orc(mcm_handle* h, int rc) {
  if (rc) snprintf(h->err, sizeof(h->err), "%s", orc_last_error(h->o));
  return rc;
}

int mcm_encode_text_ex(mcm_handle* h, const int32_t* ids_host, int32_t K, int32_t S, int32_t normalize, float* out_dev,
                       void* stream) {
  (void)stream;
  int rc = ready(h);
  if (rc) return rc;
  if (!ids_host || !out_dev) return fail(h, MCM_EINVAL, "null pointer");
  if (K <= 0 || S <= 0) return fail(h, MCM_EINVAL, "K and S must be positive");
  if (S > h->cfg.max_positions) return fail(h, MCM_ERANGE, "sequence length exceeds max_position_embeddings");
  for (int64_t i = 0; i < (int64_t)K * S; ++i)
    if (ids_host[i] < 0 || ids_host[i] >= h->cfg.vocab_size) return fail(h, MCM_EINVAL, "token id out of range");
  return from_orc(h, orc_encode_text(h->o, ids_host, K, S, out_dev, normalize != 0));
}

int mcm_encode_text(mcm_handle* h, const int32_t* ids_host, int32_t K, int32_t S, float* out_dev, void* stream) {
  return mcm_encode_text_ex(h, ids_host, K, S, 1, out_dev, stream);
}

int mcm_encode_image_ex(mcm_handle* h, const void* pixels_dev, int32_t pixel_format, int32_t B, int32_t normalize,
                        float* out_dev, void* stream) {
  (void)stream;
  int rc = ready(h);
  if (rc) return rc;
  if (!pixels_dev || !out_dev) return fail(h, MCM_EINVAL, "null pointer");
  if (pixel_format != MCM_PIXELS_F32_NCHW) return fail(h, MCM_EINVAL, "the CPU twin takes fp32 NCHW pixels only");
  if (B <= 0 || B > h->cfg.max_batch) return fail(h, MCM_ERANGE, "batch exceeds cfg.max_batch");
  return from_orc(h, orc_encode_image(h->o, (const float*)pixels_dev, B, out_dev, normalize != 0));
}

int mcm_encode_image(mcm_handle* h, const float* pixels_dev, int32_t B, float* out_dev, void* stream) {
  return mcm_encode_image_ex(h, pixels_dev, MCM_PIXELS_F32_NCHW, B, 1, out_dev, stream);
}

int mcm_encode_image_raw(mcm_handle* h, const float* pixels_dev, int32_t B, float* out_dev, void* stream) {
  return mcm_encode_image_ex(h, pixels_dev, MCM_PIXELS_F32_NCHW, B, 0, out_dev, stream);
}

int mcm_score_features(mcm_handle* h, const float* img_feat_dev, int32_t B, const float* text_feat_dev, int32_t K,
                       float T, int32_t kind, float* scores_dev, void* stream) {
  (void)stream;
  if (!h) return MCM_EINVAL;
  if (!img_feat_dev || !text_feat_dev || !scores_dev) return fail(h, MCM_EINVAL, "null pointer");
  if (B <= 0 || K <= 0 || kind < 0 || kind > MCM_SCORE_VAR || !(T > 0.f)) return fail(h, MCM_EINVAL, "bad B / K / kind / T");
  return orc_score_features(img_feat_dev, B, text_feat_dev, K, h->cfg.proj_dim, T, kind, scores_dev);
}

int mcm_score(mcm_handle* h, const float* pixels_dev, int32_t B, const float* text_feat_dev, int32_t K, float T,
              int32_t kind, float* scores_dev, void* stream) {
  int rc = ready(h);
  if (rc) return rc;
  if (B <= 0 || B > h->cfg.max_batch) return fail(h, MCM_ERANGE, "batch exceeds cfg.max_batch");
  float* feat = (float*)malloc(sizeof(float) * (size_t)B * h->cfg.proj_dim);
  if (!feat) return fail(h, MCM_ENOMEM, "malloc");
  rc = mcm_encode_image(h, pixels_dev, B, feat, stream);
  if (!rc) rc = mcm_score_features(h, feat, B, text_feat_dev, K, T, kind, scores_dev, stream);
  free(feat);
  return rc;
}
