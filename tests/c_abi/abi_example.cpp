/* abi_example.cpp — the C ABI of libmcm_hip.so driven from a plain host program (C-style code, built as C++ by hipcc; no Python, no torch): what a non-Python host
 * of the reference's hot path would write, and the check that PyTorch really is only plumbing in this repo.
 *
 * One pass of the path (reference utils/detection_util.py:223-248): seeded parameters by HF name ->
 * mcm_set_weight / mcm_finalize_weights; K prompts -> mcm_encode_text (once); B images in HBM -> mcm_score ->
 * [B] MCM scores; AUROC / AUPR / FPR95 of two score vectors -> mcm_measures.  The same parameters, pixels and
 * token ids go through the CPU oracle (oracle/libmcm_oracle.so: TEST INFRASTRUCTURE, the checker) and the scores
 * are compared.  tests/test_gpu_c_abi.py builds and runs this with hipcc on the GPU box.
 *
 * Build: hipcc tests/c_abi/abi_example.cpp -I include -L mcm_amd -lmcm_hip -L oracle -lmcm_oracle \
 *              -Wl,-rpath,$PWD/mcm_amd -Wl,-rpath,$PWD/oracle -o /tmp/abi_example
 * Run:   /tmp/abi_example [precision: 1 = fp32 (default), 2 = fp16, 0 = bf16]
 */
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mcm.h"

/* the oracle's C interface (oracle/mcm_oracle.c) */
typedef struct orc_handle orc_handle;
#ifdef __cplusplus
extern "C" {
#endif
int orc_create(const mcm_config* cfg, orc_handle** out);
void orc_destroy(orc_handle* h);
int orc_set_weight(orc_handle* h, const char* name, const float* ptr, const int64_t* shape, int32_t ndim);
int orc_encode_image(orc_handle* h, const float* pixels, int32_t B, float* out, int32_t normalize);
int orc_encode_text(orc_handle* h, const int32_t* ids, int32_t K, int32_t S, float* out, int32_t normalize);
int orc_score_features(const float* img, int32_t B, const float* text, int32_t K, int32_t Pd, float T, int32_t kind,
                       float* scores);
#ifdef __cplusplus
}
#endif

#define CHECK(x)                                                                   \
  do {                                                                             \
    int rc_ = (x);                                                                 \
    if (rc_) {                                                                     \
      fprintf(stderr, "%s failed: rc=%d (%s)\n", #x, rc_, mcm_last_error(h));      \
      return 1;                                                                    \
    }                                                                              \
  } while (0)
#define HIPCHECK(x)                                                                \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                      \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

static uint64_t rng_state = 0x243F6A8885A308D3ull;
static float frand(void) { /* uniform [-1, 1) */
  rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
  return (float)((rng_state >> 40) & 0xffffff) / (float)0x800000 - 1.0f;
}

static mcm_handle* h = NULL;
static orc_handle* o = NULL;

/* one parameter: same values into both libraries; scale 1 +- for LayerNorm weights, small for everything else */
static int set_param(const char* name, int ndim, int64_t d0, int64_t d1, int64_t d2, int64_t d3, float center, float amp) {
  const int64_t shape[4] = {d0, d1, d2, d3};
  int64_t n = 1;
  for (int i = 0; i < ndim; ++i) n *= shape[i];
  float* v = (float*)malloc(sizeof(float) * (size_t)n);
  for (int64_t i = 0; i < n; ++i) v[i] = center + amp * frand();
  int rc = mcm_set_weight(h, name, v, MCM_DT_F32, shape, ndim);
  if (!rc) rc = orc_set_weight(o, name, v, shape, ndim);
  if (rc) fprintf(stderr, "set %s: rc=%d (%s)\n", name, rc, mcm_last_error(h));
  free(v);
  return rc;
}

static int set_tower(const char* tower, int layers, int D, int ff) {
  char n[160];
  int rc = 0;
  for (int l = 0; l < layers && !rc; ++l) {
    const char* proj[4] = {"q_proj", "k_proj", "v_proj", "out_proj"};
    for (int p = 0; p < 4 && !rc; ++p) {
      snprintf(n, sizeof n, "%s.encoder.layers.%d.self_attn.%s.weight", tower, l, proj[p]);
      rc = set_param(n, 2, D, D, 0, 0, 0.f, 1.5f / sqrtf((float)D));
      snprintf(n, sizeof n, "%s.encoder.layers.%d.self_attn.%s.bias", tower, l, proj[p]);
      if (!rc) rc = set_param(n, 1, D, 0, 0, 0, 0.f, 0.05f);
    }
    for (int k = 1; k <= 2 && !rc; ++k) {
      snprintf(n, sizeof n, "%s.encoder.layers.%d.layer_norm%d.weight", tower, l, k);
      rc = set_param(n, 1, D, 0, 0, 0, 1.f, 0.1f);
      snprintf(n, sizeof n, "%s.encoder.layers.%d.layer_norm%d.bias", tower, l, k);
      if (!rc) rc = set_param(n, 1, D, 0, 0, 0, 0.f, 0.05f);
    }
    snprintf(n, sizeof n, "%s.encoder.layers.%d.mlp.fc1.weight", tower, l);
    if (!rc) rc = set_param(n, 2, ff, D, 0, 0, 0.f, 1.5f / sqrtf((float)D));
    snprintf(n, sizeof n, "%s.encoder.layers.%d.mlp.fc1.bias", tower, l);
    if (!rc) rc = set_param(n, 1, ff, 0, 0, 0, 0.f, 0.05f);
    snprintf(n, sizeof n, "%s.encoder.layers.%d.mlp.fc2.weight", tower, l);
    if (!rc) rc = set_param(n, 2, D, ff, 0, 0, 0.f, 1.5f / sqrtf((float)ff));
    snprintf(n, sizeof n, "%s.encoder.layers.%d.mlp.fc2.bias", tower, l);
    if (!rc) rc = set_param(n, 1, D, 0, 0, 0, 0.f, 0.05f);
  }
  return rc;
}

int main(int argc, char** argv) {
  mcm_config c;
  memset(&c, 0, sizeof c);
  c.abi_version = MCM_ABI_VERSION;
  c.device = 0;
  c.precision = argc > 1 ? atoi(argv[1]) : MCM_PREC_F32;
  c.image_size = 64; c.patch_size = 16; c.v_width = 128; c.v_heads = 2; c.v_layers = 2; c.v_mlp = 512;
  c.vocab_size = 49408; c.max_positions = 77; c.t_width = 128; c.t_heads = 2; c.t_layers = 2; c.t_mlp = 512;
  c.proj_dim = 64; c.ln_eps = 1e-5f; c.max_batch = 16; c.max_prompt_tokens = 2048;
  const int B = 12, K = 9, S = 11, P = c.proj_dim, ntok = (c.image_size / c.patch_size) * (c.image_size / c.patch_size) + 1;

  if (mcm_create(&c, &h)) { fprintf(stderr, "mcm_create: %s\n", mcm_last_error(NULL)); return 1; }
  if (orc_create(&c, &o)) { fprintf(stderr, "orc_create failed\n"); return 1; }
  int rc = 0;
  rc |= set_param("vision_model.embeddings.class_embedding", 1, c.v_width, 0, 0, 0, 0.f, 0.1f);
  rc |= set_param("vision_model.embeddings.patch_embedding.weight", 4, c.v_width, 3, c.patch_size, c.patch_size, 0.f, 0.05f);
  rc |= set_param("vision_model.embeddings.position_embedding.weight", 2, ntok, c.v_width, 0, 0, 0.f, 0.1f);
  rc |= set_param("vision_model.pre_layrnorm.weight", 1, c.v_width, 0, 0, 0, 1.f, 0.1f);
  rc |= set_param("vision_model.pre_layrnorm.bias", 1, c.v_width, 0, 0, 0, 0.f, 0.05f);
  rc |= set_tower("vision_model", c.v_layers, c.v_width, c.v_mlp);
  rc |= set_param("vision_model.post_layernorm.weight", 1, c.v_width, 0, 0, 0, 1.f, 0.1f);
  rc |= set_param("vision_model.post_layernorm.bias", 1, c.v_width, 0, 0, 0, 0.f, 0.05f);
  rc |= set_param("visual_projection.weight", 2, P, c.v_width, 0, 0, 0.f, 0.15f);
  rc |= set_param("text_model.embeddings.token_embedding.weight", 2, c.vocab_size, c.t_width, 0, 0, 0.f, 0.3f);
  rc |= set_param("text_model.embeddings.position_embedding.weight", 2, c.max_positions, c.t_width, 0, 0, 0.f, 0.1f);
  rc |= set_tower("text_model", c.t_layers, c.t_width, c.t_mlp);
  rc |= set_param("text_model.final_layer_norm.weight", 1, c.t_width, 0, 0, 0, 1.f, 0.1f);
  rc |= set_param("text_model.final_layer_norm.bias", 1, c.t_width, 0, 0, 0, 0.f, 0.05f);
  rc |= set_param("text_projection.weight", 2, P, c.t_width, 0, 0, 0.f, 0.15f);
  if (rc) return 1;
  CHECK(mcm_finalize_weights(h));
  uint64_t inexact = 0;
  int32_t split = 0;
  CHECK(mcm_weights_operand_exact(h, &inexact, &split)); /* these random weights are fp32-valued: a 16-bit mode splits them */
  printf("vision GEMM-weight elements that are not operand-dtype numbers: %llu; split-weight GEMMs: %d\n",
         (unsigned long long)inexact, (int)split);

  /* prompts: BOS r.. EOS pad(=EOS); pixels: [B,3,S,S] fp32, already normalised */
  int32_t* ids = (int32_t*)malloc(sizeof(int32_t) * K * S);
  for (int k = 0; k < K; ++k) {
    const int n = 3 + k % 6;
    for (int j = 0; j < S; ++j) ids[k * S + j] = 49407;
    ids[k * S] = 49406;
    for (int j = 1; j <= n; ++j) ids[k * S + j] = 1 + (int)((frand() * 0.5f + 0.5f) * 49000.f);
  }
  const size_t npx = (size_t)B * 3 * c.image_size * c.image_size;
  float* px = (float*)malloc(sizeof(float) * npx);
  for (size_t i = 0; i < npx; ++i) px[i] = 1.5f * frand();

  hipStream_t stream;
  HIPCHECK(hipStreamCreate(&stream));
  float *px_dev, *txt_dev, *sc_dev;
  HIPCHECK(hipMalloc((void**)&px_dev, sizeof(float) * npx));
  HIPCHECK(hipMalloc((void**)&txt_dev, sizeof(float) * K * P));
  HIPCHECK(hipMalloc((void**)&sc_dev, sizeof(float) * B));
  HIPCHECK(hipMemcpyAsync(px_dev, px, sizeof(float) * npx, hipMemcpyHostToDevice, stream));
  CHECK(mcm_encode_text(h, ids, K, S, txt_dev, stream));                 /* once per dataset */
  CHECK(mcm_score(h, px_dev, B, txt_dev, K, 1.0f, MCM_SCORE_MCM, sc_dev, stream)); /* once per batch */
  float got[16], want[16];
  HIPCHECK(hipMemcpyAsync(got, sc_dev, sizeof(float) * B, hipMemcpyDeviceToHost, stream));
  HIPCHECK(hipStreamSynchronize(stream));

  float* fi = (float*)malloc(sizeof(float) * B * P);
  float* ft = (float*)malloc(sizeof(float) * K * P);
  if (orc_encode_image(o, px, B, fi, 1) || orc_encode_text(o, ids, K, S, ft, 1) ||
      orc_score_features(fi, B, ft, K, P, 1.0f, MCM_SCORE_MCM, want)) {
    fprintf(stderr, "oracle failed\n");
    return 1;
  }
  double worst = 0;
  for (int b = 0; b < B; ++b) {
    const double d = fabs((double)got[b] - want[b]);
    if (d > worst) worst = d;
  }
  const double tol = c.precision == MCM_PREC_F32 ? 2e-6 : (c.precision == MCM_PREC_F16 ? 3e-4 : 2e-3);
  printf("scores[0..2] native %.8f %.8f %.8f | oracle %.8f %.8f %.8f | max|d| = %.3e (tol %.0e)\n", got[0], got[1], got[2],
         want[0], want[1], want[2], worst, tol);

  /* fp16 handles: the same batch through the split-activation arm (mcm_score_x2: what re-scores the images near the FPR95
   * threshold) — same weights, same workspace, fp32-grade scores */
  double worst2 = 0;
  if (mcm_x2_max_batch(h) > 0) {
    float got2[16];
    CHECK(mcm_score_x2(h, px_dev, MCM_PIXELS_F32_NCHW, B, txt_dev, K, 1.0f, MCM_SCORE_MCM, sc_dev, stream));
    HIPCHECK(hipMemcpyAsync(got2, sc_dev, sizeof(float) * B, hipMemcpyDeviceToHost, stream));
    HIPCHECK(hipStreamSynchronize(stream));
    for (int b = 0; b < B; ++b) {
      const double d = fabs((double)got2[b] - want[b]);
      if (d > worst2) worst2 = d;
    }
    printf("split-activation arm: max|d| vs the oracle = %.3e (tol 2e-06)\n", worst2);
    CHECK(mcm_score(h, px_dev, B, txt_dev, K, 1.0f, MCM_SCORE_MCM, sc_dev, stream)); /* (the metrics below: the handle's own arm) */
  } else if (c.precision == MCM_PREC_F16) {
    fprintf(stderr, "an fp16 handle without a split-activation arm\n");
    return 1;
  }

  /* detection metrics on the device: first half of the batch as "ID", second as "OOD" */
  double m[3];
  CHECK(mcm_measures(h, sc_dev, B / 2, sc_dev + B / 2, B - B / 2, 1, 0.95, m, stream));
  uint64_t sat = 0;
  CHECK(mcm_saturation_count(h, 1, &sat, stream));
  printf("AUROC %.6f AUPR %.6f FPR95 %.6f; fp16 saturation events %llu\n", m[0], m[1], m[2], (unsigned long long)sat);

  mcm_destroy(h);
  orc_destroy(o);
  const int ok = worst < tol && worst2 < 2e-6 && m[0] >= 0.0 && m[0] <= 1.0 && sat == 0;
  printf(ok ? "abi_example OK\n" : "abi_example FAILED\n");
  return ok ? 0 : 2;
}
