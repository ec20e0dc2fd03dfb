/* jpeg_ref.c — CPU restatement of the device half of the JPEG ingest (mcm_amd/csrc/jpeg.hip): quantised DCT coefficients ->
 * the RGB pixels libjpeg / libjpeg-turbo produce with their default decompression settings (what Pillow's
 * Image.open(path).convert("RGB") returns, i.e. the reference loader's input: torchvision ImageFolder,
 * utils/train_eval_util.py:96-146).  TEST INFRASTRUCTURE ONLY — the checker for jpeg.hip; pinned against Pillow itself by
 * tests/test_jpeg_oracle.py (Pillow decodes the same files in the build container and on the GPU box).
 *
 * Third-party algorithms restated (libjpeg 6b / libjpeg-turbo, IJG licence; no source under /root/reference):
 *   jidctint.c  jpeg_idct_islow   the default dct_method: 13-bit fixed-point constants, PASS1_BITS 2, two passes
 *   jdsample.c  h2v1_fancy_upsample / h2v2_fancy_upsample   the default do_fancy_upsampling triangle filters (plain
 *               replication when the chroma plane is at most two samples wide);
 *               jdmainct.c's context rows: above the first / below the last REAL chroma row the edge row is repeated
 *   jdcolor.c   build_ycc_rgb_table / ycc_rgb_convert   16-bit fixed-point YCbCr -> RGB
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define CONST_BITS 13
#define PASS1_BITS 2
#define DESCALE(x, n) (((x) + ((int64_t)1 << ((n) - 1))) >> (n))

static inline uint8_t clamp8(int64_t v) { return v < 0 ? 0 : (v > 255 ? 255 : (uint8_t)v); }

/* one 8x8 block: coef (natural order) * quant -> samples, row stride `stride` */
static void idct_islow(const int16_t* in, const uint16_t* q, uint8_t* out, int stride) {
  int64_t ws[64];
  for (int c = 0; c < 8; ++c) {
#define D(r) ((int64_t)in[(r) * 8 + c] * q[(r) * 8 + c])
    int64_t z2 = D(2), z3 = D(6);
    int64_t z1 = (z2 + z3) * 4433;
    int64_t tmp2 = z1 + z3 * (-15137), tmp3 = z1 + z2 * 6270;
    z2 = D(0); z3 = D(4);
    int64_t tmp0 = (z2 + z3) << CONST_BITS, tmp1 = (z2 - z3) << CONST_BITS;
    int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = D(7); tmp1 = D(5); tmp2 = D(3); tmp3 = D(1);
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    int64_t z4 = tmp1 + tmp3, z5 = (z3 + z4) * 9633;
    tmp0 *= 2446; tmp1 *= 16819; tmp2 *= 25172; tmp3 *= 12299;
    z1 *= -7373; z2 *= -20995; z3 *= -16069; z4 *= -3196;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    ws[0 * 8 + c] = DESCALE(tmp10 + tmp3, CONST_BITS - PASS1_BITS);
    ws[7 * 8 + c] = DESCALE(tmp10 - tmp3, CONST_BITS - PASS1_BITS);
    ws[1 * 8 + c] = DESCALE(tmp11 + tmp2, CONST_BITS - PASS1_BITS);
    ws[6 * 8 + c] = DESCALE(tmp11 - tmp2, CONST_BITS - PASS1_BITS);
    ws[2 * 8 + c] = DESCALE(tmp12 + tmp1, CONST_BITS - PASS1_BITS);
    ws[5 * 8 + c] = DESCALE(tmp12 - tmp1, CONST_BITS - PASS1_BITS);
    ws[3 * 8 + c] = DESCALE(tmp13 + tmp0, CONST_BITS - PASS1_BITS);
    ws[4 * 8 + c] = DESCALE(tmp13 - tmp0, CONST_BITS - PASS1_BITS);
#undef D
  }
  for (int r = 0; r < 8; ++r) {
    const int64_t* w = ws + r * 8;
    int64_t z2 = w[2], z3 = w[6];
    int64_t z1 = (z2 + z3) * 4433;
    int64_t tmp2 = z1 + z3 * (-15137), tmp3 = z1 + z2 * 6270;
    int64_t tmp0 = (w[0] + w[4]) << CONST_BITS, tmp1 = (w[0] - w[4]) << CONST_BITS;
    int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    int64_t z4 = tmp1 + tmp3, z5 = (z3 + z4) * 9633;
    tmp0 *= 2446; tmp1 *= 16819; tmp2 *= 25172; tmp3 *= 12299;
    z1 *= -7373; z2 *= -20995; z3 *= -16069; z4 *= -3196;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    uint8_t* o = out + r * stride;
    const int S = CONST_BITS + PASS1_BITS + 3;
    o[0] = clamp8(DESCALE(tmp10 + tmp3, S) + 128);
    o[7] = clamp8(DESCALE(tmp10 - tmp3, S) + 128);
    o[1] = clamp8(DESCALE(tmp11 + tmp2, S) + 128);
    o[6] = clamp8(DESCALE(tmp11 - tmp2, S) + 128);
    o[2] = clamp8(DESCALE(tmp12 + tmp1, S) + 128);
    o[5] = clamp8(DESCALE(tmp12 - tmp1, S) + 128);
    o[3] = clamp8(DESCALE(tmp13 + tmp0, S) + 128);
    o[4] = clamp8(DESCALE(tmp13 - tmp0, S) + 128);
  }
}

/* chroma sample at full-resolution (x, y) from a plane of dw x dh real samples (row stride `stride`) */
static int up_h2v2(const uint8_t* p, int stride, int dw, int dh, int x, int y) {
  const int r = y >> 1, rn = (y & 1) ? (r + 1 < dh ? r + 1 : dh - 1) : (r > 0 ? r - 1 : 0);
  const int c = x >> 1;
  const uint8_t *a = p + (size_t)r * stride, *b = p + (size_t)rn * stride;
  const int this_ = a[c] * 3 + b[c];
  if (x & 1) {
    if (c == dw - 1) return (this_ * 4 + 7) >> 4;
    return (this_ * 3 + (a[c + 1] * 3 + b[c + 1]) + 7) >> 4;
  }
  if (c == 0) return (this_ * 4 + 8) >> 4;
  return (this_ * 3 + (a[c - 1] * 3 + b[c - 1]) + 8) >> 4;
}
static int up_h2v1(const uint8_t* p, int stride, int dw, int x, int y) {
  const uint8_t* a = p + (size_t)y * stride;
  const int c = x >> 1;
  if (x & 1) return c == dw - 1 ? a[c] : (a[c] * 3 + a[c + 1] + 2) >> 2;
  return c == 0 ? a[c] : (a[c] * 3 + a[c - 1] + 1) >> 2;
}

/* coef[c]: int16 [hb[c]][wb[c]][64] natural order; quant: [ncomp][64] natural order; rgb: [height][width][3].
 * 0 ok, -1 unsupported sampling */
int orc_jpeg_reconstruct(const int16_t* const* coef, const uint16_t* quant, int width, int height, int ncomp, const int* hs,
                         const int* vs, const int* wb, const int* hb, uint8_t* rgb) {
  uint8_t* plane[3] = {0, 0, 0};
  for (int c = 0; c < ncomp; ++c) {
    const int stride = wb[c] * 8;
    plane[c] = (uint8_t*)malloc((size_t)stride * hb[c] * 8);
    for (int by = 0; by < hb[c]; ++by)
      for (int bx = 0; bx < wb[c]; ++bx)
        idct_islow(coef[c] + ((size_t)by * wb[c] + bx) * 64, quant + c * 64, plane[c] + (size_t)by * 8 * stride + bx * 8, stride);
  }
  int rc = 0;
  if (ncomp == 1) {
    for (int y = 0; y < height; ++y)
      for (int x = 0; x < width; ++x) {
        const uint8_t v = plane[0][(size_t)y * wb[0] * 8 + x];
        uint8_t* o = rgb + ((size_t)y * width + x) * 3;
        o[0] = o[1] = o[2] = v;
      }
  } else {
    const int H = hs[0], V = vs[0];
    if (!((H == 1 && V == 1) || (H == 2 && V == 1) || (H == 2 && V == 2)) || hs[1] != 1 || vs[1] != 1 || hs[2] != 1 || vs[2] != 1) {
      rc = -1;
    } else {
      const int dw = (width + H - 1) / H, dh = (height + V - 1) / V;
      const int s1 = wb[1] * 8, s2 = wb[2] * 8;
      for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) {
          const int yy = plane[0][(size_t)y * wb[0] * 8 + x];
          int cb, cr;
          if (H == 1) {
            cb = plane[1][(size_t)y * s1 + x];
            cr = plane[2][(size_t)y * s2 + x];
          } else if (dw <= 2) {  /* jdsample.c jinit_upsampler: the fancy filters need downsampled_width > 2; else replication */
            cb = plane[1][(size_t)(y / V) * s1 + (x >> 1)];
            cr = plane[2][(size_t)(y / V) * s2 + (x >> 1)];
          } else if (V == 1) {
            cb = up_h2v1(plane[1], s1, dw, x, y);
            cr = up_h2v1(plane[2], s2, dw, x, y);
          } else {
            cb = up_h2v2(plane[1], s1, dw, dh, x, y);
            cr = up_h2v2(plane[2], s2, dw, dh, x, y);
          }
          const int xb = cb - 128, xr = cr - 128;
          const int r = yy + (int)((91881 * (int64_t)xr + 32768) >> 16);
          const int b = yy + (int)((116130 * (int64_t)xb + 32768) >> 16);
          const int g = yy + (int)((-22554 * (int64_t)xb + 32768 + -46802 * (int64_t)xr) >> 16);
          uint8_t* o = rgb + ((size_t)y * width + x) * 3;
          o[0] = clamp8(r); o[1] = clamp8(g); o[2] = clamp8(b);
        }
    }
  }
  for (int c = 0; c < ncomp; ++c) free(plane[c]);
  return rc;
}
