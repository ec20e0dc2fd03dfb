// jpeg.hip — device half of the JPEG ingest (one step up the reference's loader): quantised DCT
// coefficients (written by the host's entropy decoder, jpeg_entropy.cpp) -> the RGB pixels libjpeg / libjpeg-turbo produce
// with their default decompression settings, i.e. what Pillow's Image.open(path).convert("RGB") hands the reference's
// transform (torchvision ImageFolder, utils/train_eval_util.py:96-146).  Integer arithmetic throughout; bit-exact against
// Pillow (tests/test_gpu_jpeg.py; the restatement is pinned on the CPU by tests/test_jpeg_oracle.py).
//
// Third-party algorithms (libjpeg 6b / libjpeg-turbo, outside /root/reference):
//   jidctint.c  jpeg_idct_islow — the default dct_method: dequantise, two 1-D passes, 13-bit constants, PASS1_BITS = 2
//   jdsample.c  h2v1 / h2v2 fancy upsampling — the default triangle filters; jdmainct.c repeats the first / last REAL chroma
//               row as the context row above / below the image
//   jdcolor.c   YCbCr -> RGB, 16-bit fixed point
// Kernels: idct_kernel — one thread per 8x8 block (128 B of coefficients in, 64 samples out into the component's plane;
// 2.4 M blocks per 512 half-megapixel images), colour_kernel — one thread per eight pixels of a row (Y + the two upsampled chroma
// samples from the planes -> six dword stores).  Both are HBM-bound streaming kernels: 1.5 int16 per pixel in, planes written and
// read once, 3 bytes per pixel out.  32-bit arithmetic like libjpeg-turbo's SIMD paths (what Pillow actually runs).
#include "common.hpp"

namespace {

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }
__device__ __forceinline__ int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// the 1-D kernel of jpeg_idct_islow on eight inputs already dequantised / taken from the workspace
#define IDCT_1D(in0, in1, in2, in3, in4, in5, in6, in7, SHIFT, o0, o1, o2, o3, o4, o5, o6, o7) \
  {                                                                                            \
    int z2 = in2, z3 = in6;                                                                    \
    int z1 = (z2 + z3) * 4433;                                                                 \
    int tmp2 = z1 + z3 * (-15137), tmp3 = z1 + z2 * 6270;                                      \
    int tmp0 = (in0 + in4) << 13, tmp1 = (in0 - in4) << 13;                                    \
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2; \
    tmp0 = in7; tmp1 = in5; tmp2 = in3; tmp3 = in1;                                            \
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;                                      \
    int z4 = tmp1 + tmp3;                                                                      \
    const int z5 = (z3 + z4) * 9633;                                                           \
    tmp0 *= 2446; tmp1 *= 16819; tmp2 *= 25172; tmp3 *= 12299;                                 \
    z1 *= -7373; z2 *= -20995; z3 *= -16069; z4 *= -3196;                                      \
    z3 += z5; z4 += z5;                                                                        \
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;                        \
    o0 = descale(tmp10 + tmp3, SHIFT); o7 = descale(tmp10 - tmp3, SHIFT);                      \
    o1 = descale(tmp11 + tmp2, SHIFT); o6 = descale(tmp11 - tmp2, SHIFT);                      \
    o2 = descale(tmp12 + tmp1, SHIFT); o5 = descale(tmp12 - tmp1, SHIFT);                      \
    o3 = descale(tmp13 + tmp0, SHIFT); o4 = descale(tmp13 - tmp0, SHIFT);                      \
  }

__global__ __launch_bounds__(256) void jpeg_idct_kernel(const JpegImageDev* __restrict__ meta, const uint16_t* __restrict__ quant,
                                                        const char* __restrict__ coef, uint8_t* __restrict__ planes) {
  const JpegImageDev m = meta[blockIdx.y];
  const int nb0 = m.wb[0] * m.hb[0], nb1 = m.ncomp > 1 ? m.wb[1] * m.hb[1] : 0, nb2 = m.ncomp > 1 ? m.wb[2] * m.hb[2] : 0;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < nb0 + nb1 + nb2; t += gridDim.x * blockDim.x) {
    const int c = t < nb0 ? 0 : (t < nb0 + nb1 ? 1 : 2);
    const int b = t - (c == 0 ? 0 : (c == 1 ? nb0 : nb0 + nb1));
    // (selected, not indexed: a dynamic index into the by-value struct sends it to scratch)
    const int wbc = c == 0 ? m.wb[0] : (c == 1 ? m.wb[1] : m.wb[2]);
    const int64_t coff = c == 0 ? m.coef_off[0] : (c == 1 ? m.coef_off[1] : m.coef_off[2]);
    const int64_t poff = c == 0 ? m.plane_off[0] : (c == 1 ? m.plane_off[1] : m.plane_off[2]);
    const int by = b / wbc, bx = b - by * wbc;
    const int4* src = (const int4*)(coef + coff + (size_t)b * 128);
    const uint16_t* q = quant + ((size_t)blockIdx.y * 3 + c) * 64;
    int v[64];
#pragma unroll
    for (int i = 0; i < 8; ++i) {  // a row of 8 coefficients = 16 bytes; times its row of the quantisation table
      const int4 w = src[i];
      const int4 qq = *(const int4*)(q + i * 8);
      const int ww[4] = {w.x, w.y, w.z, w.w}, qw[4] = {qq.x, qq.y, qq.z, qq.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[i * 8 + 2 * j] = (int)(int16_t)(ww[j] & 0xffff) * (int)(qw[j] & 0xffff);
        v[i * 8 + 2 * j + 1] = (ww[j] >> 16) * (int)((uint32_t)qw[j] >> 16);
      }
    }
#pragma unroll
    for (int cc = 0; cc < 8; ++cc)  // pass 1: columns
      IDCT_1D(v[0 + cc], v[8 + cc], v[16 + cc], v[24 + cc], v[32 + cc], v[40 + cc], v[48 + cc], v[56 + cc], 11,
              v[0 + cc], v[8 + cc], v[16 + cc], v[24 + cc], v[32 + cc], v[40 + cc], v[48 + cc], v[56 + cc]);
    const int stride = wbc * 8;
    uint8_t* out = planes + poff + ((size_t)by * 8) * stride + bx * 8;
#pragma unroll
    for (int r = 0; r < 8; ++r) {  // pass 2: rows
      int o0, o1, o2, o3, o4, o5, o6, o7;
      IDCT_1D(v[r * 8 + 0], v[r * 8 + 1], v[r * 8 + 2], v[r * 8 + 3], v[r * 8 + 4], v[r * 8 + 5], v[r * 8 + 6], v[r * 8 + 7], 18,
              o0, o1, o2, o3, o4, o5, o6, o7);
      uint2 pk;
      pk.x = (uint32_t)clamp255(o0 + 128) | ((uint32_t)clamp255(o1 + 128) << 8) | ((uint32_t)clamp255(o2 + 128) << 16) |
             ((uint32_t)clamp255(o3 + 128) << 24);
      pk.y = (uint32_t)clamp255(o4 + 128) | ((uint32_t)clamp255(o5 + 128) << 8) | ((uint32_t)clamp255(o6 + 128) << 16) |
             ((uint32_t)clamp255(o7 + 128) << 24);
      *(uint2*)(out + (size_t)r * stride) = pk;
    }
  }
}

// ---- upsampling + colour: eight consecutive pixels of one row per thread ------------------------------------------------
// The chroma samples the eight pixels need are columns 4k-1 .. 4k+4 of one (4:2:2) or two (4:2:0) chroma rows: one aligned
// dword + two edge bytes per row and plane, kept in registers; Y is one 8-byte load; the 24 output bytes leave as six dword
// stores (their address is only byte-aligned: rows of an image whose width is not a multiple of four; gfx950 global stores
// take that).  First form: one pixel per thread, ~14 byte loads and three byte stores per pixel, 685 us per 512 images —
// bound by the address processing of byte accesses; this form: 2 loads + 0.75 stores per pixel.
struct __attribute__((packed)) packed_u32 { uint32_t v; };

__device__ __forceinline__ void load6(const uint8_t* __restrict__ row, int c0, int stride, int (&v)[6]) {
  const uint32_t w = *(const uint32_t*)(row + c0);  // columns c0 .. c0+3 (c0 is a multiple of four, rows are 8-byte aligned)
  v[1] = w & 255; v[2] = (w >> 8) & 255; v[3] = (w >> 16) & 255; v[4] = w >> 24;
  v[0] = c0 > 0 ? row[c0 - 1] : 0;
  v[5] = c0 + 4 < stride ? row[c0 + 4] : 0;
}

// jdsample.c's triangle filters on the cached columns: sample at full-resolution x from cur = column x >> 1
__device__ __forceinline__ int fancy_h2v2(const int (&a)[6], const int (&b)[6], int x, int c0, int dw) {
  const int c = x >> 1, i = c - c0 + 1;
  const int cur = a[i] * 3 + b[i];
  if (x & 1) return c == dw - 1 ? (cur * 4 + 7) >> 4 : (cur * 3 + (a[i + 1] * 3 + b[i + 1]) + 7) >> 4;
  return c == 0 ? (cur * 4 + 8) >> 4 : (cur * 3 + (a[i - 1] * 3 + b[i - 1]) + 8) >> 4;
}
__device__ __forceinline__ int fancy_h2v1(const int (&a)[6], int x, int c0, int dw) {
  const int c = x >> 1, i = c - c0 + 1;
  if (x & 1) return c == dw - 1 ? a[i] : (a[i] * 3 + a[i + 1] + 2) >> 2;
  return c == 0 ? a[i] : (a[i] * 3 + a[i - 1] + 1) >> 2;
}

__global__ __launch_bounds__(256) void jpeg_colour_kernel(const JpegImageDev* __restrict__ meta, const uint8_t* __restrict__ planes,
                                                          uint8_t* __restrict__ rgb) {
  const JpegImageDev m = meta[blockIdx.y];
  const int G = (m.width + 7) >> 3, total = G * m.height;
  const uint8_t *py = planes + m.plane_off[0], *pb = planes + m.plane_off[1], *pr = planes + m.plane_off[2];
  const int s0 = m.wb[0] * 8, s1 = m.wb[1] * 8, s2 = m.wb[2] * 8;
  const int dw = (m.width + 1) >> 1, dh = (m.height + 1) >> 1;
  uint8_t* out = rgb + m.rgb_off;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
    const int y = t / G, k = t - y * G, x0 = k * 8;
    const uint2 yw = *(const uint2*)(py + (size_t)y * s0 + x0);
    int cbv[8], crv[8];
    if (m.ncomp == 3) {
      if (m.H == 1) {
        const uint2 bw = *(const uint2*)(pb + (size_t)y * s1 + x0), rw = *(const uint2*)(pr + (size_t)y * s2 + x0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          cbv[j] = ((j < 4 ? bw.x : bw.y) >> (8 * (j & 3))) & 255;
          crv[j] = ((j < 4 ? rw.x : rw.y) >> (8 * (j & 3))) & 255;
        }
      } else if (dw <= 2) {  // jdsample.c jinit_upsampler: the fancy filters need downsampled_width > 2; else plain replication
        const int r = m.V == 2 ? y >> 1 : y;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int c = min((x0 + j) >> 1, dw - 1);
          cbv[j] = pb[(size_t)r * s1 + c];
          crv[j] = pr[(size_t)r * s2 + c];
        }
      } else {
        const int c0 = x0 >> 1;
        const int r = m.V == 2 ? y >> 1 : y;
        int ab[6], ar[6];
        load6(pb + (size_t)r * s1, c0, s1, ab);
        load6(pr + (size_t)r * s2, c0, s2, ar);
        if (m.V == 2) {
          const int rn = (y & 1) ? min(r + 1, dh - 1) : max(r - 1, 0);  // jdmainct.c: the edge row is its own context row
          int bb[6], br[6];
          load6(pb + (size_t)rn * s1, c0, s1, bb);
          load6(pr + (size_t)rn * s2, c0, s2, br);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            cbv[j] = fancy_h2v2(ab, bb, x0 + j, c0, dw);
            crv[j] = fancy_h2v2(ar, br, x0 + j, c0, dw);
          }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            cbv[j] = fancy_h2v1(ab, x0 + j, c0, dw);
            crv[j] = fancy_h2v1(ar, x0 + j, c0, dw);
          }
        }
      }
    }
    uint32_t px[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int yy = ((j < 4 ? yw.x : yw.y) >> (8 * (j & 3))) & 255;
      int r = yy, g = yy, b = yy;
      if (m.ncomp == 3) {
        const int xb = cbv[j] - 128, xr = crv[j] - 128;
        r = clamp255(yy + ((91881 * xr + 32768) >> 16));
        b = clamp255(yy + ((116130 * xb + 32768) >> 16));
        g = clamp255(yy + ((-22554 * xb + 32768 - 46802 * xr) >> 16));
      }
      px[j] = (uint32_t)r | ((uint32_t)g << 8) | ((uint32_t)b << 16);
    }
    uint32_t w[6];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      w[3 * q + 0] = px[4 * q] | (px[4 * q + 1] << 24);
      w[3 * q + 1] = (px[4 * q + 1] >> 8) | (px[4 * q + 2] << 16);
      w[3 * q + 2] = (px[4 * q + 2] >> 16) | (px[4 * q + 3] << 8);
    }
    uint8_t* o = out + ((size_t)y * m.width + x0) * 3;
    if (x0 + 8 <= m.width) {
#pragma unroll
      for (int q = 0; q < 6; ++q) ((packed_u32*)(o + 4 * q))->v = w[q];
    } else {  // the last group of a row whose width is not a multiple of eight
      for (int kb = 0; kb < (m.width - x0) * 3; ++kb) o[kb] = (uint8_t)(w[kb >> 2] >> (8 * (kb & 3)));
    }
  }
}

}  // namespace

hipError_t launch_jpeg_reconstruct(const JpegImageDev* meta_dev, const uint16_t* quant_dev, const void* coef_dev,
                                   uint8_t* planes_dev, uint8_t* rgb_dev, int n, int max_blocks, int max_pixels, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  const int gx1 = max(1, min((max_blocks + 255) / 256, 4096)), gx2 = max(1, min((max_pixels / 8 + 255) / 256, 4096));
  hipLaunchKernelGGL(jpeg_idct_kernel, dim3(gx1, n), dim3(256), 0, s, meta_dev, quant_dev, (const char*)coef_dev, planes_dev);
  hipLaunchKernelGGL(jpeg_colour_kernel, dim3(gx2, n), dim3(256), 0, s, meta_dev, (const uint8_t*)planes_dev, rgb_dev);
  return hipGetLastError();
}
