// preprocess.hip — Resize(S) + CenterCrop(S) of uint8 RGB images on the device.
//
// Replaces the first two steps of the reference's loader transform (reference
// utils/train_eval_util.py:27-33: transforms.Resize(224), transforms.CenterCrop(224) on the PIL
// image; ToTensor + Normalize are already folded into mcm_score_u8's patch gather).  Both steps
// are third-party code outside /root/reference:
//   torchvision  Resize(int): short side -> S, long side int(S*long/short), BILINEAR, image left
//                untouched when short == S; CenterCrop: top/left = round((n - S) / 2), half to even
//                (computed on the host in mcm_api.hip, see prep_geometry()).
//   Pillow       Image.resize(BILINEAR) = ImagingResample (src/libImaging/Resample.c): separable,
//                antialiased (filter support max(scale, 1)), horizontal pass first, each pass in
//                22-bit fixed point and rounded back to uint8.
// The coefficient set-up is double precision; it is evaluated here with the same operations in
// the same order and with contraction off, so the integer coefficients — and with them every output
// byte — equal Pillow's.  Both passes are fused per output pixel (the horizontal result of each
// contributing row is recomputed, rounded to uint8 exactly as the intermediate image would hold
// it), so there is no intermediate image and nothing is allocated.
// Two forms in one kernel, chosen per workgroup (8 output rows of one image), same integer arithmetic:
//   LDS form (round 4)   filters of at most 16 taps per axis (scale factors up to 7.5) whose working set fits the LDS budget.
//                        A prologue kernel computes each image's coefficient tables once (zero-padded to 8 / 16 taps); a
//                        workgroup stages the source window of its rows (of 8, 4, 2 or 1 of them at a time, whichever fits)
//                        into LDS with 16-byte loads, runs the horizontal pass once per staged row into an LDS copy of
//                        Pillow's intermediate image and the vertical pass over that.  Both passes read LDS in aligned
//                        dwords (v_alignbyte re-aligns a pixel run that starts at any byte), keep every tap's load
//                        independent (fixed trip count over the zero-padded taps: the loads are all in flight before the
//                        first multiply) and multiply with the 24-bit multiplier (a coefficient is at most 2^22, a pixel
//                        2^8; the full 32-bit multiply runs at a quarter of the rate);
//   fused form (rounds 2 - 3)  per output pixel the horizontal result of every contributing row is recomputed from global
//                        byte loads — any scale factor up to 31; bound by the texture addresser (≈ 110 byte loads per
//                        output pixel at scale 2.2) and by the latency of its dependent load → multiply chains.
#include "common.hpp"

namespace {

constexpr int PBITS = 22;   // Resample.c PRECISION_BITS for 8-bit channels
constexpr int KMAX = 64;    // taps per output coordinate: 2*ceil(scale)+1 -> scale factors up to 31
constexpr int ROWS = 8;     // output rows per workgroup
constexpr int FT = 16;      // LDS form: taps per output coordinate it holds
constexpr int LDS_FORM_BYTES = 56 * 1024;  // LDS form: staged source window + horizontal-pass rows (two workgroups per CU)

#pragma clang fp contract(off)
// Resample.c precompute_coeffs + normalize_coeffs_8bpc for output coordinate xx (bilinear)
__device__ int resample_coeffs(int in_size, int out_size, int xx, int32_t* kk, int* xmin_out) {
  const double scale = (double)in_size / (double)out_size;
  const double fs = scale < 1.0 ? 1.0 : scale;
  const double support = 1.0 * fs;
  const double center = (xx + 0.5) * scale;
  const double ss = 1.0 / fs;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > in_size) xmax = in_size;
  const int n = min(xmax - xmin, KMAX);
  double ww = 0.0;
  for (int x = 0; x < n; ++x) {
    double v = (x + xmin - center + 0.5) * ss;
    if (v < 0.0) v = -v;
    ww += v < 1.0 ? 1.0 - v : 0.0;
  }
  for (int x = 0; x < n; ++x) {
    double v = (x + xmin - center + 0.5) * ss;
    if (v < 0.0) v = -v;
    double w = v < 1.0 ? 1.0 - v : 0.0;
    if (ww != 0.0) w /= ww;
    kk[x] = w < 0 ? (int32_t)(-0.5 + w * (double)(1 << PBITS)) : (int32_t)(0.5 + w * (double)(1 << PBITS));
  }
  *xmin_out = xmin;
  return n;
}

__device__ __forceinline__ int clip8(int32_t v) {
  v >>= PBITS;
  // The empty asm keeps the shift and the clamp apart.  Fused, hipcc (ROCm 7.2) turns two neighbouring clip8()s into one
  // gfx950 v_ashr_pk_u8_i32 and ORs its result into a packed word as if bits 31:16 of the destination were zero; on the
  // MI355X they are not (the upper half of the first source comes through: a constant-100 image came out 100, 100, 100|0x20,
  // 100|0x19 per dword).  Found by the bit-exact tests the first time four results were packed into one store.
  asm volatile("" : "+v"(v));
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// ---- LDS form ----------------------------------------------------------------------------------------------------------
// Per image: int32 [2][S][FT] coefficients (axis 0 = x, 1 = y; zero beyond the taps in use) + int32 [2][S][2] = (first tap, taps)
__host__ __device__ constexpr size_t prep_coef_ints(int S) { return (size_t)2 * S * (FT + 2); }

__global__ __launch_bounds__(256) void resize_coeffs_kernel(const PrepImage* __restrict__ meta, int S, int32_t* __restrict__ coef) {
  const PrepImage im = meta[blockIdx.x];
  const bool rx = im.nw != im.W, ry = im.nh != im.H;
  const int taps_x = rx ? 2 * (int)ceil(fmax((double)im.W / (double)im.nw, 1.0)) + 1 : 1;
  const int taps_y = ry ? 2 * (int)ceil(fmax((double)im.H / (double)im.nh, 1.0)) + 1 : 1;
  if (taps_x > FT || taps_y > FT) return;  // this image takes the fused form
  int32_t* base = coef + (size_t)blockIdx.x * prep_coef_ints(S);
  for (int t = threadIdx.x; t < 2 * S; t += blockDim.x) {
    const int axis = t >= S, c = t - axis * S;
    int32_t* o = base + (size_t)t * FT;  // (taps <= FT: resample_coeffs writes at most FT entries)
    int mn, n;
    if (axis ? ry : rx) {
      n = axis ? resample_coeffs(im.H, im.nh, im.top + c, o, &mn) : resample_coeffs(im.W, im.nw, im.left + c, o, &mn);
    } else {  // Pillow skips a pass that keeps the size: one tap of weight 1 gives the same bytes
      n = 1;
      mn = (axis ? im.top : im.left) + c;
      o[0] = 1 << PBITS;
    }
    for (int x = n; x < FT; ++x) o[x] = 0;
    int32_t* lim = base + (size_t)2 * S * FT + (size_t)t * 2;
    lim[0] = mn;
    lim[1] = n;
  }
}

__device__ __forceinline__ uint32_t byte_of(uint32_t w, int i) { return (w >> (8 * i)) & 0xffu; }
// acc += pixel * coefficient on the 24-bit multiplier (pixel < 2^8, 0 <= coefficient <= 2^22: exact)
__device__ __forceinline__ int32_t mad24(uint32_t px, int32_t k, int32_t acc) { return (int32_t)__umul24(px, (uint32_t)k) + acc; }

// One workgroup's rows through LDS; false when not even single rows fit the budget.  TB = taps per coordinate (8 | 16).
template <int TB>
__device__ bool lds_form(const PrepImage& im, const int32_t* __restrict__ coef, int S, uint8_t* __restrict__ dst, char* smem) {
  const int y0 = blockIdx.y * ROWS, nrow = min(ROWS, S - y0), S3 = S * 3;
  int32_t* kxf = (int32_t*)smem;            // [S][TB]
  int32_t* kyf = kxf + (size_t)S * TB;      // [ROWS][TB]
  int* xmin_f = (int*)(kyf + ROWS * TB);    // [S]
  int* ymin_f = xmin_f + S;                 // [ROWS]
  int* ny_f = ymin_f + ROWS;                // [ROWS]
  // [rows][wstride], then [rows][S3].  16-byte aligned as it stands (S % 4 == 0); NOT re-aligned through an integer cast: that
  // loses the LDS address space, the accesses turn into flat ones, and hipcc merges the horizontal pass's byte stores into
  // 16-bit flat stores at odd addresses, which the LDS aperture does not honour (wrong bytes, measured)
  uint8_t* win = (uint8_t*)(ny_f + ROWS);
  const int32_t* gx = coef, *gy = coef + (size_t)S * FT;
  const int32_t* limx = coef + (size_t)2 * S * FT, *limy = limx + (size_t)2 * S;
  for (int i = threadIdx.x; i < S * (TB / 4); i += blockDim.x) {
    const int xx = i / (TB / 4), q = i - xx * (TB / 4);
    *(int4*)(kxf + (size_t)xx * TB + q * 4) = *(const int4*)(gx + (size_t)xx * FT + q * 4);
  }
  for (int i = threadIdx.x; i < nrow * TB; i += blockDim.x) {
    const int r = i / TB, q = i - r * TB;
    kyf[r * TB + q] = gy[(size_t)(y0 + r) * FT + q];
  }
  for (int xx = threadIdx.x; xx < S; xx += blockDim.x) xmin_f[xx] = limx[xx * 2];
  if (threadIdx.x < nrow) {
    ymin_f[threadIdx.x] = limy[(y0 + threadIdx.x) * 2];
    ny_f[threadIdx.x] = limy[(y0 + threadIdx.x) * 2 + 1];
  }
  // first tap / last tap + 1 are non-decreasing in the output coordinate: the columns needed are [first xmin, last xmax)
  const int xlo = limx[0], xhi = limx[(S - 1) * 2] + limx[(S - 1) * 2 + 1];
  const int wbytes = (xhi - xlo) * 3;
  const int wstride = ((wbytes + 15 + 15) / 16 + 1) * 16;  // a row sits at its global address mod 16
  __syncthreads();
  // rows per pass: the most of 8 / 4 / 2 / 1 whose source rows + horizontal results fit
  int rs = ROWS;
  for (; rs >= 1; rs >>= 1) {
    int worst = 0;
    for (int r0 = 0; r0 < nrow; r0 += rs) {
      const int r1 = min(r0 + rs, nrow) - 1;
      worst = max(worst, ymin_f[r1] + ny_f[r1] - ymin_f[r0]);
    }
    if ((size_t)worst * (wstride + S3) <= (size_t)LDS_FORM_BYTES) break;
  }
  if (rs < 1 || wbytes <= 0) return false;
  const uint8_t* img_end = im.src + (size_t)im.H * im.W * 3;
  const int cpr = wstride / 16;
  constexpr int RD = TB * 3 / 4;  // dwords of one output's pixel run, re-aligned; RD + 1 raw ones cover any byte offset
  for (int r0 = 0; r0 < nrow; r0 += rs) {
    const int r1 = min(r0 + rs, nrow);
    const int ylo = ymin_f[r0], nrows = ymin_f[r1 - 1] + ny_f[r1 - 1] - ylo;
    uint8_t* hrow = win + (size_t)nrows * wstride;
    if (r0) __syncthreads();  // the previous pass is done with win / hrow
    for (int i = threadIdx.x; i < nrows * cpr; i += blockDim.x) {
      const int r = i / cpr, c = i - r * cpr;
      const uint8_t* g = im.src + ((size_t)(ylo + r) * im.W + xlo) * 3;
      const uint8_t* ga = (const uint8_t*)((uintptr_t)g & ~(uintptr_t)15) + (size_t)c * 16;
      uint4 v;
      if (ga >= im.src && ga + 16 <= img_end) {
        v = *(const uint4*)ga;
      } else {  // the 16-byte chunk sticks out of the image: byte by byte, zero outside (never under a non-zero tap)
        uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int t = 0; t < 16; ++t)
          if (ga + t >= im.src && ga + t < img_end) w[t >> 2] |= (uint32_t)ga[t] << (8 * (t & 3));
        v = make_uint4(w[0], w[1], w[2], w[3]);
      }
      *(uint4*)(win + (size_t)r * wstride + c * 16) = v;
    }
    __syncthreads();
    // horizontal pass: Pillow's intermediate image, rows ylo .. ylo + nrows, the S columns of the crop
    for (int i = threadIdx.x; i < nrows * S; i += blockDim.x) {
      const int r = i / S, xx = i - r * S;
      const uint32_t galign = (uint32_t)((uintptr_t)(im.src + ((size_t)(ylo + r) * im.W + xlo) * 3) & 15);
      const uint32_t p = (uint32_t)r * wstride + galign + (uint32_t)(xmin_f[xx] - xlo) * 3;
      const uint32_t* wp = (const uint32_t*)(win + (p & ~3u));
      const uint32_t sh = p & 3u;
      uint32_t raw[RD + 1], px[RD];
      int32_t k[TB];
#pragma unroll
      for (int q = 0; q <= RD; ++q) raw[q] = wp[q];
#pragma unroll
      for (int q = 0; q < TB / 4; ++q) *(int4*)(k + q * 4) = *(const int4*)(kxf + (size_t)xx * TB + q * 4);
#pragma unroll
      for (int q = 0; q < RD; ++q) px[q] = __builtin_amdgcn_alignbyte(raw[q + 1], raw[q], sh);
      int32_t a0 = 1 << (PBITS - 1), a1 = a0, a2 = a0;
#pragma unroll
      for (int t = 0; t < TB; ++t) {
        a0 = mad24(byte_of(px[(t * 3 + 0) >> 2], (t * 3 + 0) & 3), k[t], a0);
        a1 = mad24(byte_of(px[(t * 3 + 1) >> 2], (t * 3 + 1) & 3), k[t], a1);
        a2 = mad24(byte_of(px[(t * 3 + 2) >> 2], (t * 3 + 2) & 3), k[t], a2);
      }
      uint8_t* o = hrow + (size_t)r * S3 + xx * 3;
      o[0] = (uint8_t)clip8(a0); o[1] = (uint8_t)clip8(a1); o[2] = (uint8_t)clip8(a2);
    }
    __syncthreads();
    // vertical pass, four bytes of an output row per item (the filter is the same for every byte of a row)
    const int q4 = S3 / 4;
    for (int i = threadIdx.x; i < (r1 - r0) * q4; i += blockDim.x) {
      const int rr = i / q4, j = i - rr * q4, r = r0 + rr;
      const int ny = ny_f[r];
      const uint8_t* h = hrow + (size_t)(ymin_f[r] - ylo) * S3 + j * 4;
      uint32_t d[TB];
      int32_t k[TB];
#pragma unroll
      for (int y = 0; y < TB; ++y) d[y] = *(const uint32_t*)(h + (size_t)min(y, ny - 1) * S3);  // (zero taps re-read the last row)
#pragma unroll
      for (int q = 0; q < TB / 4; ++q) *(int4*)(k + q * 4) = *(const int4*)(kyf + r * TB + q * 4);
      int32_t v0 = 1 << (PBITS - 1), v1 = v0, v2 = v0, v3 = v0;
#pragma unroll
      for (int y = 0; y < TB; ++y) {
        v0 = mad24(byte_of(d[y], 0), k[y], v0);
        v1 = mad24(byte_of(d[y], 1), k[y], v1);
        v2 = mad24(byte_of(d[y], 2), k[y], v2);
        v3 = mad24(byte_of(d[y], 3), k[y], v3);
      }
      *(uint32_t*)(dst + ((size_t)blockIdx.x * S + y0 + r) * S3 + j * 4) =
          (uint32_t)clip8(v0) | ((uint32_t)clip8(v1) << 8) | ((uint32_t)clip8(v2) << 16) | ((uint32_t)clip8(v3) << 24);
    }
  }
  return true;
}

__global__ __launch_bounds__(256) void resize_crop_kernel(const PrepImage* __restrict__ meta, const int32_t* __restrict__ coef,
                                                          int S, uint8_t* __restrict__ dst, int fused_only) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int32_t* kx = (int32_t*)smem;                 // [S][KMAX]
  int32_t* ky = kx + (size_t)S * KMAX;          // [ROWS][KMAX]
  int* xmin_s = (int*)(ky + ROWS * KMAX);       // [S]
  int* nx_s = xmin_s + S;                       // [S]
  int* ymin_s = nx_s + S;                       // [ROWS]
  int* ny_s = ymin_s + ROWS;                    // [ROWS]
  const PrepImage im = meta[blockIdx.x];
  const int y0 = blockIdx.y * ROWS;
  const bool rx = im.nw != im.W, ry = im.nh != im.H;  // Pillow skips a pass that keeps the size
  // ---- LDS form -----------------------------------------------------------------------------------------------------
  if (coef && !fused_only && (S & 3) == 0 && (rx || ry)) {  // (no resample at all: the fused form is a plain crop copy)
    const int taps_x = rx ? 2 * (int)ceil(fmax((double)im.W / (double)im.nw, 1.0)) + 1 : 1;
    const int taps_y = ry ? 2 * (int)ceil(fmax((double)im.H / (double)im.nh, 1.0)) + 1 : 1;
    if (taps_x <= FT && taps_y <= FT) {
      const bool done = (taps_x <= 8 && taps_y <= 8) ? lds_form<8>(im, coef + (size_t)blockIdx.x * prep_coef_ints(S), S, dst, smem)
                                                     : lds_form<16>(im, coef + (size_t)blockIdx.x * prep_coef_ints(S), S, dst, smem);
      if (done) return;
      __syncthreads();  // no sub-block fits the budget: the fused form below re-uses the LDS
    }
  }
  // ---- fused form ---------------------------------------------------------------------------------------------------
  for (int xx = threadIdx.x; xx < S; xx += blockDim.x) {
    if (rx) {
      nx_s[xx] = resample_coeffs(im.W, im.nw, im.left + xx, kx + (size_t)xx * KMAX, &xmin_s[xx]);
    } else {
      nx_s[xx] = 1;
      xmin_s[xx] = im.left + xx;
      kx[(size_t)xx * KMAX] = 1 << PBITS;
    }
  }
  if (threadIdx.x < ROWS && y0 + threadIdx.x < S) {
    const int r = threadIdx.x;
    if (ry) {
      ny_s[r] = resample_coeffs(im.H, im.nh, im.top + y0 + r, ky + r * KMAX, &ymin_s[r]);
    } else {
      ny_s[r] = 1;
      ymin_s[r] = im.top + y0 + r;
      ky[r * KMAX] = 1 << PBITS;
    }
  }
  __syncthreads();
  const int nrow = min(ROWS, S - y0);
  for (int i = threadIdx.x; i < nrow * S; i += blockDim.x) {
    const int r = i / S, xx = i - r * S;
    const int xmin = xmin_s[xx], nx = nx_s[xx], ymin = ymin_s[r], ny = ny_s[r];
    const int32_t* kxx = kx + (size_t)xx * KMAX;
    const int32_t* kyy = ky + r * KMAX;
    int32_t v0 = 1 << (PBITS - 1), v1 = v0, v2 = v0;
    for (int y = 0; y < ny; ++y) {
      const uint8_t* row = im.src + ((size_t)(ymin + y) * im.W + xmin) * 3;
      int h0, h1, h2;
      if (rx) {
        int32_t a0 = 1 << (PBITS - 1), a1 = a0, a2 = a0;
        for (int x = 0; x < nx; ++x) {
          const int32_t k = kxx[x];
          a0 += (int32_t)row[x * 3 + 0] * k;
          a1 += (int32_t)row[x * 3 + 1] * k;
          a2 += (int32_t)row[x * 3 + 2] * k;
        }
        h0 = clip8(a0); h1 = clip8(a1); h2 = clip8(a2);
      } else {
        h0 = row[0]; h1 = row[1]; h2 = row[2];
      }
      const int32_t k = kyy[y];
      v0 += h0 * k; v1 += h1 * k; v2 += h2 * k;
    }
    uint8_t* o = dst + (((size_t)blockIdx.x * S + y0 + r) * S + xx) * 3;
    if (ry) {
      o[0] = (uint8_t)clip8(v0); o[1] = (uint8_t)clip8(v1); o[2] = (uint8_t)clip8(v2);
    } else {  // single tap of weight 1: the horizontal result as is
      o[0] = (uint8_t)((v0 - (1 << (PBITS - 1))) >> PBITS);
      o[1] = (uint8_t)((v1 - (1 << (PBITS - 1))) >> PBITS);
      o[2] = (uint8_t)((v2 - (1 << (PBITS - 1))) >> PBITS);
    }
  }
}

}  // namespace

int prep_max_taps() { return KMAX; }

size_t prep_coef_bytes(int max_batch, int S) { return (size_t)max_batch * prep_coef_ints(S) * sizeof(int32_t); }

hipError_t launch_resize_crop(const PrepImage* meta_dev, int32_t* coef_dev, int B, int S, uint8_t* dst, hipStream_t s,
                              bool fused_only) {
  if (B <= 0 || S <= 0) return hipErrorInvalidValue;
  const int lds_fused = (S * KMAX + ROWS * KMAX) * 4 + (2 * S + 2 * ROWS) * 4;
  const int lds_form = (S * FT + ROWS * FT) * 4 + (S + 2 * ROWS) * 4 + 16 + LDS_FORM_BYTES + 64;
  const int lds = lds_fused > lds_form ? lds_fused : lds_form;
  static PerDeviceFlag attr_set;
  if (!attr_set.get()) {
    hipError_t e = hipFuncSetAttribute((const void*)resize_crop_kernel,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) return e;
    attr_set.set();
  }
  if (lds > 150 * 1024) return hipErrorInvalidValue;
  const bool lds_ok = coef_dev && !fused_only && (S & 3) == 0;
  if (lds_ok) hipLaunchKernelGGL(resize_coeffs_kernel, dim3(B), dim3(256), 0, s, meta_dev, S, coef_dev);
  hipLaunchKernelGGL(resize_crop_kernel, dim3(B, (S + ROWS - 1) / ROWS), dim3(256), lds, s, meta_dev,
                     lds_ok ? (const int32_t*)coef_dev : (const int32_t*)nullptr, S, dst, fused_only ? 1 : 0);
  return hipGetLastError();
}
