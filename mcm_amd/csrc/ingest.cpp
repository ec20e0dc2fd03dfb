// ingest.cpp — host side of the uint8 ingest: packing a batch of decoded images into ONE pinned
// buffer, so that the batch crosses PCIe as one asynchronous copy (mcm_amd/ingest.py::PackedImagePipe) and
// mcm_resize_crop_u8 reads the images through pointers into the packed device buffer.
//
// The reference's loader hands every batch to the GPU with a synchronous `.cuda()` of one collated fp32 tensor
// (utils/detection_util.py:222-223); here the images stay the decoder's own uint8 arrays of their own sizes, and what
// remains on the host is this memcpy: 512 images of ~0.65 MB = 334 MB per batch.  Python threads reach ~13 GB/s on it
// (numpy slice assignments, one image each): 25 ms per batch, which bounded the raw-image ingest at 18k img/s while the
// device idled a third of the time.  A pool of native threads copying byte ranges — the batch is cut by BYTES, not by
// images, so every thread moves the same amount — runs at the host's memory bandwidth.
#include <stdint.h>
#include <string.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include <algorithm>
#include <thread>
#include <vector>

#include "../../include/mcm.h"

namespace {
// memcpy whose stores bypass the cache: the pinned buffer is written once and next read by the DMA engine, so pulling its
// lines into the cache first (read-for-ownership) is a third of the memory traffic for nothing
void copy_streaming(uint8_t* d, const uint8_t* s, size_t n) {
#if defined(__SSE2__)
  if (n >= 4096) {
    const size_t head = (16 - ((uintptr_t)d & 15)) & 15;
    memcpy(d, s, head);
    d += head, s += head, n -= head;
    for (; n >= 64; d += 64, s += 64, n -= 64) {
      const __m128i a = _mm_loadu_si128((const __m128i*)s), b = _mm_loadu_si128((const __m128i*)(s + 16));
      const __m128i c = _mm_loadu_si128((const __m128i*)(s + 32)), e = _mm_loadu_si128((const __m128i*)(s + 48));
      _mm_stream_si128((__m128i*)d, a);
      _mm_stream_si128((__m128i*)(d + 16), b);
      _mm_stream_si128((__m128i*)(d + 32), c);
      _mm_stream_si128((__m128i*)(d + 48), e);
    }
    _mm_sfence();
  }
#endif
  memcpy(d, s, n);
}
}  // namespace

extern "C" int mcm_pack_u8(const uint8_t* const* srcs, const int64_t* sizes, const int64_t* offsets, int32_t n,
                           uint8_t* dst, int64_t dst_bytes, int32_t threads) {
  if (!srcs || !sizes || !offsets || !dst || n < 0 || threads < 1) return MCM_EINVAL;
  int64_t total = 0;
  for (int32_t i = 0; i < n; ++i) {
    if (!srcs[i] || sizes[i] < 0 || offsets[i] < 0 || offsets[i] + sizes[i] > dst_bytes) return MCM_ERANGE;
    total += sizes[i];
  }
  if (total == 0) return MCM_OK;
  const int nt = (int)std::min<int64_t>(threads, std::max<int64_t>(1, total / (1 << 20)));  // >= 1 MiB per thread
  // thread t copies bytes [t * total / nt, (t + 1) * total / nt) of the concatenation of the images
  auto work = [&](int t) {
    int64_t lo = total * t / nt, hi = total * (t + 1) / nt, pos = 0;
    for (int32_t i = 0; i < n && pos < hi; ++i) {
      const int64_t a = std::max(lo, pos), b = std::min(hi, pos + sizes[i]);
      if (a < b) copy_streaming(dst + offsets[i] + (a - pos), srcs[i] + (a - pos), (size_t)(b - a));
      pos += sizes[i];
    }
  };
  if (nt == 1) {
    work(0);
    return MCM_OK;
  }
  std::vector<std::thread> pool;
  pool.reserve((size_t)nt - 1);
  for (int t = 1; t < nt; ++t) pool.emplace_back(work, t);
  work(0);
  for (auto& th : pool) th.join();
  return MCM_OK;
}
