// jpeg_entropy.cpp — host half of the JPEG ingest (one step further up the reference's loader:
// `Image.open(path).convert("RGB")` inside torchvision's ImageFolder, utils/train_eval_util.py:96-146).
//
// A JPEG decoder is two different machines.  Entropy decoding (Huffman, one bit-serial stream per image, no parallelism
// inside a baseline scan) is a third of libjpeg's time and belongs on host threads; dequantisation + inverse DCT +
// chroma upsampling + colour conversion (two thirds, embarrassingly parallel integer arithmetic over 8x8 blocks and
// pixels) belongs on the device (jpeg.hip).  This file is the first machine: file -> markers -> Huffman -> quantised DCT
// coefficients (int16, natural order, whole MCUs) written straight into the caller's pinned upload buffer by native
// threads — no Pillow, no GIL, no worker processes, no host-side pixels at all.  The coefficients cross PCIe in place of
// the pixels (same size: 1.5 int16 per pixel for 4:2:0 against 3 bytes).
//
// Taken: 8-bit Huffman JPEGs — baseline / extended-sequential (SOF0, SOF1; one interleaved scan) and progressive (SOF2;
// spectral selection + successive approximation, any legal scan script) — grayscale or YCbCr with chroma 1x1 and luma
// 1x1 / 2x1 / 2x2 (4:4:4, 4:2:2, 4:2:0), restart intervals.  Everything else (arithmetic coding, 12-bit, CMYK / RGB-coded
// files, sequential multi-scan, other samplings) and every file that shows ANY irregularity (a scan that does not end exactly
// where a clean one ends, implausible coefficients, a progression with gaps) is reported as not taken and left to the
// caller's fallback decoder, so that libjpeg's warn-and-recover behaviour on such files stays libjpeg's.
// The third-party algorithm restated here is the JPEG standard's (ITU T.81 Annex F.2: DECODE, RECEIVE, EXTEND; F.1.2.1.1
// DC prediction; E.2 restart; Annex G progressive scans) — the bit-exact parts that depend on libjpeg's arithmetic are all on the device side.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/mcm.h"

namespace {

const uint8_t ZIGZAG[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                            41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                            30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {
  bool set = false;
  uint8_t bits[17] = {0};
  uint8_t vals[256] = {0};
  // T.81 F.2.2.3 tables + a 9-bit lookahead
  int32_t maxcode[18];
  int32_t valoff[17];
  uint16_t look[512];  // (length << 8) | symbol, 0 = longer than 9 bits
  // AC tables: the next FAST bits -> a whole (run, value) when code + value bits fit in them:
  // (value << 8) | (run << 4) | total bits; 0 = take the two-step path
  static constexpr int FAST = 10;  // (12: slower — the tables stop fitting beside the data in L1)
  int16_t fast_ac[1 << FAST];
  void build_fast_ac() {
    for (int i = 0; i < (1 << FAST); ++i) {
      fast_ac[i] = 0;
      const uint16_t lk = look[i >> (FAST - 9)];
      if (!lk) continue;
      const int len = lk >> 8, rs = lk & 255, run = rs >> 4, sz = rs & 15;
      if (sz == 0 || len + sz > FAST) continue;
      int v = ((i << len) & ((1 << FAST) - 1)) >> (FAST - sz);
      if (v < (1 << (sz - 1))) v += 1 - (1 << sz);  // EXTEND
      if (v >= -128 && v <= 127) fast_ac[i] = (int16_t)((v * 256) + (run * 16) + (len + sz));
    }
  }
  bool build() {
    int code = 0, k = 0;
    memset(look, 0, sizeof look);
    for (int l = 1; l <= 16; ++l) {
      valoff[l] = k - code;
      for (int i = 0; i < bits[l]; ++i, ++k, ++code) {
        if (k >= 256 || code >= (1 << l)) return false;
        if (l <= 9) {
          const int lo = code << (9 - l), n = 1 << (9 - l);
          for (int j = 0; j < n; ++j) look[lo + j] = (uint16_t)((l << 8) | vals[k]);
        }
      }
      maxcode[l] = bits[l] ? code - 1 : -1;
      code <<= 1;
    }
    maxcode[17] = 0x7fffffff;
    return true;
  }
};

struct Parsed {
  std::vector<uint8_t> file;
  size_t scan = 0;  // first byte of the entropy-coded segment
  uint16_t q[4][64];
  bool qset[4] = {false, false, false, false};
  Huff dc[4], ac[4];
  int cq[3], cdc[3], cac[3];
  int cid[3] = {0, 0, 0};
  int restart = 0;
  bool progressive = false;
  size_t first_sos = 0;  // progressive: offset of the first SOS segment's length field
  void reset() {  // (records are re-used from call to call: their file buffers keep their capacity)
    scan = first_sos = 0;
    restart = 0;
    progressive = false;
    for (int t = 0; t < 4; ++t) qset[t] = dc[t].set = ac[t].set = false;
  }
};

inline int rd16(const uint8_t* p) { return (p[0] << 8) | p[1]; }
constexpr int SANE = 4095;  // |dequantised coefficient| of anything a JPEG encoder produces from 8-bit samples

bool parse_dqt(Parsed& p, const uint8_t* s, int sl) {
  int o = 0;
  while (o < sl) {
    const int pq = s[o] >> 4, tq = s[o] & 15;
    ++o;
    if (tq > 3 || pq > 1 || o + (pq ? 128 : 64) > sl) return false;
    for (int k = 0; k < 64; ++k) p.q[tq][ZIGZAG[k]] = pq ? (uint16_t)rd16(s + o + 2 * k) : s[o + k];
    p.qset[tq] = true;
    o += pq ? 128 : 64;
  }
  return true;
}

bool parse_dht(Parsed& p, const uint8_t* s, int sl) {
  int o = 0;
  while (o < sl) {
    if (o + 17 > sl) return false;
    const int tc = s[o] >> 4, th = s[o] & 15;
    if (tc > 1 || th > 3) return false;
    Huff& h = tc ? p.ac[th] : p.dc[th];
    int cnt = 0;
    h.bits[0] = 0;
    for (int l = 1; l <= 16; ++l) cnt += (h.bits[l] = s[o + l]);
    if (cnt > 256 || o + 17 + cnt > sl) return false;
    memcpy(h.vals, s + o + 17, (size_t)cnt);
    // libjpeg's jpeg_make_d_derived_tbl refuses a DC table with a symbol above 15 (JERR_BAD_HUFF_TABLE): such a file goes to
    // the Pillow fallback, which raises what the reference's loader raises (it used to be scored silently)
    if (!tc)
      for (int i = 0; i < cnt; ++i)
        if (h.vals[i] > 15) return false;
    if (!h.build()) return false;
    if (tc) h.build_fast_ac();
    h.set = true;
    o += 17 + cnt;
  }
  return true;
}

// markers up to and including SOS; fills m (status 0 / 1 / 2) and p
void parse(const char* path, Parsed& p, mcm_jpeg_image& m) {
  memset(&m, 0, sizeof m);
  m.status = 2;
  for (int c = 0; c < 3; ++c) m.coef_off[c] = -1;
  p.reset();
  FILE* f = fopen(path, "rb");
  if (!f) return;
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  if (n < 4) { fclose(f); return; }
  p.file.resize((size_t)n + 8);  // (+8: the bit reader may look a few bytes past the end)
  const size_t got = fread(p.file.data(), 1, (size_t)n, f);
  fclose(f);
  if (got != (size_t)n) return;
  memset(p.file.data() + n, 0, 8);
  const uint8_t* d = p.file.data();
  if (d[0] != 0xFF || d[1] != 0xD8) return;
  size_t pos = 2;
  bool sof = false;
  int cid[3] = {0, 0, 0};
  while (pos + 4 <= (size_t)n) {
    if (d[pos] != 0xFF) return;
    while (pos < (size_t)n && d[pos] == 0xFF) ++pos;  // fill bytes
    const int mk = d[pos++];
    if (mk == 0xD8 || (mk >= 0xD0 && mk <= 0xD7) || mk == 0x01) continue;
    if (pos + 2 > (size_t)n) return;
    const int len = rd16(d + pos);
    if (len < 2 || pos + len > (size_t)n) return;
    const uint8_t* s = d + pos + 2;
    const int sl = len - 2;
    if (mk == 0xDB) {
      if (!parse_dqt(p, s, sl)) return;
    } else if (mk == 0xC4) {
      if (!parse_dht(p, s, sl)) return;
    } else if (mk == 0xC0 || mk == 0xC1 || mk == 0xC2) {  // SOF0 / SOF1 / SOF2 (progressive)
      if (sl < 6 || sof) return;
      p.progressive = mk == 0xC2;
      const int prec = s[0];
      m.height = rd16(s + 1);
      m.width = rd16(s + 3);
      m.ncomp = s[5];
      if (prec != 8 || (m.ncomp != 1 && m.ncomp != 3) || m.width <= 0 || m.height <= 0) { m.status = 1; return; }
      if ((int64_t)m.width * m.height > ((int64_t)64 << 20)) { m.status = 1; return; }  // (left to the fallback's own size policy)
      if (sl < 6 + 3 * m.ncomp) return;
      for (int c = 0; c < m.ncomp; ++c) {
        cid[c] = p.cid[c] = s[6 + 3 * c];
        m.hs[c] = s[7 + 3 * c] >> 4;
        m.vs[c] = s[7 + 3 * c] & 15;
        p.cq[c] = s[8 + 3 * c];
        if (p.cq[c] > 3) return;
      }
      sof = true;
    } else if (mk >= 0xC3 && mk <= 0xCF && mk != 0xC4 && mk != 0xC8 && mk != 0xCC) {
      m.status = 1;  // lossless / arithmetic / hierarchical
      return;
    } else if (mk == 0xDD) {
      if (sl < 2) return;
      p.restart = rd16(s);
    } else if (mk == 0xEE) {  // Adobe: libjpeg reads the colour transform from it — anything but "YCbCr" goes to the fallback
      if (sl >= 12 && !memcmp(s, "Adobe", 5) && s[11] != 1) { m.status = 1; return; }
    } else if (mk == 0xDA) {  // SOS
      if (!sof) return;
      if (p.progressive) {
        for (int c = 0; c < m.ncomp; ++c)
          if (!p.qset[p.cq[c]]) { m.status = 1; return; }  // (quantisation tables are latched at the first scan here)
        p.first_sos = pos;
      } else {
      if (sl < 1 || s[0] != m.ncomp || sl < 1 + 2 * m.ncomp + 3) { m.status = 1; return; }  // not one interleaved scan
      for (int c = 0; c < m.ncomp; ++c) {
        if (s[1 + 2 * c] != cid[c]) { m.status = 1; return; }
        p.cdc[c] = s[2 + 2 * c] >> 4;
        p.cac[c] = s[2 + 2 * c] & 15;
        if (p.cdc[c] > 3 || p.cac[c] > 3 || !p.dc[p.cdc[c]].set || !p.ac[p.cac[c]].set || !p.qset[p.cq[c]]) return;
      }
      p.scan = pos + len;
      }
      // geometry this path takes
      if (m.ncomp == 1) {
        m.hs[0] = m.vs[0] = 1;
      } else {
        // (libjpeg decides YCbCr vs RGB from the markers and the component ids: ids 'R','G','B' mean RGB data)
        if (cid[0] == 'R' && cid[1] == 'G' && cid[2] == 'B') { m.status = 1; return; }
        const bool chroma11 = m.hs[1] == 1 && m.vs[1] == 1 && m.hs[2] == 1 && m.vs[2] == 1;
        const bool luma_ok = (m.hs[0] == 1 && m.vs[0] == 1) || (m.hs[0] == 2 && m.vs[0] == 1) || (m.hs[0] == 2 && m.vs[0] == 2);
        if (!chroma11 || !luma_ok) { m.status = 1; return; }
      }
      const int mw = 8 * m.hs[0], mh = 8 * m.vs[0];
      const int mx = (m.width + mw - 1) / mw, my = (m.height + mh - 1) / mh;
      for (int c = 0; c < m.ncomp; ++c) {
        m.wb[c] = mx * m.hs[c];
        m.hb[c] = my * m.vs[c];
      }
      m.status = 0;
      return;
    }
    pos += len;
  }
}

struct Bits {
  const uint8_t* p;
  const uint8_t* end;
  uint64_t acc = 0;  // the next bits, top-aligned
  int cnt = 0;       // valid bits in acc
  int pad = 0;       // ... of which this many, at the end, are zeros fed after a marker / the end of the data
  bool marker = false;  // a marker (or the end) was reached: zeros are fed from here on
  // after fill(): cnt > 32 — enough for one Huffman code (<= 16 bits) and its value bits (<= 15), so a symbol needs one check
  inline void fill() {
    while (cnt <= 32) {
      if (!marker && p + 4 <= end) {
        const uint32_t w = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
        const uint32_t v = ~w;
        if (!((v - 0x01010101u) & ~v & 0x80808080u)) {  // no 0xFF among the four bytes: no stuffing, no marker
          acc |= (uint64_t)w << (32 - cnt);
          cnt += 32;
          p += 4;
          continue;
        }
      }
      uint32_t b = 0;
      if (!marker && p < end) {
        b = *p;
        if (b == 0xFF) {
          if (p + 1 < end && p[1] == 0) {
            p += 2;
          } else {
            marker = true;
            b = 0;
          }
        } else {
          ++p;
        }
      } else {
        marker = true;
      }
      if (marker) pad += 8;
      acc |= (uint64_t)b << (56 - cnt);
      cnt += 8;
    }
  }
  // The segment just decoded ended cleanly: no fed zero was consumed, fewer than eight real bits (the padding of the last
  // byte) are left, and the marker `mk` follows at once.  Anything else — premature end, extraneous bytes, a different
  // marker — is libjpeg's warning-and-recovery territory, whose output this decoder does not reproduce: not taken.
  inline bool clean_end(int mk) const {
    if (cnt < pad || cnt - pad >= 8) return false;
    return p + 1 < end && p[0] == 0xFF && p[1] == mk;
  }
  inline uint32_t peek(int n) const { return (uint32_t)(acc >> (64 - n)); }
  inline void drop(int n) { acc <<= n; cnt -= n; }
};

// T.81 F.2.2.1 EXTEND, branch-free
inline int extend(int v, int s) { return v + (((v - (1 << (s - 1))) >> 31) & (1 - (1 << s))); }

// DECODE; the caller has made sure of 32 bits
inline int decode(Bits& b, const Huff& h) {
  const uint32_t lk = h.look[b.peek(9)];
  if (lk) {
    b.drop(lk >> 8);
    return lk & 255;
  }
  int code = (int)b.peek(10), l = 10;
  while (l <= 16 && code > h.maxcode[l]) {
    ++l;
    code = (int)b.peek(l);
  }
  if (l > 16) return -1;
  b.drop(l);
  return h.vals[(code + h.valoff[l]) & 255];
}

// one image's scan -> coefficient planes (every block of every plane is written: zeroed, then its non-zero coefficients)
bool entropy(const Parsed& p, const mcm_jpeg_image& m, uint8_t* dst) {
  Bits b;
  b.p = p.file.data() + p.scan;
  b.end = p.file.data() + p.file.size() - 8;
  int pred[3] = {0, 0, 0};
  const int mx = m.wb[0] / m.hs[0], my = m.hb[0] / m.vs[0];
  int16_t* plane[3];
  for (int c = 0; c < m.ncomp; ++c) plane[c] = (int16_t*)(dst + m.coef_off[c]);
  int until_restart = p.restart, next_rst = 0;
  // |v| <= lim[c][z] <=> |v * q| <= SANE; values of the one-look-up path are within +-128: unchecked where 128 q <= SANE
  int16_t lim[3][64];
  bool fast_unchecked[3];
  for (int c = 0; c < m.ncomp; ++c) {
    int qmax = 1;
    for (int z = 0; z < 64; ++z) {
      const int q = std::max<int>(1, p.q[p.cq[c]][z]);
      lim[c][z] = (int16_t)std::min(32767, SANE / q);
      qmax = std::max(qmax, q);
    }
    fast_unchecked[c] = 128 * qmax <= SANE;
  }
  for (int y = 0; y < my; ++y) {
    for (int x = 0; x < mx; ++x) {
      if (p.restart && until_restart == 0) {
        if (!b.clean_end(0xD0 + next_rst)) return false;  // the expected RSTn, byte-aligned, nothing in between
        b.p += 2;
        b.acc = 0;
        b.cnt = 0;
        b.pad = 0;
        b.marker = false;
        next_rst = (next_rst + 1) & 7;
        until_restart = p.restart;
        pred[0] = pred[1] = pred[2] = 0;
      }
      for (int c = 0; c < m.ncomp; ++c) {
        const Huff& hd = p.dc[p.cdc[c]];
        const Huff& ha = p.ac[p.cac[c]];
        const int16_t* lm = lim[c];
        const bool fchk = !fast_unchecked[c];
        for (int by = 0; by < m.vs[c]; ++by) {
          for (int bx = 0; bx < m.hs[c]; ++bx) {
            int16_t* blk = plane[c] + ((size_t)(y * m.vs[c] + by) * m.wb[c] + (x * m.hs[c] + bx)) * 64;
            memset(blk, 0, 128);  // (here, not as a pass over the whole plane: the block is written while it is in L1)
            if (b.cnt < 32) b.fill();
            int s = decode(b, hd);
            if (s < 0 || s > 11) return false;
            if (s) {
              pred[c] += extend((int)b.peek(s), s);
              b.drop(s);
            }
            // A dequantised coefficient of an 8-bit image stays below ~1200 (DC: 8 x 128 + rounding).  Beyond SANE the data
            // is damaged, and what libjpeg-turbo's 16-bit SIMD arithmetic makes of it is not what exact arithmetic
            // makes of it: not taken (the fallback decoder decides).
            if (pred[c] > lm[0] || pred[c] < -lm[0]) return false;
            blk[0] = (int16_t)pred[c];
            for (int k = 1; k < 64;) {
              if (b.cnt < 32) b.fill();
              const int fa = ha.fast_ac[b.peek(Huff::FAST)];
              if (fa) {  // code and value in one look-up
                k += (fa >> 4) & 15;
                if (k > 63) return false;
                b.drop(fa & 15);
                const int v = fa >> 8, z = ZIGZAG[k++];
                if (fchk && (v > lm[z] || v < -lm[z])) return false;
                blk[z] = (int16_t)v;
                continue;
              }
              const int rs = decode(b, ha);
              if (rs < 0) return false;
              const int r = rs >> 4;
              s = rs & 15;
              if (s) {
                k += r;
                if (k > 63) return false;
                const int v = extend((int)b.peek(s), s), z = ZIGZAG[k];
                if (v > lm[z] || v < -lm[z]) return false;
                blk[z] = (int16_t)v;
                b.drop(s);
                ++k;
              } else if (r == 15) {
                k += 16;
              } else {
                break;  // EOB
              }
            }
          }
        }
      }
      if (p.restart) --until_restart;
    }
  }
  return b.clean_end(0xD9);  // EOI right behind the last MCU
}

// ---- progressive (SOF2): T.81 Annex G.  Several scans refine the same coefficient planes — DC first / DC refinement over
// (possibly interleaved) MCUs, AC first / AC refinement over one component's own block raster with end-of-band runs; what
// comes out after the last scan is the coefficient array a sequential file would have carried, and libjpeg reconstructs it
// the same way (its inter-block smoothing only applies while coefficients are still incomplete: a file is taken only if
// every coefficient of every component reached its last bit, in a progression without gaps).
struct BitsP : Bits {
  inline int bit() {
    if (cnt < 1) fill();
    const int v = (int)(acc >> 63);
    drop(1);
    return v;
  }
  inline int bits(int n) {  // n <= 16
    if (cnt < n) fill();
    const int v = (int)peek(n);
    drop(n);
    return v;
  }
  // any marker but RSTn, right behind a cleanly ended scan
  inline bool clean_scan_end() const {
    if (cnt < pad || cnt - pad >= 8) return false;
    return p + 1 < end && p[0] == 0xFF && p[1] != 0 && !(p[1] >= 0xD0 && p[1] <= 0xD7);
  }
};

bool entropy_progressive(Parsed& p, const mcm_jpeg_image& m, uint8_t* dst) {
  const uint8_t* d = p.file.data();
  const size_t n = p.file.size() - 8;
  int16_t* plane[3];
  for (int c = 0; c < m.ncomp; ++c) {
    plane[c] = (int16_t*)(dst + m.coef_off[c]);
    memset(plane[c], 0, (size_t)m.wb[c] * m.hb[c] * 128);
  }
  // successive-approximation state per coefficient: -1 = not seen yet, else the Al of its last scan
  int8_t al_of[3][64];
  memset(al_of, -1, sizeof al_of);
  // which AC coefficients of a block are non-zero so far, bit k = zigzag position k: a refinement scan visits the non-zero
  // ones (a correction bit each) and counts the zero ones in between, instead of looking at all 63 positions of every block
  std::vector<uint64_t> nzbits[3];
  for (int c = 0; c < m.ncomp; ++c) nzbits[c].assign((size_t)m.wb[c] * m.hb[c], 0);
  const int hmax = m.hs[0], vmax = m.vs[0];  // (luma carries the largest factors in everything this path takes)
  size_t pos = p.first_sos;
  for (int nscan = 0; nscan < 1000; ++nscan) {
    if (pos + 2 > n) return false;
    const int len = rd16(d + pos);
    if (len < 8 || pos + len > n) return false;
    const uint8_t* s = d + pos + 2;
    const int ns = s[0];
    if (ns < 1 || ns > m.ncomp || len != 6 + 2 * ns) return false;
    int ci[3], td[3], ta[3];
    for (int i = 0; i < ns; ++i) {
      ci[i] = -1;
      for (int c = 0; c < m.ncomp; ++c)
        if (p.cid[c] == s[1 + 2 * i]) ci[i] = c;
      if (ci[i] < 0 || (i && ci[i] <= ci[i - 1])) return false;
      td[i] = s[2 + 2 * i] >> 4;
      ta[i] = s[2 + 2 * i] & 15;
      if (td[i] > 3 || ta[i] > 3) return false;
    }
    const int Ss = s[1 + 2 * ns], Se = s[2 + 2 * ns], Ah = s[3 + 2 * ns] >> 4, Al = s[3 + 2 * ns] & 15;
    if (Ss > Se || Se > 63 || Al > 13 || (Ss == 0 && Se != 0) || (Ss > 0 && ns != 1) || (Ah && Ah != Al + 1)) return false;
    for (int i = 0; i < ns; ++i) {
      for (int k = Ss; k <= Se; ++k) {  // a progression without gaps: first scans first, then one bit at a time
        int8_t& prev = al_of[ci[i]][k];
        if (Ah == 0 ? prev != -1 : prev != Ah) return false;
        prev = (int8_t)Al;
      }
      if (Ss == 0 ? (Ah == 0 && !p.dc[td[i]].set) : !p.ac[ta[i]].set) return false;
    }
    BitsP b;
    b.p = d + pos + len;
    b.end = d + n;
    int pred[3] = {0, 0, 0}, eobrun = 0, until_restart = p.restart, next_rst = 0;
    const int p1 = 1 << Al, m1 = -(1 << Al);
    auto restart_if_due = [&]() -> bool {
      if (!p.restart || until_restart) return true;
      if (!b.clean_end(0xD0 + next_rst)) return false;
      b.p += 2;
      b.acc = 0; b.cnt = 0; b.pad = 0; b.marker = false;
      next_rst = (next_rst + 1) & 7;
      until_restart = p.restart;
      pred[0] = pred[1] = pred[2] = 0;
      eobrun = 0;
      return true;
    };
    if (Ss == 0) {  // ---- DC scans
      const bool inter = ns > 1;
      const int c0 = ci[0];
      const int mx = inter ? m.wb[0] / m.hs[0] : (((m.width * m.hs[c0] + hmax - 1) / hmax) + 7) / 8;
      const int my = inter ? m.hb[0] / m.vs[0] : (((m.height * m.vs[c0] + vmax - 1) / vmax) + 7) / 8;
      for (int y = 0; y < my; ++y)
        for (int x = 0; x < mx; ++x) {
          if (!restart_if_due()) return false;
          for (int i = 0; i < ns; ++i) {
            const int c = ci[i], nh = inter ? m.hs[c] : 1, nv = inter ? m.vs[c] : 1;
            for (int by = 0; by < nv; ++by)
              for (int bx = 0; bx < nh; ++bx) {
                int16_t* blk = plane[c] + ((size_t)(y * nv + by) * m.wb[c] + (x * nh + bx)) * 64;
                if (Ah == 0) {
                  if (b.cnt < 32) b.fill();
                  const int sz = decode(b, p.dc[td[i]]);
                  if (sz < 0 || sz > 11) return false;
                  if (sz) {
                    pred[c] += extend((int)b.peek(sz), sz);
                    b.drop(sz);
                  }
                  const int v = pred[c] * p1;
                  if (v > 32767 || v < -32768) return false;
                  blk[0] = (int16_t)v;
                } else if (b.bit()) {
                  blk[0] |= (int16_t)p1;
                }
              }
          }
          if (p.restart) --until_restart;
        }
    } else {  // ---- AC scans: one component, its own block raster (not padded to whole MCUs)
      const int c = ci[0];
      const Huff& ha = p.ac[ta[0]];
      const int bw = (((m.width * m.hs[c] + hmax - 1) / hmax) + 7) / 8, bh = (((m.height * m.vs[c] + vmax - 1) / vmax) + 7) / 8;
      for (int y = 0; y < bh; ++y)
        for (int x = 0; x < bw; ++x) {
          if (!restart_if_due()) return false;
          int16_t* blk = plane[c] + ((size_t)y * m.wb[c] + x) * 64;
          uint64_t& nz = nzbits[c][(size_t)y * m.wb[c] + x];
          if (Ah == 0) {  // G.1.2.2: first scan of the band
            if (eobrun > 0) {
              --eobrun;
            } else {
              for (int k = Ss; k <= Se; ++k) {
                if (b.cnt < 32) b.fill();
                const int rs = decode(b, ha);
                if (rs < 0) return false;
                const int r = rs >> 4, sz = rs & 15;
                if (sz) {
                  k += r;
                  if (k > Se) return false;
                  const int v = extend((int)b.peek(sz), sz) * p1;
                  b.drop(sz);
                  if (v > 32767 || v < -32768) return false;
                  blk[ZIGZAG[k]] = (int16_t)v;
                  nz |= (uint64_t)1 << k;
                } else if (r == 15) {
                  k += 15;
                } else {
                  eobrun = 1 << r;
                  if (r) eobrun += b.bits(r);
                  --eobrun;
                  break;
                }
              }
            }
          } else {  // G.1.2.3: refinement — new +-1s between correction bits of the coefficients that are already non-zero
            const uint64_t band = (Se == 63 ? ~(uint64_t)0 : (((uint64_t)1 << (Se + 1)) - 1));
            auto correct = [&](int kk) {  // one correction bit for the non-zero coefficient at zigzag position kk
              int16_t& co = blk[ZIGZAG[kk]];
              if (b.bit() && (co & p1) == 0) co = (int16_t)(co >= 0 ? co + p1 : co + m1);
            };
            int k = Ss;
            if (eobrun == 0) {
              for (; k <= Se; ++k) {
                if (b.cnt < 32) b.fill();
                const int rs = decode(b, ha);
                if (rs < 0) return false;
                int r = rs >> 4, sz = rs & 15, val = 0;
                if (sz) {
                  if (sz != 1) return false;
                  val = b.bit() ? p1 : m1;
                } else if (r != 15) {
                  eobrun = 1 << r;
                  if (r) eobrun += b.bits(r);
                  break;
                }
                // over r still-zero coefficients (and every non-zero one on the way, with its correction bit) to the next zero
                for (;;) {
                  const uint64_t ahead = (nz & band) >> k;                       // non-zero coefficients at k, k+1, ...
                  const int gap = ahead ? __builtin_ctzll(ahead) : Se + 1 - k;   // zeros in front of the next one
                  if (gap > r) {
                    k += r;
                    break;
                  }
                  r -= gap;
                  k += gap;
                  if (k > Se) break;
                  correct(k);
                  ++k;
                  if (k > Se) break;
                }
                if (val) {
                  if (k > Se) return false;
                  blk[ZIGZAG[k]] = (int16_t)val;
                  nz |= (uint64_t)1 << k;
                }
              }
            }
            if (eobrun > 0) {  // the rest of the band: correction bits only
              if (k <= Se) {
                for (uint64_t rest = (nz & band) >> k << k; rest; rest &= rest - 1) correct(__builtin_ctzll(rest));
              }
              --eobrun;
            }
          }
          if (p.restart) --until_restart;
        }
    }
    if (!b.clean_scan_end()) return false;
    // the markers between this scan and the next (tables may change), or EOI
    pos = (size_t)(b.p - d);
    for (;;) {
      if (pos + 2 > n || d[pos] != 0xFF) return false;
      const int mk = d[pos + 1];
      pos += 2;
      if (mk == 0xD9) {  // EOI: complete?
        for (int c = 0; c < m.ncomp; ++c) {
          for (int k = 0; k < 64; ++k)
            if (al_of[c][k] != 0) return false;
          const uint16_t* qt = p.q[p.cq[c]];
          const int16_t* co = plane[c];
          for (size_t i = 0, e = (size_t)m.wb[c] * m.hb[c] * 64; i < e; ++i) {
            const int v = co[i] * (int)qt[i & 63];
            if (v > SANE || v < -SANE) return false;
          }
        }
        return true;
      }
      if (pos + 2 > n) return false;
      const int sl = rd16(d + pos);
      if (sl < 2 || pos + sl > n) return false;
      if (mk == 0xDA) break;  // the next scan: pos is at its length field
      if (mk == 0xC4) {
        if (!parse_dht(p, d + pos + 2, sl - 2)) return false;
      } else if (mk == 0xDD) {
        if (sl < 4) return false;
        p.restart = rd16(d + pos + 2);
      } else if (!((mk >= 0xE0 && mk <= 0xEF) || mk == 0xFE)) {
        return false;  // (DQT mid-image, DNL, anything unusual: the fallback's business)
      }
      pos += sl;
    }
  }
  return false;
}

}  // namespace

extern "C" int mcm_jpeg_entropy_decode(const char* const* paths, int32_t n, void* dst, int64_t dst_bytes,
                                       mcm_jpeg_image* meta, uint16_t* quant, int32_t threads, int64_t* bytes_used) {
  if (!paths || !meta || !quant || !bytes_used || n < 0 || threads < 1 || (!dst && dst_bytes > 0)) return MCM_EINVAL;
  // per-image parse records (tables, file bytes: 30 KB + the file each), kept between calls — their allocation and first
  // touch would otherwise be a serial millisecond or two in front of every batch.  One cached set per process; a call that
  // finds it in use (two pipes decoding at once) works on records of its own.
  static std::mutex cache_mu;
  static std::vector<Parsed> cache;
  std::unique_lock<std::mutex> lk(cache_mu, std::try_to_lock);
  std::vector<Parsed> own;
  std::vector<Parsed>& parsed = lk.owns_lock() ? cache : own;
  if (parsed.size() < (size_t)n) parsed.resize((size_t)n);
  const int nt = std::max(1, std::min<int>(threads, n));
  auto parallel = [&](auto&& body) {
    std::atomic<int> next{0};
    auto run = [&] {
      for (int i; (i = next.fetch_add(1)) < n;) body(i);
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(run);
    run();
    for (auto& th : pool) th.join();
  };
  parallel([&](int i) { parse(paths[i], parsed[(size_t)i], meta[i]); });
  int64_t off = 0;
  for (int32_t i = 0; i < n; ++i) {
    if (meta[i].status) continue;
    for (int c = 0; c < meta[i].ncomp; ++c) {
      meta[i].coef_off[c] = off;
      off += (int64_t)meta[i].wb[c] * meta[i].hb[c] * 128;
      memcpy(quant + ((size_t)i * 3 + c) * 64, parsed[(size_t)i].q[parsed[(size_t)i].cq[c]], 128);
    }
  }
  *bytes_used = off;
  if (off > dst_bytes) return MCM_ERANGE;
  parallel([&](int i) {
    mcm_jpeg_image& m = meta[i];
    if (m.status) return;
    Parsed& pp = parsed[(size_t)i];
    if (!(pp.progressive ? entropy_progressive(pp, m, (uint8_t*)dst) : entropy(pp, m, (uint8_t*)dst))) m.status = 2;
  });
  return MCM_OK;
}
