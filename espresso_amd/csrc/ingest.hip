// Waveform ingestion for the raw-audio training path (host code, no kernels): PCM WAV and FLAC files decoded to 16-bit mono samples
// straight into a caller-provided (pinned) staging buffer, many files in parallel, without the Python interpreter lock.
//
// Replaces, for the `wave` entries of the data json: fairseq/data/audio/audio_utils.py:74-118 get_waveform (soundfile.read of a
// WAV / FLAC file, always_2d, mono = channel 0 as espresso/tools/utils.py:438-440 takes it, normalization=False -> int16 scale) as
// called per utterance by espresso/data/feat_text_dataset.py:128-155 from `dataset.num_workers` DataLoader worker processes.
// The GPU front-end (fbank.hip) consumes the int16 samples directly; nothing is converted or padded on the host.
//
// FLAC: a self-contained decoder of the subset libFLAC's encoder produces for 8 - 24 bit streams (RFC 9639: CONSTANT / VERBATIM /
// FIXED / LPC subframes, partitioned Rice residuals with escape partitions, wasted bits, independent / left-side / side-right /
// mid-side stereo, frame CRC-8 / CRC-16 verified, optional MD5 check of the decoded audio against STREAMINFO).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <thread>
#include <vector>

#include "espresso_amd.h"

namespace {

struct Bytes {
  std::vector<uint8_t> d;
};
static bool read_file(const char* path, Bytes& b) {
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  if (fseek(f, 0, SEEK_END) != 0) { fclose(f); return false; }
  const long n = ftell(f);
  if (n < 0) { fclose(f); return false; }
  rewind(f);
  b.d.resize((size_t)n);
  const size_t got = n > 0 ? fread(b.d.data(), 1, (size_t)n, f) : 0;
  fclose(f);
  return got == (size_t)n;
}
static bool read_head(const char* path, uint8_t* buf, size_t want, size_t* got) {
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  *got = fread(buf, 1, want, f);
  fclose(f);
  return true;
}
static inline uint32_t le32(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
static inline uint16_t le16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static inline int16_t clamp16(long v) { return (int16_t)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }

struct Info {
  long frames = 0;  // samples per channel
  int rate = 0, channels = 0, bits = 0;
};

// ---- PCM WAV ---------------------------------------------------------------------------------------------------------------
struct WavFmt {
  int format = 0, channels = 0, rate = 0, bits = 0, block = 0;
  size_t data_off = 0, data_len = 0;
};
// walks the RIFF chunks inside the first `n` bytes; `file_len` bounds the data chunk when the buffer holds only the head
static int wav_parse(const uint8_t* p, size_t n, size_t file_len, WavFmt& w) {
  if (n < 12 || memcmp(p, "RIFF", 4) != 0 || memcmp(p + 8, "WAVE", 4) != 0) return -3;
  size_t off = 12;
  bool have_fmt = false;
  while (off + 8 <= n) {
    const uint32_t len = le32(p + off + 4);
    if (memcmp(p + off, "fmt ", 4) == 0) {
      if (off + 8 + 16 > n) return -3;
      const uint8_t* q = p + off + 8;
      w.format = le16(q); w.channels = le16(q + 2); w.rate = (int)le32(q + 4); w.block = le16(q + 12); w.bits = le16(q + 14);
      if (w.format == 0xFFFE) {  // WAVE_FORMAT_EXTENSIBLE: the sub-format GUID's first word — only when the chunk really holds it
        if (len < 26 || off + 8 + 26 > n) return -3;
        w.format = le16(q + 24);
      }
      have_fmt = true;
    } else if (memcmp(p + off, "data", 4) == 0) {
      if (!have_fmt) return -3;
      w.data_off = off + 8;
      size_t avail = file_len > w.data_off ? file_len - w.data_off : 0;
      w.data_len = len == 0xFFFFFFFFu || len > avail ? avail : len;  // (streamed files carry a dummy length)
      return 0;
    }
    off += 8 + (size_t)len + (len & 1);
  }
  return -3;
}
static int wav_info(const WavFmt& w, Info& in) {
  if (w.format != 1 || w.channels < 1 || (w.bits != 8 && w.bits != 16 && w.bits != 24 && w.bits != 32)) return -4;  // PCM integers only
  const int bps = w.bits / 8 * w.channels;
  in.frames = (long)(w.data_len / (size_t)bps);
  in.rate = w.rate; in.channels = w.channels; in.bits = w.bits;
  return 0;
}
static long wav_decode(const Bytes& b, int16_t* dst, long cap, Info& in) {
  WavFmt w;
  int rc = wav_parse(b.d.data(), b.d.size(), b.d.size(), w);
  if (rc == 0) rc = wav_info(w, in);
  if (rc) return rc;
  if (!dst || in.frames == 0) return in.frames;  // (header-only use: ea_audio_verify)
  if (in.frames > cap) return -5;
  const uint8_t* p = b.d.data() + w.data_off;
  const int step = w.bits / 8 * w.channels;
  if (w.bits == 16 && w.channels == 1) {
    memcpy(dst, p, (size_t)in.frames * 2);  // (little-endian host)
    return in.frames;
  }
  for (long i = 0; i < in.frames; ++i, p += step) {
    switch (w.bits) {
      case 8: dst[i] = (int16_t)(((int)p[0] - 128) * 256); break;
      case 16: dst[i] = (int16_t)le16(p); break;
      case 24: dst[i] = (int16_t)(p[1] | (p[2] << 8)); break;  // top 16 bits
      default: dst[i] = (int16_t)(p[2] | (p[3] << 8)); break;
    }
  }
  return in.frames;
}

// ---- FLAC -------------------------------------------------------------------------------------------------------------------
// MSB-first bit reader over a byte buffer with a 64-bit window (the Rice residuals are ~90 % of a FLAC stream: one refill per
// 7 bytes, the unary part by count-leading-zeros on the window)
struct BitReader {
  const uint8_t* p;
  size_t n, next = 0;   // next byte to load into the window
  uint64_t cache = 0;   // upcoming bits, left-aligned
  int cbits = 0;        // valid bits in the window
  bool ok = true;
  BitReader(const uint8_t* p_, size_t n_) : p(p_), n(n_) {}
  inline size_t pos() const { return next * 8 - (size_t)cbits; }  // bits consumed so far
  inline void refill() {
    if (next + 8 <= n) {  // one unaligned big-endian load tops the window up to 57 .. 64 bits
      uint64_t w;
      memcpy(&w, p + next, 8);
      w = __builtin_bswap64(w);
      const int take = (64 - cbits) >> 3;  // whole bytes that fit
      if (take > 0) {
        cache |= cbits ? (take == 8 ? w : (w >> (64 - 8 * take))) << (64 - cbits - 8 * take) : w;
        next += (size_t)take;
        cbits += 8 * take;
      }
      return;
    }
    while (cbits <= 56 && next < n) {
      cache |= (uint64_t)p[next++] << (56 - cbits);
      cbits += 8;
    }
  }
  inline uint64_t bits(int k) {  // k <= 57
    if (k == 0) return 0;
    if (cbits < k) {
      refill();
      if (cbits < k) { ok = false; cache = 0; cbits = 0; return 0; }
    }
    const uint64_t v = cache >> (64 - k);
    cache <<= k;
    cbits -= k;
    return v;
  }
  inline uint32_t bit() { return (uint32_t)bits(1); }
  inline int64_t sbits(int k) {
    if (k == 0) return 0;
    const uint64_t v = bits(k);
    const uint64_t m = 1ULL << (k - 1);
    return (int64_t)((v ^ m) - m);
  }
  inline uint32_t unary() {  // zeros before the next 1 (the 1 is consumed)
    uint32_t q = 0;
    for (;;) {
      if (cbits == 0) {
        refill();
        if (cbits == 0) { ok = false; return q; }
      }
      if (cache) {
        const int lz = __builtin_clzll(cache);
        if (lz < cbits) {
          q += (uint32_t)lz;
          cache = lz == 63 ? 0 : cache << (lz + 1);
          cbits -= lz + 1;
          return q;
        }
      }
      q += (uint32_t)cbits;  // only zeros in the valid part of the window
      cache = 0;
      cbits = 0;
    }
  }
  inline void align() {
    const int drop = cbits & 7;
    cache <<= drop;
    cbits -= drop;
  }
};
static uint8_t crc8(const uint8_t* p, size_t n) {
  uint8_t c = 0;
  for (size_t i = 0; i < n; ++i) {
    c ^= p[i];
    for (int b = 0; b < 8; ++b) c = (uint8_t)((c & 0x80) ? (c << 1) ^ 0x07 : (c << 1));
  }
  return c;
}
struct Crc16Table {  // tab[256 k + i]: the CRC of byte i followed by k zero bytes
  uint16_t tab[2048];
  Crc16Table() {
    for (int i = 0; i < 256; ++i) {
      uint16_t c = (uint16_t)(i << 8);
      for (int b = 0; b < 8; ++b) c = (uint16_t)((c & 0x8000) ? (c << 1) ^ 0x8005 : (c << 1));
      tab[i] = c;
    }
    for (int k = 1; k < 8; ++k)
      for (int i = 0; i < 256; ++i) {
        const uint16_t v = tab[256 * (k - 1) + i];
        tab[256 * k + i] = (uint16_t)((v << 8) ^ tab[v >> 8]);
      }
  }
};
static uint16_t crc16(const uint8_t* p, size_t n) {
  static const Crc16Table T;  // (function-local static: built once, thread-safe by C++11 — the batch reader's threads all start here)
  const uint16_t* tab = T.tab;
  uint16_t c = 0;
  size_t i = 0;
  for (; i + 8 <= n; i += 8) {  // slicing by eight: the table lookups of a step do not depend on each other
    const uint16_t t = (uint16_t)(c ^ (uint16_t)((p[i] << 8) | p[i + 1]));
    c = (uint16_t)(tab[1792 + (t >> 8)] ^ tab[1536 + (t & 0xff)] ^ tab[1280 + p[i + 2]] ^ tab[1024 + p[i + 3]] ^ tab[768 + p[i + 4]] ^
                   tab[512 + p[i + 5]] ^ tab[256 + p[i + 6]] ^ tab[p[i + 7]]);
  }
  for (; i < n; ++i) c = (uint16_t)((c << 8) ^ tab[(c >> 8) ^ p[i]]);
  return c;
}

struct Md5 {
  uint32_t a = 0x67452301, b = 0xefcdab89, c = 0x98badcfe, d = 0x10325476;
  uint64_t len = 0;
  uint8_t buf[64];
  size_t fill = 0;
  static inline uint32_t rol(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }
  void block(const uint8_t* p) {
    static const uint32_t K[64] = {
        0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af, 0xffff5bb1,
        0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453,
        0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a, 0xfffa3942,
        0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70, 0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05,
        0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665, 0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d,
        0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
    static const int S[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9,  14, 20, 5, 9,  14, 20, 5, 9,  14, 20, 5, 9,  14, 20,
                              4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
    uint32_t m[16];
    for (int i = 0; i < 16; ++i) m[i] = le32(p + 4 * i);
    uint32_t A = a, B = b, C = c, D = d;
    for (int i = 0; i < 64; ++i) {
      uint32_t f;
      int g;
      if (i < 16) { f = (B & C) | (~B & D); g = i; }
      else if (i < 32) { f = (D & B) | (~D & C); g = (5 * i + 1) & 15; }
      else if (i < 48) { f = B ^ C ^ D; g = (3 * i + 5) & 15; }
      else { f = C ^ (B | ~D); g = (7 * i) & 15; }
      const uint32_t t = D;
      D = C; C = B;
      B = B + rol(A + f + K[i] + m[g], S[i]);
      A = t;
    }
    a += A; b += B; c += C; d += D;
  }
  void update(const uint8_t* p, size_t n) {
    len += n;
    while (n) {
      const size_t k = n < 64 - fill ? n : 64 - fill;
      memcpy(buf + fill, p, k);
      fill += k; p += k; n -= k;
      if (fill == 64) { block(buf); fill = 0; }
    }
  }
  void final(uint8_t out[16]) {
    const uint64_t bits = len * 8;
    const uint8_t one = 0x80, zero = 0;
    update(&one, 1);
    while (fill != 56) update(&zero, 1);
    uint8_t l[8];
    for (int i = 0; i < 8; ++i) l[i] = (uint8_t)(bits >> (8 * i));
    update(l, 8);
    const uint32_t v[4] = {a, b, c, d};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) out[4 * i + j] = (uint8_t)(v[i] >> (8 * j));
  }
};

struct FlacStream {
  Info info;
  int min_block = 0, max_block = 0;
  uint8_t md5[16];
  size_t first_frame = 0;
};
static int flac_header(const uint8_t* p, size_t n, FlacStream& s) {
  if (n < 4 + 4 + 34 || memcmp(p, "fLaC", 4) != 0) return -3;
  size_t off = 4;
  bool have = false;
  for (;;) {
    if (off + 4 > n) return have ? 1 : -3;  // (1: the metadata continues past this buffer — enough for a probe)
    const int last = p[off] >> 7, type = p[off] & 0x7f;
    const size_t len = ((size_t)p[off + 1] << 16) | ((size_t)p[off + 2] << 8) | p[off + 3];
    off += 4;
    if (type == 0) {
      if (len < 34 || off + 34 > n) return -3;
      const uint8_t* q = p + off;
      s.min_block = (q[0] << 8) | q[1];
      s.max_block = (q[2] << 8) | q[3];
      s.info.rate = (q[10] << 12) | (q[11] << 4) | (q[12] >> 4);
      s.info.channels = ((q[12] >> 1) & 7) + 1;
      s.info.bits = (((q[12] & 1) << 4) | (q[13] >> 4)) + 1;
      s.info.frames = (long)((((uint64_t)(q[13] & 15)) << 32) | ((uint64_t)q[14] << 24) | (q[15] << 16) | (q[16] << 8) | q[17]);
      memcpy(s.md5, q + 18, 16);
      have = true;
    }
    off += len;
    if (last) break;
  }
  if (!have) return -3;
  s.first_frame = off;
  return 0;
}

// residual of one subframe into out[order ..]; returns false on a malformed stream
static bool flac_residual(BitReader& br, int32_t* out, int blocksize, int order) {
  const int method = (int)br.bits(2);
  if (method > 1) return false;
  const int pbits = method == 0 ? 4 : 5, esc = method == 0 ? 15 : 31;
  const int po = (int)br.bits(4);
  const int parts = 1 << po;
  if ((blocksize >> po) << po != blocksize && po > 0) return false;
  int i = order;
  for (int part = 0; part < parts; ++part) {
    int cnt = (blocksize >> po) - (part == 0 ? order : 0);
    if (cnt < 0) return false;
    const int k = (int)br.bits(pbits);
    if (k == esc) {
      const int nb = (int)br.bits(5);
      for (; cnt > 0; --cnt) out[i++] = (int32_t)br.sbits(nb);
    } else {
      // the window lives in locals for the partition: `out` is an int32_t*, the reader's counters are ints — through the struct
      // every store to out[] would force them back to memory
      uint64_t cache = br.cache;
      int cbits = br.cbits;
      size_t next = br.next;
      const uint8_t* const bp = br.p;
      const size_t bn = br.n;
      int32_t* o = out + i;
      int32_t* const oe = o + cnt;
      while (o < oe) {
        if (cbits < 48 && next + 8 <= bn) {  // top the window up to 57 .. 64 bits with one unaligned big-endian load
          uint64_t w;
          memcpy(&w, bp + next, 8);
          w = __builtin_bswap64(w);
          const int take = (64 - cbits) >> 3;
          cache |= cbits ? (w >> (64 - 8 * take)) << (64 - cbits - 8 * take) : w;
          next += (size_t)take;
          cbits += 8 * take;
        }
        const int lz = cache ? __builtin_clzll(cache) : 64;
        uint32_t q, r;
        if (k > 0 && lz + 1 + k <= cbits && lz < 32) {  // the whole code word (unary part, stop bit, k binary bits) sits in the window
          q = (uint32_t)lz;
          const uint64_t rest = cache << (lz + 1);
          r = (uint32_t)(rest >> (64 - k));
          cache = rest << k;
          cbits -= lz + 1 + k;
        } else {  // k = 0, a long code word or the tail of the buffer: the general reader
          br.cache = cache; br.cbits = cbits; br.next = next;
          q = br.unary();
          r = (uint32_t)br.bits(k);
          cache = br.cache; cbits = br.cbits; next = br.next;
          if (!br.ok) break;
        }
        const uint32_t u = (q << k) | r;
        *o++ = (int32_t)(u >> 1) ^ -(int32_t)(u & 1);
      }
      br.cache = cache; br.cbits = cbits; br.next = next;
      i += cnt;
    }
    if (!br.ok) return false;
  }
  return i == blocksize;
}
// out[i] += (sum_j coef[j] * out[i - 1 - j]) >> shift for i >= order, with the order known at compile time (the loop over j unrolls,
// the products overlap; only the last one waits for out[i - 1]).  Wrapping 64-bit sums: a corrupt stream must not reach signed
// overflow; valid streams never wrap.
template <int ORDER>
static void lpc_restore_n(int32_t* out, int blocksize, const int32_t* coef, int shift) {
  int64_t c[ORDER];
  for (int j = 0; j < ORDER; ++j) c[j] = coef[j];
  for (int i = ORDER; i < blocksize; ++i) {
    uint64_t acc = 0;
#pragma unroll
    for (int j = ORDER - 1; j >= 0; --j) acc += (uint64_t)(c[j] * out[i - 1 - j]);  // (oldest sample first: the newest arrives last)
    out[i] = (int32_t)(uint32_t)((uint64_t)(int64_t)out[i] + (uint64_t)((int64_t)acc >> shift));
  }
}
static void lpc_restore(int32_t* out, int blocksize, int order, const int32_t* coef, int shift) {
  switch (order) {
#define EA_LPC(N) case N: lpc_restore_n<N>(out, blocksize, coef, shift); return;
    EA_LPC(1) EA_LPC(2) EA_LPC(3) EA_LPC(4) EA_LPC(5) EA_LPC(6) EA_LPC(7) EA_LPC(8) EA_LPC(9) EA_LPC(10) EA_LPC(11) EA_LPC(12)
    EA_LPC(13) EA_LPC(14) EA_LPC(15) EA_LPC(16) EA_LPC(17) EA_LPC(18) EA_LPC(19) EA_LPC(20) EA_LPC(21) EA_LPC(22) EA_LPC(23) EA_LPC(24)
    EA_LPC(25) EA_LPC(26) EA_LPC(27) EA_LPC(28) EA_LPC(29) EA_LPC(30) EA_LPC(31) EA_LPC(32)
#undef EA_LPC
    default: return;
  }
}
static bool flac_subframe(BitReader& br, int32_t* out, int blocksize, int bps) {
  if (br.bit()) return false;  // padding bit
  const int type = (int)br.bits(6);
  int wasted = 0;
  if (br.bit()) wasted = (int)br.unary() + 1;
  if (wasted > 31 || wasted >= bps) return false;  // (a shift by the type width is undefined; a subframe keeps at least one bit)
  bps -= wasted;
  if (bps < 1 || bps > 33) return false;
  if (type == 0) {
    const int32_t v = (int32_t)br.sbits(bps);
    for (int i = 0; i < blocksize; ++i) out[i] = v;
  } else if (type == 1) {
    for (int i = 0; i < blocksize; ++i) out[i] = (int32_t)br.sbits(bps);
  } else if (type >= 8 && type <= 12) {
    const int order = type - 8;
    if (order > blocksize) return false;
    for (int i = 0; i < order; ++i) out[i] = (int32_t)br.sbits(bps);
    if (!flac_residual(br, out, blocksize, order)) return false;
    switch (order) {  // (64-bit intermediates: 24-bit side channels overflow 32 bits at order 4)
      case 1: for (int i = 1; i < blocksize; ++i) out[i] = (int32_t)((int64_t)out[i] + out[i - 1]); break;
      case 2: for (int i = 2; i < blocksize; ++i) out[i] = (int32_t)((int64_t)out[i] + 2LL * out[i - 1] - out[i - 2]); break;
      case 3: for (int i = 3; i < blocksize; ++i) out[i] = (int32_t)((int64_t)out[i] + 3LL * out[i - 1] - 3LL * out[i - 2] + out[i - 3]); break;
      case 4:
        for (int i = 4; i < blocksize; ++i)
          out[i] = (int32_t)((int64_t)out[i] + 4LL * out[i - 1] - 6LL * out[i - 2] + 4LL * out[i - 3] - out[i - 4]);
        break;
      default: break;
    }
  } else if (type >= 32) {
    const int order = (type & 31) + 1;
    if (order > blocksize) return false;
    for (int i = 0; i < order; ++i) out[i] = (int32_t)br.sbits(bps);
    const int prec = (int)br.bits(4) + 1;
    if (prec == 16) return false;
    const int shift = (int)br.sbits(5);
    if (shift < 0) return false;
    int32_t coef[32];
    for (int j = 0; j < order; ++j) coef[j] = (int32_t)br.sbits(prec);
    if (!flac_residual(br, out, blocksize, order)) return false;
    lpc_restore(out, blocksize, order, coef, shift);
  } else {
    return false;  // reserved subframe type
  }
  if (wasted) for (int i = 0; i < blocksize; ++i) out[i] = (int32_t)((uint32_t)out[i] << wasted);
  return br.ok;
}

// Decodes the whole stream.  dst (may be NULL): channel 0 as int16 (bits > 16: top 16 bits, bits < 16: scaled up), at most `cap`
// samples.  md5_ok (may be NULL): all channels hashed as libFLAC does and compared with STREAMINFO (all-zero signature = absent -> 1).
static long flac_decode(const Bytes& b, int16_t* dst, long cap, Info& in, int* md5_ok) {
  FlacStream s;
  int rc = flac_header(b.d.data(), b.d.size(), s);
  if (rc != 0) return -3;
  in = s.info;
  if (in.bits < 4 || in.bits > 32 || in.channels < 1 || in.channels > 8) return -4;
  if (dst && in.frames > cap) return -5;
  const uint8_t* p = b.d.data();
  const size_t n = b.d.size();
  size_t off = s.first_frame;
  std::vector<int32_t> ch((size_t)in.channels * 65536);
  std::vector<uint8_t> pcm;
  Md5 md5;
  long written = 0;
  const int bytes_ps = (in.bits + 7) / 8;
  while (off + 2 <= n) {
    if (p[off] != 0xFF || (p[off + 1] & 0xFE) != 0xF8) {  // lost sync (padding / trailing tag): stop at the declared length
      if (in.frames > 0 && written >= in.frames) break;
      ++off;
      continue;
    }
    BitReader br(p + off, n - off);
    br.bits(14);
    if (br.bit()) { ++off; continue; }
    br.bit();  // blocking strategy (only changes what the coded number counts)
    const int bs_code = (int)br.bits(4), sr_code = (int)br.bits(4), ca = (int)br.bits(4), ss_code = (int)br.bits(3);
    if (br.bit()) { ++off; continue; }
    {  // UTF-8-like coded frame / sample number
      const uint32_t first = (uint32_t)br.bits(8);
      int extra = 0;
      if (first >= 0xFE) extra = 6; else if (first >= 0xFC) extra = 5; else if (first >= 0xF8) extra = 4; else if (first >= 0xF0) extra = 3;
      else if (first >= 0xE0) extra = 2; else if (first >= 0xC0) extra = 1; else if (first >= 0x80) { ++off; continue; }
      for (int i = 0; i < extra; ++i) br.bits(8);
    }
    int blocksize;
    if (bs_code == 0) { ++off; continue; }
    else if (bs_code == 1) blocksize = 192;
    else if (bs_code <= 5) blocksize = 576 << (bs_code - 2);
    else if (bs_code == 6) blocksize = (int)br.bits(8) + 1;
    else if (bs_code == 7) blocksize = (int)br.bits(16) + 1;
    else blocksize = 256 << (bs_code - 8);
    if (sr_code == 12) br.bits(8);
    else if (sr_code == 13 || sr_code == 14) br.bits(16);
    else if (sr_code == 15) { ++off; continue; }
    int bps = in.bits;
    switch (ss_code) {
      case 0: break;
      case 1: bps = 8; break;
      case 2: bps = 12; break;
      case 4: bps = 16; break;
      case 5: bps = 20; break;
      case 6: bps = 24; break;
      case 7: bps = 32; break;
      default: bps = -1; break;
    }
    if (bps < 0 || !br.ok) { ++off; continue; }
    const bool bps_mismatch = bps != in.bits;  // (decided after the CRC-8: only a REAL frame header may fail the stream)
    const size_t hdr_bytes = br.pos() >> 3;
    const uint8_t want8 = (uint8_t)br.bits(8);
    if (!br.ok || crc8(p + off, hdr_bytes) != want8) { ++off; continue; }  // not a frame header after all
    const int nch = ca < 8 ? ca + 1 : 2;
    if (nch != in.channels || blocksize > 65536 || bps_mismatch) return -6;  // frames must agree with STREAMINFO (channels, sample size)
    for (int c = 0; c < nch; ++c) {
      const int side = (ca == 8 && c == 1) || (ca == 9 && c == 0) || (ca == 10 && c == 1);
      if (!flac_subframe(br, ch.data() + (size_t)c * 65536, blocksize, bps + side)) return -6;
    }
    br.align();
    const size_t body = br.pos() >> 3;
    const uint16_t want16 = (uint16_t)br.bits(16);
    if (!br.ok || crc16(p + off, body) != want16) return -7;
    int32_t* c0 = ch.data();
    int32_t* c1 = ch.data() + 65536;
    // (two's-complement wrapping arithmetic: exact for valid streams, defined for corrupt ones)
    if (ca == 8) { for (int i = 0; i < blocksize; ++i) c1[i] = (int32_t)((uint32_t)c0[i] - (uint32_t)c1[i]); }
    else if (ca == 9) { for (int i = 0; i < blocksize; ++i) c0[i] = (int32_t)((uint32_t)c0[i] + (uint32_t)c1[i]); }
    else if (ca == 10) {
      for (int i = 0; i < blocksize; ++i) {
        const int64_t side = c1[i];
        const int64_t mid = (int64_t)c0[i] * 2 + (side & 1);
        c0[i] = (int32_t)((mid + side) >> 1);
        c1[i] = (int32_t)((mid - side) >> 1);
      }
    }
    long take = blocksize;
    if (in.frames > 0 && written + take > in.frames) take = in.frames - written;
    if (dst) {
      if (written + take > cap) return -5;
      if (in.bits == 16) for (long i = 0; i < take; ++i) dst[written + i] = (int16_t)c0[i];
      else if (in.bits > 16) for (long i = 0; i < take; ++i) dst[written + i] = clamp16(c0[i] >> (in.bits - 16));
      else for (long i = 0; i < take; ++i) dst[written + i] = clamp16((long)c0[i] * (1L << (16 - in.bits)));
    }
    if (md5_ok) {
      pcm.resize((size_t)take * nch * bytes_ps);
      size_t k = 0;
      for (long i = 0; i < take; ++i)
        for (int c = 0; c < nch; ++c) {
          const int32_t v = ch[(size_t)c * 65536 + i];
          for (int bb = 0; bb < bytes_ps; ++bb) pcm[k++] = (uint8_t)((uint32_t)v >> (8 * bb));
        }
      md5.update(pcm.data(), pcm.size());
    }
    written += take;
    off += body + 2;
    if (in.frames > 0 && written >= in.frames) break;
  }
  if (in.frames == 0) in.frames = written;  // (unknown length in STREAMINFO)
  if (written != in.frames) return -6;
  if (md5_ok) {
    uint8_t got[16];
    md5.final(got);
    bool zero = true;
    for (int i = 0; i < 16; ++i) zero = zero && s.md5[i] == 0;
    *md5_ok = zero || memcmp(got, s.md5, 16) == 0 ? 1 : 0;
  }
  return written;
}

static long decode_any(const char* path, int16_t* dst, long cap, Info& in, int* md5_ok) {
  Bytes b;
  if (!read_file(path, b)) return -1;
  if (b.d.size() >= 4 && memcmp(b.d.data(), "fLaC", 4) == 0) return flac_decode(b, dst, cap, in, md5_ok);
  if (md5_ok) *md5_ok = 1;
  return wav_decode(b, dst, cap, in);
}

}  // namespace

// Header only: sample count per channel, sample rate, channels, bits per sample.  0 on success; -1 unreadable, -3 not a WAV / FLAC
// file, -4 unsupported sample format.
extern "C" int ea_audio_probe(const char* path, long* num_samples, int* sample_rate, int* channels, int* bits) {
  uint8_t head[4096];
  size_t got = 0;
  if (!path || !read_head(path, head, sizeof(head), &got)) return -1;
  Info in;
  if (got >= 4 && memcmp(head, "fLaC", 4) == 0) {
    FlacStream s;
    const int rc = flac_header(head, got, s);
    if (rc < 0) return rc;
    in = s.info;
    if (in.frames == 0) {  // length not recorded: decode to count
      const long n = decode_any(path, nullptr, 0, in, nullptr);
      if (n < 0) return (int)n;
    }
  } else {
    FILE* f = fopen(path, "rb");
    if (!f) return -1;
    fseek(f, 0, SEEK_END);
    const long flen = ftell(f);
    fclose(f);
    WavFmt w;
    int rc = wav_parse(head, got, flen > 0 ? (size_t)flen : got, w);
    if (rc == 0) rc = wav_info(w, in);
    if (rc) return rc;
  }
  if (num_samples) *num_samples = in.frames;
  if (sample_rate) *sample_rate = in.rate;
  if (channels) *channels = in.channels;
  if (bits) *bits = in.bits;
  return 0;
}

// Channel 0 of the file as int16 into dst[0 .. capacity); returns the number of samples, or < 0: -1 unreadable, -3 not WAV / FLAC,
// -4 unsupported format, -5 capacity too small, -6 malformed FLAC stream, -7 FLAC frame checksum mismatch.
extern "C" long ea_audio_read_i16(const char* path, int16_t* dst, long capacity, int* sample_rate) {
  if (!path || !dst) return -1;
  Info in;
  const long n = decode_any(path, dst, capacity, in, nullptr);
  if (n >= 0 && sample_rate) *sample_rate = in.rate;
  return n;
}

// FLAC self-check: decode every channel and compare the MD5 of the audio with the signature in STREAMINFO (1 = equal or no
// signature recorded, 0 = different, < 0 = decode error).  WAV files return 1.
extern "C" int ea_audio_verify(const char* path) {
  Info in;
  int ok = 0;
  const long n = decode_any(path, nullptr, 0, in, &ok);
  return n < 0 ? (int)n : ok;
}

// n files decoded in parallel by `num_threads` host threads (dataset.num_workers), file i into dst[offsets[i] .. offsets[i+1]).
// lengths[i] receives the sample count (or the negative error of that file), sample_rates[i] its rate.  Returns the number of
// files that failed.  No Python objects are touched: the interpreter lock is released for the whole call (ctypes).
extern "C" int ea_audio_read_batch_i16(const char* const* paths, int n, int16_t* dst, const long* offsets, int num_threads, long* lengths,
                                       int* sample_rates) {
  if (n <= 0) return 0;
  if (!paths || !dst || !offsets || !lengths) return n;
  std::atomic<int> next{0}, failed{0};
  auto work = [&]() {
    for (;;) {
      const int i = next.fetch_add(1);
      if (i >= n) return;
      Info in;
      const long got = decode_any(paths[i], dst + offsets[i], offsets[i + 1] - offsets[i], in, nullptr);
      lengths[i] = got;
      if (sample_rates) sample_rates[i] = got >= 0 ? in.rate : 0;
      if (got < 0) failed.fetch_add(1);
    }
  };
  const int nt = num_threads < 1 ? 1 : (num_threads > n ? n : num_threads);
  if (nt == 1) {
    work();
  } else {
    std::vector<std::thread> th;
    th.reserve((size_t)nt - 1);
    for (int t = 1; t < nt; ++t) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
  }
  return failed.load();
}
