/*
 * Host-side packing of reads to 2 bits per base: what a batch of reads held one
 * base per byte (fmitest.c:150-153, search_reads.c:102-111) is turned into on
 * its way to the device, a quarter of the bytes.  Read r becomes (m + 15) / 16
 * words, base k at bits 2 (k % 16) of word k / 16, zero above the last base.
 * Called from several host threads at once on disjoint ranges of reads.
 */
#include <stdint.h>
#include <string.h>

#include <immintrin.h>

/* eight bases, one per byte (low two bits), to sixteen bits: base k at bits 2k */
static inline uint32_t pack8(uint64_t x) {
  x &= 0x0303030303030303ull;
  x = (x | (x >> 6)) & 0x000F000F000F000Full;
  x = (x | (x >> 12)) & 0x000000FF000000FFull;
  return (uint32_t)((x | (x >> 24)) & 0xFFFFull);
}

static inline uint32_t pack16_plain(const uint8_t *p) {
  uint64_t lo, hi;
  memcpy(&lo, p, 8);
  memcpy(&hi, p + 8, 8);
  return pack8(lo) | (pack8(hi) << 16);
}

/* The same with SSSE3: pairs of bases by a multiply-add of bytes (b0 + 4 b1),
 * pairs of pairs by a multiply-add of words (+ 16 times the next), then the
 * four result bytes gathered from the four dwords. */
__attribute__((target("ssse3"))) static inline uint32_t pack16_ssse3(const uint8_t *p) {
  const __m128i x = _mm_and_si128(_mm_loadu_si128((const __m128i *)p), _mm_set1_epi8(3));
  const __m128i t = _mm_maddubs_epi16(x, _mm_set1_epi16(0x0401));
  const __m128i u = _mm_madd_epi16(t, _mm_set1_epi32(0x00100001));
  const __m128i g = _mm_shuffle_epi8(u, _mm_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1));
  return (uint32_t)_mm_cvtsi128_si32(g);
}

/* Reads [b, e) of the batch at pats (read r at pats + r * stride) -> out + r * words.
 * A word is packed from the 16 bytes at its first base; for the last word of a
 * read those run past the read's end, into the next read, and are masked off.
 * Only where they would run past `limit`, the end of the whole buffer, are the
 * bytes copied out first. */
#define PACK_RANGE_BODY(PACK16)                                                        \
  const uint32_t words = (m + 15) / 16, tail = m & 15u;                                \
  const uint32_t tail_mask = tail ? (1u << (2 * tail)) - 1u : ~0u;                     \
  for (uint64_t r = b; r < e; ++r) {                                                   \
    const uint8_t *src = pats + r * stride;                                            \
    uint32_t *o = out + r * words;                                                     \
    for (uint32_t w = 0; w < words; ++w) {                                             \
      const uint8_t *p = src + 16 * w;                                                 \
      uint8_t tmp[16];                                                                 \
      if (p + 16 > limit) {                                                            \
        memset(tmp, 0, sizeof tmp);                                                    \
        memcpy(tmp, p, (size_t)(limit - p));                                           \
        p = tmp;                                                                       \
      }                                                                                \
      const uint32_t v = PACK16(p);                                                    \
      o[w] = w + 1 == words ? v & tail_mask : v;                                       \
    }                                                                                  \
  }

__attribute__((target("ssse3"))) static void pack_range_ssse3(const uint8_t *pats, uint32_t m, uint64_t stride,
                                                              uint64_t b, uint64_t e, const uint8_t *limit,
                                                              uint32_t *out) {
  PACK_RANGE_BODY(pack16_ssse3)
}

static void pack_range_plain(const uint8_t *pats, uint32_t m, uint64_t stride, uint64_t b, uint64_t e,
                             const uint8_t *limit, uint32_t *out) {
  PACK_RANGE_BODY(pack16_plain)
}

void bwt_pack_read_range(const uint8_t *pats, uint32_t m, uint64_t stride, uint64_t b, uint64_t e,
                         const uint8_t *limit, uint32_t *out) {
  if (__builtin_cpu_supports("ssse3")) pack_range_ssse3(pats, m, stride, b, e, limit, out);
  else pack_range_plain(pats, m, stride, b, e, limit, out);
}
