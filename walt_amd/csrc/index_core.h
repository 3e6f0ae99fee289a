// index_core.h -- per-element builders of the derived HBM index structures
// (Ent keys, BAD bitmap, base-3 directory) and of the packed read records.
// Pure inline functions shared by the HIP kernels (index_dev.hip, pack.hip) and
// the g++ CPU unit-test harness (tests/host_harness.cpp).
#ifndef WALT_AMD_INDEX_CORE_H_
#define WALT_AMD_INDEX_CORE_H_

#include "core.h"

namespace walt {

// getHashValue (util.hpp:175-182) of the genome at pos, from the packed genome.
// Caller guarantees pos + 34 < genome_len (indexed positions satisfy
// pos < chrom_end - 36, reference.cpp:202-203).
WALT_HD uint32_t hash_at(const uint32_t* g2, uint64_t pos) {
  uint32_t h = 0;
  for (uint32_t i = 0; i < kKeyWeight; ++i) h = (h << 2) | g2_code(g2, pos + care_pos(i));
  return h;
}

// Ent for index slot value `pos`: key = genome chars at care positions 12..43.
// touches_end is set when one of them lies at or beyond genome_len (that char is
// stored as 0 and the bucket must be marked BAD).
WALT_HD Ent make_ent(const uint32_t* g2, uint32_t genome_len, uint32_t pos, bool& touches_end) {
  uint64_t key = 0;
  touches_end = false;
  for (uint32_t p = kKeyWeight; p < kKeyWeight + kKeyChars; ++p) {
    uint64_t q = (uint64_t)pos + care_pos(p);
    uint32_t c = 0;
    if (q < genome_len) c = g2_code(g2, q); else touches_end = true;
    key = (key << 2) | c;
  }
  Ent e;
  e.key_hi = (uint32_t)(key >> 32);
  e.key_lo = (uint32_t)key;
  e.pos = pos;
  return e;
}

// Directory entry for base-3 key K over 12+D digits: lower bound, inside the
// 4^12 bucket of its first 12 digits, of the D-digit sub-prefix among the Ent
// keys.  (For BAD buckets the value is only ever used as the end of the
// preceding bucket; sub-prefix 0 gives the bucket start there.)
WALT_HD uint32_t dir_entry(const uint32_t* cnt, const Ent* ent, uint32_t D, uint32_t ga, uint32_t K) {
  uint32_t sub = K % pow3(D);
  uint32_t top = K / pow3(D);
  // 12 base-3 digits -> base-4 bucket
  uint32_t h = 0;
  uint32_t div = pow3(kKeyWeight - 1);
  for (uint32_t i = 0; i < kKeyWeight; ++i) {
    uint32_t dgt = top / div;
    top -= dgt * div;
    div /= 3;
    h = (h << 2) | code_of_digit3(dgt, ga);
  }
  uint32_t lo = cnt[h], hi = cnt[h + 1];
  if (D == 0 || sub == 0) return lo;
  uint64_t T = 0;
  uint32_t sdiv = pow3(D - 1);
  for (uint32_t i = 0; i < D; ++i) {
    uint32_t dgt = sub / sdiv;
    sub -= dgt * sdiv;
    sdiv /= 3;
    T = (T << 2) | code_of_digit3(dgt, ga);
  }
  T <<= (64 - 2 * D);
  uint64_t M = key_mask(D);
  while (lo < hi) {
    uint32_t mid = lo + ((hi - lo) >> 1);
    if ((ent_key(ent[mid]) & M) < T) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// ---------------------------------------------------------------------------
// Packed read record.  `bases` are the sanitised read characters as the
// loader leaves them (mapping.cpp:101-103).  Returns false on a non-ACGT char.
// out has packed_fields(nw) words with stride `stride` (SoA over reads).
// ---------------------------------------------------------------------------
WALT_HD bool pack_read(const uint8_t* bases, uint32_t len, uint32_t ga, uint32_t D, uint32_t nw,
                       uint32_t* out, uint64_t stride) {
  bool ok = true;
  out[0] = len;
  for (uint32_t w = 0; w < nw; ++w) {
    uint32_t v = 0;
    for (uint32_t k = 0; k < 16; ++k) {
      uint32_t i = 16 * w + k;
      if (i < len) {
        uint32_t c = base_code(bases[i]);
        if (c > 3) { ok = false; c = 0; }
        v |= convert_code(c, ga) << (2 * k);
      }
    }
    out[(1 + w) * stride] = v;
  }
  uint32_t seed_len = len >= kMinReadLen ? seed_repeats(len) : 0;
  for (uint32_t s = 0; s < 3; ++s) {
    uint32_t care[kCareWords] = {0, 0, 0, 0};
    uint32_t n = seed_len > kKeyWeight ? seed_len - kKeyWeight : 0;
    uint32_t d = D < n ? D : n;
    uint32_t K = 0;
    for (uint32_t p = 0; p < seed_len; ++p) {
      uint32_t i = s + care_pos(p);  // < len, see DESIGN.md
      uint32_t c = base_code(bases[i]);
      if (c > 3) c = 0;
      c = convert_code(c, ga);
      care[p >> 4] |= c << (30 - 2 * (p & 15));
      if (p < kKeyWeight + d) K = K * 3 + digit3(c, ga);
    }
    uint32_t base = 1 + nw + s * kPerSeedWords;
    for (uint32_t w = 0; w < kCareWords; ++w) out[(base + w) * stride] = care[w];
    out[(base + kCareWords) * stride] = K * pow3(D - d);
  }
  return ok;
}

}  // namespace walt
#endif  // WALT_AMD_INDEX_CORE_H_
