// index_core.h -- per-element builders of the derived HBM index structures
// (Ent keys, BAD bitmap, base-3 directory) and of the packed read records.
// Pure inline functions shared by the HIP kernels (index_dev.hip, pack.hip) and
// the g++ CPU unit-test harness (tests/host_harness.cpp).
#ifndef WALT_AMD_INDEX_CORE_H_
#define WALT_AMD_INDEX_CORE_H_

#include <vector>

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

// (first_beyond: core.h)

// Code prefix (first Bd bits, zero padded) of the care characters behind an index
// entry: characters 0..11 from the genome, 12..43 from the entry key.
WALT_HD uint32_t ent_prefix(const uint32_t* g2, const Ent& e, uint32_t ga, uint32_t Bd) {
  uint64_t acc = 0;
  uint32_t nb = 0;
  const uint64_t key = ent_key(e);
  for (uint32_t i = 0; i < kKeyWeight + kKeyChars && nb < Bd; ++i) {
    const uint32_t c = i < kKeyWeight ? g2_code(g2, (uint64_t)e.pos + care_pos(i))
                                      : (uint32_t)((key >> (2 * (kKeyWeight + kKeyChars - 1 - i))) & 3u);
    const uint32_t l = pcode_len(c, ga);
    acc = (acc << l) | pcode_bits(c, ga);
    nb += l;
  }
  return (uint32_t)(acc >> (nb - Bd));  // 44 characters carry >= 44 > Bd bits
}

// ---------------------------------------------------------------------------
// Packed read record.  `bases` are the sanitised read characters as the
// loader leaves them (mapping.cpp:101-103).  Returns false on a non-ACGT char.
// out has packed_fields(nw) words with stride `stride` (SoA over reads).
// ---------------------------------------------------------------------------
WALT_HD bool pack_read(const uint8_t* bases, uint32_t len, uint32_t ga, uint32_t Bd, uint32_t nw,
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
  uint32_t seed_len = len >= kMinReadLen ? seed_len_of(seed_repeats(len)) : 0;
  for (uint32_t s = 0; s < kPat; ++s) {
    uint32_t care[kCareWords] = {};
    for (uint32_t p = 0; p < care_len_of(seed_len); ++p) {
      uint32_t i = s + care_pos(p);  // < len, see DESIGN.md (pattern 7, 23/24-base reads: beyond -> 0)
      uint32_t c = i < len ? base_code(bases[i]) : 0u;
      if (c > 3) c = 0;
      c = convert_code(c, ga);
      care[p >> 4] |= c << (30 - 2 * (p & 15));
    }
    uint32_t v_lo = 0, span = 0;
    if (seed_len) dir_range(care, care_len_of(seed_len), ga, Bd, v_lo, span);
    uint32_t base = 1 + nw + s * kPerSeedWords;
    for (uint32_t w = 0; w < kCareWords; ++w) out[(base + w) * stride] = care[w];
    out[(base + kCareWords) * stride] = seed_len ? dir_top(Bd) - v_lo : 0u;  // index into the reversed directory
    out[(base + kCareWords + 1) * stride] = span;
  }
  return ok;
}

// Host side: the bucket -> first-outlier table of StrandView::outl_dir for outliers sorted by bucket.
// Returns the number of {tag, index} pairs (a power of two, at most half full).
template <class Vec>
inline uint32_t build_outlier_dir(const Outlier* outl, uint32_t n_outl, Vec& dir) {
  uint32_t distinct = 0;
  for (uint32_t i = 0; i < n_outl; ++i) distinct += (i == 0 || outl[i].h != outl[i - 1].h) ? 1u : 0u;
  uint32_t pairs = 16;
  while (pairs < 2 * distinct) pairs <<= 1;
  dir.assign((size_t)pairs * 2, 0u);
  for (uint32_t i = 0; i < n_outl; ++i) {
    if (i && outl[i].h == outl[i - 1].h) continue;
    uint32_t slot = outl_dir_hash(outl[i].h) & (pairs - 1);
    while (dir[2 * slot]) slot = (slot + 1) & (pairs - 1);
    dir[2 * slot] = outl[i].h + 1;
    dir[2 * slot + 1] = i;
  }
  return pairs;
}

// Host side: the level table of StrandView::olev for outliers sorted by (bucket, q, key).  Returns the number of
// entries (a power of two, at most half full).  Two different keys with one fingerprint (never, in practice: 63 bits)
// cannot both be held: their buckets are reported in `collided` and the caller marks them BAD (literal search, no table).
// beyond_buckets: the buckets of the entries whose care characters (of the 44 a key holds) run over the end of the GENOME
// -- the last chromosome's end entries; their bucket entries get pad = 1 (core.h olev_relevant)
template <class VecE, class VecU>
inline uint32_t build_outlier_levels(const Outlier* outl, uint32_t n_outl, VecE& table, VecU& collided,
                                     const std::vector<uint32_t>& beyond_buckets = std::vector<uint32_t>()) {
  uint32_t groups = 0, buckets = 0;
  for (uint32_t i = 0; i < n_outl; ++i) {
    const Outlier& o = outl[i];
    const bool nb = i == 0 || outl[i - 1].h != o.h;
    buckets += nb ? 1u : 0u;
    const uint64_t km = (((uint64_t)o.key_hi << 32) | o.key_lo) & key_mask_fwd(o.q - kKeyWeight);
    const bool ng = nb || outl[i - 1].q != o.q ||
                    ((((uint64_t)outl[i - 1].key_hi << 32) | outl[i - 1].key_lo) & key_mask_fwd(o.q - kKeyWeight)) != km;
    groups += ng ? 1u : 0u;
  }
  uint32_t size = 16;
  while (size < 2 * (groups + buckets)) size <<= 1;
  OlevEnt zero; zero.fp_lo = zero.fp_hi = zero.value = zero.pad = 0;
  table.assign((size_t)size, zero);
  std::vector<uint32_t> owner((size_t)size, 0xFFFFFFFFu);  // bucket of the key that holds a slot
  auto put = [&](uint32_t h, uint64_t fp, uint32_t value, bool merge_or) {
    uint32_t slot = olev_slot(fp, size - 1);
    for (;;) {
      OlevEnt& e = table[slot];
      if (e.fp_lo == 0 && e.fp_hi == 0) {
        e.fp_lo = (uint32_t)fp; e.fp_hi = (uint32_t)(fp >> 32); e.value = value;
        owner[slot] = h;
        return;
      }
      if (e.fp_lo == (uint32_t)fp && e.fp_hi == (uint32_t)(fp >> 32)) {  // the callers insert every key once: a clash of two keys
        (void)merge_or;
        collided.push_back(h);
        collided.push_back(owner[slot]);
        return;
      }
      slot = (slot + 1) & (size - 1);
    }
  };
  uint32_t i = 0;
  while (i < n_outl) {
    const uint32_t h = outl[i].h;
    uint32_t qmask = 0;
    uint32_t j = i;
    while (j < n_outl && outl[j].h == h) {
      const uint32_t q = outl[j].q;
      const uint64_t M = key_mask_fwd(q - kKeyWeight);
      const uint64_t km = (((uint64_t)outl[j].key_hi << 32) | outl[j].key_lo) & M;
      uint32_t cnt = 0, maxx = 0, k = j;
      const uint32_t sh = 2 * (kKeyWeight + kKeyChars - 1 - (q < kKeyWeight + kKeyChars ? q : kKeyWeight + kKeyChars - 1));
      while (k < n_outl && outl[k].h == h && outl[k].q == q && ((((uint64_t)outl[k].key_hi << 32) | outl[k].key_lo) & M) == km) {
        const uint32_t x = (uint32_t)(((((uint64_t)outl[k].key_hi << 32) | outl[k].key_lo) >> sh) & 3u);
        maxx = x > maxx ? x : maxx;
        ++cnt;
        ++k;
      }
      if (q < kKeyWeight + kKeyChars) {
        put(h, olev_fp(h, q, km), (cnt & 0x3FFFFFFFu) | (maxx << 30), false);
        if (q - kKeyWeight < 32) qmask |= 1u << (q - kKeyWeight);
      }
      j = k;
    }
    if (qmask) put(h, olev_fp(h, kOlevBucket, 0), qmask, false);
    i = j;
  }
  for (uint32_t hb : beyond_buckets) {
    const uint64_t fp = olev_fp(hb, kOlevBucket, 0);
    uint32_t slot = olev_slot(fp, size - 1);
    for (;;) {
      OlevEnt& e = table[slot];
      if (e.fp_lo == 0 && e.fp_hi == 0) break;  // (no outlier of that bucket is listed: nothing to flag)
      if (e.fp_lo == (uint32_t)fp && e.fp_hi == (uint32_t)(fp >> 32)) { e.pad = 1; break; }
      slot = (slot + 1) & (size - 1);
    }
  }
  return size;
}

// the buckets of the index positions whose key characters reach over the genome's end (g2: the packed genome, or its
// tail with `base` = the base index of g2[0], a multiple of 16)
inline std::vector<uint32_t> beyond_genome_buckets(const uint32_t* g2, uint64_t base, uint32_t genome_len) {
  std::vector<uint32_t> out;
  const uint32_t reach = care_pos(kKeyWeight + kKeyChars - 1), hash_reach = care_pos(kKeyWeight - 1);
  for (uint64_t pos = genome_len > reach ? genome_len - reach : 0; pos + hash_reach < genome_len; ++pos)
    if (pos >= base) out.push_back(hash_at(g2 - (base >> 4), pos));
  return out;
}

}  // namespace walt
#endif  // WALT_AMD_INDEX_CORE_H_
