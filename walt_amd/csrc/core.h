// core.h -- per-lane building blocks of the MI355X seed-and-extend path.
//
// Everything here is a pure inline function over plain pointers, compiled
// into the HIP kernels (map_se.hip / map_pe.hip / index_dev.hip) and -- for CPU
// unit tests only (tests/host_harness.cpp) -- by g++.  No wave intrinsics live
// here; those are in the kernels.
//
// Data layout in HBM (one StrandView per strand file _CT00/_CT01/_GA10/_GA11):
//   g2   : genome, 2 bits/base (A0 C1 G2 T3 -- order preserving, as the
//          reference's char compares at mapping.cpp:172-173,188-189 need),
//          16 bases per u32, base i at bits [2(i%16), 2(i%16)+1].
//   cnt  : the .dbindex counter array, 4^12+1 bucket starts (reference.hpp:79-92).
//   ent  : one 12-byte entry per .dbindex index slot: {key_hi, key_lo, pos}.
//          pos is the original index[] value; key is a DERIVED 64-bit string of
//          the 32 genome chars at care positions 12..43 behind pos
//          (F2CAREDPOSITION[12..43], seedpattern.hpp:424-430), char 12 in the
//          top 2 bits.  It is what LowerBound/UpperBound (mapping.cpp:166-196)
//          would read through genome.sequence[index[mid] + cmp_pos].
//   outl : the "outlier" entries (struct Outlier below) sorted by bucket; a probe
//          is DANGEROUS -- must take the literal search -- only when its target
//          shares an outlier's characters 12..q-1.
//   bad  : bitmap over the 4^12 buckets for disorder that no outlier explains
//          (never the case for a makedb-built index): every probe into such a
//          bucket takes the literal search.
//   dir  : DERIVED directory over the first Bd BITS of an order-preserving
//          prefix code of the care characters.  A converted strand has three
//          letters, one of them half of all bases (T after C->T, A after G->A),
//          so the code gives that letter one bit and the others two
//          (C->T: A=00 G=01 T=1;  G->A: A=0 C=10 T=11): code bits of an iid genome
//          are uniform, slots are evenly filled, and bit-string order equals
//          the index's lexicographic order.  dir is stored REVERSED (S = 2^Bd):
//          dir[S - v] = 1 + the last index slot whose code prefix is below v
//          (round 4; until then: the first slot whose prefix is >= v -- the same
//          number wherever the index is sorted; they differ at chromosome-end
//          entries, which makedb sorts in front of where their real characters
//          belong: those pull "first >= v" down and leave "last < v" alone, which
//          is what the inferred literal search needs, stretch_bounds).
//          Replaces the per-character LowerBound/UpperBound narrowing of the
//          first ~20 care characters by one lookup.
#ifndef WALT_AMD_CORE_H_
#define WALT_AMD_CORE_H_

#include <stdint.h>
#if !defined(__HIPCC__)
#include <stdio.h>
#include <stdlib.h>
#endif

#if defined(__HIPCC__)
#define WALT_HD __host__ __device__ __forceinline__
#else
#define WALT_HD inline
#endif

namespace walt {

// Seed pattern: a compile-time choice like the reference's -D SEEDPATTERN3 / 5 / 7 (src/walt/Makefile:34,
// FAQ.md:5-13).  The default build is pattern 3 (0,1,0)*; -DWALT_SEEDPATTERN=5 / 7 give the libraries
// libwalt_amd_sp5.so / _sp7.so, whose indexes are, like the reference's, specific to their pattern.
//   pattern 3: seedpattern.hpp:355-361,424    (0 1 0)*          care offsets {1} of each 3
//   pattern 5: seedpattern.hpp:226-232,264    (1 0 1 0 0)*      care offsets {0, 2} of each 5
//   pattern 7: seedpattern.hpp:29-35,59       (1 1 1 0 1 0 0)*  care offsets {0, 1, 2, 4} of each 7
#ifndef WALT_SEEDPATTERN
#define WALT_SEEDPATTERN 3
#endif
static_assert(WALT_SEEDPATTERN == 3 || WALT_SEEDPATTERN == 5 || WALT_SEEDPATTERN == 7, "seed pattern 3, 5 or 7");
constexpr uint32_t kPat = WALT_SEEDPATTERN;                          // SEEDPATTERNLEN
constexpr uint32_t kCareW = kPat == 3 ? 1 : (kPat == 5 ? 2 : 4);     // SEEDPATTERNCAREDWEIGHT
constexpr uint32_t kNoCareW = kPat - kCareW;                         // SEEDPATTERNNOCAREDWEIGHT
constexpr uint32_t kKeyWeight = 12;     // F2SEEDKEYWEIGHT
constexpr uint32_t kNumCare = kPat == 3 ? 60 : (kPat == 5 ? 56 : 80);    // F2CAREDPOSITION_SIZE
constexpr uint32_t kMinReadLen = kPat == 3 ? 38 : (kPat == 5 ? 32 : 23);  // MINIMALREADLEN
constexpr uint32_t kMinSeedLen = kPat == 3 ? 36 : (kPat == 5 ? 30 : 21);  // MINIMALSEEDLEN
// Repeats of the pattern inside a read: the reference caps them at 50 (mapping.cpp:238), which for
// patterns 5 and 7 lets a read of more than 148 / 152 bases index F2CAREDPOSITION (56 / 80 entries) and
// F2NOCAREDPOSITION (84+s / 60+s explicit entries) beyond their ends -- undefined behaviour there,
// a refused read here (kMaxReadLen).  Pattern 3's tables cover its 50 repeats.
constexpr uint32_t kMaxRepeats = kPat == 3 ? 50 : (kPat == 5 ? 28 : 20);
constexpr uint32_t kMinRepeats = (kMinReadLen - kPat + 1) / kPat;    // 12 / 5 / 2
constexpr uint32_t kMaxReadLen = kPat == 3 ? 1024 : kPat * kMaxRepeats + 2 * kPat - 2;  // 1024 / 148 / 152
constexpr uint32_t kExitOneMismatch = kPat == 7 ? 4 : 2;  // mapping.cpp:253-262: stop once best mm == 1 and seed_i >= this
constexpr uint32_t kKeyChars = 32;      // care chars 12..43 held in Ent::key
constexpr uint32_t kMaskWords = 10;     // 160 bases of table-driven compare mask
constexpr uint32_t kNumBuckets = 1u << 24;
constexpr uint32_t kMinDirBits = 24;    // directory prefixes always cover the 12 hash characters
constexpr uint32_t kMaxDirBits = 32;    // 2^32 slots: slot numbers 1..2^32 are handled modulo 2^32 (core.h dir_top)
constexpr uint32_t kEraseBucket = 500000;  // reference.cpp:211

// F2CAREDPOSITION[i] (the tables of all three patterns follow their formula exactly; tests/golden/seedpattern*.json)
WALT_HD constexpr uint32_t care_pos(uint32_t i) {
  return kPat == 3 ? 1 + 3 * i
       : kPat == 5 ? (i / 2) * 5 + (i % 2) * 2
                   : (i / 4) * 7 + ((i % 4) == 3 ? 4u : (i % 4));
}

struct Ent {
  uint32_t key_hi, key_lo, pos;
};

struct StrandView {
  const uint32_t* g2;
  const uint32_t* cnt;
  const uint32_t* bad;
  const uint32_t* dir;
  const Ent* ent;
  uint32_t index_size;
  uint32_t genome_len;
  uint32_t ga;  // 0: C->T strand (letters A,G,T)  1: G->A strand (letters A,C,T)
  uint32_t n_outl;
  const uint64_t* bloom;  // blocked Bloom filter (64-bit blocks) over the probes that can be dangerous
  uint32_t bloom_mask;    // number of blocks - 1 (a power of two, sized by the number of keys)
  const uint32_t* pre;    // kPreBits-bit prefilter of the same keys, copied into LDS by the pass-1 kernels
  // Optional direct-mapped slot table (nullptr unless WALT_AMD_TABLE=1): one 12-byte record per directory slot,
  // tab[3 (slot - 1) ..] = the slot's only entry {key_hi, key_lo, pos} when it holds exactly one, else
  // {first index slot, number of entries, kTabMulti}.  With 0.72 entries per slot (2^32 slots at hg19
  // scale) a probe then needs ONE line where directory pair + entry needed two: half of the probes that hit
  // and two thirds of the non-empty misses end in a single-entry slot.  51.5 GB per strand at 2^32 slots.
  const uint32_t* tab;
  const struct Outlier* outl;  // chromosome-end entries, sorted by bucket (see probe_is_dangerous)
  const uint32_t* outl_dir;    // open-addressing table bucket -> first outlier: pairs {bucket + 1 (0 = free), index}
  uint32_t outl_dir_mask;      // pairs - 1 (power of two); outl_dir == nullptr: binary search
  // Outlier levels (DERIVED, round 4): which outliers does a probe share its characters 12..q-1 with?  The walk over all
  // outliers of the probe's bucket answered that with one DEPENDENT load per outlier -- hundreds in the T-rich buckets of
  // an assembly of thousands of contigs, and the slowest lane of a wavefront decides.  olev is a hash table keyed by
  // (bucket, q, the outliers' characters 12..q-1): value = how many outliers have that key, and the largest real byte
  // any of them holds at character q (bits 30..31); plus one entry per bucket under the pseudo-level kOlevBucket whose
  // value is the set of q that occur in the bucket.  A probe looks its own characters up level by level: independent
  // loads, four at a time.  nullptr: the walk.
  const struct OlevEnt* olev;
  uint32_t olev_mask;          // entries - 1 (a power of two)
  uint32_t olev_pad_;
  // Dense candidate windows (DERIVED).  A region of thousands of candidates (satellites,
  // young SINE / LINE copies: 2 % of the reads of an hg19-like genome own 90 % of all candidates) is a run of
  // consecutive index slots, but the genome windows behind them are scattered: one random 128-byte line per
  // candidate for 25 useful bytes, and the device serves ~48 G such lines per second whatever their size.
  // For every run of the index that lies inside ONE region of a 100-base read (same bucket, same first
  // kWinKeyChars key characters) and has at least kWinMinRun slots -- the regions the mapping kernels do not leave
  // to a single lane -- the windows of ALL its slots are therefore stored once more, in slot order:
  //   wbits[slot >> 6]  bit (slot & 63): the slot has a dense record
  //   wrank[slot >> 6]  number of dense records in front of this 64-slot word (record of a slot = rank + popcount)
  //   win               record r: 8 words {pos, the 112 bases from genome position pos - kWinLead}
  //   win2              record r: 4 words, the following 64 bases (reads of 111..160 bases)
  //   wcap              records that exist (the memory budget may end before the last run)
  // so that the wavefront verifying such a region streams 32 (48) contiguous bytes per candidate.  A record is a
  // copy of the g2 bits count_mismatch would read, so the mismatch counts are the same by construction.  A region
  // of a read of 100 bases or more lies inside one run, its records are consecutive, and four words tell whether
  // all of them exist (core.h dense_range); anything else takes ent[] + g2[] as before.
  const unsigned long long* wbits;
  const uint32_t* wrank;
  const uint32_t* win;
  const uint32_t* win2;
  uint32_t wcap;
  uint32_t wpad_;
  // Fence keys (DERIVED; round 3): fen[k - 1][2 i], [2 i + 1] = {key_hi, key_lo} of ent[i * 16^k], k = 1..kFenceLevels --
  // every 16th, 256th, 4096th and 65536th entry's key once more, contiguous, so that 16 consecutive fences of a level
  // are ONE aligned 128-byte line.  A slot of thousands of entries (a read from a repeat family) is then searched
  // like a static B-tree of fan-out 16 that needs no pointers: the level whose fences number at most 16 inside the
  // range, then the 15 fences of the next level between two of them, ... then at most 16 entries -- one line per level
  // where the pivots of a k-ary search over the entries themselves are a line each (core.h fence_plan).
  // 1/15 of the entries' keys: 1.55 GB per strand at hg19 scale.  nullptr: search the entries (slot_kary_search).
  const uint32_t* fen[4];
};
constexpr uint32_t kFenceLevels = 4;
// an entry with fewer than this many bases of its chromosome behind it has a care character (of the kNumCare a seed can
// have) beyond the chromosome's end: such slots end the dense runs (device_index.hip k_make_ent / k_win_break)
constexpr uint32_t kTailBreakRoom = care_pos(kNumCare - 1);
constexpr uint32_t kWinMinRun = 17;  // runs of at least this many index slots get dense records (map_se.hip kMidRegion + 1)
constexpr uint32_t kWinLead = kPat - 1;   // bases in front of pos: the largest seed shift (genome_pos = pos - seed_i)
constexpr uint32_t kWinWords = 7, kWinWords2 = 4;
constexpr uint32_t kWinKeyChars = kPat == 3 ? 20 : (kPat == 5 ? 26 : 32);  // key characters of a 100-base read's seed
// reads of up to this many bases can be verified on the record(s): win alone / win + win2
constexpr uint32_t kWinMaxLen1 = 16 * kWinWords - kWinLead, kWinMaxLen2 = 16 * (kWinWords + kWinWords2) - kWinLead;
// The part of a region [l, l + size) whose candidates have dense records: slots [lo, hi), record `rec` belongs
// to slot lo (records of consecutive slots are consecutive).  All of the region or nothing (lo == hi == l).
// Four independent loads.
struct DenseRange {
  uint32_t lo, hi;
  uint64_t rec;
};
WALT_HD uint32_t popc64(unsigned long long x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint32_t)__popcll(x);
#else
  return (uint32_t)__builtin_popcountll(x);
#endif
}
WALT_HD DenseRange dense_range(const StrandView& sv, uint32_t l, uint32_t size, bool usable) {
  DenseRange d;
  d.lo = d.hi = l;
  d.rec = 0;
#if defined(__HIP_DEVICE_COMPILE__)
  // branch-free: the four loads are issued by every lane (a lane without a region reads word 0), so that two calls
  // -- the two strands of a probe -- share one memory round trip instead of waiting for each other's branch
  if (sv.wbits == nullptr) return d;  // uniform
  const bool on = usable && size;
  const uint32_t first = on ? l : 0u, last = on ? l + size - 1 : 0u;
  const unsigned long long b0 = sv.wbits[first >> 6], b1 = sv.wbits[last >> 6];
  const uint32_t k0 = sv.wrank[first >> 6], k1 = sv.wrank[last >> 6];
  const uint32_t r0 = k0 + popc64(b0 & ((1ull << (first & 63u)) - 1ull)), r1 = k1 + popc64(b1 & ((1ull << (last & 63u)) - 1ull));
  if (on && ((b0 >> (first & 63u)) & 1ull) && ((b1 >> (last & 63u)) & 1ull) && r1 - r0 == size - 1 && r1 < sv.wcap) {
    d.hi = l + size;
    d.rec = r0;
  }
#else
  if (sv.wbits == nullptr) return d;
  if (usable && size) {
    const uint32_t last = l + size - 1;
    const unsigned long long b0 = sv.wbits[l >> 6], b1 = sv.wbits[last >> 6];
    const uint32_t k0 = sv.wrank[l >> 6], k1 = sv.wrank[last >> 6];
    const uint32_t r0 = k0 + popc64(b0 & ((1ull << (l & 63u)) - 1ull)), r1 = k1 + popc64(b1 & ((1ull << (last & 63u)) - 1ull));
    if (((b0 >> (l & 63u)) & 1ull) && ((b1 >> (last & 63u)) & 1ull) && r1 - r0 == size - 1 && r1 < sv.wcap) {
      d.hi = l + size;
      d.rec = r0;
    }
  }
#endif
  return d;
}

// An index entry whose care positions run over the end of its chromosome before
// care character 44.  makedb sorted it as if every character from index q on
// were smaller than any base (reference.cpp:271-276), but LowerBound/UpperBound
// read the real bytes there.  Characters 12..q-1 are real on both sides, so the
// entry sits correctly among entries that differ from it before q; the order is
// only unreliable inside the group that shares its characters 12..q-1.
struct Outlier {
  uint32_t h;        // 4^12 bucket
  uint32_t q;        // first care character index that lies beyond the chromosome end (12..43)
  uint32_t key_hi, key_lo;  // the entry's key (real bytes)
};

// Edge bitmap (DERIVED, round 3): bit b is set when some chromosome boundary lies within kEdgeMargin bases of the
// genome positions [b << kEdgeBlockShift, (b + 1) << kEdgeBlockShift).  A candidate position in a block whose bit is
// clear is at least kEdgeMargin bases from both ends of its chromosome, so that `offset in chromosome >= seed shift`
// and `position - shift + read length < chromosome end` (mapping.cpp:280-286) hold for every seed shift (< 7) and read
// length (<= 1024) without knowing the chromosome.  2^16 blocks at most: 8 KB.
constexpr uint32_t kEdgeBlockShift = 16;
constexpr uint32_t kEdgeMargin = 1024 + 16;
constexpr uint32_t kEdgeWords = (1u << (32 - kEdgeBlockShift)) / 32;  // 2,048 words

struct IndexView {
  StrandView s[4];               // CT00, CT01, GA10, GA11
  const uint32_t* start_index;   // n_chrom + 1 (Genome::start_index, reference.hpp:55)
  const uint32_t* edge_bits;     // kEdgeWords words (see above); nullptr: none
  uint32_t n_chrom;
  uint32_t dir_bits;             // Bd: directory prefix length in code bits
  uint32_t dir_slots;            // S = 2^Bd modulo 2^32 (0 when Bd == 32; see dir_top)
  uint32_t batch_max_len;        // per launch (the host fills its copy): the caller's max_read_len, and the bytes the
  uint64_t batch_cap_bytes;      // dense 2-bit read array has room for; a read beyond either is refused, not read
};

// BestMatch, mapping.hpp:39-52.  Same 16-byte layout.
struct BestMatch {
  uint32_t genome_pos;
  uint32_t times;
  uint32_t strand;  // low byte '+' / '-', upper bytes 0
  uint32_t mismatch;
};

// Packed read record produced by pack_reads (one per read and conversion).
//   words[NW]      converted read, 2 bits/base, 16 bases per word
//   care[s][4]     for seed shift s: the chars at read offsets s+1+3i, i<50,
//                  MSB first (char 0 in bits 31..30 of care[s][0])
//   slot[s], span[s]  the entries whose care characters start like this seed's
//                  are dir[slot[s]] .. dir[slot[s] - span[s]] (reversed directory)
// kept in SoA form in HBM: field f of read r at base[f * stride + r].
constexpr uint32_t kCareWords = (kNumCare + 15) / 16;  // 4 / 4 / 5
constexpr uint32_t kPerSeedWords = kCareWords + 2;
WALT_HD uint32_t packed_fields(uint32_t nw) { return 1 + nw + kPat * kPerSeedWords; }
// field 0: length; 1..nw: words; then per seed: care[4], slot, span

// ---------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------
WALT_HD uint32_t pow3(uint32_t e) {
  uint32_t r = 1;
  for (uint32_t i = 0; i < e; ++i) r *= 3;
  return r;
}

// seed geometry, mapping.cpp:235-239
WALT_HD uint32_t seed_repeats(uint32_t read_len) {
  uint32_t r = (read_len - kPat + 1) / kPat;
  return r < kMaxRepeats ? r : kMaxRepeats;
}
WALT_HD uint32_t seed_len_of(uint32_t repeats) { return repeats * kCareW; }  // mapping.cpp:239
// characters a care string holds: the seed's, but never fewer than the 12 that getHashValue reads
// (util.hpp:175-182 hashes F2CAREDPOSITION[0..11] whatever the seed length; patterns 5 / 7 have seeds of 10 / 8)
WALT_HD uint32_t care_len_of(uint32_t seed_len) { return seed_len ? (seed_len > kKeyWeight ? seed_len : kKeyWeight) : 0u; }

// 2-bit code of a sanitised base; 4 = not ACGT (getBits would exit, util.hpp:117-119)
WALT_HD uint32_t base_code(uint8_t c) {
  return c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 4u;
}
// read conversion, mapping.cpp:142-164 (C->T: code 1 -> 3; G->A: code 2 -> 0)
WALT_HD uint32_t convert_code(uint32_t code, uint32_t ga) {
  return ga ? (code == 2 ? 0u : code) : (code == 1 ? 3u : code);
}
// order-preserving prefix code of a converted base (see the dir comment above)
WALT_HD uint32_t pcode_len(uint32_t c, uint32_t ga) { return ga ? (c == 0 ? 1u : 2u) : (c == 3 ? 1u : 2u); }
WALT_HD uint32_t pcode_bits(uint32_t c, uint32_t ga) {
  return ga ? (c == 0 ? 0u : (c == 1 ? 2u : 3u)) : (c == 0 ? 0u : 1u);  // C->T: A 00, G 01, T 1
}

WALT_HD uint32_t g2_code(const uint32_t* g2, uint64_t pos) {
  return (g2[pos >> 4] >> (2 * (uint32_t)(pos & 15))) & 3u;
}
// genome char as the reference reads it: beyond the end compares below every base
WALT_HD int gchar(const StrandView& sv, uint64_t pos) {
  return pos < sv.genome_len ? (int)g2_code(sv.g2, pos) : -1;
}

// getChromID, reference.cpp:43-60
WALT_HD uint32_t chrom_id(const uint32_t* start_index, uint32_t n_chrom, uint32_t pos) {
  uint32_t l = 0, h = n_chrom;
  while (l < h) {
    uint32_t m = (l + h + 1) >> 1;
    if (pos >= start_index[m]) l = m; else h = m - 1;
  }
  return l;
}

WALT_HD uint64_t ent_key(const Ent& e) { return ((uint64_t)e.key_hi << 32) | e.key_lo; }

// char p (0..49) of an MSB-first care string held in 4 words
WALT_HD uint32_t care_char(const uint32_t* care, uint32_t p) {
  const uint32_t c0 = care[0], c1 = care[1], c2 = care[2], c3 = care[3];
  const uint32_t w = p >> 4;
  uint32_t v = kCareWords > 4 ? care[kCareWords - 1] : c3;
  v = (kCareWords > 4 && w == 3) ? c3 : v;
  v = w == 2 ? c2 : v;
  v = w == 1 ? c1 : v;
  v = w == 0 ? c0 : v;
  return (v >> (30 - 2 * (p & 15))) & 3u;
}

// S = 2^Bd as a 32-bit value: the reversed directory is addressed by slot = S - v for code prefix v, i.e.
// slot in [1, S]; with Bd == 32 slot S is represented by 0 and all slot arithmetic is modulo 2^32.  The
// directory pair of a slot is read at dir + (uint32_t)(slot - 1), which is exact for every slot in [1, 2^32].
WALT_HD uint32_t dir_top(uint32_t Bd) { return Bd < 32 ? (1u << Bd) : 0u; }

// Directory range of a care string of nchars characters: code prefixes
// [v_lo, v_lo + span) (span is a power of two; 1 when the string has >= Bd bits).
WALT_HD void dir_range(const uint32_t* care, uint32_t nchars, uint32_t ga, uint32_t Bd, uint32_t& v_lo,
                       uint32_t& span) {
  uint64_t acc = 0;
  uint32_t nb = 0;
  for (uint32_t i = 0; i < 32 && i < nchars && nb < Bd; ++i) {
    const uint32_t c = care_char(care, i);
    const uint32_t l = pcode_len(c, ga);
    acc = (acc << l) | pcode_bits(c, ga);
    nb += l;
  }
  if (nb >= Bd) {
    v_lo = (uint32_t)(acc >> (nb - Bd));
    span = 1;
  } else {
    v_lo = (uint32_t)(acc << (Bd - nb));
    span = 1u << (Bd - nb);
  }
}

// ---------------------------------------------------------------------------
// search
// ---------------------------------------------------------------------------
struct Region {  // inclusive [l,u]; empty when l > u (the reference's (1,0) marker)
  uint32_t l, u;
};
WALT_HD Region empty_region() { Region r; r.l = 1; r.u = 0; return r; }

// genome.sequence[index[j] + F2CAREDPOSITION[p]] as the reference reads it
// (mapping.cpp:172,188,207): care chars 12..43 come from the entry's key (one
// load gives key and pos), later ones from the packed genome; a position at or
// beyond the end of the genome compares below every base.
WALT_HD int ent_char(const StrandView& sv, uint32_t j, uint32_t p) {
  const Ent e = sv.ent[j];
  const uint64_t q = (uint64_t)e.pos + care_pos(p);
  if (q >= sv.genome_len) return -1;
  if (p < kKeyWeight + kKeyChars) return (int)((ent_key(e) >> (2 * (kKeyWeight + kKeyChars - 1 - p))) & 3u);
  return (int)g2_code(sv.g2, q);
}

// LowerBound / UpperBound, mapping.cpp:166-196, literal (used for BAD buckets
// and for care chars >= 44).
WALT_HD uint32_t lit_lower(const StrandView& sv, uint32_t low, uint32_t high, int ch, uint32_t p) {
  while (low < high) {
    uint32_t mid = low + (high - low) / 2;
    int c = ent_char(sv, mid, p);
    if (c >= ch) high = mid; else low = mid + 1;
  }
  return low;
}
WALT_HD uint32_t lit_upper(const StrandView& sv, uint32_t low, uint32_t high, int ch, uint32_t p) {
  while (low < high) {
    uint32_t mid = low + (high - low + 1) / 2;
    int c = ent_char(sv, mid, p);
    if (c <= ch) low = mid; else high = mid - 1;
  }
  return low;
}
// IndexRegion, mapping.cpp:198-222, for care chars [p0, seed_len) on inclusive [l,u].
WALT_HD Region lit_region(const StrandView& sv, const uint32_t* care, uint32_t p0, uint32_t seed_len,
                          uint32_t l, uint32_t u) {
  for (uint32_t p = p0; p < seed_len; ++p) {
    int ch = (int)care_char(care, p);
    l = lit_lower(sv, l, u, ch, p);
    u = lit_upper(sv, l, u, ch, p);
    if (l == u && ch != ent_char(sv, l, p)) return empty_region();
  }
  if (l > u) return empty_region();
  Region r; r.l = l; r.u = u; return r;
}

// first care character index (>= 12) of an entry that lies at or beyond the end of
// its chromosome; room = chromosome end - pos.  kNumCare when every character fits.
WALT_HD uint32_t first_beyond(uint32_t room) {
  uint32_t q;
  if (kPat == 3) {
    // care_pos(q) = 1 + 3 q >= room  <=>  q >= (room - 1) / 3 rounded up
    q = room <= 1 ? 0u : (room - 1 + 2) / 3;
  } else {
    q = 0;
    while (q < kNumCare && care_pos(q) < room) ++q;
  }
  return q < kKeyWeight ? kKeyWeight : q;
}


WALT_HD uint64_t key_mask_fwd(uint32_t nk) { return nk >= 32 ? ~0ull : ~(~0ull >> (2 * nk)); }  // (= key_mask)

// ---- the literal search with a memo (round 4) ---------------------------------------------------------------------
// lit_region above loads an entry for every step of every bisection: ~170 dependent loads for a dangerous probe into a
// bucket of tens of thousands of entries, and an assembly of thousands of contigs sends a fifth of its reads there.
// This is the SAME procedure -- the reference's mids, character by character (mapping.cpp:166-222) -- but the byte the
// reference would read at (mid, p) is taken from what is already known whenever that is certain:
//   * an entry that has been loaded is kept with its whole key (one load gives the characters 12..43 of an entry, the
//     reference re-reads the entry for every character) and with q, the first of its care characters that lies beyond
//     its chromosome's end (from its position and the chromosome starts);
//   * FACT S: let a < mid < b be index slots of one bucket, a and b loaded, their keys equal on the characters 12..p,
//     and q(a) > p (a's characters through p lie inside its chromosome).  makedb's order (reference.cpp:258-288) is
//     the order of the strings (real characters ..., then "beyond the chromosome's end", which ranks below every base
//     and ends the comparison).  a's sort key through p is its real characters.  b cannot carry a "beyond" mark at or
//     before p: it would share a's characters in front of the mark and rank BELOW a.  So a and b have the same,
//     mark-free sort key through p, every entry between them has it too, an entry whose sort key through p is free of
//     marks has real characters there -- the byte at (mid, p) is a's character p.  (A bucket whose order the outliers
//     do not explain is BAD and never comes here: it is searched by lit_region.)
//   * anything else is loaded, as before.
// What comes back is lit_region's result by construction: every comparison uses the byte the reference reads.
// tests/test_harness_cpu.py compares the two routes for every dangerous probe of the stress genomes.
constexpr uint32_t kMemoCap = 4;    // loaded entries kept (registers on the device: every access is an unrolled select)
struct LitMemo {
  uint32_t idx[kMemoCap], khi[kMemoCap], klo[kMemoCap], pos[kMemoCap], q[kMemoCap];
  uint32_t n;
  uint32_t loads, probes;  // (diagnostic: entry loads made / bytes the reference would have read)
};
WALT_HD void memo_init(LitMemo& m) {
  m.n = 0; m.loads = 0; m.probes = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (uint32_t k = 0; k < kMemoCap; ++k) { m.idx[k] = 0; m.khi[k] = 0; m.klo[k] = 0; m.pos[k] = 0; m.q[k] = 0; }
}
// q of an entry from the chromosome starts in device / host memory (the kernels pass a functor over their LDS copy)
struct QOfGlobal {
  const uint32_t* start_index;
  uint32_t n_chrom;
  WALT_HD uint32_t operator()(uint32_t pos) const {
    uint32_t l = 0, h = n_chrom;
    while (l < h) {
      const uint32_t m = (l + h + 1) >> 1;
      if (pos >= start_index[m]) l = m; else h = m - 1;
    }
    return first_beyond(start_index[l + 1] - pos);
  }
};
// keep (j, e); when full, drop an entry outside [l, u] (it can never serve again: it left the range because one of its
// characters differs from the probe's) -- the one farthest out -- else the one farthest from j
WALT_HD void memo_insert(LitMemo& m, uint32_t j, const Ent& e, uint32_t qv, uint32_t l, uint32_t u) {
  uint32_t at = m.n;
  if (m.n >= kMemoCap) {
    uint32_t best = 0, best_out = 0, best_far = 0;
    bool any_out = false;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (uint32_t k = 0; k < kMemoCap; ++k) {
      const uint32_t x = m.idx[k];
      const uint32_t out = x < l ? l - x : (x > u ? x - u : 0u);
      const uint32_t far = x < j ? j - x : x - j;
      const bool better = out ? (!any_out || out > best_out) : (!any_out && far > best_far);
      if (k == 0 || better) { best = k; best_out = out; best_far = far; }
      any_out = any_out || out != 0;
    }
    at = best;
  } else {
    ++m.n;
  }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (uint32_t k = 0; k < kMemoCap; ++k)
    if (k == at) { m.idx[k] = j; m.khi[k] = e.key_hi; m.klo[k] = e.key_lo; m.pos[k] = e.pos; m.q[k] = qv; }
}
// genome.sequence[index[j] + F2CAREDPOSITION[p]] (ent_char) for kKeyWeight <= p < kKeyWeight + kKeyChars, through the memo;
// [l, u] = the reference's current range (for the eviction only)
template <class QOf>
WALT_HD int memo_char(const StrandView& sv, LitMemo& m, const QOf& q_of, uint32_t j, uint32_t p, uint32_t l, uint32_t u) {
  ++m.probes;
  bool has_a = false, has_b = false;
  uint32_t a_idx = 0, b_idx = 0, a_hi = 0, a_lo = 0, a_pos = 0, a_q = 0, b_hi = 0, b_lo = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (uint32_t k = 0; k < kMemoCap; ++k) {
    const bool live = k < m.n;
    const uint32_t x = m.idx[k];
    const bool ta = live && x <= j && (!has_a || x > a_idx);
    const bool tb = live && x >= j && (!has_b || x < b_idx);
    if (ta) { has_a = true; a_idx = x; a_hi = m.khi[k]; a_lo = m.klo[k]; a_pos = m.pos[k]; a_q = m.q[k]; }
    if (tb) { has_b = true; b_idx = x; b_hi = m.khi[k]; b_lo = m.klo[k]; }
  }
  const uint32_t sh = 2 * (kKeyWeight + kKeyChars - 1 - p);
  if (has_a && a_idx == j) {  // loaded before: its real byte, as the reference reads it (beyond the genome: below every base)
    if ((uint64_t)a_pos + care_pos(p) >= sv.genome_len) return -1;
    return (int)(((((uint64_t)a_hi << 32) | a_lo) >> sh) & 3u);
  }
  if (has_a && has_b && a_q > p) {
    const uint64_t ka = ((uint64_t)a_hi << 32) | a_lo, kb = ((uint64_t)b_hi << 32) | b_lo;
    if (((ka ^ kb) & key_mask_fwd(p - kKeyWeight + 1)) == 0) return (int)((ka >> sh) & 3u);  // FACT S
  }
  const Ent e = sv.ent[j];
  ++m.loads;
  memo_insert(m, j, e, q_of(e.pos), l, u);
  if ((uint64_t)e.pos + care_pos(p) >= sv.genome_len) return -1;
  return (int)((ent_key(e) >> sh) & 3u);
}
// IndexRegion (mapping.cpp:198-222) for the characters [p0, seed_len) on [l, u], the key characters through the memo
template <class QOf>
WALT_HD Region lit_region_memo(const StrandView& sv, const uint32_t* care, uint32_t p0, uint32_t seed_len, uint32_t l,
                               uint32_t u, LitMemo& m, const QOf& q_of) {
  const uint32_t lim = seed_len < kKeyWeight + kKeyChars ? seed_len : kKeyWeight + kKeyChars;
  if (l < u && p0 < lim) {  // both ends of the range: two independent loads, usually the anchors of the first characters
    const Ent el = sv.ent[l], eu = sv.ent[u];
    m.loads += 2;
    memo_insert(m, l, el, q_of(el.pos), l, u);
    memo_insert(m, u, eu, q_of(eu.pos), l, u);
  }
  uint32_t p = p0;
  for (; p < lim; ++p) {
    const int ch = (int)care_char(care, p);
    uint32_t low = l, high = u;
    while (low < high) {  // LowerBound, mapping.cpp:166-180
      const uint32_t mid = low + (high - low) / 2;
      if (memo_char(sv, m, q_of, mid, p, l, u) >= ch) high = mid; else low = mid + 1;
    }
    l = low;
    high = u;
    while (low < high) {  // UpperBound, mapping.cpp:182-196
      const uint32_t mid = low + (high - low + 1) / 2;
      if (memo_char(sv, m, q_of, mid, p, l, u) <= ch) low = mid; else high = mid - 1;
    }
    u = low;
    if (l == u && ch != memo_char(sv, m, q_of, l, p, l, u)) return empty_region();
  }
  if (p < seed_len) return lit_region(sv, care, p, seed_len, l, u);  // characters >= 44: on the genome itself
  if (l > u) return empty_region();
  Region r; r.l = l; r.u = u; return r;
}

// Target key (care chars 12..43 of the read's care string) and the mask of its
// first nk chars.
WALT_HD uint64_t target_key(const uint32_t* care) {
  return ((uint64_t)(care[0] & 0xFFu) << 56) | ((uint64_t)care[1] << 24) | (care[2] >> 8);
}
WALT_HD uint64_t key_mask(uint32_t nk) { return nk >= 32 ? ~0ull : ~(~0ull >> (2 * nk)); }

WALT_HD bool bucket_is_bad(const StrandView& sv, uint32_t h) { return (sv.bad[h >> 5] >> (h & 31)) & 1u; }

// Pass 1 of the mapping kernels must decide "might this probe be dangerous?" without an extra
// DEPENDENT memory round trip (a slow path taken by even 1 % of the lanes stalls nearly every
// wavefront).  It tests a blocked Bloom filter: one 64-bit block per key, four bits in it, so the
// test is ONE 8-byte load that is issued together with the directory loads of the probe and hits L2
// (the filter is sized at >= 128 bits per key: 64 KB for hg19's 24 chromosomes, a few MB for an
// assembly of thousands of contigs -- an LDS-resident filter of fixed size saturates there and sends
// every read to the slow pass).  On a hit the read is deferred; the exact test
// (probe_is_dangerous) runs in pass 2.
// Filter key = (bucket, care characters 12..15) = the first 16 care characters = care[0]: an
// outlier with q >= 16 inserts its own four characters; one with q < 16 inserts every value of the
// characters from q on (a dangerous probe agrees with it on the characters before q, which are real
// on both sides; whatever the probe holds from q on, zero padding of a short seed included, is
// among the inserted values); a BAD bucket inserts all 256.  Two characters were not enough: in the
// 3-letter, T-heavy converted alphabet 1 % of the reads shared (bucket, 2 chars) with some
// chromosome end.
constexpr uint32_t kTabMulti = 0xFFFFFFFFu;  // never a genome position (genomes stop below 2^32 - 256)
constexpr uint32_t kBloomMinBlocks = 1u << 10, kBloomMaxBlocks = 1u << 20;  // 8 KB .. 8 MB
WALT_HD uint32_t bloom_blocks_for(uint64_t n_keys) {  // >= 2 blocks (128 bits) per key, power of two
  uint64_t want = 2 * n_keys;
  uint32_t b = kBloomMinBlocks;
  while (b < want && b < kBloomMaxBlocks) b <<= 1;
  return b;
}
constexpr uint32_t kBloomChars = 4;  // care characters 12..15 in the key
WALT_HD uint32_t bloom_key(uint32_t h, uint32_t chars12_15) { return (h << 8) | chars12_15; }
WALT_HD uint32_t bloom_block(uint32_t key, uint32_t mask) {
  uint32_t x = key * 0x9E3779B1u;
  x ^= x >> 15;
  x *= 0x2C1B3C6Du;
  x ^= x >> 13;
  return x & mask;
}
WALT_HD uint64_t bloom_bits(uint32_t key) {
  uint32_t y = (key ^ 0x85EBCA6Bu) * 0xC2B2AE35u;
  y ^= y >> 16;
  y *= 0x27D4EB2Fu;
  y ^= y >> 15;
  return (1ull << (y & 63)) | (1ull << ((y >> 6) & 63)) | (1ull << ((y >> 12) & 63)) | (1ull << ((y >> 18) & 63));
}
WALT_HD bool bloom_hit(uint64_t block, uint32_t key) {
  const uint64_t b = bloom_bits(key);
  return (block & b) == b;
}
WALT_HD void bloom_insert(uint64_t* bloom, uint32_t mask, uint32_t key) { bloom[bloom_block(key, mask)] |= bloom_bits(key); }
// Prefilter in front of the Bloom filter: one bit per key in a kPreBits-bit set that the pass-1 kernels
// hold in LDS (8 KB per strand).  The mapping kernels are bound by the per-lane accesses the L1 serves,
// and the Bloom block is one of ~4 per probe; with hg19's 24 chromosomes the prefilter is ~10 % full, so
// nine probes in ten skip that load.  No false negatives (superset of the Bloom filter's keys): the set
// of deferred reads is unchanged.  An assembly of thousands of contigs fills it and every probe goes on
// to the Bloom filter as before.
constexpr uint32_t kPreBits = 1u << 16;
WALT_HD uint32_t pre_hash(uint32_t key) { return (key * 0x9E3779B1u) >> 16; }
// key of a probe: bucket and care characters 12..15 of its (zero padded) care string
WALT_HD uint32_t bloom_key_of_care(const uint32_t* care) { return care[0]; }
// Second level: an outlier whose beyond-the-end characters start at q >= kBloomChars2 (most of them: q runs
// to 43) endangers only probes that agree with it on ALL of the characters 12..q-1, in particular on
// 12..19.  Such an outlier is inserted into the block its 16-character key selects, but with the bit pattern
// of its 20-character key, and a probe tests both patterns on the one block it loads: no extra memory access,
// and the four extra characters cut these outliers' chance matches by 0.375^4 = 1/50 (a converted strand has
// three letters, one of them half of all bases) -- a third fewer deferred reads on an assembly of many contigs.
constexpr uint32_t kBloomChars2 = 8;   // care characters 12..19
WALT_HD uint32_t bloom_key2(uint32_t key16, uint32_t chars16_19) {
  return (key16 * 0x85EBCA6Bu) ^ ((chars16_19 + 1u) * 0xC2B2AE35u);
}
WALT_HD uint32_t bloom_key2_of_care(const uint32_t* care) { return bloom_key2(care[0], care[1] >> 24); }
// the pass-1 test on the loaded block (0 = not loaded: no hit)
WALT_HD bool danger_filter_hit(uint64_t block, const uint32_t* care) {
  const uint64_t b1 = bloom_bits(bloom_key_of_care(care)), b2 = bloom_bits(bloom_key2_of_care(care));
  return (block & b1) == b1 || (block & b2) == b2;
}
WALT_HD uint64_t key_mask(uint32_t nk);
WALT_HD uint64_t target_key(const uint32_t* care);

// ---- outlier levels (StrandView::olev) ------------------------------------------------------------------------------
struct OlevEnt {
  uint32_t fp_lo, fp_hi;  // fingerprint of (bucket, q, characters 12..q-1); 0, 0 = free
  uint32_t value;         // level entry: count (bits 0..29) | largest real byte at character q << 30; bucket entry: the q set, bit q - 12
  uint32_t pad;
};
constexpr uint32_t kOlevBucket = 63;  // pseudo-level of the per-bucket entry
WALT_HD uint64_t olev_fp(uint32_t h, uint32_t q, uint64_t key_masked) {
  uint64_t x = key_masked + 0x9E3779B97F4A7C15ull * ((((uint64_t)h) << 6) | q);
  x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull;
  x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull;
  x ^= x >> 32;
  return x | (1ull << 63);  // never (0, 0)
}
WALT_HD uint32_t olev_slot(uint64_t fp, uint32_t mask) { return (uint32_t)(fp >> 8) & mask; }
// value of fingerprint fp, 0 when absent; `first` = the entry at fp's home slot, already loaded
WALT_HD uint32_t olev_resolve(const StrandView& sv, uint64_t fp, OlevEnt first) {
  uint32_t slot = olev_slot(fp, sv.olev_mask);
  OlevEnt e = first;
  for (;;) {
    if (e.fp_lo == 0 && e.fp_hi == 0) return 0;
    if (e.fp_lo == (uint32_t)fp && e.fp_hi == (uint32_t)(fp >> 32)) return e.value;
    slot = (slot + 1) & sv.olev_mask;
    e = sv.olev[slot];
  }
}
WALT_HD uint32_t olev_find(const StrandView& sv, uint64_t fp) { return olev_resolve(sv, fp, sv.olev[olev_slot(fp, sv.olev_mask)]); }
// The outliers a probe shares its characters 12..q-1 with, q < lim: levels = the set of their q (bit q - 12); dangerous =
// one of them holds, at its character q, a real byte that is not below the probe's (probe_danger_level's rule); q_min.
// Four independent look-ups per round.  Returns false when the strand has no level table (the caller walks).
// counts (optional): 4 bits per level q - 12 < 16, the number of outliers of that level the probe shares its characters
// with (15: that many or more -- look it up); beyond (optional): the bucket holds an entry whose care characters run over the
// end of the GENOME (the last chromosome's end entries: the only ones whose byte the reference reads as below every base).
WALT_HD bool olev_relevant(const StrandView& sv, uint32_t h, uint64_t T, uint32_t lim, uint32_t& levels, bool& dangerous,
                           uint32_t& q_min, uint64_t* counts = nullptr, bool* beyond = nullptr) {
  levels = 0; dangerous = false; q_min = 0xFFFFFFFFu;
  if (counts) *counts = 0;
  if (beyond) *beyond = true;
  if (sv.olev == nullptr) return false;
  uint32_t m;
  {
    const uint64_t fpb = olev_fp(h, kOlevBucket, 0);
    uint32_t slot = olev_slot(fpb, sv.olev_mask);
    OlevEnt e = sv.olev[slot];
    m = 0;
    bool by = false;
    for (;;) {
      if (e.fp_lo == 0 && e.fp_hi == 0) break;
      if (e.fp_lo == (uint32_t)fpb && e.fp_hi == (uint32_t)(fpb >> 32)) { m = e.value; by = e.pad != 0; break; }
      slot = (slot + 1) & sv.olev_mask;
      e = sv.olev[slot];
    }
    if (beyond) *beyond = by;
  }  // m: the q that occur in the bucket at all
  const uint32_t span = lim > kKeyWeight ? lim - kKeyWeight : 0u;
  m = span >= 32 ? m : (m & ((1u << span) - 1u));
  while (m) {
    uint32_t qs[4];
    uint64_t fps[4];
    OlevEnt es[4];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (uint32_t t = 0; t < 4; ++t) {
      uint32_t z = 0;
      if (m) {
#if defined(__HIP_DEVICE_COMPILE__)
        z = (uint32_t)__ffs((int)m) - 1u;
#else
        z = (uint32_t)__builtin_ctz(m);
#endif
        m &= m - 1;
        qs[t] = kKeyWeight + z;
      } else {
        qs[t] = 0;  // (no level: the bucket entry's slot, a load that is certain to hit the cache)
      }
      fps[t] = qs[t] ? olev_fp(h, qs[t], T & key_mask_fwd(qs[t] - kKeyWeight)) : olev_fp(h, kOlevBucket, 0);
      es[t] = sv.olev[olev_slot(fps[t], sv.olev_mask)];
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (uint32_t t = 0; t < 4; ++t) {
      if (!qs[t]) continue;
      const uint32_t v = olev_resolve(sv, fps[t], es[t]);
      if (!v) continue;
      levels |= 1u << (qs[t] - kKeyWeight);
      q_min = qs[t] < q_min ? qs[t] : q_min;
      if (counts && qs[t] - kKeyWeight < 16) {
        const uint32_t c = v & 0x3FFFFFFFu;
        *counts |= (uint64_t)(c < 15u ? c : 15u) << (4 * (qs[t] - kKeyWeight));
      }
      const uint32_t sh = 2 * (kKeyWeight + kKeyChars - 1 - qs[t]);
      if (!((uint32_t)((T >> sh) & 3u) > (v >> 30))) dangerous = true;
    }
  }
  return true;
}

// Does this probe have to take the literal LowerBound/UpperBound search?
WALT_HD uint32_t outl_dir_hash(uint32_t h) {
  uint32_t x = h * 0x9E3779B1u;
  return x ^ (x >> 15);
}
// 0: the probe is safe (directory / key search).  Otherwise the literal LowerBound / UpperBound search is needed, and
// the value says from where: kKeyWeight (12) = over the whole bucket from the first character (a BAD bucket, or an
// outlier of the probe's prefix group right behind the hash characters), or q_min > 12 = the smallest
// first-missing-character index among ALL outliers that share the probe's characters 12..q-1.  Up to character
// q_min - 1 the reference's bisection only ever compares entries on characters that are real AND in sorted order
// (an outlier with q < q_min that the probe does not share differs from the probe before its q, where it is placed
// correctly; one with q >= q_min is compared on real characters only), so after character q_min - 1 its range is the
// equal range of the probe's first q_min - 12 key characters -- one masked-key search -- and only from q_min on does
// the outcome depend on the outliers' real bytes (round 3; DESIGN.md section 4, "literal search from q_min").
WALT_HD uint32_t probe_danger_level(const StrandView& sv, const uint32_t* care, uint32_t seed_len) {
  const uint32_t h = care[0] >> 8;
  if (bucket_is_bad(sv, h)) return kKeyWeight;
  {
    uint32_t levels, q_first;
    bool dng;
    const uint32_t lim_t = seed_len < kKeyWeight + kKeyChars ? seed_len : kKeyWeight + kKeyChars;
    if (olev_relevant(sv, h, target_key(care), lim_t, levels, dng, q_first)) return dng ? q_first : 0u;
  }
  uint32_t lo = 0, hi = sv.n_outl;  // first outlier of bucket h
  if (sv.outl_dir) {
    // one or two loads instead of log2(n_outl) dependent ones (this test runs for every probe of pass 2)
    uint32_t slot = outl_dir_hash(h) & sv.outl_dir_mask;
    for (;;) {
      const uint32_t tag = sv.outl_dir[2 * slot];
      if (tag == 0) return 0;  // no outlier in this bucket
      if (tag == h + 1) { lo = sv.outl_dir[2 * slot + 1]; break; }
      slot = (slot + 1) & sv.outl_dir_mask;
    }
  } else {
    while (lo < hi) {
      uint32_t mid = lo + ((hi - lo) >> 1);
      if (sv.outl[mid].h < h) lo = mid + 1; else hi = mid;
    }
  }
  const uint32_t lim = seed_len < kKeyWeight + kKeyChars ? seed_len : kKeyWeight + kKeyChars;
  const uint64_t T = target_key(care);
  bool dangerous = false;
  uint32_t q_min = 0xFFFFFFFFu;
  for (; lo < sv.n_outl && sv.outl[lo].h == h; ++lo) {
    const Outlier o = sv.outl[lo];
    if (o.q >= lim) continue;  // the search never compares a character this entry lacks
    const uint64_t k = ((uint64_t)o.key_hi << 32) | o.key_lo;
    if (((T ^ k) & key_mask(o.q - kKeyWeight)) == 0) {
      q_min = o.q < q_min ? o.q : q_min;
      // The probe shares the outlier's characters 12..q-1.  The outlier sits at the START of that group
      // (makedb ranks its missing character q below every base) while the search reads a real byte x there.
      // If the probe's character q is GREATER than x, the outlier cannot matter: LowerBound's bisection
      // (mapping.cpp:166-180) reaches the group's first entries only when every entry it examined before
      // had a character >= the probe's, and then sees x < c and steps over the outlier -- the same range as
      // without it; UpperBound never examines its lower end; and in the directory the outlier's own prefix
      // is below the probe's, so the slots the probe reads are the undisturbed ones (or, when the prefixes
      // coincide in their first dir_bits, the outlier heads the slot as an entry smaller than the target,
      // which the key scan counts as such).  Anything else (c <= x) stays literal.  DESIGN.md section 4.
      const uint32_t sh = 2 * (kKeyWeight + kKeyChars - 1 - o.q);
      if (!(((T >> sh) & 3u) > ((k >> sh) & 3u))) dangerous = true;
    }
  }
  return dangerous ? q_min : 0u;
}
WALT_HD bool probe_is_dangerous(const StrandView& sv, const uint32_t* care, uint32_t seed_len) {
  return probe_danger_level(sv, care, seed_len) != 0;
}

// Full seed lookup for one (read, strand, seed shift): the region
// SingleEndMapping gets from counter[] + IndexRegion (mapping.cpp:265-274).
// care/slot come from the packed read.  The region is empty when the bucket is
// empty or nothing matches.
//
// Memory behaviour matters more than instruction count here: every dependent
// load of a binary search is a separate trip to HBM (the per-XCD L2 turns over
// in a few microseconds under this kernel's random traffic, so even re-touching
// a line misses).  A directory slot holds only a handful of entries, so slots of
// up to kScan entries are fetched with INDEPENDENT loads (they land in one or two
// 128-byte lines, in flight together) and searched in registers; the positions
// of the first kSmallRegion candidates come back with them.  Longer slots fall
// back to a binary search.
constexpr uint32_t kScan = 4;
constexpr uint32_t kLookupPos = 4;  // == kSmallRegion of the kernels

// equal range [a,u] of masked key T among the sorted entries [lo,hi) (slots longer than kScanMax: the slot of
// a read from a satellite or a young repeat family holds thousands of entries); false when T is not present.
// Two searches -- first entry >= T, first entry > T -- advance together, each round loading kKary pivots
// per search with INDEPENDENT loads: ~log4(n) + 1 memory round trips instead of the ~2 log2(n) + 4 of two
// binary searches (a 4,096-entry slot: 7 instead of 28), which every wavefront with such a lane waits for.
constexpr uint32_t kKary = 4;
// (bounds: b1 = first index in [lo, hi) whose masked key is >= T, b2 = first whose masked key is > T; hi when there is none)
WALT_HD void slot_kary_bounds(const StrandView& sv, uint32_t lo, uint32_t hi, uint64_t T, uint64_t M, uint32_t& b1,
                              uint32_t& b2) {
  uint32_t x1 = lo, y1 = hi;  // first index whose masked key is >= T lies in [x1, y1]
  uint32_t x2 = lo, y2 = hi;  // first index whose masked key is >  T lies in [x2, y2]
  while (y1 > x1 || y2 > x2) {
    const uint32_t n1 = y1 > x1 ? y1 - x1 : 0u, n2 = y2 > x2 ? y2 - x2 : 0u;  // (a range can only invert on an unsorted slot)
    uint32_t i1[kKary], i2[kKary];
    uint64_t k1[kKary], k2[kKary];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (uint32_t i = 0; i < kKary; ++i) {  // pivots at the odd eighths of the range (clamped copies when it is short)
      i1[i] = x1 + (uint32_t)(((uint64_t)(2 * i + 1) * n1) / (2 * kKary));
      i2[i] = x2 + (uint32_t)(((uint64_t)(2 * i + 1) * n2) / (2 * kKary));
    }
    const bool same = x1 == x2 && y1 == y2;  // the two ranges part only when they are down to the size of the region
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (uint32_t i = 0; i < kKary; ++i) {
      k1[i] = n1 ? ent_key(sv.ent[i1[i]]) & M : 0;
      k2[i] = same ? k1[i] : (n2 ? ent_key(sv.ent[i2[i]]) & M : 0);
    }
    if (n1) {
      uint32_t c = 0;  // pivots below T: a prefix of them (sorted slot)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (uint32_t i = 0; i < kKary; ++i) c += k1[i] < T ? 1u : 0u;
      uint32_t nx = x1, ny = y1;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (uint32_t i = 0; i < kKary; ++i) {
        nx = (c == i + 1) ? i1[i] + 1 : nx;
        ny = (c == i) ? i1[i] : ny;
      }
      x1 = nx; y1 = ny;
    }
    if (n2) {
      uint32_t c = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (uint32_t i = 0; i < kKary; ++i) c += k2[i] <= T ? 1u : 0u;
      uint32_t nx = x2, ny = y2;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (uint32_t i = 0; i < kKary; ++i) {
        nx = (c == i + 1) ? i2[i] + 1 : nx;
        ny = (c == i) ? i2[i] : ny;
      }
      x2 = nx; y2 = ny;
    }
  }
  b1 = x1; b2 = x2;
}
WALT_HD bool slot_kary_search(const StrandView& sv, uint32_t lo, uint32_t hi, uint64_t T, uint64_t M, uint32_t& a,
                              uint32_t& u) {
  uint32_t x1, x2;
  slot_kary_bounds(sv, lo, hi, T, M, x1, x2);
  if (x2 <= x1) return false;
  a = x1;
  u = x2 - 1;
  return true;
}

// The same two searches as a round-by-round state machine, so that a caller can advance SEVERAL slots' searches
// (both strands of a probe) in one loop: every round is eight independent, unconditional key loads per slot.
// While the two ranges coincide the eight pivots serve both searches (ranges shrink 9-fold per round); once they
// have parted each search gets four.  A finished search reads `safe`, any valid entry index.
struct KaryState {
  uint32_t x1, y1;  // first index whose masked key is >= T lies in [x1, y1]
  uint32_t x2, y2;  // first index whose masked key is >  T lies in [x2, y2]
};
WALT_HD void kary_init(KaryState& s, uint32_t lo, uint32_t hi) { s.x1 = s.x2 = lo; s.y1 = s.y2 = hi; }
WALT_HD bool kary_busy(const KaryState& s) { return s.y1 > s.x1 || s.y2 > s.x2; }
WALT_HD void kary_round(const StrandView& sv, KaryState& s, uint64_t T, uint64_t M, uint32_t safe) {
  const uint32_t n1 = s.y1 > s.x1 ? s.y1 - s.x1 : 0u, n2 = s.y2 > s.x2 ? s.y2 - s.x2 : 0u;
  const bool same = s.x1 == s.x2 && s.y1 == s.y2;
  uint32_t q[8];
  uint64_t k[8];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (uint32_t i = 0; i < 8; ++i) {
    const uint32_t both = s.x1 + (uint32_t)(((uint64_t)(2 * i + 1) * n1) / 16);
    const uint32_t own = i < 4 ? s.x1 + (uint32_t)(((uint64_t)(2 * i + 1) * n1) / 8)
                               : s.x2 + (uint32_t)(((uint64_t)(2 * (i - 4) + 1) * n2) / 8);
    const bool live = same ? n1 != 0 : (i < 4 ? n1 != 0 : n2 != 0);
    q[i] = live ? (same ? both : own) : safe;
  }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (uint32_t i = 0; i < 8; ++i) k[i] = ent_key(sv.ent[q[i]]) & M;
  // search 1 looks at pivots [0, m1), search 2 at [b2, b2 + m2)
  const uint32_t m1 = same ? 8u : 4u, b2 = same ? 0u : 4u, m2 = same ? 8u : 4u;
  if (n1) {
    uint32_t c = 0;  // pivots below T: a prefix of them (sorted slot)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (uint32_t i = 0; i < 8; ++i) c += (i < m1 && k[i] < T) ? 1u : 0u;
    uint32_t nx = s.x1, ny = s.y1;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (uint32_t i = 0; i < 8; ++i) {
      nx = (i < m1 && c == i + 1) ? q[i] + 1 : nx;
      ny = (i < m1 && c == i) ? q[i] : ny;
    }
    s.x1 = nx; s.y1 = ny;
  }
  if (n2) {
    uint32_t c = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (uint32_t i = 0; i < 8; ++i) c += (i >= b2 && i < b2 + m2 && k[i] <= T) ? 1u : 0u;
    uint32_t nx = s.x2, ny = s.y2;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (uint32_t i = 0; i < 8; ++i) {
      nx = (i >= b2 && i < b2 + m2 && c == i - b2 + 1) ? q[i] + 1 : nx;
      ny = (i >= b2 && i < b2 + m2 && c == i - b2) ? q[i] : ny;
    }
    s.x2 = nx; s.y2 = ny;
  }
}
WALT_HD bool kary_result(const KaryState& s, uint32_t& a, uint32_t& u) {
  if (s.x2 <= s.x1) return false;
  a = s.x1;
  u = s.x2 - 1;
  return true;
}

// ---- fence search (StrandView::fen) ------------------------------------------------------------------------------
// One round narrows a range [x, y] ("the first entry whose masked key is >= T -- or > T -- lies in [x, y]", as in
// KaryState) by the fences inside it: the pivots are the multiples of 2^sh in [x, y), sh the smallest of 0, 4, 8, 12,
// 16 that leaves at most 16 of them (sh == 0: the entries themselves).  Their keys are read in two dependent
// sub-rounds that touch the same one or two lines: pivots 3, 7, 11, 15 first, then the three pivots of the quarter
// those select.  c = number of pivots below the target gives the new range (fence_narrow), which lies strictly
// between two pivots, so the next level down has at most 15 fences inside it.  A 4,096-entry slot: three rounds and
// four or five lines per search, where the k-ary search over the entries took five or six rounds of eight lines.
// The result is the equal range of the masked key in a sorted slot, however it is searched.
struct FencePlan {
  uint32_t sh, first, m;  // level, fence number of pivot 0, pivots (0: the search is finished)
};
WALT_HD uint32_t fence_ceil(uint32_t x, uint32_t sh) { return x ? ((x - 1u) >> sh) + 1u : 0u; }  // ceil(x / 2^sh), no overflow
WALT_HD FencePlan fence_plan(const StrandView& sv, uint32_t x, uint32_t y) {
  FencePlan p;
  p.sh = 0; p.first = x; p.m = 0;
  if (y <= x) return p;
  uint32_t sh = 0;
  if (sv.fen[0] != nullptr) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (uint32_t k = 1; k <= kFenceLevels; ++k) {  // counts fall with the level: the last level whose predecessor has more than 16
      const uint32_t prev = 4 * (k - 1);
      sh = (fence_ceil(y, prev) - fence_ceil(x, prev) > 16u) ? 4 * k : sh;
    }
  }
  p.sh = sh;
  p.first = fence_ceil(x, sh);
  const uint32_t cnt = fence_ceil(y, sh) - p.first;  // >= 1: a level is only left for a range that holds one of its fences
  p.m = cnt < 16u ? cnt : 16u;                       // (more than 16 only without fences, or beyond 16 x 65536 entries)
  return p;
}
// where pivot i's key lies (i clamped to the plan's pivots; a finished search reads entry `safe`)
WALT_HD const uint32_t* fence_ptr(const StrandView& sv, const FencePlan& p, uint32_t i, uint32_t safe) {
  const uint32_t ii = i < p.m ? i : (p.m ? p.m - 1u : 0u);
  const uint32_t sh = p.m ? p.sh : 0u;
  const uint64_t f = p.m ? (uint64_t)p.first + ii : (uint64_t)safe;
  const uint32_t* base = reinterpret_cast<const uint32_t*>(sv.ent);
  base = sh == 4 ? sv.fen[0] : base;
  base = sh == 8 ? sv.fen[1] : base;
  base = sh == 12 ? sv.fen[2] : base;
  base = sh == 16 ? sv.fen[3] : base;
  return base + f * (sh ? 2u : 3u);
}
WALT_HD uint64_t fence_load(const uint32_t* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef uint32_t __attribute__((address_space(1))) gword;  // a global load, not a FLAT one (map_items.h load_global)
  const gword* q = reinterpret_cast<const gword*>(reinterpret_cast<uintptr_t>(p));
  const uint32_t hi = q[0], lo = q[1];
  return ((uint64_t)hi << 32) | lo;
#else
  return ((uint64_t)p[0] << 32) | p[1];
#endif
}
WALT_HD void fence_narrow(const FencePlan& p, uint32_t c, uint32_t& x, uint32_t& y) {
  if (!p.m) return;
  const uint32_t at_c = (p.first + c) << p.sh;                 // pivot c (used when c < m)
  const uint32_t past = ((p.first + c - 1u) << p.sh) + 1u;      // one past pivot c - 1 (used when c > 0)
  const uint32_t nx = c ? past : x, ny = c < p.m ? at_c : y;
  x = nx; y = ny;
}
// pivots below the target among A = {3, 7, 11, 15} (strict: key < T, else key <= T): a prefix, the quarter the B pivots come from
WALT_HD uint32_t fence_count4(const FencePlan& p, const uint64_t* k, uint32_t first_i, uint32_t step, uint32_t n, uint64_t T,
                              uint64_t M, bool strict) {
  uint32_t c = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (uint32_t j = 0; j < 4; ++j) {
    const uint64_t v = k[j] & M;
    const bool below = strict ? v < T : v <= T;
    c += (j < n && first_i + step * j < p.m && below) ? 1u : 0u;
  }
  return c;
}
// one lane, one slot, both searches in lock-step (the kernels' dual-strand form is map_common.h fence_round_dual)
WALT_HD void slot_fence_bounds(const StrandView& sv, uint32_t lo, uint32_t hi, uint64_t T, uint64_t M, uint32_t& o1, uint32_t& o2) {
  uint32_t x1 = lo, y1 = hi, x2 = lo, y2 = hi;
  while (y1 > x1 || y2 > x2) {
    const FencePlan p1 = fence_plan(sv, x1, y1), p2 = fence_plan(sv, x2, y2);
    uint64_t a1[4], a2[4], b1[4], b2[4];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (uint32_t j = 0; j < 4; ++j) {
      a1[j] = fence_load(fence_ptr(sv, p1, 4 * j + 3, lo));
      a2[j] = fence_load(fence_ptr(sv, p2, 4 * j + 3, lo));
    }
    const uint32_t q1 = fence_count4(p1, a1, 3, 4, 4, T, M, true), q2 = fence_count4(p2, a2, 3, 4, 4, T, M, false);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (uint32_t j = 0; j < 3; ++j) {
      b1[j] = fence_load(fence_ptr(sv, p1, 4 * q1 + j, lo));
      b2[j] = fence_load(fence_ptr(sv, p2, 4 * q2 + j, lo));
    }
    b1[3] = b2[3] = 0;
    fence_narrow(p1, 4 * q1 + fence_count4(p1, b1, 4 * q1, 1, 3, T, M, true), x1, y1);
    fence_narrow(p2, 4 * q2 + fence_count4(p2, b2, 4 * q2, 1, 3, T, M, false), x2, y2);
  }
  o1 = x1; o2 = x2;
}
WALT_HD bool slot_fence_search(const StrandView& sv, uint32_t lo, uint32_t hi, uint64_t T, uint64_t M, uint32_t& a, uint32_t& u) {
  uint32_t x1, x2;
  slot_fence_bounds(sv, lo, hi, T, M, x1, x2);
  if (x2 <= x1) return false;
  a = x1;
  u = x2 - 1;
  return true;
}
// [b1, b2) = the entries of [lo, hi) whose masked key equals T (b1 == b2: where it would be) -- [lo, hi) sorted on the masked key
WALT_HD void masked_bounds(const StrandView& sv, uint32_t lo, uint32_t hi, uint64_t T, uint64_t M, uint32_t& b1, uint32_t& b2) {
  if (hi <= lo) { b1 = b2 = lo; return; }
  if (sv.fen[0] != nullptr) slot_fence_bounds(sv, lo, hi, T, M, b1, b2);
  else slot_kary_bounds(sv, lo, hi, T, M, b1, b2);
}

// The same narrowing for care chars >= 44 on a key-equal range of at most kLookupPos slots whose
// genome positions are already in registers (the usual case for reads longer than ~134 bp: one
// candidate).  lit_region would fetch, per character, the slot and then the genome word -- ten
// dependent loads for five characters; here each candidate's two genome words holding chars
// 44..seed_len-1 (at most 15 bases apart) are loaded at once and the reference's
// LowerBound / UpperBound loops run on them in registers.  Same algorithm on the same data.
// char p of candidate k (0..3) from its two-word genome window; scalars by value so that the
// selects stay selects (a struct here was turned into an indexed load from scratch memory)
WALT_HD int small_char(uint64_t w0, uint64_t w1, uint64_t w2, uint64_t w3, uint64_t b0, uint64_t b1, uint64_t b2,
                       uint64_t b3, uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3, uint64_t genome_len,
                       uint32_t k, uint32_t p) {
  uint64_t wk = w3, bk = b3;
  uint32_t pk = p3;
  if (k == 2) { wk = w2; bk = b2; pk = p2; }
  if (k == 1) { wk = w1; bk = b1; pk = p1; }
  if (k == 0) { wk = w0; bk = b0; pk = p0; }
  const uint64_t q = (uint64_t)pk + care_pos(p);
  if (q >= genome_len) return -1;
  return (int)((wk >> (2 * (uint32_t)(q - bk))) & 3u);
}
WALT_HD Region lit_region_small(const StrandView& sv, const uint32_t* care, uint32_t seed_len, uint32_t a,
                                uint32_t size, uint32_t* pos /*[kLookupPos], in/out*/, uint32_t& npos) {
  const uint32_t pc = care_pos(kKeyWeight + kKeyChars);
  const uint32_t p0 = pos[0];
  const uint32_t p1 = size > 1 ? pos[1] : p0;
  const uint32_t p2 = size > 2 ? pos[2] : p0;
  const uint32_t p3 = size > 3 ? pos[3] : p0;
  // g2 carries kG2PadWords of slack behind the genome, so the two-word window is always readable
  const uint64_t x0 = ((uint64_t)p0 + pc) >> 4, x1 = ((uint64_t)p1 + pc) >> 4;
  const uint64_t x2 = ((uint64_t)p2 + pc) >> 4, x3 = ((uint64_t)p3 + pc) >> 4;
  const uint32_t a0 = sv.g2[x0], a1 = sv.g2[x0 + 1], c0 = sv.g2[x1], c1 = sv.g2[x1 + 1];
  const uint32_t d0 = sv.g2[x2], d1 = sv.g2[x2 + 1], e0 = sv.g2[x3], e1 = sv.g2[x3 + 1];
  const uint64_t w0 = (uint64_t)a0 | ((uint64_t)a1 << 32), b0 = x0 << 4;
  const uint64_t w1 = (uint64_t)c0 | ((uint64_t)c1 << 32), b1 = x1 << 4;
  const uint64_t w2 = (uint64_t)d0 | ((uint64_t)d1 << 32), b2 = x2 << 4;
  const uint64_t w3 = (uint64_t)e0 | ((uint64_t)e1 << 32), b3 = x3 << 4;
  const uint64_t glen = sv.genome_len;
  uint32_t l = 0, u = size - 1;
  for (uint32_t p = kKeyWeight + kKeyChars; p < seed_len; ++p) {
    const int ch = (int)care_char(care, p);
    uint32_t lo = l, hi = u;
    while (lo < hi) {  // LowerBound, mapping.cpp:166-180
      const uint32_t mid = lo + (hi - lo) / 2;
      if (small_char(w0, w1, w2, w3, b0, b1, b2, b3, p0, p1, p2, p3, glen, mid, p) >= ch) hi = mid; else lo = mid + 1;
    }
    l = lo;
    hi = u;
    while (lo < hi) {  // UpperBound, mapping.cpp:182-196
      const uint32_t mid = lo + (hi - lo + 1) / 2;
      if (small_char(w0, w1, w2, w3, b0, b1, b2, b3, p0, p1, p2, p3, glen, mid, p) <= ch) lo = mid; else hi = mid - 1;
    }
    u = lo;
    if (l == u && ch != small_char(w0, w1, w2, w3, b0, b1, b2, b3, p0, p1, p2, p3, glen, l, p)) { npos = 0; return empty_region(); }
  }
  if (l > u) { npos = 0; return empty_region(); }
  // surviving slots keep their positions: the verification does not have to load them again
  const uint32_t q0 = l == 0 ? p0 : l == 1 ? p1 : l == 2 ? p2 : p3;
  const uint32_t q1 = l == 0 ? p1 : l == 1 ? p2 : p3;
  const uint32_t q2 = l == 0 ? p2 : p3;
  const uint32_t q3 = p3;
  pos[0] = q0; pos[1] = q1; pos[2] = q2; pos[3] = q3;
  npos = u - l + 1;
  Region r; r.l = a + l; r.u = a + u; return r;
}

// Slots of kScan < ne <= kScanMax entries: the entries behind the first kScan are fetched kScan at a
// time with INDEPENDENT loads and counted like the first ones (one memory round trip per round; the
// binary search needs ~log2(ne) + 2 dependent ones, and in a wavefront of 64 lanes x 2 strands some lane
// has such a slot at nearly every probe: 5.8 % of the slots that hold a read's true position have more
// than four entries at 1.44 entries per slot).  The slot is sorted by masked key (no dangerous probe
// comes here), so the entries equal to T are the n_eq behind the n_lt smaller ones and the i-th equal
// entry met is slot lo + n_lt + i; the scan stops at the first larger key.
constexpr uint32_t kScanMax = 20;
WALT_HD void slot_scan_more(const StrandView& sv, uint32_t lo, uint32_t ne, uint64_t T, uint64_t M, uint32_t& n_lt,
                            uint32_t& n_eq, uint32_t* pos /*[kLookupPos]: first equal entries' positions, in/out*/) {
  uint32_t p0 = pos[0], p1 = pos[1], p2 = pos[2], p3 = pos[3];
  bool more = n_lt + n_eq == kScan;  // nothing larger than T among the first kScan
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (uint32_t base = kScan; more && base < ne; base += kScan) {
    Ent e[kScan];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (uint32_t j = 0; j < kScan; ++j) e[j] = sv.ent[lo + (base + j < ne ? base + j : ne - 1)];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (uint32_t j = 0; j < kScan; ++j) {
      const uint64_t k = ent_key(e[j]) & M;
      const bool in = base + j < ne;
      const bool eq = in && k == T;
      p0 = (eq && n_eq == 0) ? e[j].pos : p0;
      p1 = (eq && n_eq == 1) ? e[j].pos : p1;
      p2 = (eq && n_eq == 2) ? e[j].pos : p2;
      p3 = (eq && n_eq == 3) ? e[j].pos : p3;
      n_lt += (in && k < T) ? 1u : 0u;
      n_eq += eq ? 1u : 0u;
      more = more && !(in && k > T);
    }
  }
  pos[0] = p0; pos[1] = p1; pos[2] = p2; pos[3] = p3;
}

// (A/B and the harness's cross-check: 0 = every dangerous probe searched literally over its whole bucket, as before round 3)
#if defined(__HIP_DEVICE_COMPILE__)
WALT_HD bool literal_from_level() { return true; }
WALT_HD int literal_mode() { return 2; }
WALT_HD unsigned long long* memo_stats() { return nullptr; }
#else
inline bool& literal_from_level_flag() { static bool on = true; return on; }
inline bool literal_from_level() { return literal_from_level_flag(); }
// the memoised literal search (lit_region_memo) on / off, and -- harness only -- its counters: [0] dangerous probes searched,
// [1] of them with the memo, [2] entry loads the memoised searches made, [3] bytes the reference reads in them
// 0: lit_region, 1: lit_region_memo, 2: lit_region_inferred (the product's)
inline int& literal_mode_flag() { static int mode = 2; return mode; }
inline int literal_mode() { return literal_mode_flag(); }
inline unsigned long long*& memo_stats() { static unsigned long long* p = nullptr; return p; }
#endif

// ---- the literal search, inferred (round 4) ---------------------------------------------------------------------------
// IndexRegion (mapping.cpp:198-222) over a whole bucket [first, second) for a probe the key search may not answer
// (probe_danger_level), with the reference's own mids -- but an entry is only loaded where its byte cannot be known.
// Notation: T = the probe's characters; G_p = the index range of the entries whose makedb sort key (reference.cpp:258-288:
// real characters, then "beyond the chromosome's end", which ranks below every base and ends the comparison) equals T's
// on the characters 12..p-1; its HEADS = the entries of G_p whose character p lies beyond their chromosome (outliers with
// q = p that share T's characters in front of it: the outlier table lists them all, their number is k); the rest of G_p,
// its STRETCH [cl + k, cu), holds entries whose character p is real, in ascending order of it.
//   * A step p whose range is exactly G_p with no heads is an ordinary bisection over sorted characters: it returns the
//     entries of G_p whose character p is T's = G_(p+1), or the empty marker.  Runs of such steps are taken at once: the
//     equal range of T's characters 12..p'-1 inside the stretch (masked_bounds: the masked keys are monotone there -- an
//     entry with a "beyond" mark among those characters either differs from T in front of the mark, where it is sorted on
//     real characters, or is a head of one of T's own groups, and p' is chosen as the next level that has heads).
//   * Any other step -- heads in G_p, or a range that still holds entries from outside G_p (heads of earlier levels, or
//     entries the reference's bisection was misled to keep) -- is SIMULATED with the reference's mids: a mid inside the
//     stretch has character p >= T's exactly when it lies at or behind the stretch's first entry with such a character
//     (LB), <= T's exactly when it lies in front of UB (one equal-range search in the stretch gives both); any other mid
//     is loaded (through the memo) and its real byte compared, as the reference does.
// Every comparison therefore has the outcome the reference computes; tests/test_harness_cpu.py compares the result with
// lit_region over the whole bucket for every dangerous probe.  Diagnostic counters: searches made, entries loaded.
struct InferStats { uint32_t searches, loads, steps, dirs; };
// [b1, b2) = the entries of the sorted stretch [lo, hi) whose characters 12..pe equal the probe's.  While the prefix code
// of the characters 0..pe fits the directory's depth they are two directory words (independent loads, one round trip
// however large the bucket).  R[v] = 1 + the largest index of an entry whose REAL prefix is below v.  Inside the
// stretch every entry is sorted on real characters and every entry behind the stretch is larger; the entries an index
// holds out of place are chromosome-end entries, which makedb puts IN FRONT of where their real characters belong
// ("beyond the end" ranks lowest) -- they can sit in front of the stretch with a prefix at or above v, which does not
// raise R[v], or with one below v, and then they lie in front of the stretch anyway.  So R[v] is the stretch's first entry
// with prefix >= v whenever some entry of the stretch lies in front of that one, and at most lo otherwise: max(R[v], lo).
WALT_HD void stretch_bounds(const StrandView& sv, const uint32_t* care, uint32_t Bd, uint64_t T, uint32_t pe, uint32_t lo,
                            uint32_t hi, uint32_t& b1, uint32_t& b2, InferStats* st) {
  if (hi <= lo) { b1 = b2 = lo; return; }
  uint64_t acc = 0;
  uint32_t nb = 0;
  for (uint32_t i = 0; i <= pe; ++i) {
    const uint32_t c = care_char(care, i);
    const uint32_t len = pcode_len(c, sv.ga);
    acc = (acc << len) | pcode_bits(c, sv.ga);
    nb += len;
  }
  if (Bd != 0 && nb <= Bd && sv.dir != nullptr) {
    const uint32_t v_lo = (uint32_t)(acc << (Bd - nb)), span = nb < Bd ? 1u << (Bd - nb) : 1u;
    const uint32_t slot = dir_top(Bd) - v_lo;
    const uint32_t d = (sv.dir + (uint32_t)(slot - 1u))[1], e = sv.dir[(uint32_t)(slot - span)];
    if (st) ++st->dirs;
    b1 = d > lo ? d : lo;
    b2 = e > lo ? e : lo;
    b1 = b1 < hi ? b1 : hi;
    b2 = b2 < hi ? b2 : hi;
    return;
  }
  const uint64_t M = key_mask_fwd(pe - kKeyWeight + 1);
  masked_bounds(sv, lo, hi, T & M, M, b1, b2);
  if (st) ++st->searches;
}
template <class QOf>
WALT_HD Region lit_region_inferred(const StrandView& sv, const uint32_t* care, uint32_t seed_len, uint32_t first,
                                   uint32_t second, const QOf& q_of, InferStats* st = nullptr, uint32_t Bd = 0,
                                   bool levels_known = false, uint32_t levels_in = 0, uint64_t counts_in = 0,
                                   bool beyond_in = true) {
  const uint32_t h = care[0] >> 8;
  const uint32_t lim = seed_len < kKeyWeight + kKeyChars ? seed_len : kKeyWeight + kKeyChars;
  const uint64_t T = target_key(care);
  // levels with heads: bit q - 12 for every outlier with q < lim that shares T's characters 12..q-1 -- from the level
  // table (four independent look-ups a round), or by walking the bucket's outliers
  uint32_t levels = levels_in, o_lo = 0;
  uint64_t counts = counts_in;
  bool beyond = beyond_in;
  bool have_table = levels_known;  // (the caller's danger test has looked the levels up already)
  if (!levels_known) {
    bool dng;
    uint32_t q_first;
    have_table = olev_relevant(sv, h, T, lim, levels, dng, q_first, &counts, &beyond);
    if (!have_table) beyond = true;
  }
  if (!have_table) {
    uint32_t hi = sv.n_outl;
    if (sv.outl_dir) {
      uint32_t slot_o = outl_dir_hash(h) & sv.outl_dir_mask;
      for (;;) {
        const uint32_t tag = sv.outl_dir[2 * slot_o];
        if (tag == 0) { o_lo = sv.n_outl; break; }
        if (tag == h + 1) { o_lo = sv.outl_dir[2 * slot_o + 1]; break; }
        slot_o = (slot_o + 1) & sv.outl_dir_mask;
      }
    } else {
      while (o_lo < hi) {
        const uint32_t mid = o_lo + ((hi - o_lo) >> 1);
        if (sv.outl[mid].h < h) o_lo = mid + 1; else hi = mid;
      }
    }
    for (uint32_t i = o_lo; i < sv.n_outl && sv.outl[i].h == h; ++i) {
      const Outlier o = sv.outl[i];
      if (o.q >= lim) continue;
      const uint64_t k = ((uint64_t)o.key_hi << 32) | o.key_lo;
      if (((T ^ k) & key_mask_fwd(o.q - kKeyWeight)) == 0) levels |= 1u << (o.q - kKeyWeight);
    }
  }
  LitMemo memo;
  memo_init(memo);
  // (the memo serves here as a cache of loaded entries only: an entry is loaded where the stretch's bounds say nothing --
  // heads, entries from outside the group -- and no such entry is an anchor FACT S could trust.  So q is not worked out:
  // on the device that is a bisection over the chromosome starts, a dozen dependent loads per loaded entry, and a probe
  // from a low-complexity stretch loads heads at every step -- round 4: 2.5 ms for the slowest lane of every wavefront)
  struct QNone { WALT_HD uint32_t operator()(uint32_t) const { return 0u; } } q_none;
  (void)q_of;
  uint32_t p = kKeyWeight, l = first, u = second - 1, cl = first, cu = second;  // G_12 = the bucket
  while (p < lim) {
    const bool pure = l == cl && u + 1 == cu;
    if (pure && !((levels >> (p - kKeyWeight)) & 1u)) {
      // steps p .. pn - 1 have no heads: the range becomes the equal range of T's characters 12 .. pn - 1
      const uint32_t rest = levels >> (p - kKeyWeight);
      uint32_t pn = lim;
      if (rest) {
        uint32_t z = 0;
        while (!((rest >> z) & 1u)) ++z;
        pn = p + z < lim ? p + z : lim;
      }
      uint32_t b1, b2;
      stretch_bounds(sv, care, Bd, T, pn - 1, cl, cu, b1, b2, st);
      if (b2 <= b1) return empty_region();
      cl = b1; cu = b2; l = b1; u = b2 - 1;
      p = pn;
      continue;
    }
    // one simulated step
    uint32_t k = 0;
    if (((levels >> (p - kKeyWeight)) & 1u) && have_table) {
      const uint32_t small = p - kKeyWeight < 16 ? (uint32_t)((counts >> (4 * (p - kKeyWeight))) & 15u) : 15u;
      k = small < 15u ? small : (olev_find(sv, olev_fp(h, p, T & key_mask_fwd(p - kKeyWeight))) & 0x3FFFFFFFu);
    } else if ((levels >> (p - kKeyWeight)) & 1u) {
      for (uint32_t i = o_lo; i < sv.n_outl && sv.outl[i].h == h; ++i) {
        const Outlier o = sv.outl[i];
        if (o.q != p) continue;
        const uint64_t kk = ((uint64_t)o.key_hi << 32) | o.key_lo;
        if (((T ^ kk) & key_mask_fwd(p - kKeyWeight)) == 0) ++k;
      }
    }
    const uint32_t s0 = cl + k < cu ? cl + k : cu;  // the stretch [s0, cu)
    uint32_t LB, UB;
    stretch_bounds(sv, care, Bd, T, p, s0, cu, LB, UB, st);
    if (st) ++st->steps;
    const int ch = (int)care_char(care, p);
    uint32_t low = l, high = u;
    // (the probe's character is the smallest letter and no entry of this bucket reads beyond the genome's end: every byte
    // the bisection can meet is >= it, every step goes left, the lower bound stays where it is -- no entry is looked at.
    // A probe from a low-complexity stretch is in this case level after level, with heads at each.)
    if (!(ch == 0 && !beyond))
    while (low < high) {  // LowerBound, mapping.cpp:166-180
      const uint32_t mid = low + (high - low) / 2;
      const bool ge = (mid >= s0 && mid < cu) ? mid >= LB : memo_char(sv, memo, q_none, mid, p, l, u) >= ch;
      if (ge) high = mid; else low = mid + 1;
    }
    const uint32_t nl = low;
    high = u;
    while (low < high) {  // UpperBound, mapping.cpp:182-196
      const uint32_t mid = low + (high - low + 1) / 2;
      const bool le = (mid >= s0 && mid < cu) ? mid < UB : memo_char(sv, memo, q_none, mid, p, nl, u) <= ch;
      if (le) low = mid; else high = mid - 1;
    }
    l = nl;
    u = low;
    if (l == u) {  // mapping.cpp:206-211
      const bool eq = (l >= s0 && l < cu) ? (l >= LB && l < UB) : memo_char(sv, memo, q_none, l, p, l, u) == ch;
      if (!eq) { if (st) st->loads += memo.loads; return empty_region(); }
    }
    cl = LB; cu = UB;  // G_(p+1)
    ++p;
  }
  if (st) st->loads += memo.loads;
  if (p < seed_len) return lit_region(sv, care, p, seed_len, l, u);  // characters >= 44: on the genome itself
  if (l > u) return empty_region();
  Region r; r.l = l; r.u = u; return r;
}

struct Lookup {
  Region reg;
  uint32_t npos;              // pos[0..npos) are the genome positions of slots reg.l, reg.l+1, ...
  uint32_t pos[kLookupPos];
};

template <class QOf>
WALT_HD void seed_lookup_ex(const IndexView& iv, const StrandView& sv, const uint32_t* care, uint32_t slot,
                            uint32_t span, uint32_t seed_len, Lookup& out, bool known_good, const QOf& q_of) {
  out.npos = 0;
  out.reg = empty_region();
  uint32_t h = care[0] >> 8;  // getHashValue, util.hpp:175-182
  // care characters behind the 12 hashed ones; none for the shortest reads of patterns 5 / 7 (seed_len 10 / 8:
  // IndexRegion's loop over [F2SEEDKEYWEIGHT, seed_len) is empty there and the region is the whole bucket)
  uint32_t n = seed_len > kKeyWeight ? seed_len - kKeyWeight : 0u;
  if (!known_good) {
    uint32_t lv_mask = 0, q_first = 0;
    bool dng = false;
    const uint32_t lim_t = seed_len < kKeyWeight + kKeyChars ? seed_len : kKeyWeight + kKeyChars;
    uint64_t lv_counts = 0;
    bool lv_beyond = true;
    const bool lv_known = !bucket_is_bad(sv, h) && olev_relevant(sv, h, target_key(care), lim_t, lv_mask, dng, q_first, &lv_counts, &lv_beyond);
    const uint32_t level = lv_known ? (dng ? q_first : 0u) : probe_danger_level(sv, care, seed_len);
    if (level) {
      uint32_t first = sv.cnt[h], second = sv.cnt[h + 1];
      if (first == second) return;                         // mapping.cpp:271-272
      if (literal_mode() == 2 && !bucket_is_bad(sv, h)) {  // (a BAD bucket -- disorder no outlier explains -- keeps the plain search)
        InferStats ist = {0, 0, 0, 0};
        out.reg = lit_region_inferred(sv, care, seed_len, first, second, q_of, memo_stats() ? &ist : nullptr, iv.dir_bits, lv_known, lv_mask, lv_counts, lv_beyond);
        if (memo_stats()) { memo_stats()[0] += 1; memo_stats()[1] += 1; memo_stats()[2] += ist.loads; memo_stats()[3] += ist.searches; memo_stats()[4] += ist.steps;
                            const uint32_t cst = ist.loads + ist.searches + ist.dirs; memo_stats()[6 + (cst < 63 ? cst : 63)] += 1; memo_stats()[5] += 0; memo_stats()[69] += ist.dirs; }
        return;
      }
      uint32_t l0 = first, u0 = second - 1, p0 = kKeyWeight;
      if (level > kKeyWeight && literal_from_level()) {
        // the reference's range after the characters in front of `level`: the equal range of the probe's first
        // level - 12 key characters in the bucket (probe_danger_level); it holds the outliers that made the probe
        // dangerous, so it is never empty
        const uint64_t Mq = key_mask(level - kKeyWeight), Tq = target_key(care) & Mq;
        uint32_t a0 = 0, b0 = 0;
        const bool found = sv.fen[0] != nullptr ? slot_fence_search(sv, first, second, Tq, Mq, a0, b0)
                                                : slot_kary_search(sv, first, second, Tq, Mq, a0, b0);
        if (found) { l0 = a0; u0 = b0; p0 = level; }
      }
      // (a BAD bucket keeps the plain search: FACT S needs makedb's order)
      LitMemo memo;
      memo_init(memo);
      const bool use_memo = literal_mode() >= 1 && !bucket_is_bad(sv, h);
      out.reg = use_memo ? lit_region_memo(sv, care, p0, seed_len, l0, u0, memo, q_of) : lit_region(sv, care, p0, seed_len, l0, u0);
      if (memo_stats()) { memo_stats()[0] += 1; memo_stats()[1] += use_memo ? 1 : 0; memo_stats()[2] += memo.loads; memo_stats()[3] += memo.probes; }
      return;
    }
  }
  (void)iv;
  const uint32_t* dp = sv.dir + (uint32_t)(slot - 1u);  // slot S = 2^32 is held as 0 (dir_top)
  uint32_t lo = dp[1];
  uint32_t hi = span == 1 ? dp[0] : sv.dir[(uint32_t)(slot - span)];
  if (lo >= hi) return;
  uint32_t nk = n < kKeyChars ? n : kKeyChars;
  uint64_t M = key_mask(nk);
  uint64_t T = target_key(care) & M;
  uint32_t a, u;
  const uint32_t ne = hi - lo;
  if (ne <= kScanMax) {
    Ent e[kScan];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (uint32_t j = 0; j < kScan; ++j) {
      // clamped index: loads stay independent of ne and inside the slot
      e[j] = sv.ent[lo + (j < ne ? j : ne - 1)];
    }
    uint32_t n_lt = 0, n_eq = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (uint32_t j = 0; j < kScan; ++j) {
      const uint64_t k = ent_key(e[j]) & M;
      n_lt += (j < ne && k < T) ? 1u : 0u;
      n_eq += (j < ne && k == T) ? 1u : 0u;
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (uint32_t i = 0; i < kLookupPos; ++i) {
      uint32_t p = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (uint32_t j = 0; j < kScan; ++j) p = (n_lt + i == j) ? e[j].pos : p;
      out.pos[i] = p;
    }
    if (ne > kScan) slot_scan_more(sv, lo, ne, T, M, n_lt, n_eq, out.pos);
    if (n_eq == 0) return;
    a = lo + n_lt;
    u = a + n_eq - 1;
    out.npos = n_eq < kLookupPos ? n_eq : kLookupPos;
  } else {
    if (sv.fen[0] != nullptr ? !slot_fence_search(sv, lo, hi, T, M, a, u) : !slot_kary_search(sv, lo, hi, T, M, a, u)) return;
  }
  if (n > kKeyChars) {
    const uint32_t size = u - a + 1;
    if (kPat != 3 && size == 1) {
      // IndexRegion on one slot (mapping.cpp:206-211): it survives iff every remaining care character
      // matches.  No early exit, so the genome words are fetched together (pattern 7: up to 36 characters
      // over 63 bases) instead of one dependent load per character as lit_region would.
      const uint32_t pos = out.npos == 1 ? out.pos[0] : sv.ent[a].pos;
      bool ok = true;
      for (uint32_t p = kKeyWeight + kKeyChars; p < seed_len; ++p) {
        const uint64_t q = (uint64_t)pos + care_pos(p);
        const uint32_t c = g2_code(sv.g2, q < sv.genome_len ? q : 0);
        ok = ok && q < sv.genome_len && c == care_char(care, p);
      }
      if (ok) { out.reg.l = a; out.reg.u = a; out.npos = 1; out.pos[0] = pos; } else { out.npos = 0; }
    } else if (kPat == 3 && size <= kLookupPos && out.npos == size) {  // the two-word tail window is pattern 3's (15 bases)
      out.reg = lit_region_small(sv, care, seed_len, a, size, out.pos, out.npos);
    } else {
      out.npos = 0;
      out.reg = lit_region(sv, care, kKeyWeight + kKeyChars, seed_len, a, u);
    }
    return;
  }
  out.reg.l = a; out.reg.u = u;
}

WALT_HD void seed_lookup_ex(const IndexView& iv, const StrandView& sv, const uint32_t* care, uint32_t slot,
                            uint32_t span, uint32_t seed_len, Lookup& out, bool known_good = false) {
  QOfGlobal q_of;
  q_of.start_index = iv.start_index; q_of.n_chrom = iv.n_chrom;
  seed_lookup_ex(iv, sv, care, slot, span, seed_len, out, known_good, q_of);
}

WALT_HD Region seed_lookup(const IndexView& iv, const StrandView& sv, const uint32_t* care, uint32_t slot,
                           uint32_t span, uint32_t seed_len) {
  Lookup lk;
  seed_lookup_ex(iv, sv, care, slot, span, seed_len, lk);
  return lk.reg;
}

// ---------------------------------------------------------------------------
// verification: masked mismatch count, mapping.cpp:288-304.
//   rd/mask: NW words.  mask has bit 2k of word w set for every compared read
//   offset 16w+k (table part from the literal F2NOCAREDPOSITION rows + tail).
// ---------------------------------------------------------------------------
WALT_HD uint32_t popc32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __popc(x);
#else
  return (uint32_t)__builtin_popcount(x);
#endif
}
WALT_HD uint32_t funnel_r(uint32_t lo, uint32_t hi, uint32_t s) {  // (hi:lo >> s) low 32 bits, s in [0,31]
#if defined(__HIP_DEVICE_COMPILE__)
  return __funnelshift_r(lo, hi, s);
#else
  return (uint32_t)((((uint64_t)hi << 32) | lo) >> s);
#endif
}

// The same mask without a table, for any pattern (patterns 5 / 7: seeds of more than 44 care characters start at 119 / 90
// bases): the mask of seed shift 0 over ALL care characters >= 44 is a compile-time constant per word, a seed shift
// moves it up by 2 seed_i bits, and the characters at and beyond seed_len -- their read offsets rise with p -- are cut
// off at read offset `cut` = seed_i + care_pos(seed_len) (anything >= 16 kMaskWords... when seed_len == kNumCare).
constexpr uint32_t tail_care_base_word(uint32_t w) {
  uint32_t m = 0;
  for (uint32_t p = kKeyWeight + kKeyChars; p < kNumCare; ++p) {
    const uint32_t o = care_pos(p);
    if ((o >> 4) == w) m |= 1u << (2 * (o & 15u));
  }
  return m;
}
WALT_HD uint32_t tail_mask_word(uint32_t w, uint32_t lo, uint32_t hi);
template <int W>
WALT_HD uint32_t tail_care_mask_word(uint32_t seed_i, uint32_t cut) {
  constexpr uint32_t b1 = tail_care_base_word(W), b0 = W ? tail_care_base_word(W ? W - 1 : 0) : 0u;
  const uint32_t s = 2 * seed_i;  // <= 12
  const uint32_t m = (b1 << s) | (s ? b0 >> (32 - s) : 0u);
  return m & tail_mask_word(W, 0, cut);
}
// read offset from which the care characters of a seed of seed_len characters at shift seed_i stop
WALT_HD uint32_t tail_care_cut(uint32_t seed_i, uint32_t seed_len) {
  return seed_len < kNumCare ? seed_i + care_pos(seed_len) : 0xFFFFu;
}

template <int NW>
WALT_HD uint32_t count_mismatch(const uint32_t* g2, uint32_t gpos, const uint32_t* rd, const uint32_t* mask) {
  const uint32_t* g = g2 + (gpos >> 4);
  uint32_t sh = 2 * (gpos & 15);
  uint32_t mm = 0;
  uint32_t cur = g[0];
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    uint32_t nxt = g[w + 1];
    uint32_t x = funnel_r(cur, nxt, sh) ^ rd[w];
    mm += popc32((x | (x >> 1)) & mask[w]);
    cur = nxt;
  }
  return mm;
}

// the same count, and beside it the mismatches at the seed's care characters >= 44 (tail_care_mask_word): a single
// key-equal candidate survives IndexRegion iff that count is zero (mapping.cpp:206-211; patterns 5 / 7)
template <int NW, int W = 0>
WALT_HD void count_mismatch_tail_words(const uint32_t* g, uint32_t sh, const uint32_t* rd, const uint32_t* mask,
                                       uint32_t seed_i, uint32_t cut, uint32_t cur, uint32_t& mm, uint32_t& tmm) {
  if constexpr (W < NW) {
    const uint32_t nxt = g[W + 1];
    const uint32_t x = funnel_r(cur, nxt, sh) ^ rd[W];
    const uint32_t d = x | (x >> 1);
    mm += popc32(d & mask[W]);
    tmm += popc32(d & tail_care_mask_word<W>(seed_i, cut));
    count_mismatch_tail_words<NW, W + 1>(g, sh, rd, mask, seed_i, cut, nxt, mm, tmm);
  }
}
template <int NW>
WALT_HD uint32_t count_mismatch_tail(const uint32_t* g2, uint32_t gpos, const uint32_t* rd, const uint32_t* mask,
                                     uint32_t seed_i, uint32_t cut, uint32_t& tmm) {
  const uint32_t* g = g2 + (gpos >> 4);
  uint32_t mm = 0;
  tmm = 0;
  count_mismatch_tail_words<NW>(g, 2 * (gpos & 15), rd, mask, seed_i, cut, g[0], mm, tmm);
  return mm;
}

// the same count on NW + 1 window words held in registers, the window starting sh / 2 bases into g[0]
template <int NW>
WALT_HD uint32_t count_mismatch_regs(const uint32_t* g /*[NW + 1]*/, uint32_t sh, const uint32_t* rd, const uint32_t* mask) {
  uint32_t mm = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int w = 0; w < NW; ++w) {
    uint32_t x = funnel_r(g[w], g[w + 1], sh) ^ rd[w];
    mm += popc32((x | (x >> 1)) & mask[w]);
  }
  return mm;
}

// the same, and beside it the mismatches under a second mask (the seed's care characters >= 44, which the verifier
// tests for the candidates of a key-equal range: DESIGN.md section 4b); tmask all zero: tmm = 0
template <int NW>
WALT_HD uint32_t count_mismatch_regs2(const uint32_t* g /*[NW + 1]*/, uint32_t sh, const uint32_t* rd, const uint32_t* mask,
                                      const uint32_t* tmask, uint32_t& tmm) {
  uint32_t mm = 0, t = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int w = 0; w < NW; ++w) {
    const uint32_t x = funnel_r(g[w], g[w + 1], sh) ^ rd[w];
    const uint32_t d = x | (x >> 1);
    mm += popc32(d & mask[w]);
    t += popc32(d & tmask[w]);
  }
  tmm = t;
  return mm;
}
// bit 2k of word w set for every read offset 16 w + k that holds care character p of seed shift seed_i, p in [p0, p1)
WALT_HD uint32_t care_mask_word(uint32_t w, uint32_t seed_i, uint32_t p0, uint32_t p1) {
  uint32_t m = 0;
  for (uint32_t p = p0; p < p1; ++p) {
    const uint32_t o = seed_i + care_pos(p);
    if ((o >> 4) == w) m |= 1u << (2 * (o & 15u));
  }
  return m;
}

// compare mask word w for (seed shift, repeats, read_len): table part + tail
// [3*repeats + seed_i, read_len)  (mapping.cpp:299)
WALT_HD uint32_t tail_mask_word(uint32_t w, uint32_t lo, uint32_t hi) {
  uint32_t b0 = 16 * w, b1 = b0 + 16;
  uint32_t a = lo > b0 ? lo : b0;
  uint32_t b = hi < b1 ? hi : b1;
  if (a >= b) return 0;
  uint32_t na = a - b0, nb = b - b0;  // bases [na, nb) of this word
  uint32_t m = nb == 16 ? 0x55555555u : ((1u << (2 * nb)) - 1u) & 0x55555555u;
  uint32_t below = ((1u << (2 * na)) - 1u);
  return m & ~below;
}
// mask_table layout: [kPat][kMaxRepeats - kMinRepeats + 1][kMaskWords]
WALT_HD uint32_t mask_table_index(uint32_t seed_i, uint32_t repeats, uint32_t w) {
  return (seed_i * (kMaxRepeats - kMinRepeats + 1) + (repeats - kMinRepeats)) * kMaskWords + w;
}
WALT_HD uint32_t compare_mask_word(const uint32_t* mask_table, uint32_t seed_i, uint32_t repeats,
                                   uint32_t read_len, uint32_t w) {
  uint32_t t = w < kMaskWords ? mask_table[mask_table_index(seed_i, repeats, w)] : 0u;
  return t | tail_mask_word(w, kPat * repeats + seed_i, read_len);
}

// ---------------------------------------------------------------------------
// fold of one region's candidates into the running BestMatch,
// mapping.cpp:306-313.  Index slots are distinct positions, so inside ONE
// region consecutive equal-best candidates always differ; only the first of
// them can coincide with the carried-in position.
// RegionSummary is an associative, order-aware reduction over the candidates.
// ---------------------------------------------------------------------------
struct RegionSummary {
  uint32_t min_mm;  // 0xFFFFFFFF when no candidate passed the edge filters
  uint32_t count;   // candidates with mm == min_mm
  uint32_t first;   // position of the first such candidate
  uint32_t last;    // position of the last
};
WALT_HD RegionSummary summary_empty() { RegionSummary s; s.min_mm = 0xFFFFFFFFu; s.count = 0; s.first = 0; s.last = 0; return s; }
WALT_HD RegionSummary summary_one(uint32_t mm, uint32_t pos) { RegionSummary s; s.min_mm = mm; s.count = 1; s.first = pos; s.last = pos; return s; }
// a precedes b in candidate order.  Written field by field with selects: a
// struct-valued `cond ? a : b` makes hipcc spill both operands to scratch.
WALT_HD RegionSummary summary_merge(const RegionSummary& a, const RegionSummary& b) {
  const bool has_a = a.count != 0, has_b = b.count != 0;
  const bool take_a = has_a && (!has_b || a.min_mm <= b.min_mm);   // a contributes its first/min
  const bool take_b = has_b && (!has_a || b.min_mm <= a.min_mm);   // b contributes its last
  RegionSummary s;
  s.min_mm = take_a ? a.min_mm : b.min_mm;
  s.count = (take_a ? a.count : 0u) + (take_b ? b.count : 0u);
  s.first = take_a ? a.first : b.first;
  s.last = take_b ? b.last : a.last;
  return s;
}
WALT_HD void fold_region(BestMatch& best, const RegionSummary& s, uint32_t strand_char) {
  if (s.count == 0) return;
  if (s.min_mm < best.mismatch) {
    best.genome_pos = s.last; best.times = s.count; best.strand = strand_char; best.mismatch = s.min_mm;
  } else if (s.min_mm == best.mismatch) {
    uint32_t same = (s.first == best.genome_pos) ? 1u : 0u;
    if (s.count > same) {
      best.times += s.count - same;
      best.genome_pos = s.last;
      best.strand = strand_char;
    }
  }
}

// ---------------------------------------------------------------------------
// Paired-end top-k: std::priority_queue<CandidatePosition> as libstdc++ 11
// implements it (bits/stl_heap.h: __push_heap:134, __adjust_heap:223,
// __pop_heap:253), comparator = mismatch only (paired.hpp:39-41).  Entries are
// (pos, packed = mismatch | strand_bit << 31) pairs; strand_bit 1 = '-'.
// ---------------------------------------------------------------------------
struct HeapEnt {
  uint32_t pos, mms;
};
WALT_HD uint32_t heap_mm(const HeapEnt& e) { return e.mms & 0x7FFFFFFFu; }

// H is anything indexable like HeapEnt* (an HBM array, or a strided view of LDS in the paired-end kernel)
template <class H>
WALT_HD void heap_sift_up(H h, uint32_t hole, uint32_t top, HeapEnt v) {  // __push_heap
  while (hole > top) {
    uint32_t parent = (hole - 1) / 2;
    if (!(heap_mm(h[parent]) < heap_mm(v))) break;
    h[hole] = h[parent];
    hole = parent;
  }
  h[hole] = v;
}
template <class H>
WALT_HD void heap_push(H h, uint32_t& size, HeapEnt v) {  // push_back + push_heap
  heap_sift_up(h, size, 0, v);
  ++size;
}
template <class H>
WALT_HD void heap_adjust(H h, uint32_t hole, uint32_t len, HeapEnt v) {  // __adjust_heap
  const uint32_t top = hole;
  uint32_t child = hole;
  while ((int32_t)child < ((int32_t)len - 1) / 2) {
    child = 2 * (child + 1);
    if (heap_mm(h[child]) < heap_mm(h[child - 1])) --child;
    h[hole] = h[child];
    hole = child;
  }
  if ((len & 1) == 0 && (int32_t)child == ((int32_t)len - 2) / 2) {
    child = 2 * (child + 1);
    h[hole] = h[child - 1];
    hole = child - 1;
  }
  heap_sift_up(h, hole, top, v);
}
// pop_heap + pop_back; returns the removed top
template <class H>
WALT_HD HeapEnt heap_pop(H h, uint32_t& size) {
  HeapEnt topv = h[0];
  if (size > 1) {
    HeapEnt v = h[size - 1];
    h[size - 1] = topv;
    heap_adjust(h, 0, size - 1, v);
  }
  --size;
  return topv;
}
// TopCandidates::Push, paired.hpp:63-70
template <class H>
WALT_HD void topk_push(H h, uint32_t& size, uint32_t k, HeapEnt v) {
  if (size < k) {
    heap_push(h, size, v);
  } else if (heap_mm(v) < heap_mm(h[0])) {
    heap_pop(h, size);
    heap_push(h, size, v);
  }
}

// ---------------------------------------------------------------------------
// Paired-end merge, paired.cpp:98-104, 296-331, 474-545.
// ---------------------------------------------------------------------------
struct Candidate {  // CandidatePosition, paired.hpp:35-46 (12 bytes)
  uint32_t genome_pos;
  uint32_t strand;  // low byte '+' / '-'
  uint32_t mismatch;
};
struct PairResult {
  BestMatch m1, m2;
  uint32_t best_times;
  int32_t frag_len;
  int32_t best_i, best_j;
  uint32_t pair_mm;
  uint32_t pad_[3];
};

WALT_HD void forward_pos(uint32_t gp, uint32_t strand_char, uint32_t chr, uint32_t read_len,
                         const uint32_t* start_index, uint32_t& s, uint32_t& e) {
  uint32_t len = start_index[chr + 1] - start_index[chr];
  uint32_t v = gp - start_index[chr];
  s = strand_char == '+' ? v : len - v - read_len;
  e = s + read_len;
}
WALT_HD void best4single(const Candidate* r, int n, BestMatch& best) {  // paired.cpp:296-318
  for (int i = n - 1; i >= 0; --i) {
    if (r[i].mismatch < best.mismatch) {
      best.genome_pos = r[i].genome_pos; best.times = 1; best.strand = r[i].strand; best.mismatch = r[i].mismatch;
    } else if (r[i].mismatch == best.mismatch) {
      if (best.genome_pos == r[i].genome_pos) continue;
      best.genome_pos = r[i].genome_pos; best.strand = r[i].strand; best.times++;
    } else {
      break;
    }
  }
}
// length of a reported fragment (the `len` OutputBestPairedResults returns, paired.cpp:210-243): from mate 1's 5'
// end to mate 2's, i.e. the same span GetFragmentLength measured when the pair was accepted
WALT_HD int pair_len(const Candidate& r1, const Candidate& r2, uint32_t len1, uint32_t len2,
                     const uint32_t* start_index, uint32_t n_chrom) {
  uint32_t s1, e1, s2, e2;
  forward_pos(r1.genome_pos, r1.strand, chrom_id(start_index, n_chrom, r1.genome_pos), len1, start_index, s1, e1);
  forward_pos(r2.genome_pos, r2.strand, chrom_id(start_index, n_chrom, r2.genome_pos), len2, start_index, s2, e2);
  return r1.strand == '+' ? (int)(e2 - s1) : (int)(e1 - s2);
}
// Tail of MergePairedEndResults (paired.cpp:515-545): a unique best pair is reported as such, otherwise each
// mate falls back to its own best candidate (GetBestMatch4Single).
WALT_HD void pair_finish(const Candidate* r1, int n1, const Candidate* r2, int n2, uint32_t len1, uint32_t len2,
                         const uint32_t* start_index, uint32_t n_chrom, uint32_t max_mm, int bi, int bj,
                         uint32_t best_times, PairResult& out) {
  BestMatch init; init.genome_pos = 0; init.times = 0; init.strand = '+'; init.mismatch = max_mm;
  out.m1 = init; out.m2 = init;
  out.best_times = best_times; out.frag_len = 0; out.best_i = -1; out.best_j = -1; out.pair_mm = 0;
  out.pad_[0] = out.pad_[1] = out.pad_[2] = 0;
  if (best_times == 1) {
    out.best_i = bi; out.best_j = bj;
    out.frag_len = pair_len(r1[bi], r2[bj], len1, len2, start_index, n_chrom);
    out.pair_mm = r1[bi].mismatch + r2[bj].mismatch;
    out.m1.genome_pos = r1[bi].genome_pos; out.m1.times = 1; out.m1.strand = r1[bi].strand; out.m1.mismatch = r1[bi].mismatch;
    out.m2.genome_pos = r2[bj].genome_pos; out.m2.times = 1; out.m2.strand = r2[bj].strand; out.m2.mismatch = r2[bj].mismatch;
  } else {
    best4single(r1, n1, out.m1);
    best4single(r2, n2, out.m2);
  }
}
WALT_HD void pair_merge(const Candidate* r1, int n1, const Candidate* r2, int n2, uint32_t len1, uint32_t len2,
                        const uint32_t* start_index, uint32_t n_chrom, int frag_range, uint32_t max_mm,
                        PairResult& out) {
  int bi = -1, bj = -1;
  uint32_t min_mm = max_mm;
  uint64_t best_pos = 0;
  uint32_t best_times = 0;
  for (int i = n1 - 1; i >= 0; --i) {
    uint32_t c1 = 0xFFFFFFFFu;
    for (int j = n2 - 1; j >= 0; --j) {
      if (r1[i].strand == r2[j].strand) continue;
      uint32_t mm = r1[i].mismatch + r2[j].mismatch;
      if (mm > min_mm) break;
      if (c1 == 0xFFFFFFFFu) c1 = chrom_id(start_index, n_chrom, r1[i].genome_pos);
      uint32_t c2 = chrom_id(start_index, n_chrom, r2[j].genome_pos);
      if (c1 != c2) continue;
      uint32_t s1, e1, s2, e2;
      forward_pos(r1[i].genome_pos, r1[i].strand, c1, len1, start_index, s1, e1);
      forward_pos(r2[j].genome_pos, r2[j].strand, c2, len2, start_index, s2, e2);
      int frag = r1[i].strand == '+' ? (int)(e2 - s1) : (int)(e1 - s2);
      if (frag <= 0 || frag > frag_range) continue;
      uint64_t cur = ((uint64_t)r1[i].genome_pos << 32) + r2[j].genome_pos;
      if (mm < min_mm) {
        bi = i; bj = j; best_times = 1; min_mm = mm; best_pos = cur;
      } else if (mm == min_mm && cur != best_pos) {
        bi = i; bj = j; best_times++;
      }
    }
  }
  pair_finish(r1, n1, r2, n2, len1, len2, start_index, n_chrom, max_mm, bi, bj, best_times, out);
}

}  // namespace walt
#endif  // WALT_AMD_CORE_H_
