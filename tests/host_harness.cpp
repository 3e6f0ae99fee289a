// host_harness.cpp -- CPU unit-test harness for the per-lane building blocks of
// the HIP kernels (walt_amd/csrc/core.h, index_core.h).
//
// TEST CODE ONLY.  It compiles the SAME inline functions the kernels use with
// g++ and drives them one read at a time, so search / packing / verification /
// fold / heap logic can be checked against the oracle without a GPU.  The
// wave-cooperative parts of the kernels are exercised only by the -m gpu tests.
// This is not a product path: nothing in walt_amd/ links or loads it.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <random>
#include <vector>

#include "../walt_amd/csrc/host_common.h"
#include "../walt_amd/csrc/index_core.h"

using namespace walt;

struct HStrand {
  std::vector<uint32_t> g2, cnt, bad, dir;
  std::vector<Ent> ent;
  std::vector<Outlier> outl;
  std::vector<uint32_t> outl_dir;
  std::vector<OlevEnt> olev;
  std::vector<uint32_t> fen[kFenceLevels];
  StrandView view;
};
struct HIndex {
  HStrand s[4];
  bool present[4];
  std::vector<uint32_t> start;
  IndexView view;
};

template <int NW>
static void verify_region(const IndexView& iv, const StrandView& sv, const Lookup& lk, uint32_t seed_i,
                          uint32_t len, const uint32_t* rd, const uint32_t* mk, RegionSummary& sum) {
  const Region& reg = lk.reg;
  for (uint32_t j = reg.l; j <= reg.u; ++j) {
    uint32_t pos = (j - reg.l) < lk.npos ? lk.pos[j - reg.l] : sv.ent[j].pos;
    uint32_t chr = chrom_id(iv.start_index, iv.n_chrom, pos);
    if (pos - iv.start_index[chr] < seed_i) continue;
    uint32_t gp = pos - seed_i;
    if (gp + len >= iv.start_index[chr + 1]) continue;
    uint32_t mm = count_mismatch<NW>(sv.g2, gp, rd, mk);
    sum = summary_merge(sum, summary_one(mm, gp));
  }
}

template <int W>
static long tail_mask_words_differ(uint32_t seed_i, uint32_t seed_len) {
  if constexpr (W < 10) {
    const uint32_t want = care_mask_word((uint32_t)W, seed_i, kKeyWeight + kKeyChars, seed_len);
    const uint32_t got = tail_care_mask_word<W>(seed_i, tail_care_cut(seed_i, seed_len));
    return (want != got ? 1 : 0) + tail_mask_words_differ<W + 1>(seed_i, seed_len);
  } else {
    return 0;
  }
}

// entry reads of the reference's bisection over the whole bucket (mapping.cpp:166-222): what lit_region loads
static unsigned long long lit_region_steps(const StrandView& sv, const uint32_t* care, uint32_t seed_len) {
  const uint32_t h = care[0] >> 8;
  uint32_t l = sv.cnt[h], u = sv.cnt[h + 1];
  if (l == u) return 0;
  --u;
  unsigned long long n = 0;
  for (uint32_t p = kKeyWeight; p < seed_len; ++p) {
    const int ch = (int)care_char(care, p);
    uint32_t low = l, high = u;
    while (low < high) { const uint32_t mid = low + (high - low) / 2; ++n; if (ent_char(sv, mid, p) >= ch) high = mid; else low = mid + 1; }
    l = low; high = u;
    while (low < high) { const uint32_t mid = low + (high - low + 1) / 2; ++n; if (ent_char(sv, mid, p) <= ch) low = mid; else high = mid - 1; }
    u = low;
    if (l == u) { ++n; if (ch != ent_char(sv, l, p)) break; }
  }
  return n;
}
extern "C" {

void* hh_index_new(uint32_t n_chrom, const uint32_t* chrom_len, int dir_bits) {
  HIndex* h = new HIndex();
  h->start.assign(n_chrom + 1, 0);
  for (uint32_t i = 0; i < n_chrom; ++i) h->start[i + 1] = h->start[i] + chrom_len[i];
  memset(&h->view, 0, sizeof(h->view));
  h->view.n_chrom = n_chrom;
  h->view.dir_bits = (uint32_t)dir_bits;
  h->view.dir_slots = 1u << dir_bits;
  for (int i = 0; i < 4; ++i) h->present[i] = false;
  return h;
}

// returns number of BAD buckets, or -1 on an invalid index
long hh_index_add_strand(void* hp, int strand, const uint8_t* genome, uint32_t genome_len,
                         const uint32_t* counter, const uint32_t* index, uint32_t index_size) {
  HIndex* h = reinterpret_cast<HIndex*>(hp);
  HStrand& s = h->s[strand];
  const uint32_t ga = strand >= 2;
  const uint32_t nwords = (genome_len + 15) / 16;
  s.g2.assign((size_t)nwords + 96, 0);
  for (uint32_t i = 0; i < genome_len; ++i) {
    uint32_t c = base_code(genome[i]);
    if (c > 3 || c == (ga ? 2u : 1u)) return -1;
    s.g2[i >> 4] |= c << (2 * (i & 15));
  }
  s.cnt.assign(counter, counter + kNumBuckets + 1);
  s.bad.assign(kNumBuckets / 32, 0);
  s.ent.resize((size_t)index_size + 1);
  const uint32_t* start = h->start.data();
  const uint32_t n_chrom = h->view.n_chrom;
  s.outl.clear();
  for (uint32_t j = 0; j < index_size; ++j) {
    bool t;
    if ((uint64_t)index[j] + kMinSeedLen > genome_len) return -1;
    s.ent[j] = make_ent(s.g2.data(), genome_len, index[j], t);
    uint32_t hsh = hash_at(s.g2.data(), index[j]);
    if (!(s.cnt[hsh] <= j && j < s.cnt[hsh + 1])) return -1;
    const uint32_t chr = chrom_id(start, n_chrom, index[j]);
    const uint32_t room = start[chr + 1] - index[j];
    if (room <= care_pos(kKeyWeight + kKeyChars - 1)) {
      Outlier o; o.h = hsh; o.q = first_beyond(room); o.key_hi = s.ent[j].key_hi; o.key_lo = s.ent[j].key_lo;
      s.outl.push_back(o);
    }
  }
  std::sort(s.outl.begin(), s.outl.end(), [](const Outlier& x, const Outlier& y) {
    if (x.h != y.h) return x.h < y.h;
    if (x.q != y.q) return x.q < y.q;
    if (x.key_hi != y.key_hi) return x.key_hi < y.key_hi;
    return x.key_lo < y.key_lo;
  });
  for (uint32_t j = 1; j < index_size; ++j) {
    const Ent a = s.ent[j - 1], b = s.ent[j];
    if (ent_key(a) > ent_key(b)) {
      uint32_t ha = hash_at(s.g2.data(), a.pos), hb = hash_at(s.g2.data(), b.pos);
      if (ha != hb) continue;
      const uint32_t first_diff = kKeyWeight + (uint32_t)(__builtin_clzll(ent_key(a) ^ ent_key(b)) >> 1);
      bool explained = false;
      for (const Ent& e : {a, b}) {
        const uint32_t chr = chrom_id(start, n_chrom, e.pos);
        if (first_beyond(start[chr + 1] - e.pos) <= first_diff) explained = true;
      }
      if (!explained) s.bad[ha >> 5] |= 1u << (ha & 31);
    }
  }
  const uint32_t Bd = h->view.dir_bits, slots = h->view.dir_slots;
  // dir[S - v] = 1 + the largest index slot whose code prefix is below v (core.h StrandView::dir)
  s.dir.assign((size_t)slots + 1, 0u);
  for (uint32_t j = 0; j < index_size; ++j) {
    uint32_t v = ent_prefix(s.g2.data(), s.ent[j], ga, Bd);
    uint32_t& d = s.dir[slots - 1 - v];
    if (j + 1 > d) d = j + 1;
  }
  for (size_t k = slots; k-- > 0;)
    if (s.dir[k + 1] > s.dir[k]) s.dir[k] = s.dir[k + 1];
  s.view.g2 = s.g2.data(); s.view.cnt = s.cnt.data(); s.view.bad = s.bad.data(); s.view.dir = s.dir.data();
  s.view.ent = s.ent.data(); s.view.index_size = index_size; s.view.genome_len = genome_len; s.view.ga = ga;
  s.view.bloom = nullptr; s.view.bloom_mask = 0; s.view.outl = s.outl.data(); s.view.n_outl = (uint32_t)s.outl.size();
  s.view.outl_dir_mask = build_outlier_dir(s.outl.data(), s.view.n_outl, s.outl_dir) - 1;
  s.view.outl_dir = s.outl_dir.data();
  // the level table (core.h StrandView::olev); WALT_AMD_TEST_NO_OLEV=1: the walk over the bucket's outliers instead
  {
    std::vector<uint32_t> collided;
    const uint32_t ents = build_outlier_levels(s.outl.data(), s.view.n_outl, s.olev, collided, beyond_genome_buckets(s.g2.data(), 0, genome_len));
    for (uint32_t hb : collided) s.bad[hb >> 5] |= 1u << (hb & 31);
    const bool off = getenv("WALT_AMD_TEST_NO_OLEV") != nullptr;
    s.view.olev = off ? nullptr : s.olev.data();
    s.view.olev_mask = ents - 1;
  }
  s.view.wbits = nullptr; s.view.wrank = nullptr; s.view.win = nullptr; s.view.win2 = nullptr; s.view.wcap = 0;
  // fence keys (core.h StrandView::fen; device_index.hip k_make_fences); WALT_AMD_FENCE=0: the k-ary search instead
  {
    const char* e = getenv("WALT_AMD_FENCE");
    const bool on = index_size && !(e && atoi(e) == 0);
    for (uint32_t k = 0; k < kFenceLevels; ++k) {
      s.fen[k].clear();
      if (on)
        for (uint64_t j = 0; j < index_size; j += 1ull << (4 * (k + 1))) { s.fen[k].push_back(s.ent[j].key_hi); s.fen[k].push_back(s.ent[j].key_lo); }
      s.view.fen[k] = on ? s.fen[k].data() : nullptr;
    }
  }
  h->view.s[strand] = s.view;
  h->view.start_index = h->start.data();
  h->present[strand] = true;
  long nbad = 0;
  for (uint32_t w : s.bad) nbad += __builtin_popcount(w);
  return nbad + (long)s.outl.size();  // > 0 when the strand exercises the literal path at all
}

// test hook: force every bucket onto the literal path (or clear the flags)
void hh_index_force_bad(void* hp, int strand, int on) {
  HIndex* h = reinterpret_cast<HIndex*>(hp);
  for (uint32_t& w : h->s[strand].bad) w = on ? 0xFFFFFFFFu : 0u;
}

void hh_index_free(void* hp) { delete reinterpret_cast<HIndex*>(hp); }

// lane-serial emulation of k_map_se for NW = 64 words (any length <= 1024)
int hh_map_se(void* hp, const char* bases, const uint64_t* offsets, uint32_t n, int ag, uint32_t max_mm,
              uint32_t b, BestMatch* out, uint64_t* too_short) {
  HIndex* h = reinterpret_cast<HIndex*>(hp);
  const IndexView& iv = h->view;
  const std::vector<uint32_t>& mt = compare_mask_table();
  constexpr int NW = 64;
  std::vector<uint32_t> rec(packed_fields(NW));
  *too_short = 0;
  for (uint32_t r = 0; r < n; ++r) {
    uint32_t len = (uint32_t)(offsets[r + 1] - offsets[r]);
    if (len > kMaxReadLen) return -1;
    if (!pack_read(reinterpret_cast<const uint8_t*>(bases) + offsets[r], len, ag ? 1 : 0, iv.dir_bits, NW,
                   rec.data(), 1))
      return -2;
    BestMatch best; best.genome_pos = 0; best.times = 0; best.strand = '+'; best.mismatch = max_mm;
    const uint32_t* rd = &rec[1];
    if (len < kMinReadLen) {
      *too_short += 2;
    } else {
      uint32_t repeats = seed_repeats(len);
      for (uint32_t fi = 0; fi < 2; ++fi) {
        const StrandView& sv = iv.s[(ag ? 2 : 0) + fi];
        for (uint32_t seed_i = 0; seed_i < kPat; ++seed_i) {
          if (best.mismatch == 0 && seed_i) break;
          if (best.mismatch == 1 && seed_i >= kExitOneMismatch) break;
          const uint32_t* care = &rec[1 + NW + seed_i * kPerSeedWords];
          Lookup lk;
          seed_lookup_ex(iv, sv, care, care[kCareWords], care[kCareWords + 1], seed_len_of(repeats), lk);
          const Region reg = lk.reg;
          uint32_t size = reg.l <= reg.u ? reg.u - reg.l + 1 : 0;
          if (size == 0 || size > b) continue;
          uint32_t mk[NW];
          for (int w = 0; w < NW; ++w) mk[w] = compare_mask_word(mt.data(), seed_i, repeats, len, (uint32_t)w);
          RegionSummary sum = summary_empty();
          verify_region<NW>(iv, sv, lk, seed_i, len, rd, mk, sum);
          fold_region(best, sum, fi == 0 ? '+' : '-');
        }
      }
    }
    out[r] = best;
  }
  return 0;
}

// lane-serial emulation of the paired-end top-k kernel for one mate:
// ranked[n*top_k] in pop order, ranked_n[n].
int hh_pe_topk(void* hp, const char* bases, const uint64_t* offsets, uint32_t n, int ag, uint32_t max_mm,
               uint32_t b, uint32_t top_k, Candidate* ranked, uint32_t* ranked_n, uint64_t* too_short) {
  HIndex* h = reinterpret_cast<HIndex*>(hp);
  const IndexView& iv = h->view;
  const std::vector<uint32_t>& mt = compare_mask_table();
  constexpr int NW = 64;
  std::vector<uint32_t> rec(packed_fields(NW));
  std::vector<HeapEnt> heap(top_k + 1);
  *too_short = 0;
  for (uint32_t r = 0; r < n; ++r) {
    uint32_t len = (uint32_t)(offsets[r + 1] - offsets[r]);
    if (len > kMaxReadLen) return -1;
    if (!pack_read(reinterpret_cast<const uint8_t*>(bases) + offsets[r], len, ag ? 1 : 0, iv.dir_bits, NW,
                   rec.data(), 1))
      return -2;
    const uint32_t* rd = &rec[1];
    uint32_t hsize = 0;
    if (len < kMinReadLen) {
      *too_short += 2;
    } else {
      uint32_t repeats = seed_repeats(len);
      for (uint32_t fi = 0; fi < 2; ++fi) {
        const StrandView& sv = iv.s[(ag ? 2 : 0) + fi];
        for (uint32_t seed_i = 0; seed_i < kPat; ++seed_i) {
          bool full = hsize >= top_k;
          if (full && heap_mm(heap[0]) == 0 && seed_i) break;        // paired.cpp:133-135
          if (full && heap_mm(heap[0]) == 1 && seed_i >= kExitOneMismatch) break;   // paired.cpp:137-149
          const uint32_t* care = &rec[1 + NW + seed_i * kPerSeedWords];
          Region reg = seed_lookup(iv, sv, care, care[kCareWords], care[kCareWords + 1], seed_len_of(repeats));
          uint32_t size = reg.l <= reg.u ? reg.u - reg.l + 1 : 0;
          if (size == 0 || size > b) continue;
          uint32_t mk[NW];
          for (int w = 0; w < NW; ++w) mk[w] = compare_mask_word(mt.data(), seed_i, repeats, len, (uint32_t)w);
          for (uint32_t j = reg.l; j <= reg.u; ++j) {
            uint32_t pos = sv.ent[j].pos;
            uint32_t chr = chrom_id(iv.start_index, iv.n_chrom, pos);
            if (pos - iv.start_index[chr] < seed_i) continue;
            uint32_t gp = pos - seed_i;
            if (gp + len >= iv.start_index[chr + 1]) continue;
            uint32_t mm = count_mismatch<NW>(sv.g2, gp, rd, mk);
            if (mm > max_mm) continue;
            HeapEnt e; e.pos = gp; e.mms = mm | (fi << 31);
            topk_push(heap.data(), hsize, top_k, e);
          }
        }
      }
    }
    uint32_t k = 0;
    while (hsize) {
      HeapEnt e = heap_pop(heap.data(), hsize);
      Candidate c; c.genome_pos = e.pos; c.strand = (e.mms >> 31) ? '-' : '+'; c.mismatch = heap_mm(e);
      ranked[(uint64_t)r * top_k + k++] = c;
    }
    ranked_n[r] = k;
  }
  return 0;
}

void hh_pe_merge(void* hp, const Candidate* r1, const uint32_t* n1, const Candidate* r2, const uint32_t* n2,
                 uint32_t top_k, const uint64_t* off1, const uint64_t* off2, uint32_t n, int frag_range,
                 uint32_t max_mm, PairResult* out) {
  HIndex* h = reinterpret_cast<HIndex*>(hp);
  for (uint32_t j = 0; j < n; ++j)
    pair_merge(r1 + (uint64_t)j * top_k, (int)n1[j], r2 + (uint64_t)j * top_k, (int)n2[j],
               (uint32_t)(off1[j + 1] - off1[j]), (uint32_t)(off2[j + 1] - off2[j]), h->view.start_index,
               h->view.n_chrom, frag_range, max_mm, out[j]);
}

// the packed read records (SoA, stride words between fields) as index_core.h
// pack_read() defines them -- the specification of the device packing kernel
int hh_pack(const char* bases, const uint64_t* offsets, uint32_t n, int ga, uint32_t D, uint32_t nw, uint32_t* out,
            uint64_t stride) {
  int bad = 0;
  for (uint32_t r = 0; r < n; ++r)
    if (!pack_read(reinterpret_cast<const uint8_t*>(bases) + offsets[r], (uint32_t)(offsets[r + 1] - offsets[r]),
                   ga ? 1 : 0, D, nw, out + r, stride))
      ++bad;
  return bad;
}

// Region-level check of the exactness argument: for EVERY (read, strand, seed) probe
// compare the region of the directory/key search with the region of the literal LowerBound/UpperBound search
// over the whole bucket.  They must be equal whenever probe_is_dangerous() says the probe is safe.
// out[0] probes, out[1] safe probes whose regions differ (must be 0), out[2] probes called dangerous,
// out[3] safe probes that share an outlier's characters 12..q-1 (the class the refined rule releases).
int hh_region_check(void* hp, const char* bases, const uint64_t* offsets, uint32_t n, int ag, uint64_t* out4) {
  HIndex* h = reinterpret_cast<HIndex*>(hp);
  const IndexView& iv = h->view;
  constexpr int NW = 64;
  std::vector<uint32_t> rec(packed_fields(NW));
  out4[0] = out4[1] = out4[2] = out4[3] = 0;
  for (uint32_t r = 0; r < n; ++r) {
    uint32_t len = (uint32_t)(offsets[r + 1] - offsets[r]);
    if (len > kMaxReadLen) return -1;
    if (len < kMinReadLen) continue;
    if (!pack_read(reinterpret_cast<const uint8_t*>(bases) + offsets[r], len, ag ? 1 : 0, iv.dir_bits, NW,
                   rec.data(), 1))
      return -2;
    const uint32_t seed_len = seed_len_of(seed_repeats(len));
    for (uint32_t fi = 0; fi < 2; ++fi) {
      const StrandView& sv = iv.s[(ag ? 2 : 0) + fi];
      for (uint32_t seed_i = 0; seed_i < kPat; ++seed_i) {
        const uint32_t* care = &rec[1 + NW + seed_i * kPerSeedWords];
        ++out4[0];
        if (probe_is_dangerous(sv, care, seed_len)) {
          // the literal route: from the level the danger starts at (core.h probe_danger_level) against the whole bucket
          ++out4[2];
          // ... and the memoised search (core.h lit_region_memo: the product's route) against both, from the level and
          // over the whole bucket
          Lookup from_level, whole, memo_level, memo_whole, inferred;
          unsigned long long* const keep = memo_stats();
          memo_stats() = nullptr;  // (counted once: the product's route below)
          literal_mode_flag() = 0;
          literal_from_level_flag() = true;
          seed_lookup_ex(iv, sv, care, care[kCareWords], care[kCareWords + 1], seed_len, from_level, false);
          literal_from_level_flag() = false;
          seed_lookup_ex(iv, sv, care, care[kCareWords], care[kCareWords + 1], seed_len, whole, false);
          literal_mode_flag() = 1;
          seed_lookup_ex(iv, sv, care, care[kCareWords], care[kCareWords + 1], seed_len, memo_whole, false);
          literal_from_level_flag() = true;
          seed_lookup_ex(iv, sv, care, care[kCareWords], care[kCareWords + 1], seed_len, memo_level, false);
          literal_mode_flag() = 2;
          memo_stats() = keep;
          seed_lookup_ex(iv, sv, care, care[kCareWords], care[kCareWords + 1], seed_len, inferred, false);
          if (keep) keep[5] += lit_region_steps(sv, care, seed_len);  // entry reads of the reference's own search
          const bool e2 = whole.reg.l > whole.reg.u;
          for (const Lookup* x : {&from_level, &memo_level, &memo_whole, &inferred}) {
            const bool e1 = x->reg.l > x->reg.u;
            if (e1 != e2 || (!e1 && (x->reg.l != whole.reg.l || x->reg.u != whole.reg.u))) { ++out4[1]; break; }
          }
          continue;
        }
        Lookup fast;
        seed_lookup_ex(iv, sv, care, care[kCareWords], care[kCareWords + 1], seed_len, fast, true);
        Region lit = empty_region();
        const uint32_t hh = care[0] >> 8;
        const uint32_t first = sv.cnt[hh], second = sv.cnt[hh + 1];
        if (first != second) lit = lit_region(sv, care, kKeyWeight, seed_len, first, second - 1);
        const bool e1 = fast.reg.l > fast.reg.u, e2 = lit.l > lit.u;
        if (e1 != e2 || (!e1 && (fast.reg.l != lit.l || fast.reg.u != lit.u))) ++out4[1];
        // does it share some outlier's characters 12..q-1 (and is safe only by the refined rule)?
        const uint32_t lim = seed_len < kKeyWeight + kKeyChars ? seed_len : kKeyWeight + kKeyChars;
        const uint64_t T = target_key(care);
        for (uint32_t k = 0; k < sv.n_outl; ++k) {
          const Outlier& o = sv.outl[k];
          if (o.h != hh || o.q >= lim) continue;
          const uint64_t key = ((uint64_t)o.key_hi << 32) | o.key_lo;
          if (((T ^ key) & key_mask(o.q - kKeyWeight)) == 0) { ++out4[3]; break; }
        }
      }
    }
  }
  return 0;
}

// counters of the memoised literal searches made since the last call (core.h memo_stats): [0] dangerous probes searched,
// [1] of them through the memo, [2] entry loads they made, [3] bytes the reference's bisection reads in them
// counters since the last call (core.h memo_stats): [0] dangerous probes searched, [1] of them by the inferred search,
// [2] entries it loaded, [3] key searches it made, [4] steps it simulated, [5] entry reads of the reference's search
// [6 + c], c < 64: inferred searches that made c key searches + entry loads (63: that many or more)
void hh_memo_stats(uint64_t* out70) {
  static unsigned long long acc[70] = {};
  for (int i = 0; i < 70; ++i) { out70[i] = acc[i]; acc[i] = 0; }
  memo_stats() = acc;
}

// Care characters >= 44 narrowed by the verifier against IndexRegion: for every safe probe of a
// read whose seed has more than 44 care characters, the key-equal range [a, u] (care characters 12..43) is searched as
// the kernels do; when NO entry of the range is a run breaker (an entry with fewer than kTailBreakRoom + 1 bases of its
// chromosome behind it: device_index.hip k_make_ent / k_win_break end the dense runs there, so these are exactly the
// ranges the kernels may defer), the set of its candidates whose characters 44..seed_len-1 equal the read's --
// what item_stream keeps -- must be IndexRegion's [l, u]: contiguous, same ends, or both empty.
// out[0] long-seed safe probes with a non-empty key-equal range, out[1] of them deferrable (no breaker inside),
// out[2] deferrable probes whose set differs from lit_region's (must be 0), out[3] ranges that hold a breaker.
int hh_tail_check(void* hp, const char* bases, const uint64_t* offsets, uint32_t n, int ag, uint64_t* out4) {
  HIndex* h = reinterpret_cast<HIndex*>(hp);
  const IndexView& iv = h->view;
  constexpr int NW = 64;
  std::vector<uint32_t> rec(packed_fields(NW));
  out4[0] = out4[1] = out4[2] = out4[3] = 0;
  const uint32_t* start = h->start.data();
  for (uint32_t r = 0; r < n; ++r) {
    uint32_t len = (uint32_t)(offsets[r + 1] - offsets[r]);
    if (len > kMaxReadLen) return -1;
    if (len < kMinReadLen) continue;
    if (!pack_read(reinterpret_cast<const uint8_t*>(bases) + offsets[r], len, ag ? 1 : 0, iv.dir_bits, NW, rec.data(), 1)) return -2;
    const uint32_t seed_len = seed_len_of(seed_repeats(len));
    if (seed_len <= kKeyWeight + kKeyChars) continue;
    for (uint32_t fi = 0; fi < 2; ++fi) {
      const StrandView& sv = iv.s[(ag ? 2 : 0) + fi];
      for (uint32_t seed_i = 0; seed_i < kPat; ++seed_i) {
        const uint32_t* care = &rec[1 + NW + seed_i * kPerSeedWords];
        if (probe_is_dangerous(sv, care, seed_len)) continue;
        // the key-equal range, as the kernels find it: directory slot, then the equal range of the 32 key characters
        const uint32_t slot = care[kCareWords], span = care[kCareWords + 1];
        const uint32_t* dp = sv.dir + (uint32_t)(slot - 1u);
        const uint32_t lo = dp[1], hi = span == 1 ? dp[0] : sv.dir[(uint32_t)(slot - span)];
        if (lo >= hi) continue;
        const uint64_t M = key_mask(kKeyChars), T = target_key(care) & M;
        uint32_t a = 0, u = 0;
        if (!slot_fence_search(sv, lo, hi, T, M, a, u)) continue;
        ++out4[0];
        bool breaker = false;
        for (uint32_t j = a; j <= u; ++j) {
          const uint32_t pos = sv.ent[j].pos;
          const uint32_t chr = chrom_id(start, iv.n_chrom, pos);
          if (start[chr + 1] - pos <= kTailBreakRoom) breaker = true;
        }
        if (breaker) {
          ++out4[3];
          if (!getenv("WALT_AMD_TEST_DEFER_BREAKERS")) continue;  // (test of the test: without the rule, differences must show)
        }
        ++out4[1];
        // what the verifier keeps: candidates whose care characters 44 .. seed_len - 1 equal the read's
        uint32_t first = 0, last = 0, cnt = 0;
        bool contiguous = true;
        for (uint32_t j = a; j <= u; ++j) {
          bool eq = true;
          for (uint32_t p = kKeyWeight + kKeyChars; p < seed_len; ++p) {
            const uint64_t q = (uint64_t)sv.ent[j].pos + care_pos(p);
            eq = eq && q < sv.genome_len && g2_code(sv.g2, q) == care_char(care, p);
          }
          if (!eq) continue;
          if (cnt && j != last + 1) contiguous = false;
          if (!cnt) first = j;
          last = j;
          ++cnt;
        }
        const Region lit = lit_region(sv, care, kKeyWeight + kKeyChars, seed_len, a, u);
        const bool e2 = lit.l > lit.u;
        if (!contiguous || (cnt == 0) != e2 || (cnt && (first != lit.l || last != lit.u))) ++out4[2];
      }
    }
  }
  return 0;
}

// core.h kary_round (the heavy stages' round-by-round slot search, two slots advanced together) against
// slot_kary_search on random sorted slots: returns the number of differing answers.  Keys are drawn from few
// values so that equal ranges are long; masks of 1..32 key characters; targets present and absent.
long hh_kary_check(uint32_t seed, uint32_t n_cases) {
  uint64_t x = 0x9E3779B97F4A7C15ull * (seed + 1);
  auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
  long bad = 0;
  for (uint32_t c = 0; c < n_cases; ++c) {
    const uint32_t n = 1 + (uint32_t)(rnd() % (c % 7 == 0 ? 5000 : 90));
    const uint32_t distinct = 1 + (uint32_t)(rnd() % (c % 3 == 0 ? 4 : 40));
    const uint32_t nk = 1 + (uint32_t)(rnd() % kKeyChars);
    const uint64_t M = key_mask(nk);
    std::vector<uint64_t> vals(distinct);
    for (auto& v : vals) v = rnd();
    std::vector<uint64_t> keys(n);
    for (auto& k : keys) k = vals[rnd() % distinct];
    std::sort(keys.begin(), keys.end(), [&](uint64_t a, uint64_t b) { return (a & M) < (b & M) || ((a & M) == (b & M) && a < b); });
    std::vector<Ent> ent(n + 8);
    for (uint32_t i = 0; i < n; ++i) { ent[i].key_hi = (uint32_t)(keys[i] >> 32); ent[i].key_lo = (uint32_t)keys[i]; ent[i].pos = i; }
    StrandView sv;
    memset(&sv, 0, sizeof(sv));
    sv.ent = ent.data();
    sv.index_size = n;
    for (int t = 0; t < 6; ++t) {
      const uint64_t T = (t < 4 ? vals[rnd() % distinct] : rnd()) & M;
      const uint32_t lo = (uint32_t)(rnd() % n), hi = lo + 1 + (uint32_t)(rnd() % (n - lo));
      uint32_t a1 = 0, u1 = 0, a2 = 0, u2 = 0;
      const bool f1 = slot_kary_search(sv, lo, hi, T, M, a1, u1);
      KaryState ks, other;
      kary_init(ks, lo, hi);
      kary_init(other, 0, (uint32_t)(rnd() % 2 ? n : 0));  // a second search running beside it, as in probe_resolve_dual
      uint32_t rounds = 0;
      while (kary_busy(ks) || kary_busy(other)) {
        kary_round(sv, ks, T, M, lo);
        kary_round(sv, other, T ^ 1, M, 0);
        if (++rounds > 64) break;
      }
      const bool f2 = kary_result(ks, a2, u2);
      if (rounds > 64 || f1 != f2 || (f1 && (a1 != a2 || u1 != u2))) ++bad;
    }
  }
  return bad;
}

// slot_fence_search (core.h) on random sorted arrays of up to a few hundred thousand entries, random sub-ranges as slots:
// the equal range of a masked key must be std::equal_range's.  Returns the number of differing answers.
long hh_fence_check(uint32_t seed, uint32_t n_cases) {
  uint64_t x = 0x9E3779B97F4A7C15ull * (seed + 1);
  auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
  long bad = 0;
  for (uint32_t c = 0; c < n_cases; ++c) {
    const uint32_t n = 1 + (uint32_t)(rnd() % (c % 11 == 0 ? 300000 : (c % 3 == 0 ? 6000 : 300)));
    const uint32_t distinct = 1 + (uint32_t)(rnd() % (c % 3 == 0 ? 5 : (c % 5 == 0 ? 4000 : 60)));
    const uint32_t nk = 1 + (uint32_t)(rnd() % kKeyChars);
    const uint64_t M = key_mask(nk);
    std::vector<uint64_t> vals(distinct);
    for (auto& v : vals) v = rnd();
    std::vector<uint64_t> keys(n);
    for (auto& k : keys) k = vals[rnd() % distinct];
    std::sort(keys.begin(), keys.end(), [&](uint64_t a, uint64_t b) { return (a & M) < (b & M) || ((a & M) == (b & M) && a < b); });
    std::vector<Ent> ent(n + 8);
    for (uint32_t i = 0; i < n; ++i) { ent[i].key_hi = (uint32_t)(keys[i] >> 32); ent[i].key_lo = (uint32_t)keys[i]; ent[i].pos = i; }
    std::vector<uint32_t> fen[kFenceLevels];
    StrandView sv;
    memset(&sv, 0, sizeof(sv));
    sv.ent = ent.data();
    sv.index_size = n;
    for (uint32_t k = 0; k < kFenceLevels; ++k) {
      for (uint64_t j = 0; j < n; j += 1ull << (4 * (k + 1))) { fen[k].push_back(ent[j].key_hi); fen[k].push_back(ent[j].key_lo); }
      sv.fen[k] = fen[k].data();
    }
    for (int t = 0; t < 8; ++t) {
      const uint64_t T = (t < 5 ? vals[rnd() % distinct] : rnd()) & M;
      const uint32_t lo = t == 0 ? 0u : (uint32_t)(rnd() % n), hi = t == 0 ? n : lo + 1 + (uint32_t)(rnd() % (n - lo));
      uint32_t a1 = 0, u1 = 0;
      const bool f1 = slot_fence_search(sv, lo, hi, T, M, a1, u1);
      uint32_t a2 = lo, u2 = lo;  // reference: first index with masked key >= T, first with masked key > T
      while (a2 < hi && (keys[a2] & M) < T) ++a2;
      u2 = a2;
      while (u2 < hi && (keys[u2] & M) == T) ++u2;
      const bool f2 = u2 > a2;
      if (f1 != f2 || (f1 && (a1 != a2 || u1 != u2 - 1))) ++bad;
    }
  }
  return bad;
}

// core.h tail_care_mask_word / count_mismatch_tail (round 3: the seed's care characters >= 44 tested on the window the
// mismatch count loads -- patterns 5 / 7 -- and on the dense records): the mask built without a table must be the
// straightforward one (care_mask_word over [44, seed_len)) for every word, seed shift and seed length, and the tail
// count on random windows must be the number of those characters that differ.  Returns the number of differences.
long hh_tail_mask_check(uint32_t seed) {
  long bad = 0;
  for (uint32_t seed_i = 0; seed_i < kPat; ++seed_i)
    for (uint32_t seed_len = kKeyWeight + kKeyChars; seed_len <= kNumCare; ++seed_len) {
      if (care_pos(seed_len - 1) + seed_i >= 160) continue;  // beyond the ten words the long-seed kernels hold
      bad += tail_mask_words_differ<0>(seed_i, seed_len);
    }
  uint64_t x = 0x9E3779B97F4A7C15ull * (seed + 1);
  auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
  constexpr int NW = 10;
  for (int c = 0; c < 2000; ++c) {
    uint32_t g2[NW + 4], rd[NW], mk[NW];
    for (auto& v : g2) v = (uint32_t)rnd();
    const uint32_t gpos = (uint32_t)(rnd() % 16), seed_i = (uint32_t)(rnd() % kPat);
    const uint32_t seed_len = kKeyWeight + kKeyChars + (uint32_t)(rnd() % (kNumCare - kKeyWeight - kKeyChars + 1));
    if (seed_len > kKeyWeight + kKeyChars && care_pos(seed_len - 1) + seed_i >= 16 * NW) continue;
    for (int w = 0; w < NW; ++w) {  // the read: the window itself with a few characters changed
      rd[w] = funnel_r(g2[w], g2[w + 1], 2 * gpos);
      if (rnd() % 2) rd[w] ^= (uint32_t)(1 + rnd() % 3) << (2 * (rnd() % 16));
      mk[w] = (uint32_t)rnd() & 0x55555555u;
    }
    uint32_t want_mm = 0, want_t = 0;
    for (int w = 0; w < NW; ++w)
      for (int k = 0; k < 16; ++k) {
        const uint32_t a = (rd[w] >> (2 * k)) & 3u, b = g2_code(g2, gpos + 16 * w + k);
        if ((mk[w] >> (2 * k)) & 1u) want_mm += a != b;
      }
    for (uint32_t p = kKeyWeight + kKeyChars; p < seed_len; ++p) {
      const uint32_t o = seed_i + care_pos(p);
      want_t += ((rd[o >> 4] >> (2 * (o & 15))) & 3u) != g2_code(g2, gpos + o);
    }
    uint32_t got_t = 0;
    const uint32_t got_mm = count_mismatch_tail<NW>(g2, gpos, rd, mk, seed_i, tail_care_cut(seed_i, seed_len), got_t);
    if (got_mm != want_mm || got_t != want_t) ++bad;
  }
  return bad;
}

// The wavefront's candidate list (map_se.hip coop_lane_regions) on the CPU, step for step: 64 "lanes" with random
// regions on both strands, every candidate a random (passes the filters?, mismatches, position); per turn of 64
// candidates the owner of candidate q is found by the kernel's bisection over the lanes' exclusive counts, the
// one-candidate summaries go through the kernel's segmented inclusive scan (Hillis-Steele, keys = 2 * owner + strand,
// join when the key d lanes below is the same) with core.h's summary_merge, and every owner merges the summary of the
// last lane of its run in the turn into its own.  Each lane's two summaries must equal the in-order fold of its own
// candidates -- the loop the list replaced.  Returns the number of (trial, lane, strand) that differ.
long hh_candidate_list_check(uint32_t seed, uint32_t trials) {
  std::mt19937 rng(seed);
  long bad = 0;
  for (uint32_t t = 0; t < trials; ++t) {
    const uint32_t max_size = (t % 3 == 0) ? 4u : ((t % 3 == 1) ? 16u : 64u);
    const uint32_t busy = 1u + rng() % 100u;  // per cent of the lanes with a region on a strand
    uint32_t n_p[64], n_m[64], off[64], total = 0;
    for (int l = 0; l < 64; ++l) {
      n_p[l] = (rng() % 100u < busy) ? 1u + rng() % max_size : 0u;
      n_m[l] = (rng() % 100u < busy) ? 1u + rng() % max_size : 0u;
      off[l] = total;
      total += n_p[l] + n_m[l];
    }
    std::vector<RegionSummary> one(total);
    for (uint32_t q = 0; q < total; ++q)
      one[q] = (rng() % 4u) ? summary_one(rng() % 3u, 1000u + rng() % 50u) : summary_empty();  // few values: ties are the point
    RegionSummary want_p[64], want_m[64], got_p[64], got_m[64];
    for (int l = 0; l < 64; ++l) {
      want_p[l] = want_m[l] = got_p[l] = got_m[l] = summary_empty();
      for (uint32_t k = 0; k < n_p[l]; ++k) want_p[l] = summary_merge(want_p[l], one[off[l] + k]);
      for (uint32_t k = 0; k < n_m[l]; ++k) want_m[l] = summary_merge(want_m[l], one[off[l] + n_p[l] + k]);
    }
    for (uint32_t base = 0; base < total; base += 64) {
      RegionSummary acc[64];
      uint32_t key[64];
      for (uint32_t lane = 0; lane < 64; ++lane) {
        const uint32_t q = base + lane;
        const bool have = q < total;
        uint32_t own = 0;
        for (uint32_t st = 32; st; st >>= 1) own = off[own + st] <= q ? own + st : own;
        const uint32_t k = q - off[own];
        const bool on_m = have && k >= n_p[own];
        key[lane] = have ? 2u * own + (on_m ? 1u : 0u) : 0xFFFFFFFFu;
        acc[lane] = have ? one[q] : summary_empty();
      }
      for (uint32_t d = 1; d < 64; d <<= 1) {
        RegionSummary nxt[64];
        for (uint32_t lane = 0; lane < 64; ++lane) {
          const bool join = lane >= d && key[lane - d] == key[lane];
          nxt[lane] = join ? summary_merge(acc[lane - d], acc[lane]) : acc[lane];
        }
        for (uint32_t lane = 0; lane < 64; ++lane) acc[lane] = nxt[lane];
      }
      for (uint32_t l = 0; l < 64; ++l)
        for (int f = 0; f < 2; ++f) {
          const uint32_t lo = f ? off[l] + n_p[l] : off[l], hi = f ? off[l] + n_p[l] + n_m[l] : off[l] + n_p[l];
          const uint32_t a = lo > base ? lo : base, e = hi < base + 64 ? hi : base + 64;
          if (e <= a) continue;
          RegionSummary& sum = f ? got_m[l] : got_p[l];
          sum = summary_merge(sum, acc[e - 1 - base]);
        }
    }
    auto same = [](const RegionSummary& x, const RegionSummary& y) {
      return x.count == y.count && (x.count == 0 || (x.min_mm == y.min_mm && x.first == y.first && x.last == y.last));
    };
    for (int l = 0; l < 64; ++l) bad += (same(got_p[l], want_p[l]) ? 0 : 1) + (same(got_m[l], want_m[l]) ? 0 : 1);
  }
  return bad;
}

// expose the literal tables for tests/test_seedtab.py
int hh_pattern() { return (int)kPat; }
void hh_get_nocare(uint32_t* out /* kPat x 150 */) {
  for (uint32_t s = 0; s < kPat; ++s) memcpy(out + 150 * s, nocare_row((int)s), 150 * sizeof(uint32_t));
}

}  // extern "C"
