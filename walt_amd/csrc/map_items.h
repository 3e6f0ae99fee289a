// map_items.h -- work items for large candidate regions and the loop that streams them (single-end heavy stages,
// map_se.hip; paired-end complex reads, map_pe.hip).
//
// A region of more than a few candidates is not verified by the lane that found it (its 63 wave-mates would wait)
// nor by borrowing that lane's wavefront (its registers hold 64 reads' state: three wavefronts per SIMD); the lane
// writes an ITEM and a second kernel takes one item per wavefront.  An item carries everything the verification
// needs, so the verifier's only dependent loads are the candidates' own:
//   quad 0   id (the producer's: which read / probe), first slot l, size, first dense record or kItemDenseNone
//            (record numbers stay below 2^32, device_index.hip build_windows)
//   quad 1   read length, seed shift, tail, 0   (tail != 0: the region is a key-equal RANGE whose candidates still owe
//            the seed's care characters >= 44 -- the verifier tests them and counts the matches for -b; section 4b)
//   then     the converted read rd[NW] and its compare masks mk[NW] for that seed shift, padded to 16 bytes
// Items whose candidates all have dense records (core.h dense_range) are queued from the front of the array,
// the others from its back; each kind has its own verifier instance.
#ifndef WALT_AMD_MAP_ITEMS_H_
#define WALT_AMD_MAP_ITEMS_H_

#include "map_common.h"

namespace walt {

constexpr uint32_t kItemDenseNone = 0xFFFFFFFFu;
// Items a wavefront takes per visit to the queue's cursor: every visit is an atomic on ONE address, and the device
// serves about a hundred million of those per second -- a million items taken eight at a time spent half the
// verifier's time queueing for the cursor.  So: a quarter of a wavefront's fair share, at most kVerifyBatchMax.
constexpr uint32_t kVerifyBatchMax = 32;
// (item_stream's G: groups of 64 candidates per step; 4 was measured slower than 1 -- DESIGN.md section 12)

struct ItemQueue {
  uint4* items;   // cap items of item_quads<NW>() quads
  uint32_t* ctl;  // [0] dense items, [1] gather items, [2] [3] the verifiers' cursors (zeroed by the host)
  uint32_t cap;   // items the array has room for, both kinds TOGETHER: dense items fill it from the front, gather items from
                  // the back.  A caller sizes cap for the most items that can come (the stage kernels: two per read and
                  // round); an append that finds its side's count beyond cap is dropped AND counted in *ovf, and the
                  // verifiers count the two sides having met (items_overflow_check): walt_batch_check reports either
  uint32_t* ovf;  // overflow counter (the workspace's err[2]); nullptr: none
  // Largest first: dense items of more than kBigFirst candidates go to their own array and the dense verifier takes
  // them BEFORE the others (item numbers [0, bigs) are this array's), dealt round robin over the wavefronts -- a
  // region of thousands of candidates is tens of dependent steps, and a wavefront that met two or three of them at
  // the end of its share set the duration of the launch.  nullptr: no such array.
  uint4* bigs;
  uint32_t* big_n;   // items in it (zeroed by the host)
  uint32_t big_cap;
};
constexpr uint32_t kBigFirst = 1024;

template <int NW>
constexpr uint32_t item_quads() { return 2u + (2u * NW + 3u) / 4u; }

// All 64 lanes call this; lanes with `take` append one item each (one atomic per wavefront and kind).
// Returns false for a lane whose item found no room (a queue smaller than the number of items that can come:
// the giant queue; the caller puts the item elsewhere; the verifier clamps the count it reads to q.cap).
template <int NW>
__device__ __forceinline__ bool item_append(bool take, bool dense, uint32_t id, uint32_t l, uint32_t size, uint32_t rec,
                                            uint32_t len, uint32_t seed_i, const uint32_t* rd, const uint32_t* mk,
                                            const ItemQueue& q) {
  constexpr uint32_t Q = item_quads<NW>();
  const uint32_t lane = threadIdx.x & 63;
  bool placed = true;
  bool in_bigs = false;  // this lane's item went to the largest-first array
#pragma unroll
  for (int side = -1; side < 2; ++side) {  // -1: the largest-first array
    if (side < 0 && q.bigs == nullptr) continue;
    const bool mine = side < 0 ? (take && dense && size > kBigFirst) : (take && !in_bigs && (dense == (side == 0)));
    const unsigned long long m = __ballot(mine);
    if (!m) continue;
    const int leader = (int)__ffsll((long long)m) - 1;
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(side < 0 ? q.big_n : &q.ctl[side], (uint32_t)__popcll(m));
    base = bcast(base, leader);
    const uint32_t k = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    const uint32_t room = side < 0 ? q.big_cap : q.cap;
    if (side < 0) in_bigs = mine && k < room;       // no room: the item goes to the ordinary dense side below
    else if (mine && k >= room) placed = false;
    if (mine && k < room) {
      const uint64_t at = side <= 0 ? k : (uint64_t)q.cap - 1 - k;
      uint4* it = (side < 0 ? q.bigs : q.items) + Q * at;
      it[0] = make_uint4(id, l, size, rec);
      it[1] = make_uint4(len, seed_i, 0u, 0u);
      uint32_t w[4 * (Q - 2)];
#pragma unroll
      for (uint32_t t = 0; t < 4 * (Q - 2); ++t) w[t] = t < (uint32_t)NW ? rd[t] : (t < 2u * NW ? mk[t - NW] : 0u);
#pragma unroll
      for (uint32_t qd = 0; qd + 2 < Q; ++qd) it[2 + qd] = make_uint4(w[4 * qd], w[4 * qd + 1], w[4 * qd + 2], w[4 * qd + 3]);
    }
  }
  return placed;
}

// Both strands' items of a probe in ONE call: one atomic per kind for the two of them (every returning atomic is a
// dependent round trip for the wavefront, and same-address atomics run at ~10 ns each: map_common.h WaveList).
// Lane arguments with suffix 0 are the '+' strand's item, 1 the '-' strand's.
template <int NW>
__device__ __forceinline__ void item_append2(const bool* take, const bool* dense, const uint32_t* id, const uint32_t* l,
                                             const uint32_t* size, const uint32_t* rec, uint32_t len, uint32_t seed_i,
                                             const uint32_t* rd, const uint32_t* mk, const ItemQueue& q,
                                             const bool* tail = nullptr) {
  constexpr uint32_t Q = item_quads<NW>();
  const uint32_t lane = threadIdx.x & 63;
  bool in_bigs[2] = {false, false};
#pragma unroll
  for (int side = -1; side < 2; ++side) {  // -1: the largest-first array
    if (side < 0 && q.bigs == nullptr) continue;
    bool mine[2];
    unsigned long long m[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      mine[f] = side < 0 ? (take[f] && dense[f] && size[f] > kBigFirst) : (take[f] && !in_bigs[f] && (dense[f] == (side == 0)));
      m[f] = __ballot(mine[f]);
    }
    if (!(m[0] | m[1])) continue;
    const uint32_t n0 = (uint32_t)__popcll(m[0]), n1 = (uint32_t)__popcll(m[1]);
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(side < 0 ? q.big_n : &q.ctl[side], n0 + n1);
    base = bcast(base, 0);
    const uint32_t room = side < 0 ? q.big_cap : q.cap;
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const uint32_t k = base + (f ? n0 : 0u) + (uint32_t)__popcll(m[f] & ((1ull << lane) - 1ull));
      if (side < 0) in_bigs[f] = mine[f] && k < room;  // no room: the item goes to the ordinary dense side below
      if (side >= 0 && mine[f] && k >= room && q.ovf != nullptr) atomicAdd(q.ovf, 1u);  // (never, with a queue sized as above)
      if (mine[f] && k < room) {
        const uint64_t at = side <= 0 ? k : (uint64_t)q.cap - 1 - k;
        uint4* it = (side < 0 ? q.bigs : q.items) + Q * at;
        it[0] = make_uint4(id[f], l[f], size[f], rec[f]);
        it[1] = make_uint4(len, seed_i, (tail != nullptr && tail[f]) ? 1u : 0u, 0u);
        uint32_t w[4 * (Q - 2)];
#pragma unroll
        for (uint32_t t = 0; t < 4 * (Q - 2); ++t) w[t] = t < (uint32_t)NW ? rd[t] : (t < 2u * NW ? mk[t - NW] : 0u);
#pragma unroll
        for (uint32_t qd = 0; qd + 2 < Q; ++qd) it[2 + qd] = make_uint4(w[4 * qd], w[4 * qd + 1], w[4 * qd + 2], w[4 * qd + 3]);
      }
    }
  }
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) { return ~wave_min_u32(~v); }

// the two sides of the queue must not have met (one thread of a verifier launch checks the launch's counts)
__device__ __forceinline__ void items_overflow_check(const ItemQueue& q) {
  if (q.ovf != nullptr && (uint64_t)q.ctl[0] + q.ctl[1] > q.cap) atomicAdd(q.ovf, 1u);
}

// A pointer the kernel got inside a by-value struct is a generic pointer to the compiler, and a load through it a
// FLAT load: it counts as vector-memory AND as LDS traffic, so that waiting for one -- or for any LDS read
// while one is pending -- waits for every record in flight.  These loads are global by construction.
template <class T>
__device__ __forceinline__ T load_global(const T* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef uint32_t __attribute__((address_space(1))) gword;
  const gword* q = reinterpret_cast<const gword*>(reinterpret_cast<uintptr_t>(p));
  T v;
  uint32_t* w = reinterpret_cast<uint32_t*>(&v);
#pragma unroll
  for (unsigned i = 0; i < sizeof(T) / 4; ++i) w[i] = q[i];
  return v;
#else
  return *p;
#endif
}

// Tail items (reads above 134 bases) arrive as key-equal RANGES: the candidates whose care
// characters 44 .. seed_len - 1 equal the read's are a contiguous part of the range (a dense range is sorted on them),
// on the benchmark genome half of it.  Before the verifier streams a range, one wavefront brackets that part: 64
// evenly spaced records' tail characters (the 16-byte second half of a dense record holds them) are compared with the
// read's, the samples below and above it are cut off, and the item's header is rewritten to the bracket -- at most
// 1/32 of the range more than the region itself; the verifier still tests every candidate it streams and counts the
// region for -b.  One wavefront per item; `n_items` dense items of q.items and `n_big` of q.bigs.
// A range whose samples already prove more than `b` members (the samples that equal the read's characters are members,
// and so is everything between them) is a region the reference skips (mapping.cpp:275-277): its item shrinks to one
// candidate with tail = 2, which the verifier takes for an empty region -- satellite arrays put thousands of
// candidates into such ranges, a third of all the candidates of the 150-base benchmark reads.
template <int NW, class Sink>
__device__ __forceinline__ void tail_items_narrow(const IndexView& iv, uint32_t strand_base, const ItemQueue& q, uint32_t n_items,
                                                  uint32_t n_big, uint32_t b) {
  static_assert(NW > 8 && NW <= 10 && kPat == 3, "seeds beyond the 44 key characters: reads of 135 to 160 bases, pattern 3");
  constexpr uint32_t Q = item_quads<NW>();
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
  for (uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; i < n_items + n_big; i += n_waves) {
    uint4* const it = i < n_big ? q.bigs + (uint64_t)Q * i : q.items + (uint64_t)Q * (i - n_big);
    const uint4 h = load_global(it + (lane < Q ? lane : 0u));
    const uint32_t tail = bcast(h.z, 1);
    if (!tail) continue;  // (uniform)
    const uint32_t id = bcast(h.x, 0), l = bcast(h.y, 0), size = bcast(h.z, 0), rec0 = bcast(h.w, 0);
    const uint32_t len = bcast(h.x, 1), seed_i = bcast(h.y, 1);
    if (size <= 1 || rec0 == kItemDenseNone) continue;
    const uint32_t seed_len = len >= kMinReadLen ? seed_len_of(seed_repeats(len)) : 0u;
    // the read's words 8 and 9 (item words 8 + 8, 8 + 9: quads 4.x, 4.y)
    const uint32_t r8 = bcast(h.x, 4), r9 = bcast(h.y, 4);
    const unsigned long long rr = ((unsigned long long)r9 << 32) | r8;  // read offsets 128 .. 159
    // tail key of the read and of this lane's sample: characters 44 .. 49, two bits each, the first one highest;
    // characters at and beyond seed_len count as equal (0 on both sides)
    const uint32_t s_at = (uint32_t)(((unsigned long long)lane * size) >> 6);  // sample positions rise with the lane
    const StrandView& sv = iv.s[strand_base + Sink::strand(id)];
    const uint4 e = load_global(reinterpret_cast<const uint4*>(sv.win2) + ((uint64_t)rec0 + s_at));
    const unsigned long long g_lo = ((unsigned long long)e.y << 32) | e.x, g_hi = ((unsigned long long)e.w << 32) | e.z;  // bases 112..143, 144..175 of the record
    uint32_t key_r = 0, key_g = 0;
#pragma unroll
    for (uint32_t p = kKeyWeight + kKeyChars; p < kMaxRepeats; ++p) {
      const uint32_t o = seed_i + care_pos(p) - 128u;           // read offset - 128
      const uint32_t b = care_pos(p) + kWinLead - 112u;          // record base index - 112: 23 .. 38
      const uint32_t cr = (uint32_t)(rr >> (2u * o)) & 3u;
      const uint32_t cg = b < 32u ? (uint32_t)(g_lo >> (2u * b)) & 3u : (uint32_t)(g_hi >> (2u * (b - 32u))) & 3u;
      const bool on = p < seed_len;
      key_r = (key_r << 2) | (on ? cr : 0u);
      key_g = (key_g << 2) | (on ? cg : 0u);
    }
    const unsigned long long below = __ballot(key_g < key_r), not_above = __ballot(key_g <= key_r);
    const uint32_t L = (uint32_t)__popcll(below), G = (uint32_t)__popcll(not_above);  // prefixes of the lanes: the range is sorted
    // members lie behind sample L - 1 and in front of sample G
    const uint32_t s_prev = (uint32_t)(((unsigned long long)(L ? L - 1 : 0u) * size) >> 6);
    const uint32_t s_next = (uint32_t)(((unsigned long long)(G < 64u ? G : 63u) * size) >> 6);
    uint32_t lo = L ? s_prev + 1 : 0u, hi = G < 64u ? s_next : size;
    if (lo >= hi) { lo = lo < size ? lo : size - 1; hi = lo + 1; }  // no member: one candidate that is none stays (the region is empty)
    // samples L .. G - 1 equal the read's characters: they and all candidates between them are members
    const uint32_t s_first = (uint32_t)(((unsigned long long)L * size) >> 6), s_last = (uint32_t)(((unsigned long long)(G ? G - 1 : 0u) * size) >> 6);
    const bool over_b = G > L && s_last - s_first + 1 > b;
    if (over_b) hi = lo + 1;
    if (lane == 0) {
      it[0] = make_uint4(id, l + lo, hi - lo, rec0 + lo);
      if (over_b) it[1] = make_uint4(len, seed_i, 2u, 0u);
    }
  }
}

// The same for patterns 5 / 7, whose tail characters (up to 36 of them) span several words of read and record: the
// samples are compared with the read character by character (lexicographic state instead of a packed key); the
// record's words and the read shifted by the seed shift make every character's place a compile-time constant.
template <int NW, class Sink>
__device__ __forceinline__ void tail_items_narrow_wide(const IndexView& iv, uint32_t strand_base, const ItemQueue& q,
                                                       uint32_t n_items, uint32_t n_big, uint32_t b) {
  static_assert(NW <= 10 && kPat != 3 && long_seed_nw<NW>(), "seeds beyond the 44 key characters, patterns 5 and 7");
  constexpr uint32_t Q = item_quads<NW>();
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
  for (uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; i < n_items + n_big; i += n_waves) {
    uint4* const it = i < n_big ? q.bigs + (uint64_t)Q * i : q.items + (uint64_t)Q * (i - n_big);
    const uint4 h = load_global(it + (lane < Q ? lane : 0u));
    const uint32_t tail = bcast(h.z, 1);
    if (!tail) continue;  // (uniform)
    const uint32_t id = bcast(h.x, 0), l = bcast(h.y, 0), size = bcast(h.z, 0), rec0 = bcast(h.w, 0);
    const uint32_t len = bcast(h.x, 1), seed_i = bcast(h.y, 1);
    if (size <= 1 || rec0 == kItemDenseNone) continue;
    const uint32_t seed_len = len >= kMinReadLen ? seed_len_of(seed_repeats(len)) : 0u;
    uint32_t rd[NW + 1];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const int a = 8 + w;
      const uint32_t va = (a & 3) == 0 ? h.x : (a & 3) == 1 ? h.y : (a & 3) == 2 ? h.z : h.w;
      rd[w] = bcast(va, a >> 2);
    }
    rd[NW] = 0;
    uint32_t rs[NW];  // the read from its seed shift on
#pragma unroll
    for (int w = 0; w < NW; ++w) rs[w] = funnel_r(rd[w], rd[w + 1], 2 * seed_i);
    const uint32_t s_at = (uint32_t)(((unsigned long long)lane * size) >> 6);  // sample positions rise with the lane
    const StrandView& sv = iv.s[strand_base + Sink::strand(id)];
    const uint4* rp = reinterpret_cast<const uint4*>(sv.win) + 2 * ((uint64_t)rec0 + s_at);
    const uint4 ra = load_global(rp), rc = load_global(rp + 1);
    uint4 re = make_uint4(0, 0, 0, 0);
    if constexpr (NW > 7) re = load_global(reinterpret_cast<const uint4*>(sv.win2) + ((uint64_t)rec0 + s_at));
    const uint32_t gw[11] = {ra.y, ra.z, ra.w, rc.x, rc.y, rc.z, rc.w, re.x, re.y, re.z, re.w};  // bases from pos - kWinLead
    bool lt = false, gt = false;  // the sample's tail characters against the read's, first difference decides
#pragma unroll
    for (uint32_t p = kKeyWeight + kKeyChars; p < kNumCare; ++p) {
      constexpr uint32_t kLimit = 16u * (NW < 10 ? NW : 10);
      const uint32_t o = care_pos(p);              // read offset behind the seed shift
      const uint32_t g = care_pos(p) + kWinLead;   // base index in the record
      if (o < kLimit && g < 176u) {
        const uint32_t cr = (rs[o >> 4] >> (2u * (o & 15u))) & 3u;
        const uint32_t cg = (gw[g >> 4] >> (2u * (g & 15u))) & 3u;
        const bool on = p < seed_len && !lt && !gt;
        lt = lt || (on && cg < cr);
        gt = gt || (on && cg > cr);
      }
    }
    const unsigned long long below = __ballot(lt), not_above = __ballot(!gt);
    const uint32_t L = (uint32_t)__popcll(below), G = (uint32_t)__popcll(not_above);  // prefixes of the lanes: the range is sorted
    const uint32_t s_prev = (uint32_t)(((unsigned long long)(L ? L - 1 : 0u) * size) >> 6);
    const uint32_t s_next = (uint32_t)(((unsigned long long)(G < 64u ? G : 63u) * size) >> 6);
    uint32_t lo = L ? s_prev + 1 : 0u, hi = G < 64u ? s_next : size;
    if (lo >= hi) { lo = lo < size ? lo : size - 1; hi = lo + 1; }  // no member: one candidate that is none stays (the region is empty)
    const uint32_t s_first = (uint32_t)(((unsigned long long)L * size) >> 6), s_last = (uint32_t)(((unsigned long long)(G ? G - 1 : 0u) * size) >> 6);
    const bool over_b = G > L && s_last - s_first + 1 > b;
    if (over_b) hi = lo + 1;
    if (lane == 0) {
      it[0] = make_uint4(id, l + lo, hi - lo, rec0 + lo);
      if (over_b) it[1] = make_uint4(len, seed_i, 2u, 0u);
    }
  }
}

template <int W, int N>
__device__ __forceinline__ void tail_words(uint32_t (&t)[N], uint32_t seed_i, uint32_t cut) {
  if constexpr (W < N) {
    t[W] = tail_care_mask_word<W>(seed_i, cut);
    tail_words<W + 1>(t, seed_i, cut);
  }
}

// The item loop of a verifier kernel: one region per wavefront, everything about the item wave-uniform, so a lane
// carries little besides the records it has in flight and the kernel runs at high occupancy.  Wavefronts take
// items in batches from the queue's cursor (regions run from 17 to `-b` candidates: a static deal
// leaves a long tail).
//   DENSE = true:  every candidate has a dense record: 32 (48) sequential bytes each, 64 candidates per step, two
//                  record buffers used in turn (a register COPY of a loaded value waits for the load, so the buffers
//                  swap roles instead): while step s is evaluated from one, step s + 1 -- of this item, or the
//                  first of the next -- is on its way into the other.  Every step issues the same loads whatever
//                  it is (uniform selects on the addresses, no branch around a load), so the wait before the
//                  evaluation leaves exactly them in flight.
//   DENSE = false: index entry, then the genome window (coop_verify_groups' gather route).
//   FITS:          the chromosome starts are in LDS (s_start), else read from HBM.
// Sink (all calls wave-uniform):  strand(id) -> 0 / 1;  begin(id, seed_i, position in the queue, tail);  add(k, gp, mm, in) per lane
// and step (k = slot offset in the region, mm = 0xFFFFFFFF when the slot is beyond the region or fails the edge filters of
// mapping.cpp:280-286 / paired.cpp:166-171; in = the slot belongs to the region: always inside [0, size) unless the item
// is a `tail` item, whose candidates belong when their care characters >= 44 equal the read's -- the sink counts them
// for the -b test, mapping.cpp:275-277);  step() after every 64 candidates;  end().
template <int NW, bool DENSE, bool FITS, class Sink, int G = 1>
__device__ __forceinline__ void item_stream(const IndexView& iv, uint32_t strand_base, const ItemQueue& q, uint32_t n_items,
                                            const uint32_t* s_start, Sink& sink, uint32_t n_big = 0,
                                            const uint32_t* s_edge = nullptr /* LDS copy of iv.edge_bits, or none */) {
  // n_items counts the n_big items of q.bigs (taken first) and the dense (or gather) items of q.items
  constexpr uint32_t Q = item_quads<NW>();
  const uint32_t n_chrom = iv.n_chrom, top_step = top_step_of(n_chrom);
  uint32_t* const cursor = &q.ctl[DENSE ? 2 : 3];
  const uint32_t lane = threadIdx.x & 63;
  const uint4 zero = make_uint4(0, 0, 0, 0);
  auto header = [&](uint32_t i, bool on) {  // lane t < Q: quad t of item i
    const bool big = DENSE && i < n_big;
    const uint64_t at = big ? (uint64_t)i : (DENSE ? (uint64_t)(i - n_big) : (uint64_t)q.cap - 1 - i);
    // branch-free (a value loaded under a branch is waited for where the branch ends): idle lanes read quad 0
    return load_global((big ? q.bigs : q.items) + ((on && lane < Q) ? Q * at + lane : 0ull));
  };
  // The first batch of every wavefront is dealt statically (wave w: items [w * batch, (w + 1) * batch)), the cursor
  // hands out what lies behind those: a launch with few items -- the later stages -- makes no atomic at all
  // (two per wavefront, as it was, cost a nearly empty launch 0.6 ms).
  // Three quarters of the items are dealt this way, the last quarter evens out what the deal left uneven: a launch
  // of 25,000 items dealt one per wavefront and the other 17,000 fetched one atomic each spent 0.2 of its 0.4 ms
  // at the cursor.
  const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  // (A launch with many items per wavefront -- the single-end stages: a million -- keeps the finer deal: every
  // wavefront starts with one batch and fetches the others, 32 items per atomic.)
  const bool many = n_items >= 32 * n_waves;
  uint32_t batch = many ? n_items / (4 * n_waves) : 2u;  // of the dynamic part
  batch = batch > kVerifyBatchMax ? kVerifyBatchMax : batch;
  const uint32_t first_batch = many ? batch : (uint32_t)((3ull * n_items + 4ull * n_waves - 1) / (4ull * n_waves));  // >= 1
  const uint32_t dealt = n_waves * first_batch;
  auto grab = [&]() {  // lane 0 holds the batch start once the atomic has returned
    uint32_t v = 0;
    if (lane == 0) v = dealt + atomicAdd(cursor, batch);
    return v;
  };
  auto chrom_bounds = [&](uint32_t pos, uint32_t& c_lo, uint32_t& c_hi) {
    // Far from every chromosome boundary (core.h kEdgeMargin) the edge filters of mapping.cpp:280-286 hold whatever the
    // chromosome is: one bit per 64 K bases says so, and the search over the chromosome starts -- a quarter of the
    // verifier's instructions -- is left to the few candidates near a boundary.
    if (s_edge != nullptr && !((s_edge[pos >> (kEdgeBlockShift + 5)] >> ((pos >> kEdgeBlockShift) & 31u)) & 1u)) {
      c_lo = 0; c_hi = 0xFFFFFFFFu;
      return;
    }
    if constexpr (FITS) {  // through the LDS array itself: through a pointer that may be either, the reads would be FLAT loads
      const uint32_t chr = chrom_id_steps(s_start, n_chrom, top_step, pos);
      c_lo = s_start[chr]; c_hi = s_start[chr + 1];
    } else {  // more sequences than the LDS array holds: its samples, then the starts between two of them (map_common.h ChromTab)
      walt::chrom_bounds(s_start, iv.start_index, chrom_tab_of(n_chrom), pos, c_lo, c_hi);
    }
  };
  // (the statically dealt items of a wavefront are n_waves apart: the largest come first in the numbering and are
  // spread over the wavefronts this way)
  uint32_t b0 = wave, t = 0, stride = n_waves;
  uint32_t cur_batch = first_batch;  // items of the batch being worked on
  if (b0 >= n_items) return;
  uint32_t nb_v = 0;        // lane 0: start of the next batch, asked for when the current batch's last item begins
  bool asked = false;
  if (cur_batch == 1 && dealt < n_items) {  // the first item is its batch's last
    nb_v = grab();
    asked = true;
  }
  uint4 hd = header(b0, true);
  // the current item, wave-uniform
  // tail items (reads above 134 bases): the seed's care characters 44 .. seed_len - 1 sit at read offsets
  // seed_i + 1 + 3 p = 133 .. 150, i.e. in words 8 and 9 of the read: a 64-bit mask over bases 128 .. 159
  uint32_t id = 0, l = 0, size = 1, rec0 = 0, len = 0, seed_i = 0, tail = 0, rd[NW], mk[NW];
  unsigned long long tm64 = 0;
  // patterns 5 / 7: the tail characters span several words; their mask per word (core.h tail_care_mask_word), all zero
  // for an item that is no tail item
  constexpr bool kTailWide = kPat != 3 && DENSE && NW <= 10 && long_seed_nw<NW>();
  uint32_t tmw[kTailWide ? NW : 1];
  auto decode = [&](const uint4& h) {
    id = bcast(h.x, 0); l = bcast(h.y, 0); size = bcast(h.z, 0); rec0 = bcast(h.w, 0);
    len = bcast(h.x, 1); seed_i = bcast(h.y, 1); tail = bcast(h.z, 1);
    tm64 = 0;
    if constexpr (kTailWide) {
      const uint32_t seed_len = len >= kMinReadLen ? seed_len_of(seed_repeats(len)) : 0u;
      const uint32_t cut = tail ? tail_care_cut(seed_i, seed_len) : 0u;
      tail_words<0>(tmw, seed_i, cut);
    }
    if constexpr (NW > 8 && NW <= 10 && DENSE && kPat == 3) {  // only reads above 134 bases have seeds beyond the 44 characters of the keys
      const uint32_t seed_len = (tail && len >= kMinReadLen) ? seed_len_of(seed_repeats(len)) : 0u;
#pragma unroll
      for (uint32_t p = kKeyWeight + kKeyChars; p < kMaxRepeats; ++p) {
        const uint32_t o = seed_i + care_pos(p) - 128u;  // 5 .. 22 + seed shift: inside the 32 bases of words 8 and 9
        tm64 |= p < seed_len ? 1ull << (2u * o) : 0ull;
      }
    }
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const int a = 8 + w, c = 8 + NW + w;
      const uint32_t va = (a & 3) == 0 ? h.x : (a & 3) == 1 ? h.y : (a & 3) == 2 ? h.z : h.w;
      const uint32_t vc = (c & 3) == 0 ? h.x : (c & 3) == 1 ? h.y : (c & 3) == 2 ? h.z : h.w;
      rd[w] = bcast(va, a >> 2);
      mk[w] = bcast(vc, c >> 2);
    }
  };
  // the item after it: index (the batch's next, or the first of the batch grabbed meanwhile) and header load
  uint4 hdn = zero;
  bool hn = false;
  uint32_t cur_i = b0, ni = 0;  // queue positions of the current and the next item
  auto fetch_next = [&]() {
    if (t + 1 < cur_batch) {
      ++t;
      ni = b0 + t * stride;
    } else {
      b0 = asked ? bcast(nb_v, 0) : 0xFFFFFFFFu;  // (asked: the atomic was issued one item ago)
      asked = false;
      t = 0;
      ni = b0;
      cur_batch = batch;
      stride = 1;
    }
    if (t + 1 == cur_batch && ni < n_items && dealt < n_items) {  // ni is the batch's last item: ask for the batch after it
      nb_v = grab();
      asked = true;
    }
    hn = ni < n_items;
    hdn = header(ni, hn);
  };
  decode(hd);
  fetch_next();
  sink.begin(id, seed_i, cur_i, tail);
  if constexpr (DENSE && NW <= 10) {
    // G groups of 64 candidates per step (G = 4 for the queue of very large regions: a region of 5,000 candidates
    // taken 64 at a time is 79 dependent steps, and one such item set the duration of every launch)
    struct Buf { uint4 a[G], c[G], e[G]; };
    Buf b0_, b1_;
#pragma unroll
    for (int u = 0; u < G; ++u) { b0_.a[u] = b0_.c[u] = b0_.e[u] = zero; b1_.a[u] = b1_.c[u] = b1_.e[u] = zero; }
    auto issue = [&](uint32_t fi, uint32_t r0, uint32_t sz, uint32_t base, Buf& y) {
      // fi is wave-uniform; saying so keeps the strand's record pointer a SCALAR load -- as a vector load it was
      // waited for with vmcnt(0), i.e. together with every record in flight (the paired-end sink's strand())
      fi = (uint32_t)__builtin_amdgcn_readfirstlane((int)fi);
      r0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)r0);
      sz = (uint32_t)__builtin_amdgcn_readfirstlane((int)sz);
      const StrandView& sv = iv.s[strand_base + fi];
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const uint32_t k = base + 64 * u + lane;
        const uint64_t rec = (uint64_t)r0 + (k < sz ? k : sz - 1);
        const uint4* rp = reinterpret_cast<const uint4*>(sv.win) + 2 * rec;
        y.a[u] = load_global(rp);
        y.c[u] = load_global(rp + 1);
        if constexpr (NW > 7) y.e[u] = load_global(reinterpret_cast<const uint4*>(sv.win2) + rec);
      }
    };
    uint32_t base = 0;
    bool done = false;
    auto step = [&](Buf& x, Buf& y) {
      const bool last = base + 64 * G >= size;
      // what the other buffer gets: this item's next step, the next item's first, or (nothing left) a repeat
      uint32_t n_fi = Sink::strand(id), n_rec0 = rec0, n_size = size;
      const uint32_t n_base = last ? 0u : base + 64 * G;
      if (last && hn) { n_fi = Sink::strand(bcast(hdn.x, 0)); n_size = bcast(hdn.z, 0); n_rec0 = bcast(hdn.w, 0); }
      issue(n_fi, n_rec0, n_size, n_base, y);
#pragma unroll
      for (int u = 0; u < G; ++u) {
        if (G > 1 && base + 64 * u >= size) break;  // uniform
        const uint32_t k = base + 64 * u + lane;
        const uint32_t pos = x.a[u].x;
        uint32_t c_lo, c_hi;
        chrom_bounds(pos, c_lo, c_hi);
        const uint32_t g = pos - seed_i;
        const bool ok = k < size && (pos - c_lo >= seed_i) && (g + len < c_hi);
        uint32_t wv[NW + 1];
        const uint32_t first[11] = {x.a[u].y, x.a[u].z, x.a[u].w, x.c[u].x, x.c[u].y, x.c[u].z, x.c[u].w,
                                    NW > 7 ? x.e[u].x : 0u, NW > 7 ? x.e[u].y : 0u, NW > 7 ? x.e[u].z : 0u, NW > 7 ? x.e[u].w : 0u};
#pragma unroll
        for (int w = 0; w <= NW; ++w) wv[w] = w < 11 ? first[w] : 0u;
        uint32_t m;
        bool in = k < size;
        if constexpr (kTailWide) {  // the candidate's care characters >= 44 against the read's
          uint32_t tmm;
          m = count_mismatch_regs2<NW>(wv, 2 * (kWinLead - seed_i), rd, mk, tmw, tmm);
          in = in && tmm == 0 && tail != 2u;  // (tail == 2: the region was proved larger than -b, tail_items_narrow_wide)
        } else {
          m = count_mismatch_regs<NW>(wv, 2 * (kWinLead - seed_i), rd, mk);
        }
        if constexpr (NW > 8 && NW <= 10 && kPat == 3) {  // the candidate's care characters >= 44 against the read's (words 8, 9)
          const uint32_t shv = 2 * (kWinLead - seed_i);
          const uint32_t x8 = funnel_r(wv[8], wv[9], shv) ^ rd[8], x9 = funnel_r(wv[9], wv[10], shv) ^ rd[9];
          const unsigned long long d = ((unsigned long long)(x9 | (x9 >> 1)) << 32) | (x8 | (x8 >> 1));
          in = in && (d & tm64) == 0 && tail != 2u;  // (tail == 2: the region was proved larger than -b, tail_items_narrow)
        }
        sink.add(k, (ok && in) ? g : 0u, (ok && in) ? m : 0xFFFFFFFFu, in);
        sink.step();
      }
      if (last) {
        sink.end();
        if (hn) {
          decode(hdn);
          cur_i = ni;
          fetch_next();
          sink.begin(id, seed_i, cur_i, tail);
          base = 0;
        } else {
          done = true;
        }
      } else {
        base += 64 * G;
      }
    };
    issue(Sink::strand(id), rec0, size, 0u, b0_);
    for (;;) {
      step(b0_, b1_);
      if (done) break;
      step(b1_, b0_);
      if (done) break;
    }
  } else {
    const uint32_t* si = s_start;
    for (;;) {
      const StrandView& sv = iv.s[strand_base + Sink::strand(id)];
      DenseRange none;
      none.lo = none.hi = l;
      none.rec = 0;
      for (uint32_t base = 0; base < size; base += 64) {
        uint32_t gp[1], mm[1];
        coop_verify_groups<NW, 1>(sv, si, iv.start_index, n_chrom, l, size, base, seed_i, len, rd, mk, lane, none, gp, mm);
        sink.add(base + lane, gp[0], mm[0], base + lane < size);
        sink.step();
      }
      sink.end();
      if (!hn) break;
      decode(hdn);
      cur_i = ni;
      fetch_next();
      sink.begin(id, seed_i, cur_i, tail);
    }
  }
}

}  // namespace walt
#endif  // WALT_AMD_MAP_ITEMS_H_
