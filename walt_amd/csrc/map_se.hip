// map_se.hip -- single-end seed-and-extend on MI355X.
//
// Replaces, for a whole batch and BOTH strand passes, the loop
//     for fi in {0,1}: ReadIndex(...); #pragma omp parallel for: SingleEndMapping(...)
// of ProcessSingledEndReads (reference mapping.cpp:486-500) and the body of
// SingleEndMapping (mapping.cpp:224-316).
//
// Work decomposition: one read per LANE for the seed
// lookup (the lookup is a chain of dependent gathers, so 64 independent chains
// per wave keep 64 HBM requests in flight), candidates of small regions verified
// by the owning lane, regions larger than kSmallRegion verified by the whole
// wavefront (lane k takes slot l+k, l+64+k, ...) with a ballot/min reduction
// that reproduces the sequential BestMatch fold (mapping.cpp:306-313).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "map_common.h"
#include "map_items.h"

namespace walt {

// ASCII -> dense 2-bit array (map_common.h).  The byte stream is offsets[0] ..
// offsets[n]; thread t converts bytes [16 t, 16 t + 16) of it.  err[0] counts reads
// ... bytes that are not ACGT (getBits would exit, util.hpp:117-119).
static __global__ __launch_bounds__(kBlock) void k_ascii_to_2bit(const uint8_t* __restrict__ bases,
                                                                  const uint64_t* __restrict__ offsets, uint32_t n,
                                                                  uint32_t* __restrict__ codes2, uint64_t cap_bytes,
                                                                  uint32_t* __restrict__ err) {
  // never beyond the room the workspace has (n x max_read_len bytes): a batch whose reads are longer than the
  // caller said is refused (walt_batch_check: WALT_EINVAL), its surplus is not converted and not read
  const uint64_t o0 = offsets[0], all = offsets[n] - o0, total = all < cap_bytes ? all : cap_bytes;
  if (all > cap_bytes && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(err + 1, 1u);
  const uint8_t* src = bases + o0;
  const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(src) & 15);  // 16-byte loads need an aligned address
  const uint64_t nwords = (total + 15) / 16;
  uint32_t bad_total = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords;
       i += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t qs[4] = {0, 0, 0, 0};
    const uint64_t b0 = 16 * i;
    if (mis == 0 && b0 + 16 <= total) {
      const uint4 q = *reinterpret_cast<const uint4*>(src + b0);
      qs[0] = q.x; qs[1] = q.y; qs[2] = q.z; qs[3] = q.w;
    } else {
      for (uint64_t k = b0; k < b0 + 16; ++k) {
        const uint32_t c = k < total ? src[k] : 0x41u;  // pad with 'A' (never part of a read)
        qs[(k - b0) >> 2] |= c << (8 * (k & 3));
      }
    }
    uint32_t word = 0, bad = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t c8, b4;
      ascii4_to_codes(qs[j], c8, b4);
      word |= c8 << (8 * j);
      bad |= b4;
    }
    codes2[i] = word;
    bad_total += bad ? 1u : 0u;
  }
  bad_total = wave_sum_u32(bad_total);
  if ((threadIdx.x & 63) == 0 && bad_total) atomicAdd(err, bad_total);
  if (blockIdx.x == 0 && threadIdx.x < 8) codes2[nwords + threadIdx.x] = 0;  // slack read by the last lanes
}
void launch_ascii_to_2bit(const uint8_t* d_bases, const uint64_t* d_offsets, uint32_t n, uint32_t* d_codes2,
                          uint64_t cap_bytes, uint32_t* d_err, hipStream_t stream) {
  hipLaunchKernelGGL(k_ascii_to_2bit, dim3(256 * 16), dim3(kBlock), 0, stream, d_bases, d_offsets, n, d_codes2, cap_bytes,
                     d_err);
}

// ---------------------------------------------------------------------------
// Wave-cooperative verification of one large region owned by lane `owner`.
// All 64 lanes call this with the same (uniform) arguments broadcast from the
// owner.  Returns the RegionSummary of the region in candidate order.
// ---------------------------------------------------------------------------
constexpr int kCoopGroupsSe = 2;  // 128 candidates per step (map_common.h coop_verify_groups)
template <int NW>
__device__ __forceinline__ RegionSummary coop_region(const StrandView& sv, const uint32_t* si, const uint32_t* __restrict__ gs,
                                                     uint32_t n_chrom, uint32_t l, uint32_t size, uint32_t seed_i, uint32_t len,
                                                     const uint32_t* rd, const uint32_t* mk, uint32_t lane,
                                                     uint32_t& n_verified, const DenseRange* known = nullptr,
                                                     uint32_t abl = 0) {
  RegionSummary acc = summary_empty();
  if (abl & 16u) return acc;  // diagnostic (WALT_AMD_ABLATE, results invalid): no cooperative verification at all
  const DenseRange rb = known ? *known : dense_range(sv, l, size, win_usable<NW>(sv, len));
  for (uint32_t base = 0; base < size; base += 64 * kCoopGroupsSe) {
    uint32_t gp[kCoopGroupsSe], mm[kCoopGroupsSe];
    if (abl & 64u) {  // diagnostic: no loads
#pragma unroll
      for (int u = 0; u < kCoopGroupsSe; ++u) { gp[u] = base + lane; mm[u] = (base + 64 * u + lane) < size ? (lane & 7u) : 0xFFFFFFFFu; }
    } else {
      coop_verify_groups<NW, kCoopGroupsSe>(sv, si, gs, n_chrom, l, size, base, seed_i, len, rd, mk, lane, rb, gp, mm);
    }
    if (abl & 32u) {  // diagnostic: loads and counts, no reduction
      uint32_t x = 0;
#pragma unroll
      for (int u = 0; u < kCoopGroupsSe; ++u) x += mm[u] + gp[u];
      acc.first += x;
      continue;
    }
#pragma unroll
    for (int u = 0; u < kCoopGroupsSe; ++u) {
      if (base + 64 * u >= size) break;
      n_verified += mm[u] != 0xFFFFFFFFu ? 1u : 0u;
      const uint32_t mn = wave_min_u32(mm[u]);
      if (mn != 0xFFFFFFFFu) {
        const unsigned long long eq = __ballot(mm[u] == mn);
        RegionSummary c;
        c.min_mm = mn;
        c.count = (uint32_t)__popcll(eq);
        c.first = bcast(gp[u], (int)__ffsll((long long)eq) - 1);
        c.last = bcast(gp[u], 63 - (int)__clzll((long long)eq));
        acc = summary_merge(acc, c);
      }
    }
  }
  return acc;
}

// In-kernel phase timing (diagnostic, WALT_AMD_STAMPS=1): s_memtime at phase
// boundaries with every outstanding memory operation drained first, summed per
// wave and added to a device buffer.  The stamped run is slower (no overlap across
// phases); read its SHARES, not its length.  Phases: 0 read record, 1 care/slot
// loads, 2 bloom/bad test, 3 lookup (dir + entries), 4 masks, 5 own-lane verify,
// 6 wave-cooperative regions, 7 store, 8 total.
constexpr int kStampPhases = 9;
template <bool ON>
struct StampsT {
  unsigned long long* buf;
  unsigned long long last;
  unsigned long long acc[kStampPhases];
};
template <>
struct StampsT<false> {};  // production kernels carry no stamp state or code
__device__ __forceinline__ unsigned long long stamp_now() {
  unsigned long long t;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
__device__ __forceinline__ void stamp_begin(StampsT<true>& st, unsigned long long* buf) {
  st.buf = buf;
  for (int i = 0; i < kStampPhases; ++i) st.acc[i] = 0;
  st.last = stamp_now();
}
__device__ __forceinline__ void stamp(StampsT<true>& st, int phase) {
  const unsigned long long t = stamp_now();
  st.acc[phase] += t - st.last;
  st.acc[8] += t - st.last;
  st.last = t;
}
__device__ __forceinline__ void stamp_end(StampsT<true>& st) {
  if ((threadIdx.x & 63) == 0)
    for (int i = 0; i < kStampPhases; ++i) atomicAdd(&st.buf[i], st.acc[i]);
}
__device__ __forceinline__ void stamp_begin(StampsT<false>&, unsigned long long*) {}
__device__ __forceinline__ void stamp(StampsT<false>&, int) {}
__device__ __forceinline__ void stamp_end(StampsT<false>&) {}

// Work for one read per lane.  LITERAL = false (pass 1): a lane whose probe
// lands in a BAD bucket (literal LowerBound/UpperBound search, ~100x the
// dependent loads of the key search) stops and appends its read to the deferred
// list, so that one slow lane cannot hold up its 63 wave-mates.  LITERAL = true
// (pass 2) maps the deferred reads from scratch with the literal search enabled.
struct MapCounters {
  uint32_t probes, verified, big;
};
template <int NW>
__device__ __forceinline__ void coop_lane_regions(const IndexView& iv, const BlockShared& sh, const StrandView& svp,
                                                  const StrandView& svm, uint32_t n_p, uint32_t n_m, uint32_t l_p,
                                                  uint32_t l_m, uint32_t pos0_p, uint32_t pos0_m, bool reg_p, bool reg_m,
                                                  uint32_t sd, const LaneRead<NW>& lr, const uint32_t* mk,
                                                  bool tail_p, bool tail_m, const uint32_t* care, uint32_t n_chrom,
                                                  RegionSummary& sum_p, RegionSummary& sum_m, uint32_t& n_verified);
// strand-major kernel: regions of up to this many slots go through the wavefront's candidate list (coop_lane_regions),
// larger ones take the wavefront one at a time (coop_region: dense records where they exist)
constexpr uint32_t kListRegion = 64;

template <int NW, bool LITERAL, bool DIAG>
__device__ __forceinline__ void se_process(const IndexView& iv, BlockShared& sh, const uint32_t* si,
                                           const uint32_t* __restrict__ codes2, const uint64_t* __restrict__ offsets,
                                           uint32_t* __restrict__ err, uint32_t r,
                                           bool valid, uint32_t strand_base, uint32_t max_mm, uint32_t b,
                                           BestMatch* __restrict__ out, uint32_t* __restrict__ defer_count,
                                           uint32_t* __restrict__ defer_list, MapCounters& ctr, uint32_t& len_out,
                                           uint32_t ablate_rt, StampsT<DIAG>& st) {
  const uint32_t ablate = DIAG ? ablate_rt : 0u;
  const uint32_t n_chrom = iv.n_chrom;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t ga = strand_base >> 1, Bd = iv.dir_bits;
  LaneRead<NW> lr;
  {
    uint64_t o = 0, oe = 0;
    if (valid) { o = offsets[r]; oe = offsets[r + 1]; }
    lane_load_read<NW>(lr, codes2, offsets[0], o, oe, valid, ga, err, iv);
  }
  len_out = lr.len;
  bool mappable = valid && lr.len >= kMinReadLen;
  bool deferred = false;
  uint32_t defer_iter = 0;
  stamp(st, 0);

  BestMatch best;  // mapping.cpp:486
  best.genome_pos = 0; best.times = 0; best.strand = '+'; best.mismatch = max_mm;

  for (uint32_t fi = 0; fi < 2; ++fi) {
    const StrandView& sv = iv.s[strand_base + fi];
    const uint32_t strand_char = fi == 0 ? '+' : '-';
#pragma unroll 1
    for (uint32_t seed_i = 0; seed_i < kPat; ++seed_i) {
      // mapping.cpp:250-262 (a `break` there leaves every later seed skipped too,
      // which these per-seed predicates reproduce because best only improves)
      bool act = mappable && !(best.mismatch == 0 && seed_i) && !(best.mismatch == 1 && seed_i >= kExitOneMismatch);
      Lookup lk;
      lk.npos = 0;
      lk.reg = empty_region();
      {
        uint32_t care[kCareWords] = {};
        uint32_t slot = 0, span = 0;
        if (act) seed_query<NW>(lr.rd, seed_len_of(lr.repeats), seed_i, ga, Bd, sh.pcode4, care, slot, span);
        stamp(st, 1);
        bool is_bad = false;
        if (act && !LITERAL) {
          const uint32_t bk = bloom_key_of_care(care);
          is_bad = danger_filter_hit(sv.bloom[bloom_block(bk, sv.bloom_mask)], care);
        }
        stamp(st, 2);
        if (act) {
          if (is_bad) {
            deferred = true;
            mappable = false;
            defer_iter = fi * kPat + seed_i;
          } else if (ablate & 4u) {                       // diagnostic: no lookup at all
          } else if (ablate & 2u) {                       // diagnostic: directory only
            uint32_t lo = (sv.dir + (uint32_t)(slot - 1u))[1], hi = sv.dir[(uint32_t)(slot - span)];
            if (lo > hi) lk.reg.l = 0;
          } else {
            seed_lookup_ex(iv, sv, care, slot, span, seed_len_of(lr.repeats), lk, !LITERAL);
          }
        }
        stamp(st, 3);
      }
      const Region reg = lk.reg;
      uint32_t size = reg.l <= reg.u ? reg.u - reg.l + 1 : 0;
      if (size) ++ctr.probes;
      if (size > b) size = 0;  // mapping.cpp:275-277
      if (ablate & 1u) size = 0;                        // diagnostic: no verification
      uint32_t mk[NW];
      make_masks<NW>(mk, sh.mask_table, seed_i, lr.repeats >= kMinRepeats ? lr.repeats : kMinRepeats, lr.len);
      stamp(st, 4);

      // regions of up to kListRegion slots: ONE candidate list over the wavefront (round 4, late).  Before, a lane walked
      // its region of up to four slots by itself and every larger one took the whole wavefront for itself, one after the
      // other -- a region of five candidates cost the wavefront a turn like one of sixty-four, and a wavefront of this
      // kernel has a dozen of them per probe phase.
      {
        const uint32_t own = (size && size <= kListRegion) ? size : 0u;
        if (__ballot(own != 0)) {
          RegionSummary sum = summary_empty(), none = summary_empty();
          const uint32_t care0[kCareWords] = {};
          coop_lane_regions<NW>(iv, sh, sv, sv, own, 0u, reg.l, 0u, lk.pos[0], 0u, lk.npos != 0, false, seed_i, lr, mk, false, false,
                                care0, n_chrom, sum, none, ctr.verified);
          fold_region(best, sum, strand_char);
          if (own > kSmallRegion) ++ctr.big;  // (counted as before: regions the owning lane does not walk alone)
        }
      }
      stamp(st, 5);
      // large regions: the whole wave verifies one owner's region at a time
      unsigned long long big = __ballot(size > kListRegion);
      while (big) {
        const int owner = (int)__ffsll((long long)big) - 1;
        big &= big - 1;
        uint32_t o_rd[NW], o_mk[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          o_rd[w] = bcast(lr.rd[w], owner);
          o_mk[w] = bcast(mk[w], owner);
        }
        const uint32_t o_l = bcast(reg.l, owner), o_size = bcast(size, owner), o_len = bcast(lr.len, owner);
        uint32_t nv = 0;
        RegionSummary s = coop_region<NW>(sv, si, iv.start_index, n_chrom, o_l, o_size, seed_i, o_len, o_rd, o_mk, lane, nv);
        ctr.verified += nv;
        if ((int)lane == owner) {
          fold_region(best, s, strand_char);
          ++ctr.big;
        }
      }
      stamp(st, 6);
    }
  }
  if (!LITERAL && deferred) {
    defer_list[atomicAdd(defer_count, 1u)] = r <= kDeferMask ? (r | (defer_iter << kDeferShift)) : r;
  } else if (valid) {
    out[r] = best;
  }
  stamp(st, 7);
}

// ---------------------------------------------------------------------------
// Pass 1, seed-major: for each seed shift the '+' and '-' strand probes of a read
// are issued TOGETHER (directory loads of both strands, then both slots' entries,
// then both genome windows), because the kernel is bound by dependent round trips
// (Little's law: ~330 k lanes in flight / ~19 dependent trips per read), not by
// instruction issue.  The reference order is strand-major (+s0 +s1 +s2 -s0 -s1 -s2,
// mapping.cpp:491-499, 248-263; written for pattern 3 and its three seed shifts, the same for the five / seven of
// patterns 5 / 7 since round 3) and its early exits depend on the running best, so:
//   * '+' summaries are folded immediately (their need is known exactly);
//   * a '-' probe is computed when it MIGHT be needed (best so far, including the
//     '-' summaries already seen, has not reached the exit value) -- a superset of
//     the reference's probes -- and its RegionSummary is kept;
//   * after the last seed the '-' summaries are folded in order under the exact exit
//     conditions.  A RegionSummary does not depend on the running best (the
//     reference's mismatch-loop cut-off only truncates counts that lose anyway),
//     so the result is identical; unused speculative probes are only extra work.
// A lane whose probe lands in a BAD bucket is deferred to the literal pass.
// ---------------------------------------------------------------------------
// SlotProbe / probe_issue / probe_entries / probe_resolve / verify_nobranch live in map_common.h (shared with map_pe.hip)

constexpr uint32_t kMidRegion = 16;  // heavy pass: regions up to this size are verified by their own lane, in batches
#ifndef WALT_COOP_REGION
#define WALT_COOP_REGION 16
#endif
constexpr uint32_t kCoopRegion = WALT_COOP_REGION;  // ... with the wavefront's candidate list (pattern 3: coop_lane_regions): up to this size

// ---- staged heavy pass (round 4: run until blocked) ------------------------------------------------------------
// The heavy list is mapped chunk by chunk (hcap reads) in ROUNDS.  In a round a lane takes one read and goes through
// its seed shifts exactly as the reference does (mapping.cpp:248-263: '+' folded at once, '-' kept per seed) until a
// probe produces a region too large for the lane -- a work ITEM for k_se_verify (one region per wavefront) -- and then
// stops: it stores its folded state and the read goes onto the list of the next round, which starts behind the
// verifier launch that wrote the item's RegionSummary.  A read that produces no item is finished in ONE visit (half of
// the heavy reads of an hg19-like genome); the others are visited once more per seed shift that blocked them, and a
// revisit finds everything it needs -- the packed read, the running BestMatch, the pending summaries -- in arrays
// indexed by the read's position j in the chunk (coalesced 16-byte loads, no offsets -> bases -> conversion chain and
// no replay of the earlier seeds' summaries; round 3 visited every read kPat + 1 times and started each visit over).
// Round 0 does not start over either: pass 1 hands over the seed shift it gave up at together with its BestMatch and
// '-' summaries (SeCarry), because the reference probes a seed once (mapping.cpp:265-277).
struct HeavyStage {
  uint4* st;         // [hcap]: {best.genome_pos, best.times, best.mismatch, smallest mismatch count of the kept '-' summaries}
  uint4* rdq;        // [rd_quads<NW>()][hcap]: the converted read: {len, rd[0..2]}, {rd[3..6]}, ...
  uint4* sums;       // [kPat + 1][hcap]: row 0 = '+' summary of the seed the read is blocked at, row 1 + k = '-' summary of seed k
  uint4* items;      // 2 * hcap items of item_quads<NW>() 16-byte words; dense items from the front, gather items from the back
  uint4* giants;     // hcap / 8 items: the dense regions of more than kBigFirst candidates, verified first (count: ctl[5])
  uint32_t* ctl;     // this (chunk, round)'s counters: [0] dense items, [1] gather items, [2] [3] the verifiers' cursors,
                     // [4] reads that go on to the next round, [5] giants, [6] items that found no room (must stay 0)
  const uint32_t* list_in;  // rounds >= 1: the reads this round visits (stage_entry; count in count_in[4]); round 0: every j
  const uint32_t* count_in;
  uint32_t* list_out;       // rounds < kPat: the reads this round blocks
  uint32_t hcap;     // reads per chunk
  uint32_t first;    // heavy-list index of the chunk's first read (even == 0)
  uint32_t chunk;    // even != 0: the heavy list is cut into an EVEN number of equal chunks of at most hcap reads, and
  uint32_t even;     // this is chunk number `chunk` of them (heavy_chunk_span; the list's length is known on the device only)
  uint32_t round;    // 0 .. kPat
  uint32_t defer_min;  // long seeds: key-equal ranges of more slots than this are narrowed by the verifier (0xFFFFFFFF: never)
  uint32_t lit_ablate; // (measurement only, option se_lit_ablate; results are WRONG when set) bit 0: the literal rounds skip the reference's search
};
// What pass 1 hands to round 0 for a read it gives up (arrays indexed by the read's number in the batch, nullptr: the
// heavy pass starts over at seed 0): the heavy-list entry carries the seed shift and best.strand (heavy_entry), st the
// BestMatch after the '+' folds of the seeds before it and the '-' bound, neg[k] the '-' summary of seed k < that seed.
struct SeCarry {
  uint4* st;    // [n]
  uint4* neg;   // [kPat - 1][stride]
  uint64_t stride;
};
constexpr uint32_t kHeavySeedShift = 28;  // heavy-list entry: read (n <= 2^28) | seed shift pass 1 gave it up at << 28
__device__ __forceinline__ uint32_t heavy_entry(uint32_t r, uint32_t seed) { return r | (seed << kHeavySeedShift); }
// list entry of the staged rounds: chunk position j (hcap <= 2^26) | seed shift the read is blocked at << 26
constexpr uint32_t kStageJBits = 26;
__device__ __forceinline__ uint32_t stage_entry(uint32_t j, uint32_t seed) { return j | (seed << kStageJBits); }
template <int NW>
constexpr uint32_t rd_quads() { return (NW + 1 + 3) / 4; }
// Regions of kSmallRegion < size <= kMidRegion candidates stay with their lane (heavy kernels): all positions in one
// round of loads, then the genome windows four at a time.  A lane usually has one such region, on either strand, so the
// strands are not taken in turn: in the first pass every lane works on its '+' mid region, or on its '-' one if it has
// no '+'; the second pass (skipped unless some lane has both) takes the remaining '-' ones.  All loads of a round are
// unconditional (idle candidates read the first words of the genome): a load under `if (candidate ok)` is waited for
// at the end of its branch, which made the four candidates of a round four round trips.
// MULTI (patterns 5 / 7, long seeds): a region may be a key-equal RANGE whose candidates still owe their tail
// characters (map_common.h probe_resolve_dual, multi_max): those that match are counted (nin: the region's size for
// -b), one that fails an edge filter asks for the literal narrowing (fb).
struct MidTail {
  bool multi_p = false, multi_m = false;
  bool fb_p = false, fb_m = false;
  uint32_t nin_p = 0, nin_m = 0;
};
template <int NW, bool MULTI>
__device__ __forceinline__ void se_mid_regions(const IndexView& iv, const BlockShared& sh, const StrandView& svp,
                                               const StrandView& svm, uint32_t nmid_p, uint32_t nmid_m, uint32_t l_p,
                                               uint32_t l_m, uint32_t seed_i, uint32_t len, const uint32_t* rd,
                                               const uint32_t* mk, uint32_t n_chrom, uint32_t top_step, uint32_t tail_cut,
                                               MidTail& mt, RegionSummary& sum_p, RegionSummary& sum_m, uint32_t& n_verified) {
#pragma unroll 1
  for (uint32_t pass = 0; pass < 2; ++pass) {
    const bool on_m = pass == 0 ? (nmid_p == 0 && nmid_m != 0) : (nmid_p != 0 && nmid_m != 0);
    const bool on_p = pass == 0 && nmid_p != 0;
    const uint32_t nmid = on_p ? nmid_p : (on_m ? nmid_m : 0u);
    if (!__ballot(nmid != 0)) continue;
    const Ent* const ent = on_m ? svm.ent : svp.ent;
    const uint32_t* const g2 = on_m ? svm.g2 : svp.g2;
    const uint32_t my_l = nmid ? (on_m ? l_m : l_p) : 0u;
    uint32_t posb[kMidRegion];
#pragma unroll
    for (uint32_t k = 0; k < kMidRegion; ++k) posb[k] = ent[k < nmid ? my_l + k : 0u].pos;  // (what a lane does not have: entry 0, one broadcast access)
    RegionSummary acc = summary_empty();
    const ChromTab ct = chrom_tab_of(n_chrom);
#pragma unroll 1
    for (uint32_t k0 = 0; k0 < kMidRegion; k0 += 4) {
      if (!__ballot(k0 < nmid)) break;
      bool ok[4];
      uint32_t gpv[4], win[4][NW + 1];
#pragma unroll
      for (uint32_t jj = 0; jj < 4; ++jj) {
        // posb[k0 + jj] by selects (k0 is not a compile-time constant: indexing would go to scratch)
        uint32_t pos = 0;
#pragma unroll
        for (uint32_t k = jj; k < kMidRegion; k += 4) pos = (k == k0 + jj) ? posb[k] : pos;
        uint32_t c_lo, c_hi;
        (void)top_step;
        chrom_bounds(sh.start_index, iv.start_index, ct, pos, c_lo, c_hi);
        const uint32_t g = pos - seed_i;
        ok[jj] = k0 + jj < nmid && (pos - c_lo >= seed_i) && (g + len < c_hi);  // mapping.cpp:280-286
        if constexpr (MULTI) {
          if (k0 + jj < nmid && !ok[jj]) { mt.fb_p = mt.fb_p || (on_p && mt.multi_p); mt.fb_m = mt.fb_m || (on_m && mt.multi_m); }
        }
        gpv[jj] = ok[jj] ? g : 0u;
        const uint32_t* gw = g2 + (gpv[jj] >> 4);
#pragma unroll
        for (int w = 0; w <= NW; w += 4) {
          constexpr int kAll = NW + 1;
          const int cnt = kAll - w < 4 ? kAll - w : 4;
          uint32_t q[4] = {0, 0, 0, 0};
          __builtin_memcpy(q, gw + w, 4 * cnt);
#pragma unroll
          for (int t = 0; t < cnt; ++t) win[jj][w + t] = q[t];
        }
      }
#pragma unroll
      for (uint32_t jj = 0; jj < 4; ++jj) {
        uint32_t mm = 0;
        bool keep = ok[jj];
        if constexpr (MULTI) {
          uint32_t tmm = 0;
          count_mismatch_regs_tail<NW>(win[jj], 2 * (gpv[jj] & 15u), rd, mk, seed_i, tail_cut, mm, tmm);
          const bool mine = on_p ? mt.multi_p : (on_m && mt.multi_m);
          keep = keep && (!mine || tmm == 0);
          if (mine && keep) { if (on_p) ++mt.nin_p; else ++mt.nin_m; }
        } else {
          mm = count_mismatch_regs<NW>(win[jj], 2 * (gpv[jj] & 15u), rd, mk);
        }
        if (keep) {
          acc = summary_merge(acc, summary_one(mm, gpv[jj]));
          ++n_verified;
        }
      }
    }
    if (on_p) sum_p = acc;
    if (on_m) sum_m = acc;
  }
}

// Round 4 (late), pattern 3: the candidates the lanes of a wavefront verify themselves -- regions of up to kMidRegion
// index slots on either strand -- as ONE list in lane order (lane 0's '+' candidates, its '-' ones, lane 1's ...):
// candidate q goes to lane q % 64 of turn q / 64, whichever lane owns it.  Before, every lane walked its own regions
// (four candidates of both strands one after the other, then up to sixteen of a mid region four at a time) while the
// lanes with shorter or no regions waited: the phase stamps put a third of the stage kernel there.  A turn:
//   1. the owner of q: the last lane whose exclusive candidate count is <= q (bisection over the lanes, ds_bpermute),
//   2. the owner's read, masks and region start by ds_bpermute; the slot's position (the line the probe just read),
//      edge filters of mapping.cpp:280-286, genome window, mismatch count -- verify_nobranch's steps,
//   3. a segmented inclusive scan of the one-candidate summaries with summary_merge (keys = owner and strand, which
//      rise along the lanes; the merge is associative and order-aware, core.h), and
//   4. every owner merges the summary of the LAST lane of its run in this turn into its own, turn after turn in list
//      order -- the order its own loop had.
// Every lane runs every turn (ds_bpermute reads nothing from a lane outside EXEC).
template <int NW>
__device__ __forceinline__ void coop_lane_regions(const IndexView& iv, const BlockShared& sh, const StrandView& svp,
                                                  const StrandView& svm, uint32_t n_p, uint32_t n_m, uint32_t l_p,
                                                  uint32_t l_m, uint32_t pos0_p, uint32_t pos0_m, bool reg_p, bool reg_m,
                                                  uint32_t sd, const LaneRead<NW>& lr, const uint32_t* mk,
                                                  bool tail_p, bool tail_m, const uint32_t* care, uint32_t n_chrom,
                                                  RegionSummary& sum_p, RegionSummary& sum_m, uint32_t& n_verified) {
  constexpr bool kLong = long_seed_nw<NW>();
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t total;
  const uint32_t off = wave_excl_scan_u32(n_p + n_m, lane, total);
  const ChromTab ct = chrom_tab_of(n_chrom);
  // reg_x: the region's first position is in pos0_x already (a slot-table record that holds its single entry has no
  // index slot to read it from: map_common.h probe_issue)
  const uint32_t meta = sd | (lr.repeats << 3) | (lr.len << 9) | (tail_p ? 1u << 20 : 0u) | (tail_m ? 1u << 21 : 0u) |
                        (reg_p ? 1u << 22 : 0u) | (reg_m ? 1u << 23 : 0u);
#pragma unroll 1
  for (uint32_t base = 0; base < total; base += 64) {
    WALT_DIAG_COUNT(4, 1);                                            // turns of the candidate list
    WALT_DIAG_COUNT(5, total - base < 64u ? total - base : 64u);      // ... and the candidates in them
    const uint32_t q = base + lane;
    const bool have = q < total;
    uint32_t own = 0;
#pragma unroll
    for (uint32_t s = 32; s; s >>= 1) {
      const uint32_t o = shfl_pin(off, own + s);
      own = o <= q ? own + s : own;
    }
    const uint32_t o_off = shfl_pin(off, own), o_np = shfl_pin(n_p, own);
    const uint32_t o_lp = shfl_pin(l_p, own), o_lm = shfl_pin(l_m, own);
    const uint32_t o_meta = shfl_pin(meta, own);
    const uint32_t k = q - o_off;
    const bool on_m = have && k >= o_np;
    const uint32_t kk = on_m ? k - o_np : k;
    const bool from_reg = have && kk == 0 && ((o_meta >> (on_m ? 23 : 22)) & 1u);
    const uint32_t slot = (have && !from_reg) ? (on_m ? o_lm : o_lp) + kk : 0u;
    const Ent* const ent = on_m ? svm.ent : svp.ent;
    const uint32_t* const g2 = on_m ? svm.g2 : svp.g2;
    const uint32_t o_p0 = shfl_pin(pos0_p, own), o_m0 = shfl_pin(pos0_m, own);
    uint32_t pos = ent[slot].pos;  // (a lane without a candidate: entry 0, one broadcast access)
    pos = from_reg ? (on_m ? o_m0 : o_p0) : pos;
    uint32_t o_rd[NW], o_mk[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      o_rd[w] = shfl_pin(lr.rd[w], own);
      o_mk[w] = shfl_pin(mk[w], own);
    }
    const uint32_t o_sd = o_meta & 7u, o_len = (o_meta >> 9) & 2047u;
    uint32_t c_lo, c_hi;
    chrom_bounds(sh.start_index, iv.start_index, ct, pos, c_lo, c_hi);
    const uint32_t g = pos - o_sd;
    bool ok = have && (pos - c_lo >= o_sd) && (g + o_len < c_hi);  // mapping.cpp:280-286
    const uint32_t gp = ok ? g : 0u;
    uint32_t mm = 0;
    if (ok) mm = count_mismatch<NW>(g2, gp, o_rd, o_mk);
    if constexpr (kLong) {  // a single key-equal candidate of a long seed still owes its care characters >= 44 (probe_resolve)
      uint32_t o_care[kCareWords];
#pragma unroll
      for (uint32_t w = 0; w < kCareWords; ++w) o_care[w] = shfl_pin(care[w], own);
      const bool tail = (o_meta >> (on_m ? 21 : 20)) & 1u;
      if (__ballot(ok && tail)) {
        const StrandView& sv0 = svp;
        const bool t = on_m ? tail_care_ok(svm, ok ? pos : 0u, o_care, seed_len_of((o_meta >> 3) & 63u))
                            : tail_care_ok(sv0, ok ? pos : 0u, o_care, seed_len_of((o_meta >> 3) & 63u));
        ok = ok && (!tail || t);
      }
    }
    n_verified += ok ? 1u : 0u;
    // ---- the turn's summaries, run by run
    uint32_t key = have ? 2u * own + (on_m ? 1u : 0u) : 0xFFFFFFFFu;
    RegionSummary acc = summary_empty();
    if (ok) acc = summary_one(mm, gp);
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      RegionSummary prev;
      prev.min_mm = shfl_up_pin(acc.min_mm, (uint32_t)d);
      prev.count = shfl_up_pin(acc.count, (uint32_t)d);
      prev.first = shfl_up_pin(acc.first, (uint32_t)d);
      prev.last = shfl_up_pin(acc.last, (uint32_t)d);
      const uint32_t pkey = shfl_up_pin(key, (uint32_t)d);
      const RegionSummary both = summary_merge(prev, acc);
      const bool join = lane >= (uint32_t)d && pkey == key;
      acc.min_mm = join ? both.min_mm : acc.min_mm;
      acc.count = join ? both.count : acc.count;
      acc.first = join ? both.first : acc.first;
      acc.last = join ? both.last : acc.last;
    }
    // ---- back to the owners: the last lane of my '+' run and of my '-' run in this turn
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const uint32_t lo = f ? off + n_p : off, hi = f ? off + n_p + n_m : off + n_p;
      const uint32_t a = lo > base ? lo : base, e = hi < base + 64 ? hi : base + 64;
      const bool mine = e > a;
      const int src = mine ? (int)(e - 1 - base) : 0;
      RegionSummary run;
      run.min_mm = shfl_pin(acc.min_mm, (uint32_t)src);
      run.count = shfl_pin(acc.count, (uint32_t)src);
      run.first = shfl_pin(acc.first, (uint32_t)src);
      run.last = shfl_pin(acc.last, (uint32_t)src);
      RegionSummary& sum = f ? sum_m : sum_p;
      const RegionSummary both = summary_merge(sum, run);
      sum.min_mm = mine ? both.min_mm : sum.min_mm;
      sum.count = mine ? both.count : sum.count;
      sum.first = mine ? both.first : sum.first;
      sum.last = mine ? both.last : sum.last;
    }
  }
}

// does the reference probe seed shift seed_i when the best mismatch count so far is mm?  (mapping.cpp:248-263: never
// again after an exact match; after a one-mismatch match only the first kExitOneMismatch shifts.)  Monotone: a larger
// mm never needs fewer seeds, a later seed is never needed when an earlier one is not.
__device__ __forceinline__ bool seed_needed(uint32_t mm, uint32_t seed_i) {
  return seed_i == 0 || (mm != 0 && !(mm == 1 && seed_i >= kExitOneMismatch));
}

// chunk c of the heavy list of H reads (HeavyStage::even): first read and number of reads
__device__ __forceinline__ void heavy_chunk_span(uint32_t H, uint32_t hcap, uint32_t c, uint32_t& first, uint32_t& len) {
  uint32_t k = (H + hcap - 1) / hcap;
  k = (k + 1) & ~1u;                               // the two halves of the pipeline get the same number of chunks
  const uint32_t per = k ? (((H + k - 1) / k + 63u) & ~63u) : 0u;  // <= hcap (a multiple of 64)
  first = c * per;
  len = first < H ? (H - first < per ? H - first : per) : 0u;
}

template <int NW, bool DIAG, bool HEAVY>
__device__ __forceinline__ void se_process_dual(const IndexView& iv, BlockShared& sh, const PreFilter& pf, const uint32_t* si,
                                                const uint32_t* __restrict__ codes2, uint64_t o_first,
                                                uint64_t o_read, uint64_t oe_read, uint32_t* __restrict__ err,
                                                uint32_t r,
                                                bool valid, uint32_t strand_base, uint32_t max_mm, uint32_t b,
                                                BestMatch* __restrict__ out, uint32_t* __restrict__ defer_count,
                                                uint32_t* __restrict__ defer_list, uint32_t* __restrict__ heavy_count,
                                                uint32_t* __restrict__ heavy_list, MapCounters& ctr_out,
                                                uint32_t& len_out, uint32_t ablate_rt, StampsT<DIAG>& st,
                                                WaveList* wl_heavy = nullptr, const SeCarry& carry = SeCarry()) {
  const uint32_t ablate = DIAG ? ablate_rt : 0u;
  const uint32_t n_chrom = iv.n_chrom;
  const uint32_t top_step = top_step_of(n_chrom);
  MapCounters ctr = {0, 0, 0};  // this read's work; counted once, by the pass that completes the read
  bool heavy = false;
  uint32_t heavy_seed = 0;  // pass 1: the seed shift it gives the read up at (seeds before it are complete: SeCarry)
  const uint32_t lane = threadIdx.x & 63;
  const StrandView& svp = iv.s[strand_base];
  const StrandView& svm = iv.s[strand_base + 1];
  const uint32_t ga = strand_base >> 1, Bd = iv.dir_bits;
  LaneRead<NW> lr;
  lane_load_read<NW>(lr, codes2, o_first, o_read, oe_read, valid, ga, err, iv);
  len_out = lr.len;
  bool mappable = valid && lr.len >= kMinReadLen;
  bool deferred = false;
  uint32_t defer_iter = 0;
  stamp(st, 0);

  BestMatch best;  // mapping.cpp:486
  best.genome_pos = 0; best.times = 0; best.strand = '+'; best.mismatch = max_mm;
  RegionSummary mneg[kPat];  // '-' strand, per seed
#pragma unroll
  for (uint32_t k = 0; k < kPat; ++k) mneg[k] = summary_empty();
  uint32_t minus_lb = 0xFFFFFFFFu;  // smallest mismatch count any kept '-' summary holds
  constexpr bool kLong = long_seed_nw<NW>();  // seeds of more than the 44 characters of hash + key exist
  const uint32_t seed_len = seed_len_of(lr.repeats);

#pragma unroll 1
  for (uint32_t seed_i = 0; seed_i < kPat; ++seed_i) {
    RegionSummary sum_p = summary_empty(), sum_m = summary_empty();
    const MapCounters ctr_seed = ctr;  // (a seed pass 1 gives up at is counted by the heavy pass, which does it again)
    const bool was_heavy = heavy;
    // '+': exact (mapping.cpp:250-257 with the state after the '+' folds so far)
    bool need_p = mappable && seed_needed(best.mismatch, seed_i);
    // '-': superset of the reference's decision (see header comment)
    const uint32_t lb = best.mismatch < minus_lb ? best.mismatch : minus_lb;
    bool need_m = mappable && seed_needed(lb, seed_i);
    if (ablate & 4u) need_p = need_m = false;

    uint32_t care[kCareWords] = {0, 0, 0, 0};
    uint32_t slot = 0, span = 0;
    if (need_p || need_m) seed_query<NW>(lr.rd, seed_len, seed_i, ga, Bd, sh.pcode4, care, slot, span);
    stamp(st, 1);
    // Bloom blocks of both strands and the directory pairs of both strands: independent loads, one wait
    const uint32_t bkey = bloom_key_of_care(care);
    uint64_t bw_p = 0, bw_m = 0;
    if (need_p && prefilter_hit(pf, 0, bkey)) bw_p = svp.bloom[bloom_block(bkey, svp.bloom_mask)];
    if (need_m && prefilter_hit(pf, 1, bkey)) bw_m = svm.bloom[bloom_block(bkey, svm.bloom_mask)];
    SlotProbe pp, pm;
    uint32_t hi_p, hi_m;
    probe_issue(svp, need_p, slot, span, pp, hi_p);
    probe_issue(svm, need_m, slot, span, pm, hi_m);
    const bool bad_p = need_p && bw_p && danger_filter_hit(bw_p, care);
    const bool bad_m = need_m && bw_m && danger_filter_hit(bw_m, care);
    if constexpr (DIAG) if (ablate & 8u) {
      // self-check (WALT_AMD_STAMPS=1 WALT_AMD_ABLATE=8; results stay valid): the filter must flag every probe
      // the exact test calls dangerous; violations are counted in stamp word 15
      if (need_p && !bad_p && probe_is_dangerous(svp, care, seed_len_of(lr.repeats))) atomicAdd(&st.buf[15], 1ull);
      if (need_m && !bad_m && probe_is_dangerous(svm, care, seed_len_of(lr.repeats))) atomicAdd(&st.buf[15], 1ull);
      if (need_p) atomicAdd(&st.buf[14], 1ull);  // probes checked
    }
    if (bad_p || bad_m) {
      if constexpr (!HEAVY) {
        // a filter hit is a SUPERSET of the dangerous probes (1.3 % of the reads of a 93-sequence genome against
        // ~0.3 % truly dangerous ones): the heavy pass applies the exact test before anything goes to the literal pass
        heavy = true;
        mappable = false;
        need_p = need_m = false;
      } else {
        const bool dng_p = bad_p && probe_is_dangerous(svp, care, seed_len_of(lr.repeats));
        const bool dng_m = bad_m && probe_is_dangerous(svm, care, seed_len_of(lr.repeats));
        if (dng_p || dng_m) {
          deferred = true;
          mappable = false;
          need_p = need_m = false;
          defer_iter = seed_i + (dng_p ? 0u : kPat);  // (grouping only: map_common.h kDeferShift)
          defer_iter = defer_iter < 7u ? defer_iter : 7u;
        }
      }
    }
    stamp(st, 2);
    pp.ne = (need_p && hi_p > pp.lo) ? hi_p - pp.lo : 0u;
    pm.ne = (need_m && hi_m > pm.lo) ? hi_m - pm.lo : 0u;
    if (ablate & 2u) pp.ne = pm.ne = 0;
    if (!HEAVY && (pp.ne > kScanMax || pm.ne > kScanMax)) {  // a long slot: the heavy pass searches it
      heavy = true;
      mappable = false;
      need_p = need_m = false;
      pp.ne = pm.ne = 0;
    }
    probe_entries(svp, pp);
    probe_entries(svm, pm);
    Lookup lp, lm;
    bool tail_p, tail_m;
    if constexpr (HEAVY) {
      probe_resolve_dual<kLong>(svp, svm, pp, pm, care, seed_len, lp, lm, tail_p, tail_m);
    } else {
      bool unres_p = false, unres_m = false;  // patterns 5 / 7: a key-equal range of several slots still owes its tail characters
      probe_resolve<kLong>(svp, pp, care, seed_len, lp, tail_p, &unres_p);
      probe_resolve<kLong>(svm, pm, care, seed_len, lm, tail_m, &unres_m);
      if (unres_p || unres_m) {  // the heavy pass narrows it
        heavy = true;
        mappable = false;
        lp.reg = empty_region(); lm.reg = empty_region();
        lp.npos = lm.npos = 0;
      }
    }
    stamp(st, 3);

    uint32_t size_p = lp.reg.l <= lp.reg.u ? lp.reg.u - lp.reg.l + 1 : 0;
    uint32_t size_m = lm.reg.l <= lm.reg.u ? lm.reg.u - lm.reg.l + 1 : 0;
    ctr.probes += (size_p ? 1u : 0u) + (size_m ? 1u : 0u);
    if (size_p > b) size_p = 0;  // mapping.cpp:275-277
    if (size_m > b) size_m = 0;
    if (ablate & 1u) size_p = size_m = 0;
    if (!HEAVY && (size_p > kSmallRegion || size_m > kSmallRegion)) {  // a large region: the heavy pass verifies it
      heavy = true;
      mappable = false;
      size_p = size_m = 0;
    }
    uint32_t mk[NW];
    make_masks<NW>(mk, sh.mask_table, seed_i, lr.repeats >= kMinRepeats ? lr.repeats : kMinRepeats, lr.len);
    stamp(st, 4);

    // small regions: candidate k of both strands checked side by side
    const bool small_p = size_p && size_p <= kSmallRegion, small_m = size_m && size_m <= kSmallRegion;
    if (small_p || small_m) {
      const uint32_t kmax = (small_p ? size_p : 0u) > (small_m ? size_m : 0u) ? size_p : (small_m ? size_m : size_p);
#pragma unroll 1
      for (uint32_t k = 0; k < kmax; ++k) {  // rolled (code size); pos[] picked by selects, not indexing
        const bool act_p = small_p && k < size_p, act_m = small_m && k < size_m;
        uint32_t pos_p = k == 0 ? lp.pos[0] : k == 1 ? lp.pos[1] : k == 2 ? lp.pos[2] : lp.pos[3];
        uint32_t pos_m = k == 0 ? lm.pos[0] : k == 1 ? lm.pos[1] : k == 2 ? lm.pos[2] : lm.pos[3];
        if (act_p && k >= lp.npos) pos_p = svp.ent[lp.reg.l + k].pos;
        if (act_m && k >= lm.npos) pos_m = svm.ent[lm.reg.l + k].pos;
        bool ok_p, ok_m;
        uint32_t gp_p, gp_m, mm_p, mm_m;
        if constexpr (kLong && kPat != 3) {  // long seeds: a single key-equal candidate still owes its care chars >= 44 (probe_resolve)
          bool t_p, t_m;
          const uint32_t cut = tail_care_cut(seed_i, seed_len);
          verify_nobranch_tail<NW>(svp, sh, si, iv.start_index, n_chrom, top_step, act_p, pos_p, seed_i, lr.len, lr.rd, mk, cut, ok_p, gp_p, mm_p, t_p);
          verify_nobranch_tail<NW>(svm, sh, si, iv.start_index, n_chrom, top_step, act_m, pos_m, seed_i, lr.len, lr.rd, mk, cut, ok_m, gp_m, mm_m, t_m);
          ok_p = ok_p && (!tail_p || t_p);
          ok_m = ok_m && (!tail_m || t_m);
        } else {
        verify_nobranch<NW>(svp, sh, si, iv.start_index, n_chrom, top_step, act_p, pos_p, seed_i, lr.len, lr.rd, mk, ok_p, gp_p, mm_p);
        verify_nobranch<NW>(svm, sh, si, iv.start_index, n_chrom, top_step, act_m, pos_m, seed_i, lr.len, lr.rd, mk, ok_m, gp_m, mm_m);
        if (kLong) {  // long seeds: a single key-equal candidate still owes its care chars >= 44 (probe_resolve)
          // inactive lanes carry no valid position: read from 0 like verify_nobranch does
          const bool t_p = tail_care_ok(svp, act_p ? pos_p : 0u, care, seed_len), t_m = tail_care_ok(svm, act_m ? pos_m : 0u, care, seed_len);
          ok_p = ok_p && (!tail_p || t_p);
          ok_m = ok_m && (!tail_m || t_m);
        }
        }
        if (ok_p) { sum_p = summary_merge(sum_p, summary_one(mm_p, gp_p)); ++ctr.verified; }
        if (ok_m) { sum_m = summary_merge(sum_m, summary_one(mm_m, gp_m)); ++ctr.verified; }
      }
    }
    stamp(st, 5);
    if constexpr (HEAVY) {
    // The one-kernel heavy pass (WALT_AMD_HEAVY=mono in the diagnostic build; kept for comparison, the product path is
    // the staged pass below): a lane with a large region fetches its dense range (two loads, all lanes together);
    // regions of up to kMidRegion candidates stay with their lane (se_mid_regions); the larger ones take the whole
    // wavefront, one owner at a time.
    DenseRange dr_p = dense_range(svp, lp.reg.l, size_p, size_p > kMidRegion && win_usable<NW>(svp, lr.len));
    DenseRange dr_m = dense_range(svm, lm.reg.l, size_m, size_m > kMidRegion && win_usable<NW>(svm, lr.len));
    {
      const uint32_t nmid_p = (size_p > kSmallRegion && size_p <= kMidRegion) ? size_p : 0u;
      const uint32_t nmid_m = (size_m > kSmallRegion && size_m <= kMidRegion) ? size_m : 0u;
      MidTail none;
      se_mid_regions<NW, false>(iv, sh, svp, svm, nmid_p, nmid_m, lp.reg.l, lm.reg.l, seed_i, lr.len, lr.rd, mk, n_chrom, top_step,
                                0u, none, sum_p, sum_m, ctr.verified);
    }
#pragma unroll 1
    for (uint32_t fi = 0; fi < 2; ++fi) {
      const StrandView& sv = fi ? svm : svp;
      const uint32_t my_size = fi ? size_m : size_p;
      const uint32_t my_l = fi ? lm.reg.l : lp.reg.l;
      unsigned long long big = __ballot(my_size > kMidRegion);
      while (big) {
        const int owner = (int)__ffsll((long long)big) - 1;
        big &= big - 1;
        uint32_t o_rd[NW], o_mk[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          o_rd[w] = bcast(lr.rd[w], owner);
          o_mk[w] = bcast(mk[w], owner);
        }
        const uint32_t o_l = bcast(my_l, owner), o_size = bcast(my_size, owner), o_len = bcast(lr.len, owner);
        DenseRange o_dr;
        o_dr.lo = bcast(fi ? dr_m.lo : dr_p.lo, owner);
        o_dr.hi = bcast(fi ? dr_m.hi : dr_p.hi, owner);
        o_dr.rec = bcast((uint32_t)(fi ? dr_m.rec : dr_p.rec), owner);  // record numbers stay below 2^32 (build_windows)
        uint32_t nv = 0;
        RegionSummary s = coop_region<NW>(sv, si, iv.start_index, n_chrom, o_l, o_size, seed_i, o_len, o_rd, o_mk, lane, nv, &o_dr, ablate);
        ctr_out.verified += nv;  // candidates of the OWNER's region that this lane verified
        if ((int)lane == owner) {
          if (fi) sum_m = s; else sum_p = s;
          ++ctr.big;
        }
      }
    }
    }
    stamp(st, 6);
    if (!HEAVY && heavy && !was_heavy) {  // given up at this seed: the seeds before it are what pass 1 hands over
      heavy_seed = seed_i;
      ctr = ctr_seed;
    }
    fold_region(best, sum_p, '+');  // empty when the '+' probe was not needed
#pragma unroll
    for (uint32_t k = 0; k < kPat; ++k)  // (selects: an index into the array would send it to scratch)
      if (seed_i == k) mneg[k] = sum_m;
    if (sum_m.count && sum_m.min_mm < minus_lb) minus_lb = sum_m.min_mm;
  }
  if constexpr (!HEAVY) {
    // a read given to the heavy pass: its state after the seeds pass 1 completed (best: '+' folds only; the '-' summaries
    // wait for the end as they do here) -- the staged rounds go on from there (SeCarry)
    if (heavy && !deferred && carry.st != nullptr) {
      carry.st[r] = make_uint4(best.genome_pos, best.times, best.mismatch, minus_lb);
#pragma unroll
      for (uint32_t k = 0; k + 1 < kPat; ++k)
        if (k < heavy_seed) carry.neg[k * carry.stride + r] = make_uint4(mneg[k].min_mm, mneg[k].count, mneg[k].first, mneg[k].last);
    }
    const bool carried = heavy && !deferred && carry.st != nullptr;
    // (only '+' folds have reached `best` here, so its strand is '+': nothing more to hand over)
    wavelist_append(*wl_heavy, heavy && !deferred, heavy_entry(r, carried ? heavy_seed : 0u), heavy_count, heavy_list);  // (buffered: map_common.h WaveList)
    if (carried) { ctr_out.probes += ctr.probes; ctr_out.verified += ctr.verified; ctr_out.big += ctr.big; }
  }
  // '-' strand folds in reference order under the exact exit conditions
  {
    bool go = true;
#pragma unroll
    for (uint32_t k = 0; k < kPat; ++k) {
      go = go && seed_needed(best.mismatch, k);
      if (go) fold_region(best, mneg[k], '-');
    }
  }
  wave_append(deferred, r <= kDeferMask ? (r | (defer_iter << kDeferShift)) : r, defer_count, defer_list);
  if (!deferred && !heavy && valid) out[r] = best;
  if (!deferred && !heavy) { ctr_out.probes += ctr.probes; ctr_out.verified += ctr.verified; ctr_out.big += ctr.big; }
  stamp(st, 7);
}

__device__ __forceinline__ void flush_counters(const MapCounters& ctr, uint32_t shortv,
                                               unsigned long long* __restrict__ shards) {
  block_flush_stats(shortv, ctr.probes, ctr.verified, ctr.big, shards);
}

// folds the shards into walt_batch_stats (accumulating) and clears them
static __global__ void k_reduce_stats(unsigned long long* __restrict__ shards, unsigned long long* __restrict__ stats) {
  const uint32_t t = threadIdx.x;  // one thread per shard
  unsigned long long v[4];
  for (int i = 0; i < 4; ++i) {
    v[i] = shards[(uint64_t)t * kStatShardWords + i];
    shards[(uint64_t)t * kStatShardWords + i] = 0;
  }
  __shared__ unsigned long long red[4][kStatShards];
  for (int i = 0; i < 4; ++i) red[i][t] = v[i];
  __syncthreads();
  if (t < 4) {
    unsigned long long sum = 0;
    for (uint32_t k = 0; k < kStatShards; ++k) sum += red[t][k];
    if (sum) atomicAdd(&stats[t], sum);
  }
}
// Counting sort of the deferred list by iteration tag (8 bins).  Each block owns
// a contiguous slice of the list and aggregates in LDS, so the global control
// words see 8 atomics per block.  ctl[0] = count, ctl[8..15] = bin totals
// (k_bin_count), ctl[16..23] = bin cursors (k_bin_scatter); zeroed per call.
constexpr unsigned kBinBlocks = 64;

__device__ __forceinline__ void bin_slice(uint32_t count, uint32_t& lo, uint32_t& hi) {
  const uint32_t per = (count + kBinBlocks - 1) / kBinBlocks;
  lo = blockIdx.x * per < count ? blockIdx.x * per : count;
  hi = lo + per < count ? lo + per : count;
}
// ctl2[0] = the list's length now (the side launch's share), ctl2[8..23] = 0 (its bins); rng = {that length, 0}
static __global__ void k_lit_snapshot(const uint32_t* __restrict__ count, uint32_t* __restrict__ ctl2, uint32_t* __restrict__ rng) {
  if (threadIdx.x == 0) { ctl2[0] = *count; rng[0] = *count; }
  if (threadIdx.x >= 8 && threadIdx.x < 24) ctl2[threadIdx.x] = 0;
}
// the second share of the side launches: the entries behind the first share (rng[0]) up to the list's length now --
// ctl3 = {their number, .., [8..23] = 0 (bins), [24..25] = their range}; rng[0] = the length now (what the last launch starts at)
static __global__ void k_lit_snapshot_rest(const uint32_t* __restrict__ count, uint32_t* __restrict__ ctl3, uint32_t* __restrict__ rng) {
  if (threadIdx.x == 0) {
    const uint32_t first = rng[0], c = *count;
    ctl3[0] = c > first ? c - first : 0u;
    ctl3[1] = first;
    ctl3[24] = first;
    ctl3[25] = c > first ? c : first;
    rng[0] = c > first ? c : first;
  }
  if (threadIdx.x >= 8 && threadIdx.x < 24) ctl3[threadIdx.x] = 0;
}
static __global__ void k_lit_end(const uint32_t* __restrict__ count, uint32_t* __restrict__ rng) { rng[1] = *count; }
// rng = {first, end}: the entries of the deferred list behind the `done` the literal rounds have mapped
static __global__ void k_lit_rest(const uint32_t* __restrict__ count, uint32_t* __restrict__ rng, uint32_t done) {
  const uint32_t c = *count;
  rng[0] = done < c ? done : c;
  rng[1] = c;
}
// (first != nullptr: the ctl[0] entries from *first on, sorted into the same places of `sorted`)
static __global__ void k_bin_count(uint32_t* __restrict__ ctl, const uint32_t* __restrict__ list, const uint32_t* __restrict__ first) {
  __shared__ uint32_t bins[8];
  if (threadIdx.x < 8) bins[threadIdx.x] = 0;
  __syncthreads();
  if (first != nullptr) list += *first;
  uint32_t lo, hi;
  bin_slice(ctl[0], lo, hi);
  const uint32_t lane = threadIdx.x & 63;
  for (uint32_t base = lo; base < hi; base += blockDim.x) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t bin = i < hi ? (list[i] >> kDeferShift) & 7u : 8u;
    for (uint32_t k = 0; k < 8; ++k) {
      const unsigned long long m = __ballot(bin == k);
      if (lane == 0 && m) atomicAdd(&bins[k], (uint32_t)__popcll(m));
    }
  }
  __syncthreads();
  if (threadIdx.x < 8 && bins[threadIdx.x]) atomicAdd(&ctl[8 + threadIdx.x], bins[threadIdx.x]);
}
static __global__ void k_bin_scatter(uint32_t* __restrict__ ctl, const uint32_t* __restrict__ list,
                                     uint32_t* __restrict__ sorted, const uint32_t* __restrict__ first) {
  __shared__ uint32_t bins[8], cursor[8];
  if (threadIdx.x < 8) bins[threadIdx.x] = 0;
  __syncthreads();
  if (first != nullptr) { list += *first; sorted += *first; }
  uint32_t lo, hi;
  bin_slice(ctl[0], lo, hi);
  const uint32_t lane = threadIdx.x & 63;
  for (uint32_t base = lo; base < hi; base += blockDim.x) {  // this block's bin counts
    const uint32_t i = base + threadIdx.x;
    const uint32_t bin = i < hi ? (list[i] >> kDeferShift) & 7u : 8u;
    for (uint32_t k = 0; k < 8; ++k) {
      const unsigned long long m = __ballot(bin == k);
      if (lane == 0 && m) atomicAdd(&bins[k], (uint32_t)__popcll(m));
    }
  }
  __syncthreads();
  if (threadIdx.x < 8) {  // reserve this block's range inside every bin
    uint32_t start = 0;
    for (uint32_t k = 0; k < threadIdx.x; ++k) start += ctl[8 + k];
    cursor[threadIdx.x] = start + (bins[threadIdx.x] ? atomicAdd(&ctl[16 + threadIdx.x], bins[threadIdx.x]) : 0u);
  }
  __syncthreads();
  for (uint32_t base = lo; base < hi; base += blockDim.x) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t e = i < hi ? list[i] : 0u;
    const uint32_t bin = i < hi ? (e >> kDeferShift) & 7u : 8u;
    for (uint32_t k = 0; k < 8; ++k) {
      const unsigned long long m = __ballot(bin == k);
      if (!m) continue;
      uint32_t off = 0;
      if (lane == 0) off = atomicAdd(&cursor[k], (uint32_t)__popcll(m));
      off = bcast(off, 0);
      if (bin == k) sorted[off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = e & kDeferMask;
    }
  }
}
void launch_bin_deferred(uint32_t* d_ctl, const uint32_t* d_list, uint32_t* d_sorted, hipStream_t stream, const uint32_t* d_first) {
  hipLaunchKernelGGL(k_bin_count, dim3(kBinBlocks), dim3(kBlock), 0, stream, d_ctl, d_list, d_first);
  hipLaunchKernelGGL(k_bin_scatter, dim3(kBinBlocks), dim3(kBlock), 0, stream, d_ctl, d_list, d_sorted, d_first);
}

void launch_reduce_stats(unsigned long long* d_shards, unsigned long long* d_stats, hipStream_t stream) {
  hipLaunchKernelGGL(k_reduce_stats, dim3(1), dim3(kStatShards), 0, stream, d_shards, d_stats);
}

// pass 1: every read of the batch, one per lane (HEAVY = false); the one-kernel heavy pass over the heavy list (HEAVY = true)
template <int NW, bool DIAG, bool HEAVY>
__global__ __launch_bounds__(kBlock, HEAVY ? (NW <= 8 ? 3 : (NW <= 10 ? 2 : 1)) : (NW <= 8 ? 4 : (NW <= 10 ? 3 : 1))) void k_map_se(IndexView iv, const uint32_t* __restrict__ codes2,
                                                    const uint64_t* __restrict__ offsets,
                                                    uint32_t* __restrict__ err,
                                                    uint32_t n_all, uint32_t strand_base,
                                                    uint32_t max_mm, uint32_t b,
                                                    const uint32_t* __restrict__ mask_table,
                                                    BestMatch* __restrict__ out,
                                                    unsigned long long* __restrict__ stats,
                                                    uint32_t* __restrict__ defer_count,
                                                    uint32_t* __restrict__ defer_list,
                                                    uint32_t* __restrict__ heavy_count,
                                                    uint32_t* __restrict__ heavy_list, uint32_t ablate,
                                                    unsigned long long* __restrict__ stamps,
                                                    SeCarry carry = SeCarry()) {
  const uint32_t n = HEAVY ? *heavy_count : n_all;
  if (HEAVY && n == 0) return;
  __shared__ BlockShared sh;
  __shared__ PreFilter pf;
  // per-wavefront list buffers of pass 1's heavy list
  __shared__ uint32_t s_wl[HEAVY ? 1 : (kBlock / 64) * kWaveBuf];
  WaveList wl_heavy;
  wl_heavy.buf = s_wl + (HEAVY ? 0 : (threadIdx.x >> 6) * kWaveBuf);
  wl_heavy.n = 0;
  wl_heavy.cap = kWaveBuf;
  prefilter_stage(pf, iv, strand_base);
  const uint32_t* si = block_prologue(sh, iv, mask_table, strand_base);
  // persistent blocks: the LDS prologue (mask table, chromosome starts, Bloom
  // filters: ~25 KB) is paid once per block, not once per 256 reads
  MapCounters ctr = {0, 0, 0};
  uint32_t shortv = 0;
  StampsT<DIAG> st;
  stamp_begin(st, stamps);
  // each block walks its own contiguous slice of the batch (consecutive 256-read
  // chunks share pages: a strided assignment made every load a TLB miss)
  const uint64_t chunks = ((uint64_t)n + blockDim.x - 1) / blockDim.x;
  const uint64_t per_block = (chunks + gridDim.x - 1) / gridDim.x;
  const uint64_t c_lo = (uint64_t)blockIdx.x * per_block;
  const uint64_t c_hi = c_lo + per_block < chunks ? c_lo + per_block : chunks;
  const uint64_t o_first = offsets[0];
  // this lane's read offsets are fetched one chunk ahead
  uint64_t o_nx = 0, oe_nx = 0;
  uint32_t r_nx = 0;
  auto fetch = [&](uint64_t i) {
    r_nx = HEAVY ? (heavy_list[i] & kDeferMask) : (uint32_t)i;  // (heavy_entry: the seed pass 1 stopped at rides on top)
    o_nx = offsets[r_nx];
    oe_nx = offsets[(uint64_t)r_nx + 1];
  };
  if (c_lo < c_hi && c_lo * blockDim.x + threadIdx.x < n) fetch(c_lo * blockDim.x + threadIdx.x);
  for (uint64_t c = c_lo; c < c_hi; ++c) {
    const uint64_t i64 = c * blockDim.x + threadIdx.x;
    const bool valid = i64 < n;
    const uint32_t r = valid ? r_nx : 0;
    const uint64_t o_cur = o_nx, oe_cur = oe_nx;
    if (c + 1 < c_hi && i64 + blockDim.x < n) fetch(i64 + blockDim.x);
    uint32_t len;
    se_process_dual<NW, DIAG, HEAVY>(iv, sh, pf, si, codes2, o_first, o_cur, oe_cur, err, r, valid, strand_base, max_mm, b,
                                     out, defer_count, defer_list, heavy_count, heavy_list, ctr, len, ablate, st, &wl_heavy, carry);
    // too_short is counted once per strand pass (mapping.cpp:230-233); pass 1 sees every read
    if (!HEAVY) shortv += (valid && len < kMinReadLen) ? 2u : 0u;
  }
  if constexpr (!HEAVY) wavelist_flush(wl_heavy, heavy_count, heavy_list);
  stamp_end(st);
  flush_counters(ctr, shortv, stats);
}

#if defined(WALT_DIAG)
// Diagnostic build: phase sums of k_se_stage (WALT_AMD_STAMPS=4; walt_profile_stage_stamps).  Phases: 0 taking a read
// (record / state loads), 1 seed query + filter + directory loads, 2 exact danger test, 3 entries + resolve, 4 masks +
// small regions, 5 dense range + mid regions, 6 work items, 7 state store / lists / finished reads, 8 total.
__device__ unsigned long long g_stage_stamps[16];
__device__ uint32_t g_stage_stamps_on;
#define STG_DECL StampsT<true> sst; const bool sst_on = (g_stage_stamps_on & 1u) != 0; stamp_begin(sst, g_stage_stamps)
#define STG(k) do { if (sst_on) stamp(sst, (k)); } while (0)
#define STG_END do { if (sst_on) stamp_end(sst); } while (0)
#else
#define STG_DECL
#define STG(k)
#define STG_END
#endif
// ---------------------------------------------------------------------------
// k_se_stage: one ROUND of the staged heavy pass over one chunk of the heavy list (HeavyStage).  One read per lane;
// a lane runs its seed shifts until a probe produces a work item, a dangerous probe sends the read to the literal list,
// or the read is finished.  What the seed loop does per seed is what pass 1 does, with long slots searched (both
// strands together: map_common.h probe_resolve_dual) and regions of up to kMidRegion candidates verified in the lane.
// ---------------------------------------------------------------------------
// LIT (the literal rounds, launch_map_se): the rounds over the reads the ordinary rounds gave up at a truly dangerous probe;
// here such a probe's region comes from the reference's own search (core.h seed_lookup_ex -> lit_region_inferred: a few
// key searches, hardly an entry load) and the read goes on like any other.  Until round 4 these reads were mapped from
// scratch by the strand-major kernel of round 1 (k_map_se_literal: ~12 ns a read whatever its probes cost -- 120 ms for
// the 9.5 M such reads of an hg19-scale assembly of 3,000 contigs).
// MIDS = false (measured in round 4, not instantiated): regions of 5 .. 16 candidates become (gather) work items like the
// larger ones instead of being verified by their lane: 4 spilled registers fewer, 41.5 -> 43.7 ms per 50 M reads.
template <int NW, int OCC = 0, bool LIT = false, bool MIDS = true>  // OCC: wavefronts per SIMD the registers are capped for (0: the default)
__global__ __launch_bounds__(kBlock, OCC ? OCC : (LIT ? (NW <= 8 ? 3 : (NW <= 10 ? 2 : 1)) : (NW <= 8 ? 4 : (NW <= 10 ? 3 : 1)))) void k_se_stage(
    IndexView iv, const uint32_t* __restrict__ codes2, const uint64_t* __restrict__ offsets, uint32_t* __restrict__ err,
    uint32_t strand_base, uint32_t max_mm, uint32_t b, const uint32_t* __restrict__ mask_table, BestMatch* __restrict__ out,
    unsigned long long* __restrict__ stats, uint32_t* __restrict__ defer_count, uint32_t* __restrict__ defer_list,
    const uint32_t* __restrict__ heavy_count, const uint32_t* __restrict__ heavy_list, HeavyStage hs, SeCarry carry) {
  uint32_t n = *heavy_count, h_first = 0;
  if (hs.even) {
    heavy_chunk_span(n, hs.hcap, hs.chunk, h_first, n);
  } else {
    h_first = hs.first;
    n = n > hs.first ? n - hs.first : 0u;
    n = n < hs.hcap ? n : hs.hcap;
  }
  if (hs.round) n = hs.count_in[4] < hs.hcap ? hs.count_in[4] : hs.hcap;
  if (n == 0) return;
  __shared__ BlockShared sh;
  __shared__ PreFilter pf;
  __shared__ uint32_t s_wl[(kBlock / 64) * kWaveBuf];  // per-wavefront buffers of the next round's list
  WaveList wl;
  wl.buf = s_wl + (threadIdx.x >> 6) * kWaveBuf;
  wl.n = 0;
  wl.cap = kWaveBuf;
  prefilter_stage(pf, iv, strand_base);
  const uint32_t* si = block_prologue(sh, iv, mask_table, strand_base);
  constexpr uint32_t RQ = rd_quads<NW>();
  constexpr bool kLong = long_seed_nw<NW>();                    // seeds of more than the 44 characters of hash + key exist
  constexpr bool kDefer = kLong && NW <= 10;                    // ... and the verifier narrows their key-equal ranges (section 4b)
  constexpr bool kMulti = kLong && kPat != 3 && NW <= 10;       // patterns 5 / 7: short key-equal ranges verified side by side in the lane
  const uint32_t n_chrom = iv.n_chrom, top_step = top_step_of(n_chrom);
  const StrandView& svp = iv.s[strand_base];
  const StrandView& svm = iv.s[strand_base + 1];
  const uint32_t ga = strand_base >> 1, Bd = iv.dir_bits;
  const uint64_t hcap = hs.hcap;
  heavy_list += h_first;
  MapCounters ctr = {0, 0, 0};
  ItemQueue q;
  q.items = hs.items; q.ctl = hs.ctl; q.cap = 2 * hs.hcap; q.ovf = err + 2;
  q.bigs = hs.giants; q.big_n = hs.ctl + 5; q.big_cap = hs.hcap / 8;
  const uint64_t chunks = ((uint64_t)n + blockDim.x - 1) / blockDim.x;
  const uint64_t per_block = (chunks + gridDim.x - 1) / gridDim.x;
  const uint64_t c_lo = (uint64_t)blockIdx.x * per_block;
  const uint64_t c_hi = c_lo + per_block < chunks ? c_lo + per_block : chunks;
  const uint64_t o_first = offsets[0];
  // The block's slice of the list goes to its wavefronts in equal parts, and a wavefront hands ITS part out read by read:
  // the lanes that need a read take the next ones in lane order (round 4, late; with a fixed column of reads per lane --
  // c * 256 + thread -- the lanes whose reads were done early waited for the lane with the most seed shifts: 42 of 64
  // lanes in a seed shift).  The list entries (round 0: and the reads' offsets) are fetched a window of 64 ahead, lane L
  // holding entry `win + L` (A) and `win + 64 + L` (B); a taker gets its entry by ds_bpermute.
  const uint32_t lane_id = threadIdx.x & 63u;
  const uint64_t b_lo = c_lo * blockDim.x, b_hi = c_hi * blockDim.x < n ? c_hi * blockDim.x : n;
  const uint64_t per_wave = ((((b_hi > b_lo ? b_hi - b_lo : 0) + (blockDim.x >> 6) - 1) / (blockDim.x >> 6)) + 63) & ~63ull;
  const uint64_t w_lo = b_lo + (threadIdx.x >> 6) * per_wave < b_hi ? b_lo + (threadIdx.x >> 6) * per_wave : b_hi;
  const uint64_t w_hi = w_lo + per_wave < b_hi ? w_lo + per_wave : b_hi;
  uint64_t cursor = w_lo, win = w_lo;  // (wave-uniform) next read to hand out; first entry of window A
  uint32_t e_a = 0, e_b = 0;
  uint64_t o_a = 0, oe_a = 0, o_b = 0, oe_b = 0;
  auto fetch = [&](uint64_t i, uint32_t& e, uint64_t& o, uint64_t& oe) {
    e = 0; o = 0; oe = 0;
    if (i < w_hi) {
      if (hs.round == 0) {
        e = heavy_list[i];
        const uint32_t r = e & kDeferMask;
        o = offsets[r];
        oe = offsets[(uint64_t)r + 1];
      } else {
        e = hs.list_in[i];
      }
    }
  };
  fetch(win + lane_id, e_a, o_a, oe_a);
  fetch(win + 64 + lane_id, e_b, o_b, oe_b);
  // Every LANE has a read of its own, takes it one seed shift further per turn of the loop below, and takes its next
  // read as soon as this one is blocked, deferred or finished (round 4, late): until then the 64 reads of a wavefront
  // iteration went through the seed shifts together, and a lane whose read was done after the first shift idled while some
  // other lane ran a second and a third -- 29 of 64 lanes per vector instruction.  Per-lane state of the current read:
  bool have = false;
  LaneRead<NW> lr;
  lr.len = 0; lr.repeats = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) lr.rd[w] = 0;
  BestMatch best;  // mapping.cpp:486; only '+' folds reach it before the end, so its strand is '+' throughout
  best.genome_pos = 0; best.times = 0; best.strand = '+'; best.mismatch = max_mm;
  uint32_t minus_lb = 0xFFFFFFFFu;  // smallest mismatch count any kept '-' summary holds
  uint32_t j = 0, r = 0, seed_i = 0, seed_len = 0, defer_iter = 0;
  bool active = false, fin = false, deferred = false;
  STG_DECL;
  for (;;) {
    const uint64_t takers = __ballot(!have);
    const uint64_t i64 = cursor + __builtin_amdgcn_mbcnt_hi((uint32_t)(takers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)takers, 0u));
    const bool valid = !have && i64 < w_hi;  // this lane takes its next read now
    if (__ballot(valid)) {
    // its list entry: in window A or B (cursor < win + 64 and at most 64 takers: i64 < win + 128)
    const uint32_t rel = valid ? (uint32_t)(i64 - win) : 0u;
    const int src = (int)(rel & 63u);
    const bool in_b = rel >= 64u;
    const uint32_t ea = shfl_pin(e_a, (uint32_t)src), eb = shfl_pin(e_b, (uint32_t)src);
    const uint32_t e_cur = in_b ? eb : ea;
    uint64_t o_cur = 0, oe_cur = 0;
    if (hs.round == 0) {  // (uniform)
      const uint32_t ola = shfl_pin((uint32_t)o_a, (uint32_t)src), olb = shfl_pin((uint32_t)o_b, (uint32_t)src);
      const uint32_t oha = shfl_pin((uint32_t)(o_a >> 32), (uint32_t)src), ohb = shfl_pin((uint32_t)(o_b >> 32), (uint32_t)src);
      const uint32_t ela = shfl_pin((uint32_t)oe_a, (uint32_t)src), elb = shfl_pin((uint32_t)oe_b, (uint32_t)src);
      const uint32_t eha = shfl_pin((uint32_t)(oe_a >> 32), (uint32_t)src), ehb = shfl_pin((uint32_t)(oe_b >> 32), (uint32_t)src);
      o_cur = (uint64_t)(in_b ? olb : ola) | ((uint64_t)(in_b ? ohb : oha) << 32);
      oe_cur = (uint64_t)(in_b ? elb : ela) | ((uint64_t)(in_b ? ehb : eha) << 32);
    }
    cursor += (uint64_t)__popcll(takers);
    cursor = cursor < w_hi ? cursor : w_hi;
    if (cursor >= win + 64) {  // (uniform) window A is used up: B becomes A, the next 64 entries are fetched
      e_a = e_b; o_a = o_b; oe_a = oe_b;
      win += 64;
      fetch(win + 64 + lane_id, e_b, o_b, oe_b);
    }

    // ---- the read and its state (into t_*: the lanes that keep their read keep everything)
    LaneRead<NW> t_lr;
    BestMatch t_best;
    t_best.genome_pos = 0; t_best.times = 0; t_best.strand = '+'; t_best.mismatch = max_mm;
    uint32_t t_minus_lb = 0xFFFFFFFFu;
    uint32_t t_j = 0, t_r = 0, t_seed_i = 0;
    {
    LaneRead<NW>& lr = t_lr;
    BestMatch& best = t_best;
    uint32_t& minus_lb = t_minus_lb;
    uint32_t& j = t_j; uint32_t& r = t_r; uint32_t& seed_i = t_seed_i;
    if (hs.round == 0) {
      j = (uint32_t)i64;
      r = valid ? e_cur & kDeferMask : 0u;
      lane_load_read<NW>(lr, codes2, o_first, o_cur, oe_cur, valid, ga, err, iv);
      if (valid) {
        if (carry.st != nullptr) {  // (uniform) what pass 1 worked out before it gave the read up
          seed_i = (e_cur >> kHeavySeedShift) & 7u;
          const uint4 cs = load_global(carry.st + r);
          uint4 ng[kPat > 1 ? kPat - 1 : 1];
#pragma unroll
          for (uint32_t k = 0; k + 1 < kPat; ++k)
            if (k < seed_i) ng[k] = load_global(carry.neg + k * carry.stride + r);
          best.genome_pos = cs.x; best.times = cs.y; best.mismatch = cs.z;
          minus_lb = cs.w;
#pragma unroll
          for (uint32_t k = 0; k + 1 < kPat; ++k)
            if (k < seed_i) hs.sums[(uint64_t)(1 + k) * hcap + j] = ng[k];
        }
        uint32_t w[4 * RQ];
#pragma unroll
        for (uint32_t t = 0; t < 4 * RQ; ++t) w[t] = t == 0 ? lr.len : (t <= (uint32_t)NW ? lr.rd[t - 1] : 0u);
#pragma unroll
        for (uint32_t t = 0; t < RQ; ++t) hs.rdq[(uint64_t)t * hcap + j] = make_uint4(w[4 * t], w[4 * t + 1], w[4 * t + 2], w[4 * t + 3]);
      }
    } else {
      lr.len = 0; lr.repeats = 0;
#pragma unroll
      for (int w = 0; w < NW; ++w) lr.rd[w] = 0;
      if (valid) {  // everything by j: one round of loads
        j = e_cur & ((1u << kStageJBits) - 1u);
        const uint32_t s_blocked = (e_cur >> kStageJBits) & 7u;
        const uint32_t he = heavy_list[j];
        const uint4 stv = load_global(hs.st + j);
        const uint4 sp = load_global(hs.sums + j), sm = load_global(hs.sums + (uint64_t)(1 + s_blocked) * hcap + j);
        uint4 rq[RQ];
#pragma unroll
        for (uint32_t t = 0; t < RQ; ++t) rq[t] = load_global(hs.rdq + (uint64_t)t * hcap + j);
        r = he & kDeferMask;
        uint32_t w[4 * RQ];
#pragma unroll
        for (uint32_t t = 0; t < RQ; ++t) { w[4 * t] = rq[t].x; w[4 * t + 1] = rq[t].y; w[4 * t + 2] = rq[t].z; w[4 * t + 3] = rq[t].w; }
        lr.len = w[0];
        lr.repeats = lr.len >= kMinReadLen ? seed_repeats(lr.len) : 0;
#pragma unroll
        for (int t = 0; t < NW; ++t) lr.rd[t] = w[t + 1];
        best.genome_pos = stv.x; best.times = stv.y; best.mismatch = stv.z;
        minus_lb = stv.w;
        // the seed the read was blocked at: its '+' summary (the lane's own, or the verifier's) is folded now, its '-'
        // summary stays in its row for the end and counts for the '-' bound (mapping.cpp:306-313, 250-257)
        RegionSummary a; a.min_mm = sp.x; a.count = sp.y; a.first = sp.z; a.last = sp.w;
        fold_region(best, a, '+');
        if (sm.y && sm.x < minus_lb) minus_lb = sm.x;
        seed_i = s_blocked + 1;
      }
    }
    }
    if (valid) {
      lr.len = t_lr.len; lr.repeats = t_lr.repeats;
#pragma unroll
      for (int w = 0; w < NW; ++w) lr.rd[w] = t_lr.rd[w];
      best = t_best;
      minus_lb = t_minus_lb;
      j = t_j; r = t_r; seed_i = t_seed_i;
      seed_len = seed_len_of(lr.repeats);
      have = true;
      active = lr.len >= kMinReadLen && seed_i < kPat;
      fin = !active;  // finished: the '-' folds and the record are left (seed_i = seeds that have been done)
      deferred = false;
      defer_iter = 0;
    }
    }
    STG(0);
    if (!__ballot(have)) break;  // the wavefront's part of the list is used up

    // ---- one seed shift for the lanes whose read goes on (a lane's seed_i is its own)
    if (__ballot(active)) {
      // '+': exact (mapping.cpp:250-257 with the state after the '+' folds so far); '-': a superset of the reference's
      // decision (se_process_dual's header comment).  Monotone in the seed: when neither is needed, nothing later is.
      bool need_p = active && seed_needed(best.mismatch, seed_i);
      const uint32_t lb = best.mismatch < minus_lb ? best.mismatch : minus_lb;
      bool need_m = active && seed_needed(lb, seed_i);
      if (active && !need_p && !need_m) { fin = true; active = false; }
      if (__ballot(active)) {
      WALT_DIAG_COUNT(2, 1);                                // seed steps of a wavefront
      WALT_DIAG_COUNT(3, __popcll(__ballot(active)));       // ... and the lanes in them
      const uint32_t sd = active ? seed_i : 0u;  // (an idle lane computes along on seed 0)
      RegionSummary sum_p = summary_empty(), sum_m = summary_empty();
      bool pend_p = false, pend_m = false;  // the summary comes from k_se_verify

      uint32_t care[kCareWords] = {0, 0, 0, 0};
      uint32_t slot = 0, span = 0;
      if (need_p || need_m) seed_query<NW>(lr.rd, seed_len, sd, ga, Bd, sh.pcode4, care, slot, span);
      // Bloom blocks of both strands and the directory pairs of both strands: independent loads, one wait
      const uint32_t bkey = bloom_key_of_care(care);
      uint64_t bw_p = 0, bw_m = 0;
      if (need_p && prefilter_hit(pf, 0, bkey)) bw_p = svp.bloom[bloom_block(bkey, svp.bloom_mask)];
      if (need_m && prefilter_hit(pf, 1, bkey)) bw_m = svm.bloom[bloom_block(bkey, svm.bloom_mask)];
      SlotProbe pp, pm;
      uint32_t hi_p, hi_m;
      probe_issue(svp, need_p, slot, span, pp, hi_p);
      probe_issue(svm, need_m, slot, span, pm, hi_m);
      const bool bad_p = need_p && bw_p && danger_filter_hit(bw_p, care);
      const bool bad_m = need_m && bw_m && danger_filter_hit(bw_m, care);
      STG(1);
      bool lit_p = false, lit_m = false;  // LIT: this strand's region comes from the reference's search
      if (bad_p || bad_m) {  // a filter hit is a superset of the dangerous probes: the exact test
        const bool dng_p = bad_p && probe_is_dangerous(svp, care, seed_len);
        const bool dng_m = bad_m && probe_is_dangerous(svm, care, seed_len);
        if constexpr (LIT) {
          lit_p = dng_p;
          lit_m = dng_m;
        } else if (dng_p || dng_m) {  // the literal rounds take the read up again
          deferred = true;
          active = false;
          need_p = need_m = false;
          defer_iter = sd + (dng_p ? 0u : kPat);  // (grouping only: map_common.h kDeferShift)
          defer_iter = defer_iter < 7u ? defer_iter : 7u;
        }
      }
      STG(2);
      pp.ne = (need_p && !lit_p && hi_p > pp.lo) ? hi_p - pp.lo : 0u;
      pm.ne = (need_m && !lit_m && hi_m > pm.lo) ? hi_m - pm.lo : 0u;
#if defined(WALT_DIAG)
      if (WALT_DIAG_TWICE(3)) {  // (other lines of the same slots: the entries behind the first four)
        SlotProbe p2 = pp, m2 = pm;
        p2.lo += p2.ne > 64 ? 48 : 0; m2.lo += m2.ne > 64 ? 48 : 0;
        probe_entries(svp, p2);
        probe_entries(svm, m2);
        if (p2.e[1].pos == 0xFFFFFFF0u && m2.e[1].pos == 0xFFFFFFF1u) pp.lo = 0;  // (keeps the loads alive)
      }
#endif
      probe_entries(svp, pp);
      probe_entries(svm, pm);
      Lookup lp, lm;
      bool tail_p, tail_m;
      bool defer_p = false, defer_m = false;  // long seeds: the verifier narrows the key-equal range (map_common.h DEFER)
      if constexpr (kDefer) {
        probe_resolve_dual<true, true>(svp, svm, pp, pm, care, seed_len, lp, lm, tail_p, tail_m, &defer_p, &defer_m, hs.defer_min,
                                       win_usable<NW>(svp, lr.len) && win_usable<NW>(svm, lr.len), kPat != 3 ? kMidRegion : 0u);
      } else {
        probe_resolve_dual<kLong>(svp, svm, pp, pm, care, seed_len, lp, lm, tail_p, tail_m);
      }
      if constexpr (LIT) {  // IndexRegion as the reference runs it (positions are fetched with the candidates)
        if (lit_p && need_p && !(hs.lit_ablate & 1u)) { seed_lookup_ex(iv, svp, care, slot, span, seed_len, lp, false); tail_p = false; defer_p = false; }
        if (lit_m && need_m && !(hs.lit_ablate & 1u)) { seed_lookup_ex(iv, svm, care, slot, span, seed_len, lm, false); tail_m = false; defer_m = false; }
      }
      uint32_t size_p = lp.reg.l <= lp.reg.u ? lp.reg.u - lp.reg.l + 1 : 0;
      uint32_t size_m = lm.reg.l <= lm.reg.u ? lm.reg.u - lm.reg.l + 1 : 0;
      STG(3);
      ctr.probes += (size_p ? 1u : 0u) + (size_m ? 1u : 0u);
      // patterns 5 / 7: a key-equal range of several slots whose candidates owe their tail characters (probe_resolve_dual)
      MidTail mt;
      mt.multi_p = kMulti && tail_p && size_p > 1;
      mt.multi_m = kMulti && tail_m && size_m > 1;
      if (size_p > b && !defer_p && !mt.multi_p) size_p = 0;  // mapping.cpp:275-277 (a deferred range: the verifier counts the region)
      if (size_m > b && !defer_m && !mt.multi_m) size_m = 0;
      uint32_t mk[NW];
      make_masks<NW>(mk, sh.mask_table, sd, lr.repeats >= kMinRepeats ? lr.repeats : kMinRepeats, lr.len);
      const uint32_t tail_cut = tail_care_cut(sd, seed_len);

      constexpr bool kCoop = kPat == 3;  // the lanes' own regions as one list over the wavefront (coop_lane_regions)
      constexpr uint32_t kLaneMax = kCoop ? kCoopRegion : ((MIDS || kMulti) ? kMidRegion : kSmallRegion);  // largest region a lane verifies itself
      const bool big_p = size_p > kLaneMax || defer_p, big_m = size_m > kLaneMax || defer_m;
      if constexpr (kCoop) {
        const uint32_t own_p = big_p ? 0u : size_p, own_m = big_m ? 0u : size_m;
#if defined(WALT_DIAG)
        if (WALT_DIAG_TWICE(2) && __ballot(own_p | own_m)) {
          RegionSummary d_p = summary_empty(), d_m = summary_empty();
          uint32_t d_n = 0;
          coop_lane_regions<NW>(iv, sh, svp, svm, own_p, own_m, lp.reg.l, lm.reg.l, lp.pos[0], lm.pos[0], lp.npos != 0, lm.npos != 0,
                                sd, lr, mk, tail_p, tail_m, care, n_chrom, d_p, d_m, d_n);
          if (d_n == 0xFFFFFFF0u) sum_p = d_p;  // (keeps the copy alive)
        }
#endif
        if (__ballot(own_p | own_m))
          coop_lane_regions<NW>(iv, sh, svp, svm, own_p, own_m, lp.reg.l, lm.reg.l, lp.pos[0], lm.pos[0], lp.npos != 0, lm.npos != 0,
                                sd, lr, mk, tail_p, tail_m, care, n_chrom, sum_p, sum_m, ctr.verified);
      }
      // small regions: candidate k of both strands checked side by side
      const bool small_p = !kCoop && size_p && size_p <= kSmallRegion, small_m = !kCoop && size_m && size_m <= kSmallRegion;
      if (__ballot(small_p || small_m)) {
        const uint32_t kmax = (small_p ? size_p : 0u) > (small_m ? size_m : 0u) ? size_p : (small_m ? size_m : 0u);
#pragma unroll 1
        for (uint32_t k = 0; __ballot(k < kmax); ++k) {  // rolled (code size); pos[] picked by selects, not indexing
          const bool act_p = small_p && k < size_p, act_m = small_m && k < size_m;
          uint32_t pos_p = k == 0 ? lp.pos[0] : k == 1 ? lp.pos[1] : k == 2 ? lp.pos[2] : lp.pos[3];
          uint32_t pos_m = k == 0 ? lm.pos[0] : k == 1 ? lm.pos[1] : k == 2 ? lm.pos[2] : lm.pos[3];
          if (act_p && k >= lp.npos) pos_p = svp.ent[lp.reg.l + k].pos;
          if (act_m && k >= lm.npos) pos_m = svm.ent[lm.reg.l + k].pos;
          bool ok_p, ok_m;
          uint32_t gp_p, gp_m, mm_p, mm_m;
          if constexpr (kLong && kPat != 3) {  // long seeds: a key-equal candidate still owes its care chars >= 44 (probe_resolve)
            bool t_p, t_m;
            verify_nobranch_tail<NW>(svp, sh, si, iv.start_index, n_chrom, top_step, act_p, pos_p, sd, lr.len, lr.rd, mk, tail_cut, ok_p, gp_p, mm_p, t_p);
            verify_nobranch_tail<NW>(svm, sh, si, iv.start_index, n_chrom, top_step, act_m, pos_m, sd, lr.len, lr.rd, mk, tail_cut, ok_m, gp_m, mm_m, t_m);
            mt.fb_p = mt.fb_p || (mt.multi_p && act_p && !ok_p);
            mt.fb_m = mt.fb_m || (mt.multi_m && act_m && !ok_m);
            mt.nin_p += (mt.multi_p && ok_p && t_p) ? 1u : 0u;
            mt.nin_m += (mt.multi_m && ok_m && t_m) ? 1u : 0u;
            ok_p = ok_p && (!tail_p || t_p);
            ok_m = ok_m && (!tail_m || t_m);
          } else {
            verify_nobranch<NW>(svp, sh, si, iv.start_index, n_chrom, top_step, act_p, pos_p, sd, lr.len, lr.rd, mk, ok_p, gp_p, mm_p);
            verify_nobranch<NW>(svm, sh, si, iv.start_index, n_chrom, top_step, act_m, pos_m, sd, lr.len, lr.rd, mk, ok_m, gp_m, mm_m);
            if (kLong) {  // long seeds: a single key-equal candidate still owes its care chars >= 44 (probe_resolve)
              // inactive lanes carry no valid position: read from 0 like verify_nobranch does
              const bool t_p = tail_care_ok(svp, act_p ? pos_p : 0u, care, seed_len), t_m = tail_care_ok(svm, act_m ? pos_m : 0u, care, seed_len);
              ok_p = ok_p && (!tail_p || t_p);
              ok_m = ok_m && (!tail_m || t_m);
            }
          }
          if (ok_p) { sum_p = summary_merge(sum_p, summary_one(mm_p, gp_p)); ++ctr.verified; }
          if (ok_m) { sum_m = summary_merge(sum_m, summary_one(mm_m, gp_m)); ++ctr.verified; }
        }
      }
      // larger regions: the dense range of those that become items (two loads, all lanes together); regions of up to
      // kMidRegion candidates stay with their lane (a deferred range -- long seeds -- is an item whatever its size: only
      // the verifier can narrow it)
      STG(4);
      const DenseRange dr_p = dense_range(svp, lp.reg.l, size_p, big_p && win_usable<NW>(svp, lr.len));
      const DenseRange dr_m = dense_range(svm, lm.reg.l, size_m, big_m && win_usable<NW>(svm, lr.len));
      {
        const uint32_t nmid_p = (!kCoop && size_p > kSmallRegion && !big_p) ? size_p : 0u;
        const uint32_t nmid_m = (!kCoop && size_m > kSmallRegion && !big_m) ? size_m : 0u;
        if constexpr ((MIDS || kMulti) && !kCoop)
        if (__ballot(nmid_p | nmid_m))
          se_mid_regions<NW, kMulti>(iv, sh, svp, svm, nmid_p, nmid_m, lp.reg.l, lm.reg.l, sd, lr.len, lr.rd, mk, n_chrom, top_step,
                                     tail_cut, mt, sum_p, sum_m, ctr.verified);
      }
      if constexpr (kMulti) {
        // the region's size is the number of candidates whose tail characters match (mapping.cpp:275-277); a range with a
        // candidate at a chromosome's edge is narrowed the reference's way and verified one candidate after the other (rare)
        if (mt.multi_p && !mt.fb_p && mt.nin_p > b) sum_p = summary_empty();
        if (mt.multi_m && !mt.fb_m && mt.nin_m > b) sum_m = summary_empty();
        if (__ballot(mt.fb_p || mt.fb_m)) {
#pragma unroll 1
          for (uint32_t fi = 0; fi < 2; ++fi) {
            const bool mine = fi ? mt.fb_m : mt.fb_p;
            if (!mine) continue;
            const StrandView& sv = fi ? svm : svp;
            const Region rg = lit_region(sv, care, kKeyWeight + kKeyChars, seed_len, fi ? lm.reg.l : lp.reg.l, fi ? lm.reg.u : lp.reg.u);
            RegionSummary acc = summary_empty();
            if (rg.l <= rg.u && rg.u - rg.l + 1 <= b) {
              for (uint32_t sl = rg.l; sl <= rg.u; ++sl) {
                uint32_t gp_c, mm_c;
                if (verify_candidate<NW>(sv, si, iv.start_index, n_chrom, sv.ent[sl].pos, sd, lr.len, lr.rd, mk, gp_c, mm_c))
                  acc = summary_merge(acc, summary_one(mm_c, gp_c));
              }
            }
            if (fi) sum_m = acc; else sum_p = acc;
          }
        }
      }
      STG(5);
      if (__ballot(big_p || big_m)) {  // both strands' large regions become work items: one atomic for the two (map_items.h item_append2)
        const bool big2[2] = {big_p, big_m};
        const bool dense2[2] = {big_p && dr_p.hi > dr_p.lo, big_m && dr_m.hi > dr_m.lo};
        const uint32_t id2[2] = {j, j | (1u << 31)}, l2[2] = {lp.reg.l, lm.reg.l}, size2[2] = {size_p, size_m};
        const uint32_t rec2[2] = {dense2[0] ? (uint32_t)dr_p.rec : kItemDenseNone, dense2[1] ? (uint32_t)dr_m.rec : kItemDenseNone};
        const bool tail2[2] = {defer_p, defer_m};
        item_append2<NW>(big2, dense2, id2, l2, size2, rec2, lr.len, sd, lr.rd, mk, q, tail2);
        if (big_p) { ++ctr.big; pend_p = true; }
        if (big_m) { ++ctr.big; pend_m = true; }
      }
      STG(6);
      // ---- the seed is done, or the read waits for the verifier
      bool blocked = false;
      if (active) {
        // this seed's '-' summary: the lane's own now, an item's when k_se_verify has run
        if (!pend_m) hs.sums[(uint64_t)(1 + seed_i) * hcap + j] = make_uint4(sum_m.min_mm, sum_m.count, sum_m.first, sum_m.last);
        if (pend_p || pend_m) {
          if (!pend_p) hs.sums[j] = make_uint4(sum_p.min_mm, sum_p.count, sum_p.first, sum_p.last);
          hs.st[j] = make_uint4(best.genome_pos, best.times, best.mismatch, minus_lb);
          blocked = true;
          active = false;
        } else {
          fold_region(best, sum_p, '+');  // empty when the '+' probe was not needed
          if (sum_m.count && sum_m.min_mm < minus_lb) minus_lb = sum_m.min_mm;
          ++seed_i;
          if (seed_i == kPat) { fin = true; active = false; }
        }
      }
      wavelist_append(wl, blocked, stage_entry(j, seed_i), &hs.ctl[4], hs.list_out);
      }
    }
    // ---- the lanes whose read is blocked, deferred or finished leave it (and take the next one at the loop's top)
    const bool leave = have && !active;
    fin = fin && leave;
    wave_append(leave && deferred, r | (defer_iter << kDeferShift), defer_count, defer_list);
    // ---- finished reads: the '-' strand's summaries folded in reference order under the exact exit conditions
    if (__ballot(fin)) {
      uint4 ng[kPat];
#pragma unroll
      for (uint32_t k = 0; k < kPat; ++k) {
        ng[k] = make_uint4(0xFFFFFFFFu, 0u, 0u, 0u);
        if (fin && k < seed_i) ng[k] = load_global(hs.sums + (uint64_t)(1 + k) * hcap + j);
      }
      bool go = true;
#pragma unroll
      for (uint32_t k = 0; k < kPat; ++k) {
        go = go && seed_needed(best.mismatch, k);
        RegionSummary a; a.min_mm = ng[k].x; a.count = ng[k].y; a.first = ng[k].z; a.last = ng[k].w;
        if (go) fold_region(best, a, '-');
      }
      if (fin) out[r] = best;
    }
    if (leave) { have = false; fin = false; deferred = false; }
    STG(7);
  }
  STG_END;
  if (hs.list_out != nullptr) wavelist_flush(wl, &hs.ctl[4], hs.list_out);
  flush_counters(ctr, 0u, stats);
}

// ---------------------------------------------------------------------------
// k_se_verify: the work items of one heavy stage (HeavyStage), one region per wavefront (map_items.h item_stream).
// The RegionSummary of a region is order-free once the slot order is kept in the lanes: minimum, number of
// candidates holding it, the first and the last of them in slot order (core.h summary_merge) -- accumulated per
// lane over its slots (ascending) and reduced once per item.
// ---------------------------------------------------------------------------
struct LaneBest {  // this lane's slots of one region, ascending
  uint32_t mm, cnt, f_k, f_gp, l_k, l_gp;
};
__device__ __forceinline__ void lane_best_add(LaneBest& a, uint32_t k, uint32_t gp, uint32_t mm) {
  const bool has = mm != 0xFFFFFFFFu;
  const bool better = has && mm < a.mm, same = has && mm == a.mm;
  a.cnt = better ? 1u : a.cnt + (same ? 1u : 0u);
  a.f_k = better ? k : a.f_k;
  a.f_gp = better ? gp : a.f_gp;
  a.l_k = (better || same) ? k : a.l_k;
  a.l_gp = (better || same) ? gp : a.l_gp;
  a.mm = better ? mm : a.mm;
}
__device__ __forceinline__ uint4 lane_best_reduce(const LaneBest& a) {
  const uint32_t mn = wave_min_u32(a.mm);
  if (mn == 0xFFFFFFFFu) return make_uint4(0xFFFFFFFFu, 0u, 0u, 0u);
  const bool eq = a.mm == mn;
  const uint32_t cnt = wave_sum_u32(eq ? a.cnt : 0u);
  const uint32_t fk = wave_min_u32(eq ? a.f_k : 0xFFFFFFFFu), lk = wave_max_u32(eq ? a.l_k : 0u);
  const unsigned long long fm = __ballot(eq && a.f_k == fk), lm = __ballot(eq && a.l_k == lk);
  return make_uint4(mn, cnt, bcast(a.f_gp, (int)__ffsll((long long)fm) - 1), bcast(a.l_gp, (int)__ffsll((long long)lm) - 1));
}
// item id = chunk position j | strand << 31; the summary goes to row 0 ('+': the seed the read is blocked at) or row
// 1 + seed ('-') of HeavyStage::sums
struct SummarySink {
  uint4* sums;
  uint32_t hcap;
  uint32_t b;        // -b: a region of more candidates is skipped (mapping.cpp:275-277); tested here for tail items
  uint32_t id, seed_i, tail;
  LaneBest acc;
  uint32_t in_region;  // this lane's candidates that belong to the region (tail items: whose care characters >= 44 match)
  uint32_t n_verified;
  static __device__ __forceinline__ uint32_t strand(uint32_t id) { return id >> 31; }
  __device__ __forceinline__ void begin(uint32_t id_, uint32_t seed_, uint32_t, uint32_t tail_) {
    id = id_; seed_i = seed_; tail = tail_;
    acc = {0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u};
    in_region = 0;
  }
  __device__ __forceinline__ void add(uint32_t k, uint32_t gp, uint32_t mm, bool in) {
    n_verified += mm != 0xFFFFFFFFu ? 1u : 0u;
    in_region += in ? 1u : 0u;
    lane_best_add(acc, k, gp, mm);
  }
  __device__ __forceinline__ void step() {}
  __device__ __forceinline__ void end() {
    uint4 res = lane_best_reduce(acc);
    if (tail && wave_sum_u32(in_region) > b) res = make_uint4(0xFFFFFFFFu, 0u, 0u, 0u);  // (uniform) the narrowed region exceeds -b
    if ((threadIdx.x & 63) == 0) sums[(uint64_t)((id >> 31) ? 1u + seed_i : 0u) * hcap + (id & 0x7FFFFFFFu)] = res;
  }
};

// the tail items of a stage bracketed before the verifier streams them (map_items.h tail_items_narrow)
template <int NW>
__global__ __launch_bounds__(kBlock) void k_se_tail_narrow(IndexView iv, uint32_t strand_base, HeavyStage hs, uint32_t b) {
  if constexpr (NW <= 10 && long_seed_nw<NW>()) {
    ItemQueue q;
    q.items = hs.items; q.ctl = hs.ctl; q.cap = 2 * hs.hcap; q.ovf = nullptr;
    q.bigs = hs.giants; q.big_n = hs.ctl + 5; q.big_cap = hs.hcap / 8;
    uint32_t n_big = *q.big_n;
    n_big = n_big < q.big_cap ? n_big : q.big_cap;
    uint32_t n_items = q.ctl[0];
    n_items = n_items < q.cap ? n_items : q.cap;
    if constexpr (kPat == 3) tail_items_narrow<NW, SummarySink>(iv, strand_base, q, n_items, n_big, b);
    else tail_items_narrow_wide<NW, SummarySink>(iv, strand_base, q, n_items, n_big, b);
  }
}

template <int NW, bool DENSE, int G = 1>
__global__ __launch_bounds__(kBlock, G > 1 ? (NW <= 8 ? 3 : 2) : DENSE ? (NW <= 8 ? 6 : 4) : (NW <= 8 ? 4 : (NW <= 10 ? 2 : 1))) void k_se_verify(
    IndexView iv, uint32_t strand_base, unsigned long long* __restrict__ stats, HeavyStage hs, uint32_t b, uint32_t* __restrict__ err) {
  static_assert(item_quads<NW>() <= 64, "an item header is fetched by one wavefront load");
  ItemQueue q;
  q.items = hs.items; q.ctl = hs.ctl; q.cap = 2 * hs.hcap; q.ovf = err + 2;
  q.bigs = hs.giants; q.big_n = hs.ctl + 5; q.big_cap = hs.hcap / 8;
  if (DENSE && blockIdx.x == 0 && threadIdx.x == 0) items_overflow_check(q);
  uint32_t n_big = DENSE ? *q.big_n : 0u;
  n_big = n_big < q.big_cap ? n_big : q.big_cap;
  const uint32_t n_items = q.ctl[DENSE ? 0 : 1] + n_big;
  if (n_items == 0) return;
  __shared__ uint32_t s_start[kLdsChroms + 1];
  __shared__ uint32_t s_edge[DENSE ? kEdgeWords : 1];  // (the dense verifier: core.h edge bitmap)
  const bool fits = iv.n_chrom <= kLdsChroms;  // every chromosome start is in LDS (else every 2^shift-th: ChromTab)
  chrom_tab_stage(s_start, iv.start_index, chrom_tab_of(iv.n_chrom));
  if (DENSE && iv.edge_bits != nullptr)
    for (uint32_t i = threadIdx.x; i < kEdgeWords; i += blockDim.x) s_edge[i] = iv.edge_bits[i];
  __syncthreads();
  const uint32_t* const edge = (DENSE && iv.edge_bits != nullptr) ? s_edge : nullptr;
  SummarySink sink;
  sink.sums = hs.sums; sink.hcap = hs.hcap; sink.n_verified = 0; sink.b = b; sink.tail = 0; sink.in_region = 0;
  if (fits) item_stream<NW, DENSE, true, SummarySink, G>(iv, strand_base, q, n_items, s_start, sink, n_big, edge);
  else item_stream<NW, DENSE, false, SummarySink, G>(iv, strand_base, q, n_items, s_start, sink, n_big, edge);
  flush_counters({0u, sink.n_verified, 0u}, 0u, stats);
}

#ifndef WALT_LIT_OCC
#define WALT_LIT_OCC 5  // wavefronts per SIMD the literal kernel's registers are capped for (measured: see DESIGN.md section 12b)
#endif
// pass 2: the deferred reads (grid-stride over the list; count is on the device)
template <int NW, bool LITERAL = true>
__global__ __launch_bounds__(kBlock, NW <= 8 ? WALT_LIT_OCC : 1) void k_map_se_literal(IndexView iv, const uint32_t* __restrict__ codes2,
                                                            const uint64_t* __restrict__ offsets,
                                                            uint32_t* __restrict__ err, uint32_t strand_base,
                                                            uint32_t max_mm,
                                                            uint32_t b, const uint32_t* __restrict__ mask_table,
                                                            BestMatch* __restrict__ out,
                                                            unsigned long long* __restrict__ stats,
                                                            uint32_t* __restrict__ defer_count,
                                                            uint32_t* __restrict__ defer_list,
                                                            uint32_t all_reads, const uint32_t* __restrict__ range = nullptr) {
  // range != nullptr: the list entries [range[0], range[1]) only (launch_map_se: what was deferred after the side launch)
  uint32_t n_entries = 0;
  if (range != nullptr) {
    const uint32_t lo_entry = range[0];
    n_entries = range[1] > lo_entry ? range[1] - lo_entry : 0u;
    if (n_entries == 0) return;
    defer_list += lo_entry;
  }
  __shared__ BlockShared sh;
  const uint32_t* si = block_prologue(sh, iv, mask_table, strand_base);
  // all_reads != 0 (seed patterns 5 and 7, which have no seed-major pass 1): every read 0 .. all_reads-1 with
  // LITERAL = false (directory/key search; a Bloom hit appends the read to defer_list), then the list with
  // LITERAL = true
  const uint32_t count = all_reads ? all_reads : (range != nullptr ? n_entries : *defer_count);
  MapCounters ctr = {0, 0, 0};
  uint32_t shortv = 0;
  for (uint32_t base = blockIdx.x * blockDim.x; base < count; base += gridDim.x * blockDim.x) {
    const uint32_t i = base + threadIdx.x;
    const bool valid = i < count;
    const uint32_t r = valid ? (all_reads ? i : (defer_list[i] & kDeferMask)) : 0;
    uint32_t len;
#if defined(WALT_DIAG)
    // diagnostic library: phase sums of this kernel (se_process's phases, map_se.hip kStampPhases) when bit 1 of the switch is set
    StampsT<true> st;
    stamp_begin(st, g_stage_stamps);
    se_process<NW, LITERAL, true>(iv, sh, si, codes2, offsets, err, r, valid, strand_base, max_mm, b, out,
                                  LITERAL ? nullptr : defer_count, LITERAL ? nullptr : defer_list, ctr, len, 0u, st);
    if (g_stage_stamps_on & 2u) stamp_end(st);
#else
    StampsT<false> st;
    se_process<NW, LITERAL, false>(iv, sh, si, codes2, offsets, err, r, valid, strand_base, max_mm, b, out,
                                   LITERAL ? nullptr : defer_count, LITERAL ? nullptr : defer_list, ctr, len, 0u, st);
#endif
    // too_short is counted once per strand pass (mapping.cpp:230-233); pass 1 counts it when there is one
    if (all_reads) shortv += (valid && len < kMinReadLen) ? 2u : 0u;
  }
  flush_counters(ctr, shortv, stats);
}

// ---------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------
static inline uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

uint64_t se_stride(uint32_t n) { return align_up(n ? n : 1, 64); }
// Staged heavy pass: reads per chunk of the heavy list and the bytes of its state behind the dense read array:
// [control words] then per state slot [kPat round lists][st][packed reads][kPat + 1 summaries][2 x hcap items][giants].
// The list is mapped in two halves on two streams (launch_map_se), each with its own state slot: chunks of an eighth
// of the batch (the heavy reads of an hg19-like genome, a sixth of all, make one chunk per half), cut evenly on the
// device (heavy_chunk_span).  Option se_pipe = 0: one stream, one slot, chunks of a quarter.
// The geometry a call uses follows from (n, read length, the index's options); the workspace a caller must provide
// (walt_se_workspace_bytes, which knows no index) is that of the DEFAULT options, and no option may need more: an
// option value that would is not applied (se_geometry).
constexpr uint32_t kLitChunks = 2;  // chunks of the deferred list the literal rounds take (launch_map_se)
constexpr uint32_t kHeavyCtlWords = 64 * kPat + 8 * kPat * kLitChunks + 32 + 8;  // 8 words per (chunk, round): up to 8 chunks x kPat rounds, the literal rounds' chunks, the side launch's second share, the rest launch's range
struct SeGeometry {
  uint32_t hcap;     // reads per chunk
  uint32_t chunks;   // launched chunks (<= 8; even when piped)
  bool piped;
  uint64_t slot_bytes, heavy_bytes;
};
static uint64_t se_heavy_slot_bytes(uint64_t hcap, int nw) {
  const uint64_t item_q = 2 + (2 * (uint64_t)nw + 3) / 4;  // item_quads<NW>()
  const uint64_t rd_q = ((uint64_t)nw + 1 + 3) / 4;         // rd_quads<NW>()
  return kPat * hcap * 4 + hcap * (1 + rd_q + kPat + 1) * 16 + (2 * hcap + hcap / 8) * item_q * 16;
}
static SeGeometry se_geometry_for(uint32_t n, int nw, bool pipe, uint64_t chunk_hook) {
  SeGeometry g;
  const uint64_t share = ((uint64_t)n + (pipe ? 7 : 3)) / (pipe ? 8 : 4);
  g.hcap = (uint32_t)align_up(n <= 65536 ? (n ? n : 1) : (share > 65536 ? share : 65536), 64);
  if (chunk_hook > 0 && chunk_hook * 8 >= n) g.hcap = (uint32_t)align_up(chunk_hook, 64);  // at most 8 chunks are launched
  const uint32_t k = (uint32_t)(((uint64_t)n + g.hcap - 1) / g.hcap);
  g.piped = pipe && k > 1;
  g.chunks = g.piped ? (k + 1u) & ~1u : k;
  g.slot_bytes = se_heavy_slot_bytes(g.hcap, nw);
  g.heavy_bytes = 16 + kHeavyCtlWords * 4 + (g.piped ? 2 : 1) * g.slot_bytes;
  return g;
}
// bytes of the staged state a caller's workspace holds: the larger of the two default schedules
static uint64_t se_heavy_bytes(uint32_t n, int nw) {
  const uint64_t a = se_geometry_for(n, nw, true, 0).heavy_bytes, c = se_geometry_for(n, nw, false, 0).heavy_bytes;
  return a > c ? a : c;
}
static SeGeometry se_geometry(uint32_t n, int nw, const walt_options& opt) {
  SeGeometry g = se_geometry_for(n, nw, opt.se_pipe != 0, opt.se_heavy_chunk > 0 ? (uint64_t)opt.se_heavy_chunk : 0);
  if (g.heavy_bytes > se_heavy_bytes(n, nw) || g.hcap > (1u << kStageJBits)) g = se_geometry_for(n, nw, opt.se_pipe != 0, 0);
  return g;
}
// what pass 1 hands to the staged rounds (SeCarry): kPat 16-byte words per read of the batch
static uint64_t se_carry_bytes(uint32_t n) { return (uint64_t)kPat * se_stride(n) * 16; }

constexpr unsigned kLiteralGrid = 2048;  // blocks of the deferred-read pass (grid-stride)

#if defined(WALT_DIAG)
// Diagnostic build only (make diag: libwalt_amd_diag.so).  WALT_AMD_ABLATE (results are WRONG when bits other than 8
// are set): bit 0 skips verification, bit 1 stops after the directory lookup, bit 2 skips the lookup, bit 3 checks the
// danger filter against the exact test (results stay valid).  WALT_AMD_STAMPS=1: in-kernel phase times.
// WALT_AMD_SYNC_DEBUG=1: wait after every launch of the single-end path and say which one returned.
static uint32_t g_ablate = 0;
static unsigned long long* g_stamps = nullptr;
static void debug_sync(const char* what, hipStream_t stream) {
  static const bool on = getenv("WALT_AMD_SYNC_DEBUG") != nullptr;
  if (!on) return;
  const hipError_t e = hipStreamSynchronize(stream);
  fprintf(stderr, "[walt_amd sync] %s: %s\n", what, hipGetErrorString(e));
}
#else
static void debug_sync(const char*, hipStream_t) {}
#endif

unsigned persistent_grid(const walt_index* idx) {
  return idx->opt.grid > 0 ? (unsigned)idx->opt.grid : (unsigned)idx->n_cu * 8u;
}

template <int NW>
static int launch_map_se(walt_index* idx, const IndexView& view, const uint32_t* codes2, const uint64_t* offsets, uint32_t* err,
                         uint32_t n, uint32_t strand_base, uint32_t max_mm, uint32_t b, BestMatch* out,
                         unsigned long long* stats, uint32_t* defer_count, uint32_t* defer_list, uint64_t stride,
                         uint32_t* heavy_area, uint4* carry_area, hipStream_t stream) {
  const walt_options& opt = idx->opt;
  const unsigned pg = persistent_grid(idx);
  const unsigned g1 = grid_for(n) < pg ? grid_for(n) : pg;
  uint32_t* heavy_count = defer_count + 24;   // control block word (zeroed with it)
  uint32_t* heavy_list = defer_list + 2 * stride;
  const bool mono = opt.se_heavy_mono != 0;
  SeCarry carry;
  carry.st = (opt.se_carry && !mono) ? carry_area : nullptr;
  carry.neg = carry_area + stride;
  carry.stride = stride;
  // walt_profile_detail: an event after every kernel group (profiling on: bench.py)
  idx->n_detail = 0;
  auto mark = [&](unsigned char kind) {
    if (!idx->profile || idx->n_detail >= walt_index::kDetailEvents) return;
    hipEvent_t& e = idx->ev_detail[idx->n_detail];
    if (!e && hipEventCreate(&e) != hipSuccess) return;
    if (hipEventRecord(e, stream) != hipSuccess) return;
    idx->ev_kind[idx->n_detail++] = kind;
  };
  mark(255);  // start
#if defined(WALT_DIAG)
  const bool diag = g_ablate != 0 || g_stamps != nullptr;  // diagnostic instantiation (stamps / ablation)
  // WALT_AMD_STAMPS=2: phase stamps of the (one-kernel) heavy pass only, 3: of pass 1 only (1: both, summed)
  const char* sm = getenv("WALT_AMD_STAMPS");
  const int stamp_mode = sm ? atoi(sm) : 0;
  const bool diag1 = diag && stamp_mode != 2, diag2 = diag && stamp_mode != 3;
  if (diag1)
    hipLaunchKernelGGL((k_map_se<NW, true, false>), dim3(g1), dim3(kBlock), 0, stream, view, codes2, offsets, err, n,
                       strand_base, max_mm, b, idx->d_mask_table, out, stats, defer_count, defer_list, heavy_count,
                       heavy_list, g_ablate, g_stamps, carry);
  else
#else
  const bool diag = false;
#endif
    hipLaunchKernelGGL((k_map_se<NW, false, false>), dim3(g1), dim3(kBlock), 0, stream, view, codes2, offsets, err, n,
                       strand_base, max_mm, b, idx->d_mask_table, out, stats, defer_count, defer_list, heavy_count,
                       heavy_list, 0u, nullptr, carry);
  debug_sync("pass 1", stream);
  mark(0);
  bool lit_done = false;  // the literal rounds have mapped the deferred reads (all but what the rest launch below takes)
  bool lit_side = false;  // the strand-major literal kernel runs on a side stream beside the end of the heavy pass
  bool forked = false;  // work is queued on the index's own streams: an error return must not leave it running
  auto unwind = [&]() {
    if (!forked) return;
    if (idx->se_pipe) (void)hipStreamSynchronize(idx->se_pipe);
    if (idx->se_side) (void)hipStreamSynchronize(idx->se_side);
  };
#define WALT_HIP_FORKED(expr)                                                                \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      unwind();                                                                              \
      return walt::fail(WALT_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));       \
    }                                                                                        \
  } while (0)
  uint32_t* const rng = heavy_area + kHeavyCtlWords - 8;  // {first entry, end} of the deferred list the rest launch maps
  if (mono) {
    // the one-kernel heavy pass (large regions verified by the whole wavefront of their read's lane) instead of the
    // staged one (large regions streamed by k_se_verify); same results, kept for comparison
#if defined(WALT_DIAG)
    if (diag2)
      hipLaunchKernelGGL((k_map_se<NW, true, true>), dim3(g1), dim3(kBlock), 0, stream, view, codes2, offsets, err, n,
                         strand_base, max_mm, b, idx->d_mask_table, out, stats, defer_count, defer_list, heavy_count,
                         heavy_list, g_ablate, g_stamps, SeCarry());
    else
#endif
      hipLaunchKernelGGL((k_map_se<NW, false, true>), dim3(g1), dim3(kBlock), 0, stream, view, codes2, offsets, err, n,
                         strand_base, max_mm, b, idx->d_mask_table, out, stats, defer_count, defer_list, heavy_count,
                         heavy_list, 0u, nullptr, SeCarry());
  } else {
    const SeGeometry geo = se_geometry(n, NW, opt);
    const uint32_t hcap = geo.hcap;
    // two halves on two streams: odd chunks on idx->se_pipe with the second state slot.  Every launch is a set of
    // persistent blocks that fill the device by their registers, so two launches do not share a SIMD for long; what
    // the second stream buys is that one half's launch starts into the other's tail.  Diagnostic runs keep one stream.
    const bool piped = geo.piped && !diag;
    const uint32_t chunks = geo.chunks;  // <= 8
    HeavyStage hs;
    hs.ctl = heavy_area;
    hs.hcap = hcap;
    hs.even = piped ? 1u : 0u;
    const uint64_t slot_words = geo.slot_bytes / 4;
    auto use_slot = [&](uint32_t k, uint32_t*& lists) {
      lists = heavy_area + kHeavyCtlWords + k * slot_words;  // [kPat][hcap]: what round k hands to round k + 1
      hs.st = reinterpret_cast<uint4*>(lists + (uint64_t)kPat * hcap);
      hs.rdq = hs.st + hcap;
      hs.sums = hs.rdq + (uint64_t)rd_quads<NW>() * hcap;
      hs.items = hs.sums + (uint64_t)(kPat + 1) * hcap;  // 2 * hcap items of item_quads<NW>() quads
      hs.giants = hs.items + (uint64_t)2 * hcap * item_quads<NW>();
    };
    // long seeds: key-equal ranges of more slots than this go to the verifier unnarrowed (option se_defer_min: 0 = never
    // (A/B: every range narrowed by lit_region in the stage kernel); measured 4 / 8 / 16 -> 32.5 / 33.2 / 34.2 ms per 25 M 150-base reads)
    hs.defer_min = opt.se_defer_min < 0 ? (uint32_t)kSmallRegion
                 : opt.se_defer_min == 0 ? 0xFFFFFFFFu
                 : (uint32_t)(opt.se_defer_min < (long long)kSmallRegion ? (long long)kSmallRegion : opt.se_defer_min);
    hs.lit_ablate = (uint32_t)opt.se_lit_ablate;
    uint32_t* const ctl0 = heavy_area;
    static const int vb_dense = [] {
      int nb = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_se_verify<NW, NW <= 10>, kBlock, 0) != hipSuccess || nb < 1) nb = 4;
      return nb;
    }();
    static const int vb_gather = [] {
      int nb = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_se_verify<NW, false>, kBlock, 0) != hipSuccess || nb < 1) nb = 2;
      return nb;
    }();
    const unsigned vg_dense = (unsigned)vb_dense * (unsigned)idx->n_cu, vg_gather = (unsigned)vb_gather * (unsigned)idx->n_cu;
    // The reads with a truly dangerous probe.  Default: the strand-major kernel (k_map_se_literal, the reference's search
    // inferred: core.h lit_region_inferred) on a low-priority side stream: after the last look-up round of the first chunk
    // (of each half) the list is complete but for what later chunks add (nothing, when the heavy list fits these
    // chunks); it is sorted and mapped while the main streams run the last verifier launches and the last round; what
    // later chunks defer is mapped at the end.  Option se_lit_staged = 1: through staged rounds instead (below).
    const bool lit_staged = opt.se_lit_staged != 0 && !diag;
    lit_side = !lit_staged && opt.se_lit_side != 0 && !diag && n <= kDeferMask;
    uint32_t* const ctl2 = ctl0 + 64 * kPat;  // [0] count of the side launch, [8..23] its bins (the literal rounds' words: one or the other)
    uint32_t* const rng_side = ctl2 + 32;     // {entries the side launch took, the list's final length}
    static_assert(8 * kPat * kLitChunks >= 34, "the side launch's control words fit the literal rounds' area");
    if (lit_side && !idx->se_side) {
      int lo_pri = 0, hi_pri = 0;
      WALT_HIP(hipDeviceGetStreamPriorityRange(&lo_pri, &hi_pri));
      WALT_HIP(hipStreamCreateWithPriority(&idx->se_side, hipStreamNonBlocking, lo_pri));
      WALT_HIP(hipEventCreateWithFlags(&idx->se_fork, hipEventDisableTiming));
      WALT_HIP(hipEventCreateWithFlags(&idx->se_join, hipEventDisableTiming));
    }
    if (piped && !idx->se_pipe) {
      WALT_HIP(hipStreamCreateWithFlags(&idx->se_pipe, hipStreamNonBlocking));
      for (hipEvent_t& e : idx->se_pipe_ev) WALT_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    // se_lit_side = 2 (default): the side launch starts HERE, on what pass 1 deferred (most of the list: pass 1 sees every
    // read, the staged rounds the heavy sixth), and runs beside the whole heavy pass -- a chain of dependent look-ups in
    // few wavefronts that fills a fraction of the device; what the staged rounds defer is mapped at the end.  The
    // snapshot is taken before any staged kernel of either half can append to the list.
    const bool lit_early = lit_side && opt.se_lit_side >= 2;
    if (lit_early) {
      hipLaunchKernelGGL(k_lit_snapshot, dim3(1), dim3(64), 0, stream, defer_count, ctl2, rng_side);
      WALT_HIP(hipEventRecord(idx->se_fork, stream));
      WALT_HIP(hipStreamWaitEvent(idx->se_side, idx->se_fork, 0));
      forked = true;
      launch_bin_deferred(ctl2, defer_list, defer_list + stride, idx->se_side);
      const unsigned g_lit = grid_for(n) < kLiteralGrid ? grid_for(n) : kLiteralGrid;
      hipLaunchKernelGGL(k_map_se_literal<NW>, dim3(g_lit), dim3(kBlock), 0, idx->se_side, view, codes2, offsets, err,
                         strand_base, max_mm, b, idx->d_mask_table, out, stats, ctl2, defer_list + stride, 0u);
      WALT_HIP_FORKED(hipEventRecord(idx->se_join, idx->se_side));
    }
    if (piped) {  // (pass 1 done)
      WALT_HIP(hipEventRecord(idx->se_pipe_ev[0], stream));
      WALT_HIP(hipStreamWaitEvent(idx->se_pipe, idx->se_pipe_ev[0], 0));
    }
    // the rounds of one chunk: look-up stage, then the verifiers of its items, kPat times, and the final fold
    auto run_chunk = [&](hipStream_t cs, uint32_t* ctl_base, uint32_t* lists, bool lit, const uint32_t* count, const uint32_t* list,
                         const SeCarry& cy, bool marks, uint32_t c_side) -> int {
      // (options se_stage_blocks / se_verify_blocks: blocks per compute unit of these launches -- a launch that leaves
      // wavefront slots free lets the other half's launch of the other kind run beside it; A/B)
      const unsigned pgs = opt.se_stage_blocks > 0 ? (unsigned)opt.se_stage_blocks * (unsigned)idx->n_cu : pg;
      const unsigned gh = grid_for(hcap) < pgs ? grid_for(hcap) : pgs;
      const unsigned vgd = opt.se_verify_blocks > 0 ? (unsigned)opt.se_verify_blocks * (unsigned)idx->n_cu : vg_dense;
      for (uint32_t round = 0; round <= kPat; ++round) {
        hs.round = round;
        hs.ctl = ctl_base + 8 * (round < kPat ? round : 0);
        hs.list_in = round ? lists + (uint64_t)(round - 1) * hcap : nullptr;
        hs.count_in = round ? ctl_base + 8 * (round - 1) : nullptr;
        hs.list_out = round < kPat ? lists + (uint64_t)round * hcap : nullptr;
        if (lit)
          hipLaunchKernelGGL((k_se_stage<NW, 0, true>), dim3(gh), dim3(kBlock), 0, cs, view, codes2, offsets, err,
                             strand_base, max_mm, b, idx->d_mask_table, out, stats, defer_count, defer_list, count, list, hs, cy);

        // One wavefront per SIMD fewer and no spilled registers (reads of up to 128 bases: 3 instead of 4; measured after
        // the read hand-out and the candidate list went in: 38.6 against 39.6 ms per 50 M reads) -- unless the genome has
        // more sequences than the LDS table of chromosome starts holds (map_common.h ChromTab): then every look-up
        // bisects in HBM and the extra wavefront is worth more than the registers (3,000 contigs: 106.7 against 109.8 ms).
        // Option se_stage_occ chooses explicitly (A/B).
        else if (NW <= 10 && (opt.se_stage_occ ? opt.se_stage_occ == (NW <= 8 ? 3 : 2) : (NW <= 8 && view.n_chrom <= 1023u)))
          hipLaunchKernelGGL((k_se_stage<NW, (NW <= 8 ? 3 : 2)>), dim3(gh), dim3(kBlock), 0, cs, view, codes2, offsets, err,
                             strand_base, max_mm, b, idx->d_mask_table, out, stats, defer_count, defer_list, count, list, hs, cy);
        else
          hipLaunchKernelGGL((k_se_stage<NW>), dim3(gh), dim3(kBlock), 0, cs, view, codes2, offsets, err,
                             strand_base, max_mm, b, idx->d_mask_table, out, stats, defer_count, defer_list, count, list, hs, cy);
        if (marks) mark(lit ? 3 : 1);
        if (round == kPat) break;
        if (!lit && round == 0 && piped && c_side == 0 && opt.se_stagger != 0) {  // the other half starts one stage behind (A/B)
          WALT_HIP_FORKED(hipEventRecord(idx->se_pipe_ev[0], stream));
          WALT_HIP_FORKED(hipStreamWaitEvent(idx->se_pipe, idx->se_pipe_ev[0], 0));
        }
        if (!lit && round == kPat - 1) {  // (round kPat probes nothing: this chunk has made its last deferrals)
          if (piped && c_side == 0) WALT_HIP_FORKED(hipEventRecord(idx->se_pipe_ev[1], stream));
          if (lit_early && c_side == (piped ? 1u : 0u)) {  // the second share: what the staged rounds deferred
            if (piped) WALT_HIP_FORKED(hipStreamWaitEvent(cs, idx->se_pipe_ev[1], 0));
            uint32_t* const ctl3 = ctl0 + 64 * kPat + 8 * kPat * kLitChunks;
            hipLaunchKernelGGL(k_lit_snapshot_rest, dim3(1), dim3(64), 0, cs, defer_count, ctl3, rng_side);
            WALT_HIP_FORKED(hipEventRecord(idx->se_fork, cs));
            WALT_HIP_FORKED(hipStreamWaitEvent(idx->se_side, idx->se_fork, 0));
            launch_bin_deferred(ctl3, defer_list, defer_list + stride, idx->se_side, ctl3 + 1);
            const unsigned g_lit = grid_for(n) < kLiteralGrid ? grid_for(n) : kLiteralGrid;
            hipLaunchKernelGGL(k_map_se_literal<NW>, dim3(g_lit), dim3(kBlock), 0, idx->se_side, view, codes2, offsets, err,
                               strand_base, max_mm, b, idx->d_mask_table, out, stats, defer_count, defer_list + stride, 0u, ctl3 + 24);
            WALT_HIP_FORKED(hipEventRecord(idx->se_join, idx->se_side));
          }
          if (lit_side && !lit_early && c_side == (piped ? 1u : 0u)) {
            if (piped) WALT_HIP_FORKED(hipStreamWaitEvent(cs, idx->se_pipe_ev[1], 0));
            hipLaunchKernelGGL(k_lit_snapshot, dim3(1), dim3(64), 0, cs, defer_count, ctl2, rng_side);
            WALT_HIP_FORKED(hipEventRecord(idx->se_fork, cs));
            WALT_HIP_FORKED(hipStreamWaitEvent(idx->se_side, idx->se_fork, 0));
            forked = true;
            launch_bin_deferred(ctl2, defer_list, defer_list + stride, idx->se_side);
            const unsigned g_lit = grid_for(n) < kLiteralGrid ? grid_for(n) : kLiteralGrid;
            hipLaunchKernelGGL(k_map_se_literal<NW>, dim3(g_lit), dim3(kBlock), 0, idx->se_side, view, codes2, offsets, err,
                               strand_base, max_mm, b, idx->d_mask_table, out, stats, ctl2, defer_list + stride, 0u);
            WALT_HIP_FORKED(hipEventRecord(idx->se_join, idx->se_side));
          }
        }
        if constexpr (NW <= 10 && long_seed_nw<NW>())
          hipLaunchKernelGGL((k_se_tail_narrow<NW>), dim3(pg), dim3(kBlock), 0, cs, view, strand_base, hs, b);
        if constexpr (NW <= 10) {
          hipLaunchKernelGGL((k_se_verify<NW, true>), dim3(vgd), dim3(kBlock), 0, cs, view, strand_base, stats, hs, b, err);
        }
        hipLaunchKernelGGL((k_se_verify<NW, false>), dim3(vg_gather), dim3(kBlock), 0, cs, view, strand_base, stats, hs, b, err);
        if (marks) mark(lit ? 3 : 2);
      }
      return WALT_OK;
    };
    for (uint32_t c = 0; c < chunks; ++c) {
      const bool odd = piped && (c & 1u);
      forked = forked || odd;
      uint32_t* lists = nullptr;
      use_slot(odd ? 1u : 0u, lists);
      hs.first = c * hcap;
      hs.chunk = c;
      // chunks behind the first pair append to the deferred list: not while the side launch's share is being fixed
      if (piped && lit_side && !lit_early && c == 2) WALT_HIP_FORKED(hipStreamWaitEvent(stream, idx->se_fork, 0));
      const int rc_chunk = run_chunk(odd ? idx->se_pipe : stream, ctl0 + 8 * kPat * c, lists, false, heavy_count, heavy_list, carry, !odd, c);
      if (rc_chunk) return rc_chunk;
    }
    if (piped) {  // the second half joins
      WALT_HIP_FORKED(hipEventRecord(idx->se_pipe_ev[2], idx->se_pipe));
      WALT_HIP_FORKED(hipStreamWaitEvent(stream, idx->se_pipe_ev[2], 0));
    }
    // ---- the literal rounds: the reads the rounds above gave up at a truly dangerous probe (0.9 % of the reads of a
    // 93-sequence genome, a fifth of those of a 3,000-contig assembly) go through the same rounds once more with the
    // reference's search switched on (k_se_stage<.., LIT>), kLitChunks chunks of the deferred list; what lies behind
    // them -- nothing, unless more than 2 x hcap reads were deferred -- is left to the strand-major kernel below.
    if (lit_staged) {
      hs.even = 0;
      for (uint32_t c = 0; c < kLitChunks && c < chunks; ++c) {
        uint32_t* lists = nullptr;
        use_slot(geo.piped ? (c & 1u) : 0u, lists);
        hs.first = c * hcap;
        hs.chunk = c;
        const int rc_chunk = run_chunk(stream, ctl0 + 64 * kPat + 8 * kPat * c, lists, true, defer_count, defer_list, SeCarry(), true, 0xFFFFFFFFu);
        if (rc_chunk) return rc_chunk;
      }
      hipLaunchKernelGGL(k_lit_rest, dim3(1), dim3(1), 0, stream, defer_count, rng,
                         (kLitChunks < chunks ? kLitChunks : chunks) * hcap);
      lit_done = true;
    }
  }
  debug_sync("heavy pass", stream);
  mark(1);  // (the one-kernel heavy pass, when that is what ran)
  if (lit_side) {  // what the chunks behind the first one deferred (usually nothing), then the side launch ends the call
    uint32_t* const rng_side = heavy_area + 64 * kPat + 32;
    hipLaunchKernelGGL(k_lit_end, dim3(1), dim3(1), 0, stream, defer_count, rng_side);
    unsigned g3 = grid_for(n) < kLiteralGrid ? grid_for(n) : kLiteralGrid;
    hipLaunchKernelGGL(k_map_se_literal<NW>, dim3(g3), dim3(kBlock), 0, stream, view, codes2, offsets, err,
                       strand_base, max_mm, b, idx->d_mask_table, out, stats, defer_count, defer_list, 0u, rng_side);
    WALT_HIP_FORKED(hipStreamWaitEvent(stream, idx->se_join, 0));
    mark(3);
    return WALT_OK;
  }
  if (lit_done) {  // deferred reads behind the literal rounds' chunks (usually none)
    unsigned g3 = grid_for(n) < kLiteralGrid ? grid_for(n) : kLiteralGrid;
    hipLaunchKernelGGL(k_map_se_literal<NW>, dim3(g3), dim3(kBlock), 0, stream, view, codes2, offsets, err,
                       strand_base, max_mm, b, idx->d_mask_table, out, stats, defer_count, defer_list, 0u, rng);
    mark(3);
    return WALT_OK;
  }
#undef WALT_HIP_FORKED
  uint32_t* defer_sorted = defer_list + stride;
  if (n <= kDeferMask) launch_bin_deferred(defer_count, defer_list, defer_sorted, stream);
  else defer_sorted = defer_list;
  debug_sync("bin", stream);
  unsigned g2 = grid_for(n) < kLiteralGrid ? grid_for(n) : kLiteralGrid;
  hipLaunchKernelGGL(k_map_se_literal<NW>, dim3(g2), dim3(kBlock), 0, stream, view, codes2, offsets, err,
                     strand_base, max_mm, b, idx->d_mask_table, out, stats, defer_count, defer_sorted, 0u);
  debug_sync("literal pass", stream);
  mark(3);
  return WALT_OK;
}

int map_se_device(walt_index* idx, const void* d_bases, const void* d_offsets, uint32_t n, uint32_t max_read_len,
                  int ag, uint32_t max_mm, uint32_t b, void* d_out, void* d_stats, void* d_workspace,
                  size_t workspace_bytes, hipStream_t stream) {
  if (!idx) return fail(WALT_EINVAL, "null index");
  const unsigned need = ag ? WALT_STRANDS_GA : WALT_STRANDS_CT;
  if ((idx->strand_mask & need) != need)
    return fail(WALT_EINVAL, ag ? "index opened without the _GA10/_GA11 strands" : "index opened without the _CT00/_CT01 strands");
  if (n == 0) return WALT_OK;
  if (n > kDeferMask + 1) return fail(WALT_EINVAL, "more than 2^28 reads in one batch (the reference's -N limit is 10^8, walt.cpp:236-239)");
  // One single-end call at a time per index: the call's side streams and events belong to the index (include/walt_amd.h)
  std::unique_lock<std::mutex> busy(idx->se_busy, std::try_to_lock);
  if (!busy.owns_lock()) return fail(WALT_EINVAL, "walt_map_se_batch: another single-end call is running on this index (an index is not re-entrant)");
#if defined(WALT_DIAG)
  {
    const char* ab = getenv("WALT_AMD_ABLATE");
    g_ablate = ab ? (uint32_t)atoi(ab) : 0u;
    if (g_ablate & ~8u) fprintf(stderr, "[walt_amd] WALT_AMD_ABLATE=%u: DIAGNOSTIC RUN, mapping results are not valid\n", g_ablate);
    if ((getenv("WALT_AMD_STAMPS") || g_ablate) && !g_stamps) {  // the diagnostic kernels always write their stamps
      WALT_HIP(hipMalloc(reinterpret_cast<void**>(&g_stamps), 16 * sizeof(unsigned long long)));
      WALT_HIP(hipMemset(g_stamps, 0, 16 * sizeof(unsigned long long)));
    }
  }
#endif
  const int nw = nw_for_len(max_read_len);
  if (!nw) return fail(WALT_EINVAL, "read length above 1024 is not supported (reference line limit is 1000, util.hpp:43)");
  if (max_read_len > kMaxReadLen)
    return fail(WALT_EINVAL, "reads longer than " + std::to_string(kMaxReadLen) + " bases are outside the tables of seed pattern " +
                                 std::to_string(kPat) + " (seedpattern.hpp)");
  if (workspace_bytes < walt_se_workspace_bytes(n, max_read_len))
    return fail(WALT_EINVAL, "walt_map_se_batch_device: the workspace holds " + std::to_string(workspace_bytes) + " bytes, the call needs " +
                                 std::to_string(walt_se_workspace_bytes(n, max_read_len)) + " (walt_se_workspace_bytes)");
  WALT_HIP(hipSetDevice(idx->device));
  const uint64_t stride = se_stride(n);
  // workspace: [64 words: read errors, deferral control] [statistic shards] [deferred list] [sorted deferred
  // list] [heavy list] [dense 2-bit reads] [state of the staged heavy pass] [what pass 1 hands over]
  uint32_t* err = reinterpret_cast<uint32_t*>(d_workspace);
  unsigned long long* shards = reinterpret_cast<unsigned long long*>(err + 64);
  uint32_t* defer_count = err + 32;  // control block: [0] count, [8..15] bin counts, [16..23] bin cursors
  uint32_t* defer_list = err + 64 + kStatShardBytes / 4;
  uint32_t* codes2 = defer_list + 3 * stride;  // deferred list, its sorted copy, heavy list
  // state of the staged heavy pass behind the dense reads, 16-byte aligned
  uint32_t* heavy_area = reinterpret_cast<uint32_t*>(
      align_up(reinterpret_cast<uint64_t>(codes2 + codes2_words((uint64_t)n * max_read_len)), 16));
  uint4* carry_area = reinterpret_cast<uint4*>(align_up(reinterpret_cast<uint64_t>(heavy_area) + se_heavy_bytes(n, nw), 16));
  WALT_HIP(hipMemsetAsync(err, 0, 64 * sizeof(uint32_t) + kStatShardBytes, stream));
  WALT_HIP(hipMemsetAsync(heavy_area, 0, kHeavyCtlWords * sizeof(uint32_t), stream));
  const uint8_t* bases = reinterpret_cast<const uint8_t*>(d_bases);
  const uint64_t* offsets = reinterpret_cast<const uint64_t*>(d_offsets);
  if (idx->profile) WALT_HIP(hipEventRecord(idx->ev[0], stream));
  IndexView view = idx->view;  // this launch's copy: the limits lane_load_read enforces
  view.batch_max_len = max_read_len;
  view.batch_cap_bytes = (uint64_t)n * max_read_len;
  launch_ascii_to_2bit(bases, offsets, n, codes2, view.batch_cap_bytes, err, stream);
  if (idx->profile) WALT_HIP(hipEventRecord(idx->ev[1], stream));
  BestMatch* out = reinterpret_cast<BestMatch*>(d_out);
  const uint32_t sb = ag ? 2u : 0u;
  int rc;
#define WALT_SE_CASE(NWV) rc = launch_map_se<NWV>(idx, view, codes2, offsets, err, n, sb, max_mm, b, out, shards, defer_count, defer_list, stride, heavy_area, carry_area, stream)
#if defined(WALT_ONLY_NW)  // (development: one kernel instance, for quick resource checks -- tools/kernel_resources.sh)
  WALT_SE_CASE(WALT_ONLY_NW);
  (void)nw;
#else
  switch (nw) {
    case 7: WALT_SE_CASE(7); break;
    case 8: WALT_SE_CASE(8); break;
#if WALT_SEEDPATTERN == 3  // patterns 5 / 7 stop at kMaxReadLen = 148 / 152 bases
    case 10: WALT_SE_CASE(10); break;
    case 16: WALT_SE_CASE(16); break;
    case 32: WALT_SE_CASE(32); break;
    default: WALT_SE_CASE(64); break;
#else
    default: WALT_SE_CASE(10); break;
#endif
  }
#endif
#undef WALT_SE_CASE
  if (rc) return rc;
  launch_reduce_stats(shards, reinterpret_cast<unsigned long long*>(d_stats), stream);
  if (idx->profile) {
    WALT_HIP(hipEventRecord(idx->ev[2], stream));
    idx->ev_valid = true;
  }
  WALT_HIP(hipGetLastError());
  return WALT_OK;
}

// err[0] / err[1] (non-ACGT reads / over-long reads) live at the start of the workspace
int check_read_errors(const void* d_workspace, hipStream_t stream) {
  uint32_t herr[3] = {0, 0, 0};
  WALT_HIP(hipMemcpyAsync(herr, d_workspace, sizeof(herr), hipMemcpyDeviceToHost, stream));
  WALT_HIP(hipStreamSynchronize(stream));
  if (herr[2]) return fail(WALT_EHIP, "internal: a work-item queue of the heavy pass overflowed (" + std::to_string(herr[2]) + " items dropped); results are incomplete");
  if (herr[1]) return fail(WALT_EINVAL, "reads longer than max_read_len, or more bases than n x max_read_len (" + std::to_string(herr[1]) + " refusals)");
  if (herr[0]) return fail(WALT_EBASE, "the batch holds non-ACGT nucleotides (in " + std::to_string(herr[0]) + " of its 16-base words)");
  return WALT_OK;
}

}  // namespace walt

using namespace walt;

extern "C" {

int walt_batch_check(const void* d_workspace, void* stream) {
  if (!d_workspace) return fail(WALT_EINVAL, "walt_batch_check: bad argument");
  return check_read_errors(d_workspace, reinterpret_cast<hipStream_t>(stream));
}

int walt_profile_enable(walt_index* idx, int on) {
  if (!idx) return fail(WALT_EINVAL, "null index");
  WALT_HIP(hipSetDevice(idx->device));
  if (on && !idx->ev[0])
    for (int i = 0; i < 3; ++i) WALT_HIP(hipEventCreate(&idx->ev[i]));
  idx->profile = on != 0;
  idx->ev_valid = false;
  return WALT_OK;
}

// diagnostic: read and clear the WALT_AMD_STAMPS phase sums (cycles summed over waves)
int walt_profile_stamps(unsigned long long* out16) {
#if defined(WALT_DIAG)
  if (!g_stamps) return fail(WALT_EINVAL, "WALT_AMD_STAMPS is not enabled");
  WALT_HIP(hipDeviceSynchronize());
  WALT_HIP(hipMemcpy(out16, g_stamps, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  WALT_HIP(hipMemset(g_stamps, 0, 16 * sizeof(unsigned long long)));
  return WALT_OK;
#else
  (void)out16;
  return fail(WALT_EINVAL, "phase stamps exist in the diagnostic build only (make diag: libwalt_amd_diag.so)");
#endif
}

#if defined(WALT_DIAG)
// diagnostic library only: switch the phase stamps of k_se_stage on / off; read and clear their sums
extern "C" int walt_profile_stage_stamps(int on, unsigned long long* out16) {
  WALT_HIP(hipDeviceSynchronize());
  unsigned long long zero[16] = {0};
  if (out16) {
    unsigned long long ctr[16];
    WALT_HIP(hipMemcpyFromSymbol(out16, HIP_SYMBOL(walt::g_stage_stamps), sizeof(zero)));
    WALT_HIP(hipMemcpyFromSymbol(ctr, HIP_SYMBOL(walt::g_diag_ctr), sizeof(ctr)));
    for (int i = 0; i < 7; ++i) out16[9 + i] = ctr[i];
  }
  WALT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(walt::g_stage_stamps), zero, sizeof(zero)));
  WALT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(walt::g_diag_ctr), zero, sizeof(zero)));
  const uint32_t v = (uint32_t)on & 3u, tw = (uint32_t)on >> 8;  // bit 0: k_se_stage's stamps, bit 1: k_map_se_literal's
  WALT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(walt::g_stage_stamps_on), &v, sizeof(v)));
  WALT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(walt::g_diag_twice), &tw, sizeof(tw)));
  return WALT_OK;
}
#endif

int walt_profile_detail(walt_index* idx, float* out4) {
  if (!idx || !out4 || !idx->profile || !idx->ev_valid) return fail(WALT_EINVAL, "no profiled call recorded");
  WALT_HIP(hipEventSynchronize(idx->ev[2]));
  for (int k = 0; k < 4; ++k) out4[k] = 0.f;
  for (int i = 1; i < idx->n_detail; ++i) {
    float ms = 0.f;
    WALT_HIP(hipEventElapsedTime(&ms, idx->ev_detail[i - 1], idx->ev_detail[i]));
    if (idx->ev_kind[i] < 4) out4[idx->ev_kind[i]] += ms;
  }
  return WALT_OK;
}

int walt_profile_last(walt_index* idx, float* pack_ms, float* map_ms) {
  if (!idx || !idx->profile || !idx->ev_valid) return fail(WALT_EINVAL, "no profiled call recorded");
  WALT_HIP(hipEventSynchronize(idx->ev[2]));
  float a = 0, b = 0;
  WALT_HIP(hipEventElapsedTime(&a, idx->ev[0], idx->ev[1]));
  WALT_HIP(hipEventElapsedTime(&b, idx->ev[1], idx->ev[2]));
  if (pack_ms) *pack_ms = a;
  if (map_ms) *map_ms = b;
  return WALT_OK;
}

size_t walt_se_workspace_bytes(uint32_t n, uint32_t max_read_len) {
  int nw = nw_for_len(max_read_len);
  if (!nw) nw = 64;
  return 64 * sizeof(uint32_t) + kStatShardBytes + 3 * se_stride(n) * sizeof(uint32_t) +
         codes2_words((uint64_t)n * max_read_len) * sizeof(uint32_t) + 16 + se_heavy_bytes(n, nw) + 16 + se_carry_bytes(n);
}

int walt_map_se_batch_device(walt_index* idx, const void* d_bases, const void* d_offsets, uint32_t n,
                             uint32_t max_read_len, int ag_wildcard, uint32_t max_mismatches, uint32_t b,
                             void* d_out, void* d_stats, void* d_workspace, size_t workspace_bytes, void* stream) {
  return map_se_device(idx, d_bases, d_offsets, n, max_read_len, ag_wildcard, max_mismatches, b, d_out, d_stats,
                       d_workspace, workspace_bytes, reinterpret_cast<hipStream_t>(stream));
}

int walt_map_se_batch(walt_index* idx, const char* bases, const uint64_t* offsets, uint32_t n, int ag_wildcard,
                      uint32_t max_mismatches, uint32_t b, walt_best_match* out, walt_batch_stats* stats) {
  if (!idx || !offsets || (!bases && n && offsets[n] > 0) || (!out && n)) return fail(WALT_EINVAL, "walt_map_se_batch: bad argument");
  if (stats) memset(stats, 0, sizeof(*stats));
  if (n == 0) return WALT_OK;
  WALT_HIP(hipSetDevice(idx->device));
  uint32_t max_len = 0;
  for (uint32_t i = 0; i < n; ++i) {
    if (offsets[i + 1] < offsets[i]) return fail(WALT_EINVAL, "offsets not non-decreasing");
    uint64_t l = offsets[i + 1] - offsets[i];
    if (l > 1024) return fail(WALT_EINVAL, "read length above 1024 is not supported");
    if (l > max_len) max_len = (uint32_t)l;
  }
  const uint64_t nbytes = offsets[n] - offsets[0];
  void *d_bases = nullptr, *d_off = nullptr, *d_out = nullptr, *d_stats = nullptr, *d_ws = nullptr;
  int rc = WALT_OK;
  hipError_t e;
  // offsets as the kernels want them: relative to the first read of the batch (a caller that shards a batch
  // over several devices passes a slice of its offsets array)
  const uint64_t* off_src = offsets;
  std::vector<uint64_t> rel;
  if (offsets[0] != 0) {
    rel.resize((size_t)n + 1);
    for (uint32_t i = 0; i <= n; ++i) rel[i] = offsets[i] - offsets[0];
    off_src = rel.data();
  }
  if ((e = host_api_buffer(idx, 0, nbytes + 16, &d_bases)) != hipSuccess ||
      (e = host_api_buffer(idx, 1, ((size_t)n + 1) * sizeof(uint64_t), &d_off)) != hipSuccess ||
      (e = host_api_buffer(idx, 2, (size_t)n * sizeof(walt_best_match), &d_out)) != hipSuccess ||
      (e = host_api_buffer(idx, 3, sizeof(walt_batch_stats), &d_stats)) != hipSuccess ||
      (e = host_api_buffer(idx, 4, walt_se_workspace_bytes(n, max_len), &d_ws)) != hipSuccess)
    return fail(WALT_ENOMEM, std::string("hipMalloc failed: ") + hipGetErrorString(e));
  if ((e = hipMemcpyAsync(d_bases, bases + offsets[0], nbytes, hipMemcpyHostToDevice, nullptr)) != hipSuccess ||
      (e = hipMemcpyAsync(d_off, off_src, ((size_t)n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, nullptr)) != hipSuccess ||
      (e = hipMemsetAsync(d_stats, 0, sizeof(walt_batch_stats), nullptr)) != hipSuccess)
    return fail(WALT_EHIP, std::string("upload failed: ") + hipGetErrorString(e));
  rc = map_se_device(idx, d_bases, d_off, n, max_len, ag_wildcard, max_mismatches, b, d_out, d_stats, d_ws,
                     walt_se_workspace_bytes(n, max_len), nullptr);
  if (!rc) rc = check_read_errors(d_ws, nullptr);
  if (!rc) {
    if ((e = hipMemcpy(out, d_out, (size_t)n * sizeof(walt_best_match), hipMemcpyDeviceToHost)) != hipSuccess)
      rc = fail(WALT_EHIP, std::string("download failed: ") + hipGetErrorString(e));
    walt_batch_stats st;
    if (!rc && hipMemcpy(&st, d_stats, sizeof(st), hipMemcpyDeviceToHost) == hipSuccess && stats) *stats = st;
  }
  return rc;
}

}  // extern "C"
