// map_pe.hip -- paired-end seed-and-extend, top-k and pair merge on MI355X.
//
// Replaces, per batch: the mate / strand loops around PairEndMapping
// (reference paired.cpp:642-672, body 106-201), the per-read
// std::priority_queue top-k (paired.hpp:51-74) including its libstdc++ tie
// order, the heap drain (paired.cpp:685-692) and the pair search + single-mate
// fallback of MergePairedEndResults (paired.cpp:474-545), which the reference
// runs serially on the host.
//
// Kernels (per mate):
//   k_pe_topk_dual  pass 1, every read, one per lane, seed-major like map_se.hip's pass 1:
//               the '+' and '-' probes of a seed shift are issued together.  It completes
//               the reads whose heap can never fill (every region <= 4 slots, at most
//               kFastCands candidates, fewer than top_k): with a heap that is never full
//               the reference takes no early exit (paired.cpp:133-141), so all six probes
//               are needed and only the PUSH ORDER (+s0 +s1 +s2 -s0 -s1 -s2) has to be
//               reproduced.  '+' candidates are pushed on arrival, '-' candidates wait in
//               registers; the heap lives in LDS and is popped there, so these reads write
//               their ranked list directly.
//               Other reads go to one of two lists.
//   k_pe_topk_list<.., false>  "complex" reads (a large region, many candidates, small
//               top_k): strand-major sequential probing with the exact early exits,
//               candidates pushed IN CANDIDATE ORDER into the read's heap (LDS, top_k slots);
//               large regions are verified by the whole wave and pushed by the owner lane.
//               A short list is spread down to one read per wavefront.
//   k_pe_topk_list<.., true>   reads whose probe met the Bloom filter of chromosome-end
//               entries: the same, with the literal search of the reference.
//               Every kernel pops its heaps itself (paired.cpp:685-692) and writes the ranked
//               lists in pop order; no heap ever lives in HBM.
//   k_pe_merge / k_pe_merge_heavy  pair search, one pair per lane / per wavefront.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "map_common.h"
#include "map_items.h"

namespace walt {

// Pairs processed per workspace pass.  The list kernels are latency-bound (a handful of slow reads, one per
// wavefront), so their cost per pass is nearly constant: large passes amortise it.  The ranked lists
// (2 x top_k x 12 B per pair) bound the pass by a ~10 GB workspace budget.
constexpr uint32_t kPeChunkMax = 1u << 23;
// The geometry of a call -- pairs per pass, staged reads per round, rounds, survivor pool -- follows from (n, top_k),
// the index's options and ONE more bit, `roomy`: larger passes taken in fewer rounds when the device has the room
// (0 = 8 M-pair passes, staged lists in four rounds: 24 GB of workspace at -k 50; 1 = 10 M-pair passes in one round:
// 50 GB, every launch of the staged kernels gets four times the items and the passes are a third fewer -- 273 -> 247 ms
// per 50 M pairs).  Nothing about it is kept in the process: the CALLER's workspace size decides (a call is roomy when
// the workspace it was given holds the roomy geometry; walt_pe_workspace_bytes_best sizes one from the free memory of
// the index's device), so a sizing call and a mapping call cannot disagree and two indexes on two devices do not share
// a decision (round 3 kept it in a process-global).
struct PeGeometry {
  uint32_t chunk;        // pairs per pass
  uint32_t rounds;       // rounds a pass takes its staged list in
  uint32_t ccap;         // staged reads per round, pass and mate
  uint32_t pool_chunks;  // survivor chunks of a round
  bool roomy;
};
constexpr uint32_t kCoopUnroll = 2;    // 64-candidate groups of a large region verified per step (4 cost a wave per SIMD in registers)

// One read per lane.  LITERAL as in map_se.hip: pass 1 defers reads that hit a
// BAD bucket, pass 2 maps them from scratch (their heap restarts empty).
template <int NW, bool LITERAL>
__device__ __forceinline__ void pe_process(const IndexView& iv, BlockShared& sh, const uint32_t* si,
                                           const uint32_t* __restrict__ codes2, const uint64_t* __restrict__ offsets,
                                           uint32_t* __restrict__ err, uint32_t r,
                                           bool valid, uint32_t strand_base, uint32_t max_mm, uint32_t b,
                                           uint32_t top_k, HeapEnt* heap /* this lane's top_k slots (LDS) */,
                                           Candidate* __restrict__ ranked,
                                           uint32_t* __restrict__ heap_n, uint32_t* __restrict__ defer_count,
                                           uint32_t* __restrict__ defer_list, uint32_t& n_probe,
                                           uint32_t& n_verified, uint32_t& n_big, uint32_t& len_out,
                                           uint32_t heap_cap = 0xFFFFFFFFu, uint32_t* __restrict__ over_count = nullptr,
                                           uint32_t* __restrict__ over_list = nullptr) {
  // heap_cap < top_k: this lane's heap has room for heap_cap candidates only (more reads per wavefront share
  // the LDS); a read that collects more is handed to over_list and mapped again with a full heap.  Below
  // top_k entries TopCandidates::Push only appends, so nothing else changes.
  bool overflow = false;
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
  uint32_t hsize = 0;

  for (uint32_t fi = 0; fi < 2; ++fi) {
    const StrandView& sv = iv.s[strand_base + fi];
#pragma unroll 1
    for (uint32_t seed_i = 0; seed_i < kPat; ++seed_i) {
      // paired.cpp:133-149: stop once the heap is full of exact (seed >= 1) or
      // one-mismatch (seed >= 2; pattern 7: >= 4) candidates; top only decreases, so per-seed
      // predicates equal the reference's `break`.
      const bool full = hsize >= top_k;
      const uint32_t top_mm = hsize ? heap_mm(heap[0]) : 0xFFFFFFFFu;
      bool act = mappable && !(full && top_mm == 0 && seed_i) && !(full && top_mm == 1 && seed_i >= kExitOneMismatch);
      Lookup lk;
      lk.npos = 0;
      lk.reg = empty_region();
      if (act) {
        uint32_t care[kCareWords];
        uint32_t slot, span;
        seed_query<NW>(lr.rd, seed_len_of(lr.repeats), seed_i, ga, Bd, sh.pcode4, care, slot, span);
        const uint32_t bk = bloom_key_of_care(care);
        if (!LITERAL && danger_filter_hit(sv.bloom[bloom_block(bk, sv.bloom_mask)], care)) {
          deferred = true;
          mappable = false;
          defer_iter = fi * kPat + seed_i;
        } else {
          seed_lookup_ex(iv, sv, care, slot, span, seed_len_of(lr.repeats), lk, !LITERAL);
        }
      }
      const Region reg = lk.reg;
      uint32_t size = reg.l <= reg.u ? reg.u - reg.l + 1 : 0;
      if (size) ++n_probe;
      if (size > b) size = 0;  // paired.cpp:161-163
      uint32_t mk[NW];
      make_masks<NW>(mk, sh.mask_table, seed_i, lr.repeats >= kMinRepeats ? lr.repeats : kMinRepeats, lr.len);

      if (size && size <= kSmallRegion) {
#pragma unroll
        for (uint32_t k = 0; k < kSmallRegion; ++k) {  // static k: lk.pos[] stays in registers
          if (k < size) {
            uint32_t pos = k < lk.npos ? lk.pos[k] : sv.ent[reg.l + k].pos, gp, mm;
            if (verify_candidate<NW>(sv, si, iv.start_index, n_chrom, pos, seed_i, lr.len, lr.rd, mk, gp, mm)) {
              ++n_verified;
              if (mm <= max_mm) {  // paired.cpp:192-195
                HeapEnt e; e.pos = gp; e.mms = mm | (fi << 31);
                if (heap_cap < top_k && hsize >= heap_cap) { overflow = true; mappable = false; }
                if (!overflow) topk_push(heap, hsize, top_k, e);
              }
            }
          }
        }
      }
      unsigned long long big = __ballot(size > kSmallRegion);
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
        // kCoopUnroll x 64 candidates per step: the slot loads of a step are independent, then the genome
        // windows are, so a step costs two memory round trips instead of 2 x kCoopUnroll (a listed read
        // owns its wave, nothing else hides the latency); pushes stay in slot order.
        const DenseRange rb = dense_range(sv, o_l, o_size, win_usable<NW>(sv, o_len));
        for (uint32_t base = 0; base < o_size; base += 64 * kCoopUnroll) {
          uint32_t cgp[kCoopUnroll], cmm[kCoopUnroll];
          coop_verify_groups<NW, (int)kCoopUnroll>(sv, si, iv.start_index, n_chrom, o_l, o_size, base, seed_i, o_len, o_rd, o_mk, lane, rb,
                                                   cgp, cmm);  // paired.cpp:166-190
#pragma unroll
          for (uint32_t u = 0; u < kCoopUnroll; ++u) n_verified += cmm[u] != 0xFFFFFFFFu ? 1u : 0u;
#pragma unroll
          for (uint32_t u = 0; u < kCoopUnroll; ++u) {
            if (base + u * 64 >= o_size) break;
            const uint32_t mm = cmm[u], gp = cgp[u];
            // candidates that can still enter the owner's heap, judged against the
            // heap state at the start of this chunk (top only decreases, so this
            // never drops a candidate TopCandidates::Push would have accepted)
            const uint32_t o_hsize = bcast(hsize, owner);
            uint32_t o_top = 0xFFFFFFFFu;
            if ((int)lane == owner && hsize) o_top = heap_mm(heap[0]);
            o_top = bcast(o_top, owner);
            const bool o_full = o_hsize >= top_k;
            unsigned long long push = __ballot(mm <= max_mm && (!o_full || mm < o_top));
            while (push) {
              const int src = (int)__ffsll((long long)push) - 1;
              push &= push - 1;
              const uint32_t c_gp = bcast(gp, src), c_mm = bcast(mm, src);
              if ((int)lane == owner) {
                HeapEnt e; e.pos = c_gp; e.mms = c_mm | (fi << 31);
                if (heap_cap < top_k && hsize >= heap_cap) { overflow = true; mappable = false; }
                if (!overflow) topk_push(heap, hsize, top_k, e);
              }
            }
          }
        }
        if ((int)lane == owner) ++n_big;
      }
    }
  }
  const bool over = valid && overflow && !(!LITERAL && deferred);
  if (over_list) wave_append(over, r, over_count, over_list);  // every lane of the wave gets here
  if (!LITERAL && deferred) {
    defer_list[atomicAdd(defer_count, 1u)] = r <= kDeferMask ? (r | (defer_iter << kDeferShift)) : r;
  } else if (valid && !over) {
    // paired.cpp:685-692: pop everything; ranked[r][i] = i-th popped (descending mismatch)
    heap_n[r] = hsize;
    Candidate* out = ranked + (uint64_t)r * top_k;
    uint32_t i = 0;
    while (hsize) {
      const HeapEnt e = heap_pop(heap, hsize);
      Candidate c; c.genome_pos = e.pos; c.strand = (e.mms >> 31) ? '-' : '+'; c.mismatch = heap_mm(e);
      out[i++] = c;
    }
  }
}

__device__ __forceinline__ void pe_flush(uint32_t shortv, uint32_t n_probe, uint32_t n_verified, uint32_t n_big,
                                         unsigned long long* __restrict__ shards) {
  block_flush_stats(shortv, n_probe, n_verified, n_big, shards);
}

#if WALT_SEEDPATTERN == 3  // pass 1 exists for the default pattern only (core.h probe_is_dangerous)
// ---------------------------------------------------------------------------
// pass 1 (see the header comment)
// ---------------------------------------------------------------------------
constexpr uint32_t kFastCands = 6;          // LDS heap slots per lane: 6 x 8 B x 256 lanes = 12 KB per block

struct LdsHeap {  // entry i of this lane's heap; entries of one index are contiguous across lanes (no bank conflicts)
  HeapEnt* base;
  __device__ __forceinline__ HeapEnt& operator[](uint32_t i) const { return base[i * kBlock]; }
};

template <int NW>
__device__ __forceinline__ void pe_process_dual(const IndexView& iv, BlockShared& sh, const PreFilter& pf, const uint32_t* si, LdsHeap fast,
                                                const uint32_t* __restrict__ codes2,
                                                const uint64_t* __restrict__ offsets, uint32_t* __restrict__ err,
                                                uint32_t r, bool valid, uint32_t strand_base, uint32_t max_mm,
                                                uint32_t b, uint32_t top_k, Candidate* __restrict__ ranked,
                                                uint32_t* __restrict__ heap_n, uint32_t* __restrict__ bloom_count,
                                                uint32_t* __restrict__ bloom_list, uint32_t* __restrict__ cplx_count,
                                                uint32_t* __restrict__ cplx_list, uint32_t& n_probe,
                                                uint32_t& n_verified, uint32_t& len_out, WaveList& wl_cplx) {
  const uint32_t n_chrom = iv.n_chrom;
  const uint32_t top_step = top_step_of(n_chrom);
  const StrandView& svp = iv.s[strand_base];
  const StrandView& svm = iv.s[strand_base + 1];
  const uint32_t ga = strand_base >> 1, Bd = iv.dir_bits;
  LaneRead<NW> lr;
  {
    uint64_t o = 0, oe = 0;
    if (valid) { o = offsets[r]; oe = offsets[r + 1]; }
    lane_load_read<NW>(lr, codes2, offsets[0], o, oe, valid, ga, err, iv);
  }
  len_out = lr.len;
  bool mappable = valid && lr.len >= kMinReadLen;
  bool deferred = false, cplx = false;
  uint32_t defer_iter = 0;
  uint32_t hsize = 0;                                             // '+' candidates, pushed on arrival
  uint32_t mp0 = 0, mp1 = 0, mp2 = 0, mm0 = 0, mm1 = 0, mm2 = 0;  // '-' candidates, pushed after seed 2
  uint32_t n_minus = 0;

#pragma unroll 1
  for (uint32_t seed_i = 0; seed_i < 3; ++seed_i) {
    bool need = mappable;
    uint32_t care[kCareWords] = {0, 0, 0, 0};
    uint32_t slot = 0, span = 0;
    if (need) seed_query<NW>(lr.rd, lr.repeats, seed_i, ga, Bd, sh.pcode4, care, slot, span);
    // Bloom blocks of both strands and the directory pairs of both strands: independent loads, one wait
    const uint32_t bkey = bloom_key_of_care(care);
    uint64_t bw_p = 0, bw_m = 0;
    if (need && prefilter_hit(pf, 0, bkey)) bw_p = svp.bloom[bloom_block(bkey, svp.bloom_mask)];
    if (need && prefilter_hit(pf, 1, bkey)) bw_m = svm.bloom[bloom_block(bkey, svm.bloom_mask)];
    SlotProbe pp, pm;
    uint32_t hi_p, hi_m;
    probe_issue(svp, need, slot, span, pp, hi_p);
    probe_issue(svm, need, slot, span, pm, hi_m);
    const bool bad_p = need && bw_p && danger_filter_hit(bw_p, care);
    const bool bad_m = need && bw_m && danger_filter_hit(bw_m, care);
    if (bad_p || bad_m) {
      deferred = true;
      mappable = false;
      need = false;
      defer_iter = seed_i + (bad_p ? 0u : 3u);
    }
    pp.ne = (need && hi_p > pp.lo) ? hi_p - pp.lo : 0u;
    pm.ne = (need && hi_m > pm.lo) ? hi_m - pm.lo : 0u;
    if (pp.ne > kScanMax || pm.ne > kScanMax) {  // a long slot (repeat family): searched by the list kernel, not here
      cplx = true;
      mappable = false;
      pp.ne = pm.ne = 0;
    }
    probe_entries(svp, pp);
    probe_entries(svm, pm);
    Lookup lp, lm;
    bool tail_p, tail_m;
    probe_resolve<(NW > 8)>(svp, pp, care, lr.repeats, lp, tail_p);
    probe_resolve<(NW > 8)>(svm, pm, care, lr.repeats, lm, tail_m);
    uint32_t size_p = lp.reg.l <= lp.reg.u ? lp.reg.u - lp.reg.l + 1 : 0;
    uint32_t size_m = lm.reg.l <= lm.reg.u ? lm.reg.u - lm.reg.l + 1 : 0;
    n_probe += (size_p ? 1u : 0u) + (size_m ? 1u : 0u);
    if (size_p > b) size_p = 0;  // paired.cpp:161-163
    if (size_m > b) size_m = 0;
    if (size_p > kSmallRegion || size_m > kSmallRegion) {  // a large region: the sequential kernel's business
      cplx = true;
      mappable = false;
      size_p = size_m = 0;
    }
    uint32_t mk[NW];
    make_masks<NW>(mk, sh.mask_table, seed_i, lr.repeats >= kMinRepeats ? lr.repeats : kMinRepeats, lr.len);
    const uint32_t kmax = size_p > size_m ? size_p : size_m;
#pragma unroll 1
    for (uint32_t k = 0; k < kmax; ++k) {
      const bool act_p = k < size_p, act_m = k < size_m;
      uint32_t pos_p = k == 0 ? lp.pos[0] : k == 1 ? lp.pos[1] : k == 2 ? lp.pos[2] : lp.pos[3];
      uint32_t pos_m = k == 0 ? lm.pos[0] : k == 1 ? lm.pos[1] : k == 2 ? lm.pos[2] : lm.pos[3];
      if (act_p && k >= lp.npos) pos_p = svp.ent[lp.reg.l + k].pos;
      if (act_m && k >= lm.npos) pos_m = svm.ent[lm.reg.l + k].pos;
      bool ok_p, ok_m;
      uint32_t gp_p, gp_m, mm_p, mm_m;
      verify_nobranch<NW>(svp, sh, si, iv.start_index, n_chrom, top_step, act_p, pos_p, seed_i, lr.len, lr.rd, mk, ok_p, gp_p, mm_p);
      verify_nobranch<NW>(svm, sh, si, iv.start_index, n_chrom, top_step, act_m, pos_m, seed_i, lr.len, lr.rd, mk, ok_m, gp_m, mm_m);
      if (NW > 8) {  // long seeds: a single key-equal candidate still owes its care chars >= 44 (probe_resolve)
        // inactive lanes carry no valid position: read from 0 like verify_nobranch does
        const bool t_p = tail_care_ok(svp, act_p ? pos_p : 0u, care, lr.repeats), t_m = tail_care_ok(svm, act_m ? pos_m : 0u, care, lr.repeats);
        ok_p = ok_p && (!tail_p || t_p);
        ok_m = ok_m && (!tail_m || t_m);
      }
      if (ok_p) {
        ++n_verified;
        if (mm_p <= max_mm) {  // paired.cpp:192-195
          if (hsize < kFastCands) {
            HeapEnt e; e.pos = gp_p; e.mms = mm_p;
            heap_push(fast, hsize, e);
          } else {
            cplx = true;
          }
        }
      }
      if (ok_m) {
        ++n_verified;
        if (mm_m <= max_mm) {
          if (n_minus == 0) { mp0 = gp_m; mm0 = mm_m; }
          else if (n_minus == 1) { mp1 = gp_m; mm1 = mm_m; }
          else if (n_minus == 2) { mp2 = gp_m; mm2 = mm_m; }
          else cplx = true;
          ++n_minus;
        }
      }
    }
  }
  const uint32_t total = hsize + n_minus;
  const bool to_cplx = !deferred && valid && (cplx || total > kFastCands || total >= top_k);
  if (bloom_list == cplx_list) {  // (uniform) the staged path: filter hits, tagged, share the complex reads' list and its buffer
    wavelist_append(wl_cplx, deferred, r <= kDeferMask ? (r | (defer_iter << kDeferShift)) : r, cplx_count, cplx_list);
  } else {
    wave_append(deferred, r <= kDeferMask ? (r | (defer_iter << kDeferShift)) : r, bloom_count, bloom_list);
  }
  wavelist_append(wl_cplx, to_cplx, r, cplx_count, cplx_list);  // a fifth of the reads: buffered (map_common.h WaveList)
  if (!deferred && !to_cplx && valid) {
    // the heap never fills: plain pushes in the reference's order, then the drain of paired.cpp:685-692
    if (n_minus > 0) { HeapEnt e; e.pos = mp0; e.mms = mm0 | 0x80000000u; heap_push(fast, hsize, e); }
    if (n_minus > 1) { HeapEnt e; e.pos = mp1; e.mms = mm1 | 0x80000000u; heap_push(fast, hsize, e); }
    if (n_minus > 2) { HeapEnt e; e.pos = mp2; e.mms = mm2 | 0x80000000u; heap_push(fast, hsize, e); }
    Candidate* out = ranked + (uint64_t)r * top_k;
    uint32_t i = 0;
    while (hsize) {
      const HeapEnt e = heap_pop(fast, hsize);
      Candidate c; c.genome_pos = e.pos; c.strand = (e.mms >> 31) ? '-' : '+'; c.mismatch = heap_mm(e);
      out[i++] = c;
    }
    heap_n[r] = total;
  }
}

template <int NW>
__global__ __launch_bounds__(kBlock, (NW <= 8 ? 4 : (NW <= 10 ? 3 : 1))) void k_pe_topk_dual(
    IndexView iv, const uint32_t* __restrict__ codes2, const uint64_t* __restrict__ offsets, uint32_t* __restrict__ err,
    uint32_t n, uint32_t strand_base, uint32_t max_mm, uint32_t b, uint32_t top_k,
    const uint32_t* __restrict__ mask_table, Candidate* __restrict__ ranked, uint32_t* __restrict__ heap_n,
    unsigned long long* __restrict__ stats, uint32_t* __restrict__ bloom_count, uint32_t* __restrict__ bloom_list,
    uint32_t* __restrict__ cplx_count, uint32_t* __restrict__ cplx_list) {
  __shared__ BlockShared sh;
  __shared__ HeapEnt s_fast[kFastCands][kBlock];
  __shared__ PreFilter pf;
  constexpr uint32_t kCplxBuf = 128;  // (the kernel's 38 KB of LDS leave 2 KB at four blocks per CU)
  __shared__ uint32_t s_cplx[(kBlock / 64) * kCplxBuf];
  WaveList wl_cplx;
  wl_cplx.buf = s_cplx + (threadIdx.x >> 6) * kCplxBuf;
  wl_cplx.n = 0;
  wl_cplx.cap = kCplxBuf;
  prefilter_stage(pf, iv, strand_base);
  const uint32_t* si = block_prologue(sh, iv, mask_table, strand_base);
  LdsHeap fast;
  fast.base = &s_fast[0][threadIdx.x];
  uint32_t n_probe = 0, n_verified = 0, shortv = 0;
  // each block walks its own contiguous slice of the batch
  const uint64_t chunks = ((uint64_t)n + blockDim.x - 1) / blockDim.x;
  const uint64_t per_block = (chunks + gridDim.x - 1) / gridDim.x;
  const uint64_t c_lo = (uint64_t)blockIdx.x * per_block;
  const uint64_t c_hi = c_lo + per_block < chunks ? c_lo + per_block : chunks;
  for (uint64_t c = c_lo; c < c_hi; ++c) {
    const uint64_t r64 = c * blockDim.x + threadIdx.x;
    const bool valid = r64 < n;
    const uint32_t r = valid ? (uint32_t)r64 : 0;
    uint32_t len;
    pe_process_dual<NW>(iv, sh, pf, si, fast, codes2, offsets, err, r, valid, strand_base, max_mm, b, top_k, ranked, heap_n,
                        bloom_count, bloom_list, cplx_count, cplx_list, n_probe, n_verified, len, wl_cplx);
    // paired.cpp:112-115: too_short once per strand pass
    shortv += (valid && len < kMinReadLen) ? 2u : 0u;
  }
  wavelist_flush(wl_cplx, cplx_count, cplx_list);
  pe_flush(shortv, n_probe, n_verified, 0, stats);
}

// ---------------------------------------------------------------------------
// Staged path for the complex reads (and the reads that met the chromosome-end filter): on a repeat-rich genome a
// few per cent of the reads own nearly all candidates -- thousands each, because a heap that is not full takes no
// early exit -- and the list kernel below, which verifies a region with the wavefront of the lane that owns it
// and keeps that wavefront's heaps in LDS, runs them at a wavefront and a half per SIMD.  Here instead:
//   k_pe_stage   one read per lane: all six probes (a superset of the reference's: its exits need a full heap),
//                both strands of a seed shift looked up together; regions of up to kMidRegion candidates are
//                verified in place and their SURVIVORS (mismatches <= -m, in slot order) stored per probe;
//                larger regions become work items (map_items.h).  The exact chromosome-end test runs here;
//                only truly dangerous reads go on to the literal list.
//   k_pe_verify  one region per wavefront (item_stream): survivors appended in slot order to the probe's list.
//                Filter: a candidate is dropped when top_k earlier survivors OF THE SAME REGION have no more
//                mismatches than it has -- the heap is then full of candidates at least as good, and
//                TopCandidates::Push (paired.hpp:63-70) would refuse it whatever else the heap holds.
//   k_pe_push    one read per lane: the probes in the reference's order (+s0 +s1 +s2 -s0 -s1 -s2) under its exact
//                exits (paired.cpp:133-149; a probe after an exit is simply not read), survivors pushed in slot
//                order into the read's heap (LDS), heap popped into the ranked list.
// A read whose probe keeps more survivors than kPeChunks chunks hold, or beyond the staged capacity of the pass, goes
// to the list kernel as before.  The reads with a truly dangerous probe take the same three kernels in a second
// round (LITERAL = true: the dangerous probes' regions come from the literal search, core.h seed_lookup_ex).
// ---------------------------------------------------------------------------
constexpr uint32_t kPeMidRegion = 16;  // regions up to this size are verified by their own lane
constexpr uint32_t kPeChunkEnts = 64;  // survivors per chunk of the pool
constexpr uint32_t kPeChunks = 8;      // chunks a probe may take (512 survivors); more: the list kernel maps the read
constexpr uint32_t kPoolGrab = 4;      // pool chunks a wavefront of k_pe_verify takes per atomic (SurvivorSink::take_chunk)

// Survivors {position, mismatches} of probe p of staged read j.  A region verified in place has at most
// kPeMidRegion of them: inl[(p * kPeMidRegion + k) * ccap + j].  An item's survivors go to chunks of kPeChunkEnts
// taken from a pool as they come (their number is not known beforehand and runs from none to hundreds):
// chunk[(p * kPeChunks + c) * ccap + j] = number of the c-th chunk.  surv_n[p * ccap + j] = count, bit 31: chunked.
struct PeStage {
  uint2* inl;
  uint32_t* surv_n;
  uint32_t* cz;      // [p * ccap + j]: the probe's survivors with 0 mismatches | those with at most 1 << 16 (each capped at 65535)
  uint32_t* chunk;
  uint2* pool;
  uint32_t* pool_next;   // chunks handed out from the dynamic part
  uint32_t pool_chunks;  // chunks the pool holds
  uint32_t static_n;     // the first chunk of dense item i of seed s is chunk s * static_n + i while i < static_n (no
                         // atomic: same-address atomics run at ~10 ns each, and nearly every item takes one chunk);
                         // the dynamic part starts at 3 * static_n
  uint32_t* flag;    // [j]: 1 = gone to the literal list, 2 = a probe outgrew its chunks or the pool
  ItemQueue q;       // 2 * ccap items (emptied after every seed); id = j | probe << 24, probe = 3 * strand + seed shift
  uint32_t ccap;     // staged reads per pass and mate
  uint32_t defer_min;  // long seeds: key-equal ranges of more slots than this are narrowed by the verifier (0xFFFFFFFF: never)
  uint32_t lit_fuse;   // the literal round's three seed shifts in ONE launch when its list is short enough (k_pe_stage)
  uint32_t* fused_note;  // control word 3 of the mate: set to 1 by a literal round that ran that way (tests, bench)
};

template <int NW, bool LITERAL>
__device__ __forceinline__ void pe_stage_dual(const IndexView& iv, BlockShared& sh, const PreFilter& pf, const uint32_t* si,
                                              const uint32_t* __restrict__ codes2, const uint64_t* __restrict__ offsets,
                                              uint32_t* __restrict__ err, uint32_t r, bool valid, uint32_t j,
                                              uint32_t strand_base, uint32_t max_mm, uint32_t b, const PeStage& ps,
                                              uint32_t* __restrict__ lit_count, uint32_t* __restrict__ lit_list,
                                              uint32_t stage_seed, uint32_t strands_in, bool every_probe, uint32_t top_k,
                                              uint32_t& n_probe, uint32_t& n_verified, uint32_t& n_big) {
  const uint32_t n_chrom = iv.n_chrom;
  const uint32_t top_step = top_step_of(n_chrom);
  const StrandView& svp = iv.s[strand_base];
  const StrandView& svm = iv.s[strand_base + 1];
  const uint32_t ga = strand_base >> 1, Bd = iv.dir_bits;
  LaneRead<NW> lr;
  {
    uint64_t o = 0, oe = 0;
    if (valid) { o = offsets[r]; oe = offsets[r + 1]; }
    lane_load_read<NW>(lr, codes2, offsets[0], o, oe, valid, ga, err, iv);
  }
  bool mappable = valid && lr.len >= kMinReadLen;
  bool dead = false;
  uint32_t defer_iter = 0;
  const uint64_t ccap = ps.ccap;
  uint32_t z0_p = 0, z1_p = 0, z0_m = 0, z1_m = 0;  // this seed's in-place survivors with 0 / at most 1 mismatches
  auto keep = [&](uint32_t probe, uint32_t& cnt, uint32_t& z0, uint32_t& z1, bool ok, uint32_t gp, uint32_t mm) {  // survivors of an in-place region
    if (ok && mm <= max_mm) {  // paired.cpp:192-195
      ps.inl[((uint64_t)probe * kPeMidRegion + cnt) * ccap + j] = make_uint2(gp, mm);
      ++cnt;
      z0 += mm == 0 ? 1u : 0u;
      z1 += mm <= 1 ? 1u : 0u;
    }
  };
  // Which of this seed's two probes the reference can still make (paired.cpp:133-149: it stops a strand once the heap
  // is full of candidates with no mismatch, from seed 2 on with at most one).  The heap holds the top_k best of what
  // was pushed, so that is: top_k pushed candidates that good.  Counted over the probes of the earlier seeds that
  // were made (a '+' probe of THIS seed or a later one could only add to what the '-' strand sees): never a
  // probe dropped that the reference makes; k_pe_push applies the exact exits.
  // strands: which of the seed's two probes are this lane's (bit 0 '+', bit 1 '-').  Both, except in the literal round
  // taken in one launch (k_pe_stage), where every (read, seed, strand) has a lane of its own and every probe is made
  // (every_probe: the counts of the earlier seeds' items are not known yet -- a superset of the superset below).
  const uint32_t strands = LITERAL ? strands_in : 3u;
  const bool skip_exits = LITERAL && every_probe;
  bool need_p = mappable && (strands & 1u), need_m = mappable && (strands & 2u);
  if (stage_seed > 0 && valid && !skip_exits) {
    if (ps.flag[j] & 1u) mappable = need_p = need_m = false;  // went to the literal list at an earlier seed
    auto zeros = [&](uint32_t probe, bool made, uint32_t& a0, uint32_t& a1) {
      const uint32_t v = made ? ps.cz[(uint64_t)probe * ccap + j] : 0u;
      a0 += v & 0xFFFFu;
      a1 += v >> 16;
    };
    uint32_t p0 = 0, p1 = 0, a0 = 0, a1 = 0;  // '+' strand so far; both strands so far
    zeros(0, true, p0, p1);
    const bool made_p1 = p0 < top_k;                       // +s1 is made unless +s0 filled the heap with exact matches
    uint32_t m0 = 0, m1 = 0;
    zeros(3, true, m0, m1);
    const bool made_m1 = p0 + m0 < top_k;                  // (superset) -s1
    if (stage_seed == 1) {
      need_p = need_p && made_p1;
      need_m = need_m && made_m1;
    } else {
      zeros(1, made_p1, p0, p1);
      zeros(4, made_m1, m0, m1);
      need_p = need_p && p1 < top_k;                       // +s2: unless top_k candidates with at most one mismatch on '+'
      need_m = need_m && p1 + m1 < top_k;
    }
    (void)a0; (void)a1;
  }

#pragma unroll 1
  for (uint32_t seed_i = stage_seed; seed_i <= stage_seed; ++seed_i) {
    bool need = need_p || need_m;
    uint32_t care[kCareWords] = {0, 0, 0, 0};
    uint32_t slot = 0, span = 0;
    if (need) seed_query<NW>(lr.rd, lr.repeats, seed_i, ga, Bd, sh.pcode4, care, slot, span);
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
    bool lit_p = false, lit_m = false;  // LITERAL: this strand's region comes from the literal search
    if (bad_p || bad_m) {  // the filter is a superset: the exact test decides
      const bool dng_p = bad_p && probe_is_dangerous(svp, care, seed_len_of(lr.repeats));
      const bool dng_m = bad_m && probe_is_dangerous(svm, care, seed_len_of(lr.repeats));
      if (LITERAL) {
        lit_p = dng_p;
        lit_m = dng_m;
      } else if (dng_p || dng_m) {
        dead = true;
        mappable = false;
        need = need_p = need_m = false;
        defer_iter = seed_i + (dng_p ? 0u : 3u);
      }
    }
    pp.ne = (need_p && !lit_p && hi_p > pp.lo) ? hi_p - pp.lo : 0u;
    pm.ne = (need_m && !lit_m && hi_m > pm.lo) ? hi_m - pm.lo : 0u;
    probe_entries(svp, pp);
    probe_entries(svm, pm);
    Lookup lp, lm;
    bool tail_p, tail_m;
    bool defer_p = false, defer_m = false;  // long seeds: the verifier narrows the key-equal range (map_common.h DEFER)
    if constexpr ((NW > 8) && (NW <= 10)) {
      probe_resolve_dual<true, true>(svp, svm, pp, pm, care, lr.repeats, lp, lm, tail_p, tail_m, &defer_p, &defer_m, ps.defer_min,
                                     win_usable<NW>(svp, lr.len) && win_usable<NW>(svm, lr.len));
    } else {
      probe_resolve_dual<(NW > 8)>(svp, svm, pp, pm, care, lr.repeats, lp, lm, tail_p, tail_m);
    }
    if constexpr (LITERAL) {  // LowerBound / UpperBound as the reference runs them (the lanes of a wave come sorted by iteration)
      if (lit_p && need_p) { seed_lookup_ex(iv, svp, care, slot, span, seed_len_of(lr.repeats), lp, false); tail_p = false; defer_p = false; }
      if (lit_m && need_m) { seed_lookup_ex(iv, svm, care, slot, span, seed_len_of(lr.repeats), lm, false); tail_m = false; defer_m = false; }
    }
    uint32_t size_p = lp.reg.l <= lp.reg.u ? lp.reg.u - lp.reg.l + 1 : 0;
    uint32_t size_m = lm.reg.l <= lm.reg.u ? lm.reg.u - lm.reg.l + 1 : 0;
    n_probe += (size_p ? 1u : 0u) + (size_m ? 1u : 0u);
    if (size_p > b && !defer_p) size_p = 0;  // paired.cpp:161-163 (a deferred range: the verifier counts the region)
    if (size_m > b && !defer_m) size_m = 0;
    uint32_t mk[NW];
    make_masks<NW>(mk, sh.mask_table, seed_i, lr.repeats >= kMinRepeats ? lr.repeats : kMinRepeats, lr.len);
    const uint32_t probe_p = seed_i, probe_m = 3 + seed_i;
    uint32_t cnt_p = 0, cnt_m = 0;
    // regions of up to kSmallRegion candidates: candidate k of both strands side by side (positions in registers)
    const bool small_p = size_p && size_p <= kSmallRegion, small_m = size_m && size_m <= kSmallRegion;
    if (small_p || small_m) {
      const uint32_t kmax = (small_p ? size_p : 0u) > (small_m ? size_m : 0u) ? size_p : (small_m ? size_m : size_p);
#pragma unroll 1
      for (uint32_t k = 0; k < kmax; ++k) {
        const bool act_p = small_p && k < size_p, act_m = small_m && k < size_m;
        uint32_t pos_p = k == 0 ? lp.pos[0] : k == 1 ? lp.pos[1] : k == 2 ? lp.pos[2] : lp.pos[3];
        uint32_t pos_m = k == 0 ? lm.pos[0] : k == 1 ? lm.pos[1] : k == 2 ? lm.pos[2] : lm.pos[3];
        if (act_p && k >= lp.npos) pos_p = svp.ent[lp.reg.l + k].pos;
        if (act_m && k >= lm.npos) pos_m = svm.ent[lm.reg.l + k].pos;
        bool ok_p, ok_m;
        uint32_t gp_p, gp_m, mm_p, mm_m;
        verify_nobranch<NW>(svp, sh, si, iv.start_index, n_chrom, top_step, act_p, pos_p, seed_i, lr.len, lr.rd, mk, ok_p, gp_p, mm_p);
        verify_nobranch<NW>(svm, sh, si, iv.start_index, n_chrom, top_step, act_m, pos_m, seed_i, lr.len, lr.rd, mk, ok_m, gp_m, mm_m);
        if (NW > 8) {  // long seeds: a single key-equal candidate still owes its care chars >= 44 (probe_resolve)
          const bool t_p = tail_care_ok(svp, act_p ? pos_p : 0u, care, lr.repeats), t_m = tail_care_ok(svm, act_m ? pos_m : 0u, care, lr.repeats);
          ok_p = ok_p && (!tail_p || t_p);
          ok_m = ok_m && (!tail_m || t_m);
        }
        n_verified += (ok_p ? 1u : 0u) + (ok_m ? 1u : 0u);
        keep(probe_p, cnt_p, z0_p, z1_p, ok_p, gp_p, mm_p);
        keep(probe_m, cnt_m, z0_m, z1_m, ok_m, gp_m, mm_m);
      }
    }
    // mid regions (map_se.hip heavy stages): one pass for the lanes' '+' (or only '-') region, a second for the rest
    {
      const uint32_t nmid_p = (size_p > kSmallRegion && size_p <= kPeMidRegion && !defer_p) ? size_p : 0u;
      const uint32_t nmid_m = (size_m > kSmallRegion && size_m <= kPeMidRegion && !defer_m) ? size_m : 0u;
#pragma unroll 1
      for (uint32_t pass = 0; pass < 2; ++pass) {
        const bool on_m = pass == 0 ? (nmid_p == 0 && nmid_m != 0) : (nmid_p != 0 && nmid_m != 0);
        const bool on_p = pass == 0 && nmid_p != 0;
        const uint32_t nmid = on_p ? nmid_p : (on_m ? nmid_m : 0u);
        if (!__ballot(nmid != 0)) continue;
        const Ent* const ent = on_m ? svm.ent : svp.ent;
        const uint32_t* const g2 = on_m ? svm.g2 : svp.g2;
        const uint32_t my_l = nmid ? (on_m ? lm.reg.l : lp.reg.l) : 0u;
        const uint32_t probe = on_m ? probe_m : probe_p;
        uint32_t posb[kPeMidRegion];
#pragma unroll
        for (uint32_t k = 0; k < kPeMidRegion; ++k) posb[k] = ent[my_l + (k < nmid ? k : 0u)].pos;
        uint32_t cnt = 0, y0 = 0, y1 = 0;
#pragma unroll 1
        for (uint32_t k0 = 0; k0 < kPeMidRegion; k0 += 4) {
          if (!__ballot(k0 < nmid)) break;
          bool ok[4];
          uint32_t gpv[4], win[4][NW + 1];
#pragma unroll
          for (uint32_t jj = 0; jj < 4; ++jj) {
            uint32_t pos = 0;
#pragma unroll
            for (uint32_t k = jj; k < kPeMidRegion; k += 4) pos = (k == k0 + jj) ? posb[k] : pos;
            uint32_t c_lo, c_hi;
            chrom_bounds(sh.start_index, iv.start_index, chrom_tab_of(n_chrom), pos, c_lo, c_hi);
            const uint32_t g = pos - seed_i;
            ok[jj] = k0 + jj < nmid && (pos - c_lo >= seed_i) && (g + lr.len < c_hi);  // paired.cpp:166-171
            gpv[jj] = ok[jj] ? g : 0u;
            const uint32_t* gw = g2 + (gpv[jj] >> 4);
#pragma unroll
            for (int w = 0; w <= NW; w += 4) {
              constexpr int kAll = NW + 1;
              const int cw = kAll - w < 4 ? kAll - w : 4;
              uint32_t qq[4] = {0, 0, 0, 0};
              __builtin_memcpy(qq, gw + w, 4 * cw);
#pragma unroll
              for (int t = 0; t < cw; ++t) win[jj][w + t] = qq[t];
            }
          }
#pragma unroll
          for (uint32_t jj = 0; jj < 4; ++jj) {
            const uint32_t mm = count_mismatch_regs<NW>(win[jj], 2 * (gpv[jj] & 15u), lr.rd, mk);
            n_verified += ok[jj] ? 1u : 0u;
            keep(probe, cnt, y0, y1, ok[jj], gpv[jj], mm);
          }
        }
        if (on_p) { cnt_p = cnt; z0_p = y0; z1_p = y1; }
        if (on_m) { cnt_m = cnt; z0_m = y0; z1_m = y1; }
      }
    }
    // larger regions: work items
    const bool big_p = size_p > kPeMidRegion || defer_p, big_m = size_m > kPeMidRegion || defer_m;  // (a deferred range is an item whatever its size)
    const DenseRange dr_p = dense_range(svp, lp.reg.l, size_p, big_p && win_usable<NW>(svp, lr.len));
    const DenseRange dr_m = dense_range(svm, lm.reg.l, size_m, big_m && win_usable<NW>(svm, lr.len));
    {
      const bool dn_p = big_p && dr_p.hi > dr_p.lo, dn_m = big_m && dr_m.hi > dr_m.lo;
      const bool take2[2] = {big_p, big_m}, dense2[2] = {dn_p, dn_m};
      const uint32_t id2[2] = {j | (probe_p << 24), j | (probe_m << 24)}, l2[2] = {lp.reg.l, lm.reg.l}, size2[2] = {size_p, size_m};
      const uint32_t rec2[2] = {dn_p ? (uint32_t)dr_p.rec : kItemDenseNone, dn_m ? (uint32_t)dr_m.rec : kItemDenseNone};
      const bool tail2[2] = {defer_p, defer_m};
      item_append2<NW>(take2, dense2, id2, l2, size2, rec2, lr.len, seed_i, lr.rd, mk, ps.q, tail2);  // one atomic for both strands' items
      n_big += (big_p ? 1u : 0u) + (big_m ? 1u : 0u);
    }
    if (valid && !big_p && (strands & 1u)) {  // an item's counts come from k_pe_verify
      ps.surv_n[(uint64_t)probe_p * ccap + j] = cnt_p;
      ps.cz[(uint64_t)probe_p * ccap + j] = z0_p | (z1_p << 16);
    }
    if (valid && !big_m && (strands & 2u)) {
      ps.surv_n[(uint64_t)probe_m * ccap + j] = cnt_m;
      ps.cz[(uint64_t)probe_m * ccap + j] = z0_m | (z1_m << 16);
    }
  }
  wave_append(dead, r <= kDeferMask ? (r | (defer_iter << kDeferShift)) : r, lit_count, lit_list);
  if (valid && ((stage_seed == 0 && (strands & 1u)) || dead)) ps.flag[j] = dead ? 1u : 0u;
}

template <int NW, bool LITERAL>
__global__ __launch_bounds__(kBlock, (NW <= 8 ? 4 : (NW <= 10 ? 3 : 1))) void k_pe_stage(
    IndexView iv, const uint32_t* __restrict__ codes2, const uint64_t* __restrict__ offsets, uint32_t* __restrict__ err,
    uint32_t strand_base, uint32_t max_mm, uint32_t b, const uint32_t* __restrict__ mask_table,
    unsigned long long* __restrict__ stats, const uint32_t* __restrict__ list_count, const uint32_t* __restrict__ list,
    PeStage ps, uint32_t* __restrict__ lit_count, uint32_t* __restrict__ lit_list, uint32_t* __restrict__ fb_count,
    uint32_t* __restrict__ fb_list, uint32_t first, uint32_t last_round, uint32_t stage_seed, uint32_t top_k) {
  // this round's part of the list: [first, first + ccap); the last round also sends what lies beyond to fb_list
  uint32_t count = *list_count;
  count = count > first ? count - first : 0u;
  if (!last_round && count > ps.ccap) count = ps.ccap;
  if (count == 0) return;
  // The literal round of a pass holds a per cent of its reads, and each of its launches waits for the slowest lane's
  // literal searches -- three launches, each a chain of up to two searches deep.  When six items per read fit the queue
  // (two per read and seed is what it is sized for) the launch of seed 0 takes the whole round, one (read, seed, strand)
  // per lane, probe-major: the list comes sorted by the iteration of the dangerous probe, so the lanes of a wavefront
  // either all search literally or none does, and no lane runs more than one search.  Every probe is made (the exits
  // between the seeds need the verifier's counts: a superset of the superset the ordinary rounds make, k_pe_push
  // applies the exact exits all the same); the launches of seeds 1 and 2 return at once and their verifier launches
  // find the queue empty.  Decided here, from the list's length on the device, the same way by all three launches.
  bool fused = false;
  if constexpr (LITERAL) {
    if (ps.lit_fuse && last_round && first == 0 && (uint64_t)3 * count <= ps.ccap) {
      if (stage_seed != 0) return;
      fused = true;
      if (blockIdx.x == 0 && threadIdx.x == 0) *ps.fused_note = 1u;
    }
  }
  list += first;
  __shared__ BlockShared sh;
  __shared__ PreFilter pf;
  prefilter_stage(pf, iv, strand_base);
  const uint32_t* si = block_prologue(sh, iv, mask_table, strand_base);
  uint32_t n_probe = 0, n_verified = 0, n_big = 0;
  const uint64_t total = fused ? (uint64_t)6 * count : (uint64_t)count;
  for (uint64_t base = (uint64_t)blockIdx.x * blockDim.x; base < total; base += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t v = base + threadIdx.x;
    const bool in = v < total;
    uint32_t pr = 0;   // fused: the probe this lane makes (0..2 '+', 3..5 '-')
    uint64_t i = v;
    if (fused && in) { pr = (uint32_t)(v / count); i = v - (uint64_t)pr * count; }
    const uint32_t r = in ? (list[i] & kDeferMask) : 0u;
    const bool staged = in && i < ps.ccap;
    wave_append(in && !staged && stage_seed == 0, r, fb_count, fb_list);  // beyond the staged capacity of the pass: the list kernel's
    pe_stage_dual<NW, LITERAL>(iv, sh, pf, si, codes2, offsets, err, r, staged, (uint32_t)i, strand_base, max_mm, b, ps, lit_count,
                      lit_list, fused ? pr % 3u : stage_seed, fused ? (pr >= 3u ? 2u : 1u) : 3u, fused, top_k, n_probe, n_verified,
                      n_big);
  }
  pe_flush(0u, n_probe, n_verified, n_big, stats);
}

// id = j | probe << 24
struct SurvivorSink {
  PeStage ps;
  uint32_t max_mm, top_k;
  uint32_t* hist;  // (unused since round 4: the histogram lives in registers)
  uint32_t hreg;   // this lane's bin of the item's survivors by mismatch count: lane t counts t mismatches (63 = and more)
  uint32_t id, cnt, bound;  // bound: candidates with at least this many mismatches can no longer enter the heap
  uint32_t cur;             // number of the chunk that holds survivor cnt - 1 (wave-uniform)
  uint32_t pos, seed;       // the item's place in the queue and its seed shift (static first chunk)
  bool dense_kind;
  bool over;
  uint32_t gp_, mm_;
  uint32_t n_verified;
  uint32_t b, tail, in_region;  // -b (paired.cpp:161-163) for tail items, whose region the lanes count here
  uint32_t grab_next, grab_end;  // this wavefront's stock of pool chunks (wave-uniform): kPoolGrab per visit to the pool's counter
  static __device__ __forceinline__ uint32_t strand(uint32_t id) { return ((id >> 24) & 7u) >= 3u ? 1u : 0u; }
  __device__ __forceinline__ void begin(uint32_t id_, uint32_t seed_, uint32_t pos_, uint32_t tail_) {
    id = id_;
    seed = seed_;
    pos = pos_;
    tail = tail_;
    in_region = 0;
    cnt = 0;
    bound = 0xFFFFFFFFu;
    cur = 0;
    over = false;
    hreg = 0;
  }
  __device__ __forceinline__ void add(uint32_t, uint32_t gp, uint32_t mm, bool in) { gp_ = gp; mm_ = mm; in_region += in ? 1u : 0u; }
  // the c-th chunk of this probe: a number from the pool, noted in the probe's chunk table (wave-uniform result)
  __device__ __forceinline__ uint32_t take_chunk(uint32_t c) {
    const uint32_t j = id & 0xFFFFFFu, probe = (id >> 24) & 7u;
    uint32_t v = 0;
    const bool fixed = c == 0 && dense_kind && pos < ps.static_n;
    // A chunk from the dynamic part: the pool's counter is ONE address, and the device serves ~10^8 atomics per second
    // on one address -- twelve million chunks a step, one atomic each, were 120 ms of queueing and the whole duration
    // of this kernel (round 3: 2.3 TB/s of records against the single-end verifier's 6.3).  A wavefront now takes
    // kPoolGrab chunks per visit and hands them out itself; what it has left at the end is lost to the pass (the pool
    // is sized with that: kPoolGrab - 1 chunks per wavefront at most).
    if (!fixed && grab_next == grab_end) {  // (uniform)
      uint32_t g = 0;
      if ((threadIdx.x & 63) == 0) g = atomicAdd(ps.pool_next, kPoolGrab);
      grab_next = bcast(g, 0);
      grab_end = grab_next + kPoolGrab;
    }
    if (!fixed) v = 3 * ps.static_n + grab_next++;
    else v = seed * ps.static_n + pos;
    if ((threadIdx.x & 63) == 0) {
      if (c < kPeChunks && v < ps.pool_chunks) ps.chunk[((uint64_t)probe * kPeChunks + c) * ps.ccap + j] = v;
    }
    if (c >= kPeChunks || v >= ps.pool_chunks) { over = true; v = 0; }
    return v;
  }
  __device__ __forceinline__ void step() {
    const uint32_t lane = threadIdx.x & 63;
    n_verified += mm_ != 0xFFFFFFFFu ? 1u : 0u;
    const bool pass = mm_ <= max_mm && mm_ < bound;  // paired.cpp:192-195; bound: see the header comment
    const unsigned long long m = __ballot(pass);
    if (!m || over) return;
    const uint32_t n = (uint32_t)__popcll(m);
    const uint32_t at = cnt + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    const uint32_t c0 = cnt / kPeChunkEnts, c1 = (cnt + n - 1) / kPeChunkEnts;  // at most one chunk boundary in 64 survivors
    if (cnt % kPeChunkEnts == 0) cur = take_chunk(c0);
    const uint32_t nxt = c1 != c0 ? take_chunk(c1) : cur;
    if (pass && !over) ps.pool[(uint64_t)(at / kPeChunkEnts == c0 ? cur : nxt) * kPeChunkEnts + at % kPeChunkEnts] = make_uint2(gp_, mm_);
    cur = nxt;
    // the histogram: one ballot per mismatch count that can pass (0 .. -m), lane t keeps bin t -- no LDS atomics
    // (64 lanes on seven addresses were serialised there, step after step)
    {
      const uint32_t top = max_mm < 63u ? max_mm : 63u;
      for (uint32_t t = 0; t <= top; ++t) {  // (uniform)
        const unsigned long long mt = __ballot(pass && (mm_ < 63u ? mm_ : 63u) == t);
        hreg += lane == t ? (uint32_t)__popcll(mt) : 0u;
      }
    }
    cnt += n;
    if (cnt >= top_k) {
      // smallest t with top_k survivors of at most t mismatches: the heap is then full of candidates that good
      uint32_t total;
      const uint32_t h = hreg;
      const uint32_t before = wave_excl_scan_u32(h, lane, total);
      const unsigned long long ge = __ballot(before + h >= top_k);
      const uint32_t t = ge ? (uint32_t)__ffsll((long long)ge) - 1u : 63u;
      bound = t < 63u ? t : 0xFFFFFFFFu;  // bucket 63 lumps larger counts together: no bound from it
    }
  }
  __device__ __forceinline__ void end() {
    const uint32_t j = id & 0xFFFFFFu, probe = (id >> 24) & 7u;
    const bool skip = tail && wave_sum_u32(in_region) > b;  // (uniform) the narrowed region exceeds -b: the probe pushes nothing
    const uint32_t hv0 = bcast(hreg, 0), hv1 = bcast(hreg, 1);
    if ((threadIdx.x & 63) == 0) {
      ps.surv_n[(uint64_t)probe * ps.ccap + j] = (skip ? 0u : cnt) | 0x80000000u;
      const uint32_t h0 = skip ? 0u : hv0, h1 = skip ? 0u : h0 + hv1;
      ps.cz[(uint64_t)probe * ps.ccap + j] = (h0 < 0xFFFFu ? h0 : 0xFFFFu) | ((h1 < 0xFFFFu ? h1 : 0xFFFFu) << 16);
      if (over) atomicOr(&ps.flag[j], 2u);
    }
  }
};

// the tail items of a stage bracketed before the verifier streams them (map_items.h tail_items_narrow)
template <int NW>
__global__ __launch_bounds__(kBlock) void k_pe_tail_narrow(IndexView iv, uint32_t strand_base, PeStage ps, uint32_t b) {
  if constexpr (NW > 8 && NW <= 10) {
    uint32_t n_big = *ps.q.big_n;
    n_big = n_big < ps.q.big_cap ? n_big : ps.q.big_cap;
    uint32_t n_items = ps.q.ctl[0];
    n_items = n_items < ps.q.cap ? n_items : ps.q.cap;
    tail_items_narrow<NW, SurvivorSink>(iv, strand_base, ps.q, n_items, n_big, b);
  }
}

template <int NW, bool DENSE>
__global__ __launch_bounds__(kBlock, DENSE ? (NW <= 8 ? 6 : 4) : (NW <= 8 ? 4 : (NW <= 10 ? 2 : 1))) void k_pe_verify(
    IndexView iv, uint32_t strand_base, uint32_t max_mm, uint32_t top_k, unsigned long long* __restrict__ stats, PeStage ps,
    uint32_t b) {
  static_assert(item_quads<NW>() <= 64, "an item header is fetched by one wavefront load");
  uint32_t n_big = DENSE ? *ps.q.big_n : 0u;
  n_big = n_big < ps.q.big_cap ? n_big : ps.q.big_cap;
  const uint32_t n_items = ps.q.ctl[DENSE ? 0 : 1] + n_big;
  if (n_items == 0) return;
  __shared__ uint32_t s_start[kLdsChroms + 1];
  __shared__ uint32_t s_hist[kBlock / 64][64];
  __shared__ uint32_t s_edge[DENSE ? kEdgeWords : 1];  // (the dense verifier: core.h edge bitmap)
  const bool fits = iv.n_chrom <= kLdsChroms;  // every chromosome start is in LDS (else every 2^shift-th: ChromTab)
  chrom_tab_stage(s_start, iv.start_index, chrom_tab_of(iv.n_chrom));
  if (DENSE && iv.edge_bits != nullptr)
    for (uint32_t i = threadIdx.x; i < kEdgeWords; i += blockDim.x) s_edge[i] = iv.edge_bits[i];
  __syncthreads();
  const uint32_t* const edge = (DENSE && iv.edge_bits != nullptr) ? s_edge : nullptr;
  SurvivorSink sink;
  sink.ps = ps; sink.max_mm = max_mm; sink.top_k = top_k; sink.hist = s_hist[threadIdx.x >> 6]; sink.n_verified = 0;
  sink.gp_ = 0; sink.mm_ = 0xFFFFFFFFu; sink.dense_kind = DENSE; sink.b = b; sink.tail = 0; sink.in_region = 0;
  sink.grab_next = sink.grab_end = 0;
  if (fits) item_stream<NW, DENSE, true>(iv, strand_base, ps.q, n_items, s_start, sink, n_big, edge);
  else item_stream<NW, DENSE, false>(iv, strand_base, ps.q, n_items, s_start, sink, n_big, edge);
  pe_flush(0u, 0u, sink.n_verified, 0u, stats);
}

#endif  // WALT_SEEDPATTERN == 3

constexpr uint32_t kListHeapSlots = 768;  // HeapEnt slots per wave (6 KB): 15 heaps of top_k = 50, 2 of top_k = 300

// pass 2 / 3: the reads of a list, one per lane, strand-major with the reference's exits.
// LITERAL = false: "complex" reads of pass 1 (key/directory search; a Bloom hit sends the
// read on to the literal list).  LITERAL = true: literal search, handles everything.
template <int NW, bool LITERAL>
__global__ __launch_bounds__(kBlock) void k_pe_topk_list(IndexView iv, const uint32_t* __restrict__ codes2,
                                                          const uint64_t* __restrict__ offsets,
                                                          uint32_t* __restrict__ err, uint32_t strand_base,
                                                          uint32_t max_mm, uint32_t b, uint32_t top_k,
                                                          const uint32_t* __restrict__ mask_table,
                                                          Candidate* __restrict__ ranked, uint32_t* __restrict__ heap_n,
                                                          unsigned long long* __restrict__ stats,
                                                          const uint32_t* __restrict__ list_count,
                                                          const uint32_t* __restrict__ list,
                                                          uint32_t* __restrict__ defer_count,
                                                          uint32_t* __restrict__ defer_list, uint32_t all_reads,
                                                          uint32_t heap_cap = 0, uint32_t* __restrict__ over_count = nullptr,
                                                          uint32_t* __restrict__ over_list = nullptr) {
  __shared__ BlockShared sh;
  __shared__ HeapEnt s_heap[kBlock / 64][kListHeapSlots];  // the heaps of the reads a wave is working on
  if (!all_reads && *list_count == 0) return;  // empty list (the usual state of the overflow list): no prologue
  const uint32_t* si = block_prologue(sh, iv, mask_table, strand_base);
  // all_reads != 0 (seed patterns 5 and 7, which have no pass 1): every read 0 .. all_reads-1
  const uint32_t count = all_reads ? all_reads : *list_count;
  uint32_t n_probe = 0, n_verified = 0, n_big = 0, shortv = 0;
  // A short list is spread thin -- down to ONE read per wavefront: listed reads are the slow ones (large
  // regions verified by the whole wave, long chains of heap updates, literal searches), and 64 of them in
  // one wave run one after another.  rpw = reads per wave so that every wave of the grid has work.
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t waves_per_block = blockDim.x >> 6;
  const uint32_t total_waves = gridDim.x * waves_per_block;
  const uint32_t wave = blockIdx.x * waves_per_block + (threadIdx.x >> 6);
  uint32_t rpw = (count + total_waves - 1) / total_waves;
  // heap slots per read: top_k, or heap_cap when the launch offers small heaps with an overflow list AND the
  // list is too long for full heaps to keep every wave busy anyway (a short list gains nothing from them)
  const uint32_t rpw_full = kListHeapSlots / top_k < 64 ? kListHeapSlots / top_k : 64;
  const uint32_t want_cap = heap_cap & 0x7FFFFFFFu;  // bit 31: use the small heaps whatever the list length (tests)
  const bool small = want_cap && want_cap < top_k && over_list &&
                     ((heap_cap >> 31) || (uint64_t)count > (uint64_t)rpw_full * total_waves);
  const uint32_t cap = small ? want_cap : top_k;
  const uint32_t rpw_max = kListHeapSlots / cap < 64 ? kListHeapSlots / cap : 64;  // top_k <= 300: at least 2
  rpw = rpw < 1 ? 1 : (rpw > rpw_max ? rpw_max : rpw);
  HeapEnt* heap = &s_heap[threadIdx.x >> 6][(lane < rpw ? lane : 0) * cap];
  for (uint64_t base = (uint64_t)wave * rpw; base < count; base += (uint64_t)total_waves * rpw) {
    const uint64_t i = base + lane;
    const bool valid = lane < rpw && i < count;
    const uint32_t r = valid ? (all_reads ? (uint32_t)i : (list[i] & kDeferMask)) : 0;
    uint32_t len;
    pe_process<NW, LITERAL>(iv, sh, si, codes2, offsets, err, r, valid, strand_base, max_mm, b, top_k, heap, ranked,
                            heap_n, defer_count, defer_list, n_probe, n_verified, n_big, len,
                            cap < top_k ? cap : 0xFFFFFFFFu, over_count, over_list);
    if (all_reads) shortv += (valid && len < kMinReadLen) ? 2u : 0u;  // paired.cpp:112-115, once per strand pass
  }
  pe_flush(shortv, n_probe, n_verified, n_big, stats);
}

#if WALT_SEEDPATTERN == 3
// k_pe_push (staged path, see k_pe_stage): the staged reads of a round, one per lane, heaps in LDS like the list
// kernel's.  Most staged reads keep a handful of survivors, so the kernel runs twice: SMALL = true visits every read
// and maps those whose six probes kept at most kPushSmall survivors together -- their heap never holds more, so
// kPushSmall slots do and all 64 lanes of a wavefront work -- and lists the others; SMALL = false maps the listed
// ones with top_k slots each (kListHeapSlots / top_k reads per wavefront).
constexpr uint32_t kPushSmall = 12;
// The heaps of the second launch hold 4-byte entries -- mismatches (bits 0..10), the survivor's place = probe and
// number in the probe's survivor list (bits 11..19 number, 20..22 probe) -- instead of (position, mismatches): the
// comparator looks at the mismatches only (paired.hpp:39-41), the strand follows from the probe, and the position is
// fetched for the entries that are left when the heap is emptied.  Twice the reads per wavefront for the same LDS
// (30 at -k 50): the kernel is bound by instruction issue with a quarter of its lanes at work.
static_assert(kPeChunks * kPeChunkEnts <= 512 && kPeMidRegion <= 512, "a survivor's number takes 9 bits");
#ifndef WALT_PUSH_BIG_SLOTS
#define WALT_PUSH_BIG_SLOTS 1536
#endif
constexpr uint32_t kPushBigSlots = WALT_PUSH_BIG_SLOTS;  // 4-byte heap slots per wavefront of the second launch
// W = uint32_t, MB = 11: the layout above.  W = uint16_t, MB = 4 (round 4): mismatches in bits 0..3, the place in bits
// 4..15 -- for -m up to 15, i.e. every default run: twice the reads per wavefront again (61 of 64 lanes at -k 50).
template <class W, int MB>
struct HeapPacked {
  W* w;
  struct Ref {
    W* p;
    __device__ __forceinline__ operator HeapEnt() const {
      const uint32_t v = *p;
      HeapEnt e;
      e.pos = v >> MB;                                                          // place: probe << 9 | number
      e.mms = (v & ((1u << MB) - 1u)) | ((v >> (MB + 9)) >= 3u ? 0x80000000u : 0u);  // probes 3..5 are the '-' strand's
      return e;
    }
    __device__ __forceinline__ Ref& operator=(const HeapEnt& e) { *p = (W)((e.mms & ((1u << MB) - 1u)) | (e.pos << MB)); return *this; }
    __device__ __forceinline__ Ref& operator=(const Ref& o) { *p = *o.p; return *this; }
  };
  __device__ __forceinline__ Ref operator[](uint32_t i) const { return Ref{w + i}; }
};
template <bool SMALL, int EB = 4>  // EB: bytes of a heap entry of the second launch (4, or 2 when -m <= 15)
__global__ __launch_bounds__(kBlock) void k_pe_push(const uint32_t* __restrict__ list_count, const uint32_t* __restrict__ list,
                                                    PeStage ps, uint32_t top_k, Candidate* __restrict__ ranked,
                                                    uint32_t* __restrict__ heap_n, uint32_t* __restrict__ fb_count,
                                                    uint32_t* __restrict__ fb_list, uint32_t first,
                                                    uint32_t* __restrict__ big_count, uint32_t* __restrict__ big_list) {
  __shared__ HeapEnt s_heap[kBlock / 64][SMALL ? kListHeapSlots : kPushBigSlots / 2];  // (second launch: kPushBigSlots 4-byte entries)
  uint32_t count;
  if (SMALL) {
    count = *list_count;
    count = count > first ? count - first : 0u;
    if (count > ps.ccap) count = ps.ccap;
  } else {
    count = *big_count;
  }
  if (count == 0) return;
  list += first;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t waves_per_block = blockDim.x >> 6;
  const uint32_t total_waves = gridDim.x * waves_per_block;
  const uint32_t wave = blockIdx.x * waves_per_block + (threadIdx.x >> 6);
  const uint32_t cap = SMALL ? kPushSmall : top_k;
  constexpr uint32_t kSlots = SMALL ? kListHeapSlots : kPushBigSlots * (4 / EB);
  constexpr int MB = EB == 2 ? 4 : 11;  // mismatch bits of a packed entry
  using PackedW = typename std::conditional<EB == 2, uint16_t, uint32_t>::type;
  const uint32_t rpw_max = kSlots / cap < 64 ? kSlots / cap : 64;  // top_k <= 300: at least 2
  uint32_t rpw = (count + total_waves - 1) / total_waves;
  rpw = rpw < 1 ? 1 : (rpw > rpw_max ? rpw_max : rpw);
  HeapEnt* heap = &s_heap[threadIdx.x >> 6][(lane < rpw ? lane : 0) * (SMALL ? cap : 0u)];
  HeapPacked<PackedW, MB> heap4;
  heap4.w = reinterpret_cast<PackedW*>(&s_heap[threadIdx.x >> 6][0]) + (lane < rpw ? lane : 0) * cap;
  const uint64_t ccap = ps.ccap;
  for (uint64_t base = (uint64_t)wave * rpw; base < count; base += (uint64_t)total_waves * rpw) {
    const uint64_t i = base + lane;
    const bool in = lane < rpw && i < count;
    const uint64_t j = in ? (SMALL ? i : (uint64_t)big_list[i]) : 0;
    const uint32_t r = in ? (list[j] & kDeferMask) : 0u;
    bool go = in;
    if (SMALL) {
      const uint32_t fl = in ? ps.flag[j] : 1u;
      wave_append(in && !(fl & 1u) && (fl & 2u), r, fb_count, fb_list);  // a probe outgrew its chunks: the list kernel maps the read
      go = in && fl == 0;
      uint32_t total = 0;
#pragma unroll
      for (uint32_t probe = 0; probe < 6; ++probe) total += go ? (ps.surv_n[(uint64_t)probe * ccap + j] & 0x7FFFFFFFu) : 0u;
      const bool big = go && total > kPushSmall && top_k > kPushSmall;
      wave_append(big, (uint32_t)j, big_count, big_list);
      go = go && !big;
    }
    if (!go) continue;
    uint32_t hsize = 0;
    for (uint32_t fi = 0; fi < 2; ++fi) {
      for (uint32_t seed_i = 0; seed_i < kPat; ++seed_i) {
        // paired.cpp:133-149 (top only decreases: per-seed predicates equal the reference's `break`)
        const bool full = hsize >= top_k;
        const uint32_t top_mm = hsize ? (SMALL ? heap_mm(heap[0]) : ((uint32_t)heap4.w[0] & ((1u << MB) - 1u))) : 0xFFFFFFFFu;
        if ((full && top_mm == 0 && seed_i) || (full && top_mm == 1 && seed_i >= kExitOneMismatch)) continue;
        const uint32_t probe = 3 * fi + seed_i;
        const uint32_t sn = ps.surv_n[(uint64_t)probe * ccap + j];
        const uint32_t n = sn & 0x7FFFFFFFu;
        const bool chunked = (sn >> 31) != 0;
        uint32_t ch = 0;
        // eight survivors per round of loads (one load per push made a read with 200 survivors 200 round trips)
        for (uint32_t k0 = 0; k0 < n; k0 += 8) {
          if (chunked && k0 % kPeChunkEnts == 0) ch = ps.chunk[((uint64_t)probe * kPeChunks + k0 / kPeChunkEnts) * ccap + j];
          uint2 c[8];
#pragma unroll
          for (uint32_t t = 0; t < 8; ++t) {
            const uint32_t k = k0 + t < n ? k0 + t : n - 1;  // (a chunk holds 64: eight from k0 stay inside it)
            c[t] = chunked ? ps.pool[(uint64_t)ch * kPeChunkEnts + k % kPeChunkEnts]
                           : ps.inl[((uint64_t)probe * kPeMidRegion + k) * ccap + j];
          }
#pragma unroll
          for (uint32_t t = 0; t < 8; ++t) {
            if (k0 + t < n) {
              HeapEnt e; e.pos = c[t].x; e.mms = c[t].y | (fi << 31);
              if constexpr (SMALL) {
                topk_push(heap, hsize, top_k, e);  // paired.cpp:195
              } else {
                e.pos = (probe << 9) | (k0 + t);
                topk_push(heap4, hsize, top_k, e);
              }
            }
          }
        }
      }
    }
    // paired.cpp:685-692: pop everything; ranked[r][i] = i-th popped (descending mismatch)
    heap_n[r] = hsize;
    Candidate* out = ranked + (uint64_t)r * top_k;
    if constexpr (SMALL) {
      uint32_t i2 = 0;
      while (hsize) {
        const HeapEnt e = heap_pop(heap, hsize);
        Candidate c; c.genome_pos = e.pos; c.strand = (e.mms >> 31) ? '-' : '+'; c.mismatch = heap_mm(e);
        out[i2++] = c;
      }
    } else {
      // emptied in place: every pop moves the top to the end of the shrinking heap (__pop_heap), so afterwards entry
      // n - 1 - i is the i-th popped; then the positions, eight entries per round of loads
      const uint32_t n_ent = hsize;
      while (hsize) heap_pop(heap4, hsize);
      for (uint32_t i0 = 0; i0 < n_ent; i0 += 8) {
        uint32_t v[8], sn[8], ch[8];
        uint2 c[8];
#pragma unroll
        for (uint32_t t = 0; t < 8; ++t) {
          v[t] = heap4.w[n_ent - 1 - (i0 + t < n_ent ? i0 + t : n_ent - 1)];
          sn[t] = ps.surv_n[(uint64_t)(v[t] >> (MB + 9)) * ccap + j];
        }
#pragma unroll
        for (uint32_t t = 0; t < 8; ++t) {
          const uint32_t probe = v[t] >> (MB + 9), k = (v[t] >> MB) & 511u;
          ch[t] = (sn[t] >> 31) ? ps.chunk[((uint64_t)probe * kPeChunks + k / kPeChunkEnts) * ccap + j] : 0u;
        }
#pragma unroll
        for (uint32_t t = 0; t < 8; ++t) {
          const uint32_t probe = v[t] >> (MB + 9), k = (v[t] >> MB) & 511u;
          c[t] = (sn[t] >> 31) ? ps.pool[(uint64_t)ch[t] * kPeChunkEnts + k % kPeChunkEnts]
                               : ps.inl[((uint64_t)probe * kPeMidRegion + k) * ccap + j];
        }
#pragma unroll
        for (uint32_t t = 0; t < 8; ++t) {
          if (i0 + t < n_ent) {
            Candidate cd; cd.genome_pos = c[t].x; cd.strand = (v[t] >> (MB + 9)) >= 3u ? '-' : '+'; cd.mismatch = v[t] & ((1u << MB) - 1u);
            out[i0 + t] = cd;
          }
        }
      }
    }
  }
}
#endif  // WALT_SEEDPATTERN == 3

// Pairs whose candidate lists span more than kLightCombos (i, j) combinations are
// left to k_pe_merge_heavy: one such lane would otherwise hold its whole wave for
// thousands of iterations (repeat families fill both lists to top_k).
constexpr uint32_t kLightCombos = 16;  // (64 until the heavy kernel worked the candidates out once per pair: 202 -> 199 ms)

__global__ void k_pe_merge(IndexView iv, const Candidate* __restrict__ ranked1, const uint32_t* __restrict__ n1,
                           const Candidate* __restrict__ ranked2, const uint32_t* __restrict__ n2,
                           const uint64_t* __restrict__ off1, const uint64_t* __restrict__ off2, uint32_t n,
                           uint32_t top_k, int frag_range, uint32_t max_mm, PairResult* __restrict__ out,
                           uint32_t* __restrict__ heavy_count, uint32_t* __restrict__ heavy_list) {
  // chromosome starts in LDS when they fit: getChromID is a chain of dependent loads per candidate pair
  __shared__ uint32_t s_start[kLdsChroms + 1];
  const bool fits = iv.n_chrom <= kLdsChroms;
  if (fits)
    for (uint32_t i = threadIdx.x; i <= iv.n_chrom; i += blockDim.x) s_start[i] = iv.start_index[i];
  __syncthreads();
  const uint32_t* starts = fits ? s_start : iv.start_index;
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = r < n;
  const uint32_t a = valid ? n1[r] : 0, b = valid ? n2[r] : 0;
  const bool heavy = a * b > kLightCombos;
  if (valid && !heavy) {
    const Candidate* r1 = ranked1 + (uint64_t)r * top_k;
    const Candidate* r2 = ranked2 + (uint64_t)r * top_k;
    const uint32_t len1 = (uint32_t)(off1[r + 1] - off1[r]), len2 = (uint32_t)(off2[r + 1] - off2[r]);
    PairResult pr;
    if (a <= 4 && b <= 4) {
      // the common case (a unique read is found once per seed shift: three candidates): both lists are
      // fetched in one round of independent loads and the 16 combinations run from registers, in the
      // order of pair_merge (i and j descending; its `break` equals `continue`, see k_pe_merge_heavy)
      Candidate A[4], B[4];
      uint32_t cA[4], cB[4], sA[4], eA[4], sB[4], eB[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        A[k] = r1[k < (int)a ? k : 0];
        B[k] = r2[k < (int)b ? k : 0];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        cA[k] = chrom_id(starts, iv.n_chrom, A[k].genome_pos);
        cB[k] = chrom_id(starts, iv.n_chrom, B[k].genome_pos);
        forward_pos(A[k].genome_pos, A[k].strand, cA[k], len1, starts, sA[k], eA[k]);
        forward_pos(B[k].genome_pos, B[k].strand, cB[k], len2, starts, sB[k], eB[k]);
      }
      int bi = -1, bj = -1;
      uint32_t min_mm = max_mm, best_times = 0, best_hi = 0, best_lo = 0;
#pragma unroll
      for (int i = 3; i >= 0; --i) {
#pragma unroll
        for (int j = 3; j >= 0; --j) {
          const uint32_t mm = A[i].mismatch + B[j].mismatch;
          const int frag = A[i].strand == '+' ? (int)(eB[j] - sA[i]) : (int)(eA[i] - sB[j]);
          const bool ok = i < (int)a && j < (int)b && A[i].strand != B[j].strand && mm <= min_mm && cA[i] == cB[j] &&
                          frag > 0 && frag <= frag_range;
          const bool differs = A[i].genome_pos != best_hi || B[j].genome_pos != best_lo;
          if (ok && mm < min_mm) {
            bi = i; bj = j; best_times = 1; min_mm = mm; best_hi = A[i].genome_pos; best_lo = B[j].genome_pos;
          } else if (ok && differs) {  // mm == min_mm
            bi = i; bj = j; best_times++;
          }
        }
      }
      pair_finish(r1, (int)a, r2, (int)b, len1, len2, starts, iv.n_chrom, max_mm, bi, bj, best_times, pr);
    } else {
      pair_merge(r1, (int)a, r2, (int)b, len1, len2, starts, iv.n_chrom, frag_range, max_mm, pr);
    }
    out[r] = pr;
  }
  // heavy pairs -> list: ONE atomic per block (6 % of the pairs of a repeat-rich genome are heavy, i.e. some lane of
  // nearly every wavefront: an atomic per wavefront was 150,000 on one address per launch, most of its duration)
  __shared__ uint32_t s_cnt[kBlock / 64], s_base;
  const unsigned long long hv = __ballot(heavy);
  const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) s_cnt[wv] = (uint32_t)__popcll(hv);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t tot = 0;
    for (uint32_t k = 0; k < blockDim.x / 64; ++k) tot += s_cnt[k];
    s_base = tot ? atomicAdd(heavy_count, tot) : 0u;
  }
  __syncthreads();
  if (heavy) {
    uint32_t before = 0;
    for (uint32_t k = 0; k < wv; ++k) before += s_cnt[k];
    heavy_list[s_base + before + (uint32_t)__popcll(hv & ((1ull << lane) - 1ull))] = r;
  }
}

// One wavefront per heavy pair: 64 (i, j) combinations per step, in the reference's order
// (i descending, j descending inside, paired.cpp:478-513).  The sequential fold
//   mm < min  -> new best, times = 1;   mm == min && key != best_key -> last-wins, times++
// over one step equals: m* = smallest mm among the step's valid combinations; if m* < min the
// FIRST lane holding m* restarts the fold and the later lanes with mm == m* and a different
// key count; if m* == min the lanes with mm == min and a key different from best_key count.
// The reference's ordered `break` (486-487) only skips combinations with mm > min, which never
// change the fold, so evaluating them (and rejecting on mm > min) is equivalent.
__global__ __launch_bounds__(kBlock) void k_pe_merge_heavy(IndexView iv, const Candidate* __restrict__ ranked1,
                                                            const uint32_t* __restrict__ n1,
                                                            const Candidate* __restrict__ ranked2,
                                                            const uint32_t* __restrict__ n2,
                                                            const uint64_t* __restrict__ off1,
                                                            const uint64_t* __restrict__ off2, uint32_t top_k,
                                                            int frag_range, uint32_t max_mm,
                                                            PairResult* __restrict__ out,
                                                            const uint32_t* __restrict__ heavy_count,
                                                            const uint32_t* __restrict__ heavy_list) {
  __shared__ uint32_t s_start[kLdsChroms + 1];
  // per wavefront: the pair's two candidate lists with what every combination needs of a candidate -- position,
  // mismatches | strand, chromosome, forward start -- worked out ONCE per candidate (the chromosome search per
  // combination was 2 x 2,500 searches for a pair with two full lists of 50)
  extern __shared__ uint4 s_cand[];  // [waves][2][top_k]
  const bool fits = iv.n_chrom <= kLdsChroms;
  if (fits)
    for (uint32_t i = threadIdx.x; i <= iv.n_chrom; i += blockDim.x) s_start[i] = iv.start_index[i];
  __syncthreads();
  const uint32_t* starts = fits ? s_start : iv.start_index;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t waves_per_block = blockDim.x >> 6;
  const uint32_t count = *heavy_count;
  uint4* const ca = s_cand + (uint64_t)(threadIdx.x >> 6) * 2 * top_k;
  uint4* const cb = ca + top_k;
  for (uint32_t h = blockIdx.x * waves_per_block + (threadIdx.x >> 6); h < count; h += gridDim.x * waves_per_block) {
    const uint32_t r = heavy_list[h];
    const Candidate* r1 = ranked1 + (uint64_t)r * top_k;
    const Candidate* r2 = ranked2 + (uint64_t)r * top_k;
    const uint32_t na = n1[r], nb = n2[r];
    const uint32_t len1 = (uint32_t)(off1[r + 1] - off1[r]), len2 = (uint32_t)(off2[r + 1] - off2[r]);
    const uint32_t total = na * nb;
    __builtin_amdgcn_wave_barrier();  // (the previous pair's combinations have been read)
    for (uint32_t t = lane; t < na + nb; t += 64) {
      const bool first = t < na;
      const Candidate c = first ? r1[t] : r2[t - na];
      const uint32_t chr = chrom_id(starts, iv.n_chrom, c.genome_pos);
      uint32_t sf, ef;
      forward_pos(c.genome_pos, c.strand, chr, first ? len1 : len2, starts, sf, ef);
      (first ? ca : cb)[first ? t : t - na] = make_uint4(c.genome_pos, c.mismatch | ((c.strand & 0xFFu) == '-' ? 0x80000000u : 0u), chr, sf);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint32_t min_mm = max_mm, best_times = 0;
    uint32_t best_hi = 0, best_lo = 0;  // best_pos = (pos1 << 32) + pos2
    int bi = -1, bj = -1;
    // combination c = base + lane is (i, j) = (na - 1 - c / nb, nb - 1 - c % nb): quotient and remainder are carried from
    // step to step (one division per pair instead of one per step and lane).  Both lists are in pop order -- mismatches
    // descending with the index -- so the combinations of a step and of every later one hold at least
    // mismatches(first lane's i) + mismatches(best of list 2): once that exceeds min_mm nothing that follows can change
    // the fold (the reference's `break`, paired.cpp:486-487, leaves the same combinations out row by row).
    const uint32_t q_step = 64u / nb, r_step = 64u % nb;
    uint32_t cq = lane / nb, cr = lane % nb;
    const uint32_t best_b = cb[nb - 1].y & 0x7FFFFFFFu;
    for (uint32_t base = 0; base < total; base += 64) {
      const uint32_t c = base + lane;
      const uint32_t q0 = shfl_pin(cq, 0u);  // (uniform) base / nb < na
      if ((ca[na - 1 - q0].y & 0x7FFFFFFFu) + best_b > min_mm) break;
      bool ok = false;
      uint32_t mm = 0xFFFFFFFFu, p1 = 0, p2 = 0;
      int i = 0, j = 0;
      const uint32_t my_q = cq, my_r = cr;
      cq += q_step; cr += r_step;
      if (cr >= nb) { cr -= nb; ++cq; }
      if (c < total) {
        i = (int)(na - 1 - my_q);
        j = (int)(nb - 1 - my_r);
        const uint4 A = ca[i], B = cb[j];
        p1 = A.x; p2 = B.x;
        if ((A.y ^ B.y) >> 31) {  // opposite strands
          mm = (A.y & 0x7FFFFFFFu) + (B.y & 0x7FFFFFFFu);
          if (mm <= min_mm && A.z == B.z) {
            const int frag = (A.y >> 31) ? (int)(A.w + len1 - B.w) : (int)(B.w + len2 - A.w);
            ok = frag > 0 && frag <= frag_range;
          }
        }
      }
      const uint32_t m_star = wave_min_u32(ok ? mm : 0xFFFFFFFFu);
      if (m_star == 0xFFFFFFFFu || m_star > min_mm) continue;
      unsigned long long cnt_mask;
      if (m_star < min_mm) {
        const unsigned long long at_min = __ballot(ok && mm == m_star);
        const int f = __ffsll((long long)at_min) - 1;
        min_mm = m_star;
        best_hi = bcast(p1, f);
        best_lo = bcast(p2, f);
        bi = (int)bcast((uint32_t)i, f);
        bj = (int)bcast((uint32_t)j, f);
        cnt_mask = __ballot(ok && mm == m_star && (int)lane > f && (p1 != best_hi || p2 != best_lo));
        best_times = 1 + (uint32_t)__popcll(cnt_mask);
      } else {
        cnt_mask = __ballot(ok && mm == min_mm && (p1 != best_hi || p2 != best_lo));
        best_times += (uint32_t)__popcll(cnt_mask);
      }
      if (cnt_mask) {
        const int last = 63 - __clzll((long long)cnt_mask);
        bi = (int)bcast((uint32_t)i, last);
        bj = (int)bcast((uint32_t)j, last);
      }
    }
    // The tail of the merge (core.h pair_finish, paired.cpp:515-545) from the lists in LDS: lane 0 used to run it over
    // the lists in device memory -- GetBestMatch4Single's fold is a chain of up to top_k dependent loads per mate, and a
    // pair of a repeat family (the heavy pairs) is ambiguous more often than not: most of the kernel's duration.  The two
    // mates' folds now run side by side on lanes 0 and 1 over the LDS copies.
    {
      BestMatch mine;
      mine.genome_pos = 0; mine.times = 0; mine.strand = '+'; mine.mismatch = max_mm;
      if (best_times != 1 && lane < 2) {  // paired.cpp:296-318, the fold of core.h best4single entry for entry
        const uint4* const cl = lane ? cb : ca;
        for (int k = (int)(lane ? nb : na) - 1; k >= 0; --k) {
          const uint4 e = cl[k];
          const uint32_t emm = e.y & 0x7FFFFFFFu;
          const char es = (e.y >> 31) ? '-' : '+';
          if (emm < mine.mismatch) {
            mine.genome_pos = e.x; mine.times = 1; mine.strand = es; mine.mismatch = emm;
          } else if (emm == mine.mismatch) {
            if (mine.genome_pos == e.x) continue;
            mine.genome_pos = e.x; mine.strand = es; mine.times++;
          } else {
            break;
          }
        }
      }
      const uint32_t o_pos = shfl_pin(mine.genome_pos, 1u), o_times = shfl_pin(mine.times, 1u);
      const uint32_t o_strand = shfl_pin((uint32_t)(uint8_t)mine.strand, 1u), o_mm = shfl_pin(mine.mismatch, 1u);
      if (lane == 0) {
        PairResult pr;
        pr.m1 = mine;
        pr.m2.genome_pos = o_pos; pr.m2.times = o_times; pr.m2.strand = (char)o_strand; pr.m2.mismatch = o_mm;
        pr.best_times = best_times; pr.frag_len = 0; pr.best_i = -1; pr.best_j = -1; pr.pair_mm = 0;
        pr.pad_[0] = pr.pad_[1] = pr.pad_[2] = 0;
        if (best_times == 1) {
          const uint4 A = ca[bi], B = cb[bj];
          pr.best_i = bi; pr.best_j = bj;
          pr.frag_len = (A.y >> 31) ? (int)(A.w + len1 - B.w) : (int)(B.w + len2 - A.w);  // core.h pair_len
          pr.pair_mm = (A.y & 0x7FFFFFFFu) + (B.y & 0x7FFFFFFFu);
          pr.m1.genome_pos = A.x; pr.m1.times = 1; pr.m1.strand = (A.y >> 31) ? '-' : '+'; pr.m1.mismatch = A.y & 0x7FFFFFFFu;
          pr.m2.genome_pos = B.x; pr.m2.times = 1; pr.m2.strand = (B.y >> 31) ? '-' : '+'; pr.m2.mismatch = B.y & 0x7FFFFFFFu;
        }
        out[r] = pr;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static inline uint64_t pe_align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }
static inline uint64_t align_up(uint64_t v, uint64_t a) { return pe_align_up(v, a); }

constexpr uint32_t kPeInlineHost = 16, kPeChunksHost = 8;  // = kPeMidRegion, kPeChunks (defined with the pattern-3 kernels)
struct PeWorkspace {
  uint32_t* err;
  unsigned long long* shards[2];
  uint32_t* heap_n[2];
  uint32_t* defer_list[2];
  uint32_t* codes2[2];
  Candidate* ranked[2];
  uint64_t stride;
  uint64_t total_bytes;
  uint32_t cap_reads;  // reads per mate the pass's codes2 / ranked arrays have room for
  // staged path (k_pe_stage): per mate, the fallback list and the survivors / counts / flags / items of ccap reads
  uint32_t* fb_list[2];
  uint32_t* big_list[2];
  uint2* inl[2];
  uint32_t* surv_n[2];
  uint32_t* cz[2];
  uint32_t* chunk_tab[2];
  uint2* pool[2];
  uint32_t* sflag[2];
  uint4* items[2];
  uint4* bigs[2];
  uint32_t ccap, pool_chunks, rounds;
};
// staged reads per round, pass and mate: a sixteenth of the pass (complex reads and filter hits are a fifth of the
// reads of an hg19-like genome: four rounds; the state of a staged read is 2.2 KB and the paired-end index leaves
// little room -- with rounds of an eighth the survivor pool had to shrink and a tenth of the staged reads fell
// back to the list kernel: 359 ms against 315), all of it when the pass is small
static PeGeometry pe_geometry(uint32_t n, uint32_t top_k, const walt_options& opt, bool roomy) {
  PeGeometry g;
  g.roomy = roomy;
  const uint64_t budget = roomy ? 12ull << 30 : 10ull << 30;
  uint64_t c = budget / ((uint64_t)(top_k ? top_k : 1) * 2 * sizeof(Candidate));
  const uint64_t cmax = roomy ? 10000000ull : kPeChunkMax;
  if (c > cmax) c = cmax;
  if (c < (1u << 16)) c = 1u << 16;
  if (opt.pe_chunk > 0) c = (uint64_t)opt.pe_chunk;  // test hook: several passes on a small batch
  g.chunk = n < c ? n : (uint32_t)c;
  const long long rv = opt.pe_rounds > 0 ? opt.pe_rounds : (roomy ? 1 : 4);
  g.rounds = (uint32_t)(rv == 1 || rv == 2 ? rv : 4);
  if (opt.pe_stage_cap > 0) {  // test hook: several rounds and the list-kernel fallback on a small batch
    g.ccap = (uint32_t)pe_align_up((uint64_t)opt.pe_stage_cap, 64);
  } else if (g.chunk <= 65536) {
    g.ccap = g.chunk ? g.chunk : 1;
  } else {
    const uint32_t cc = g.chunk / (4 * g.rounds);  // the rounds together hold a quarter of the pass
    g.ccap = (uint32_t)pe_align_up(cc > 65536 ? cc : 65536, 64);
  }
  // 128 survivors per staged read on average (192 when the device is roomy); three quarters static, the dynamic
  // quarter handed out kPoolGrab chunks at a time
  const uint32_t pool = roomy ? g.ccap * 3u : g.ccap * 2u + g.ccap / 2u;
  g.pool_chunks = pool > 8192 ? pool : 8192;
  return g;
}

static PeWorkspace carve_pe(void* base, const PeGeometry& geo, int nw, uint32_t top_k, uint32_t max_read_len) {
  PeWorkspace w;
  const uint32_t chunk = geo.chunk;
  uint8_t* p = reinterpret_cast<uint8_t*>(base);
  uint64_t off = 0;
  auto take = [&](uint64_t bytes) {
    uint8_t* q = p ? p + off : nullptr;
    off += align_up(bytes, 256);
    return q;
  };
  w.stride = align_up(chunk ? chunk : 1, 64);
  w.cap_reads = chunk;
  // [0..1] pack errors, [64 + 32 m ...] deferral control of mate m, [128] heavy-pair count of the merge
  w.err = reinterpret_cast<uint32_t*>(take(192 * sizeof(uint32_t)));
  for (int m = 0; m < 2; ++m) w.shards[m] = reinterpret_cast<unsigned long long*>(take(kStatShardBytes));
  for (int m = 0; m < 2; ++m) w.heap_n[m] = reinterpret_cast<uint32_t*>(take((uint64_t)chunk * 4 + 64));
  for (int m = 0; m < 2; ++m) w.defer_list[m] = reinterpret_cast<uint32_t*>(take(3 * w.stride * 4 + 64));  // literal list, its sorted copy, complex list
  for (int m = 0; m < 2; ++m) w.codes2[m] = reinterpret_cast<uint32_t*>(take(codes2_words((uint64_t)chunk * max_read_len) * 4 + 64));
  for (int m = 0; m < 2; ++m) w.ranked[m] = reinterpret_cast<Candidate*>(take((uint64_t)chunk * top_k * sizeof(Candidate) + 64));
  w.ccap = geo.ccap;
  w.pool_chunks = geo.pool_chunks;
  w.rounds = geo.rounds;
  const uint64_t quads = 2 + (2 * (uint64_t)nw + 3) / 4;  // item_quads<NW>()
  for (int m = 0; m < 2; ++m) {
    w.fb_list[m] = reinterpret_cast<uint32_t*>(take(w.stride * 4 + 64));
    w.big_list[m] = reinterpret_cast<uint32_t*>(take((uint64_t)w.ccap * 4 + 64));
    w.inl[m] = reinterpret_cast<uint2*>(take((uint64_t)6 * kPeInlineHost * w.ccap * 8));
    w.surv_n[m] = reinterpret_cast<uint32_t*>(take((uint64_t)6 * w.ccap * 4));
    w.cz[m] = reinterpret_cast<uint32_t*>(take((uint64_t)6 * w.ccap * 4));
    w.chunk_tab[m] = reinterpret_cast<uint32_t*>(take((uint64_t)6 * kPeChunksHost * w.ccap * 4));
    w.pool[m] = reinterpret_cast<uint2*>(take((uint64_t)w.pool_chunks * 64 * 8));
    w.sflag[m] = reinterpret_cast<uint32_t*>(take((uint64_t)w.ccap * 4));
    w.items[m] = reinterpret_cast<uint4*>(take((uint64_t)2 * w.ccap * quads * 16));  // the queue is emptied after every seed: two probes per read
    w.bigs[m] = reinterpret_cast<uint4*>(take((uint64_t)(w.ccap / 8 + 64) * quads * 16));  // its largest-first array
  }
  w.total_bytes = off;
  return w;
}

template <int NW>
static int launch_pe_topk(const walt_index* idx, const IndexView& view, const uint32_t* codes2, const uint64_t* offsets, uint32_t* err,
                          uint64_t stride, uint32_t n, uint32_t sb,
                          uint32_t max_mm, uint32_t b, uint32_t top_k, uint32_t* heap_n,
                          Candidate* ranked, unsigned long long* stats, uint32_t* ctl, uint32_t* defer_list,
                          const PeWorkspace& w, int mate, hipStream_t stream) {
  // ctl: [0] literal-list count, [3] 1 when the literal round took its seed shifts in one launch, [8..23] its bins
  // (launch_bin_deferred), [24] complex-list count
  uint32_t* lit_count = ctl;
  uint32_t* cplx_count = ctl + 24;
  uint32_t* lit_list = defer_list;
  uint32_t* lit_sorted = defer_list + stride;
  uint32_t* cplx_list = defer_list + 2 * stride;
#if WALT_SEEDPATTERN != 3
  (void)lit_sorted; (void)cplx_list; (void)cplx_count; (void)w; (void)mate;
  // patterns 5 / 7: the strand-major list kernel over every read with the directory/key search, Bloom hits
  // deferred to the literal list
  hipLaunchKernelGGL((k_pe_topk_list<NW, false>), dim3(6u * (unsigned)idx->n_cu), dim3(kBlock), 0, stream, view, codes2, offsets, err, sb,
                     max_mm, b, top_k, idx->d_mask_table, ranked, heap_n, stats, nullptr, nullptr, lit_count, lit_list, n);
  hipLaunchKernelGGL((k_pe_topk_list<NW, true>), dim3(6u * (unsigned)idx->n_cu), dim3(kBlock), 0, stream, view, codes2, offsets, err, sb,
                     max_mm, b, top_k, idx->d_mask_table, ranked, heap_n, stats, lit_count, lit_list, nullptr, nullptr, 0u);
  return WALT_OK;
#else
  const unsigned pg = persistent_grid(idx);
  const unsigned g1 = grid_for(n) < pg ? grid_for(n) : pg;
  const unsigned g2 = 6u * (unsigned)idx->n_cu;  // x 4 waves: the list kernels size their per-wave share from the list length
  // option pe_mode = 1: complex reads and filter hits mapped by the list kernels only (the path before the staged one)
  const bool list_only = idx->opt.pe_mode == 1;
  uint32_t* over_count = ctl + 25;  // zeroed with the control block at the start of the pass
  if (!list_only) {
    static_assert(kPeMidRegion == kPeInlineHost && kPeChunks == kPeChunksHost && kPeChunkEnts == 64, "carve_pe sizes the survivor storage");
    // ctl: [26] fallback-list count, [27] pool chunks handed out, [28..31] the item queue's counters
    PeStage ps;
    ps.inl = w.inl[mate]; ps.surv_n = w.surv_n[mate]; ps.cz = w.cz[mate]; ps.chunk = w.chunk_tab[mate]; ps.pool = w.pool[mate];
    ps.pool_next = ctl + 27; ps.pool_chunks = w.pool_chunks; ps.static_n = w.pool_chunks / 4; ps.flag = w.sflag[mate];
    ps.q.items = w.items[mate]; ps.q.ctl = ctl + 28; ps.q.cap = 2 * w.ccap; ps.q.ovf = nullptr;
    ps.q.bigs = w.bigs[mate]; ps.q.big_n = ctl + 2; ps.q.big_cap = w.ccap / 8 + 64;
    ps.ccap = w.ccap;
    ps.lit_fuse = idx->opt.pe_lit_fuse != 0 ? 1u : 0u;
    ps.fused_note = ctl + 3;
    // option pe_defer_min = 0: never (A/B); = n: ranges of more than n slots (n >= the in-lane limit)
    const long long dm = idx->opt.pe_defer_min;
    ps.defer_min = dm < 0 ? (uint32_t)kSmallRegion : dm == 0 ? 0xFFFFFFFFu : (uint32_t)(dm < (long long)kSmallRegion ? (long long)kSmallRegion : dm);
    uint32_t* fb_count = ctl + 26;
    uint32_t* fb_list = w.fb_list[mate];
    static const int vb_dense = [] {
      int nb = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_pe_verify<NW, NW <= 10>, kBlock, 0) != hipSuccess || nb < 1) nb = 4;
      return nb;
    }();
    static const int vb_gather = [] {
      int nb = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_pe_verify<NW, false>, kBlock, 0) != hipSuccess || nb < 1) nb = 2;
      return nb;
    }();
    const unsigned vg_dense = (unsigned)vb_dense * (unsigned)idx->n_cu, vg_gather = (unsigned)vb_gather * (unsigned)idx->n_cu;
    const unsigned gs = grid_for(w.ccap) < pg ? grid_for(w.ccap) : pg;
    auto verify = [&]() {
      if constexpr (NW > 8 && NW <= 10)
        hipLaunchKernelGGL((k_pe_tail_narrow<NW>), dim3(4u * (unsigned)idx->n_cu), dim3(kBlock), 0, stream, view, sb, ps, b);
      if constexpr (NW <= 10)
        hipLaunchKernelGGL((k_pe_verify<NW, true>), dim3(vg_dense), dim3(kBlock), 0, stream, view, sb, max_mm, top_k, stats, ps, b);
      hipLaunchKernelGGL((k_pe_verify<NW, false>), dim3(vg_gather), dim3(kBlock), 0, stream, view, sb, max_mm, top_k, stats, ps, b);
    };
    // the staged state holds ccap reads: a list is taken in rounds of ccap (the list's length is on the device: a
    // fixed number of rounds, the empty ones cost a few launches), the last round hands the rest to the list kernel
    const uint32_t kRounds = w.rounds;  // together a quarter of the pass
    uint32_t* big_count = ctl + 1;  // reads of the round with more than kPushSmall survivors
    uint32_t* big_list = w.big_list[mate];
    auto clear_round = [&]() {  // pool + queue, the push kernels' list
      return hipMemsetAsync(ctl + 27, 0, 5 * sizeof(uint32_t), stream) == hipSuccess &&
             hipMemsetAsync(big_count, 0, 2 * sizeof(uint32_t), stream) == hipSuccess;  // ctl[1], and ctl[2]: the queue's largest-first count
    };
    auto push = [&](const uint32_t* count, const uint32_t* list, uint32_t* over_n, uint32_t* over_l, uint32_t first) {
      hipLaunchKernelGGL((k_pe_push<true, 4>), dim3(g2), dim3(kBlock), 0, stream, count, list, ps, top_k, ranked, heap_n, over_n, over_l,
                         first, big_count, big_list);
      if (max_mm <= 15 && idx->opt.pe_push_wide == 0)  // 2-byte heap entries: 4 mismatch bits
        hipLaunchKernelGGL((k_pe_push<false, 2>), dim3(g2), dim3(kBlock), 0, stream, count, list, ps, top_k, ranked, heap_n, over_n, over_l,
                           first, big_count, big_list);
      else
        hipLaunchKernelGGL((k_pe_push<false, 4>), dim3(g2), dim3(kBlock), 0, stream, count, list, ps, top_k, ranked, heap_n, over_n, over_l,
                           first, big_count, big_list);
    };
    auto clear_queue = [&]() {
      return hipMemsetAsync(ctl + 28, 0, 4 * sizeof(uint32_t), stream) == hipSuccess &&
             hipMemsetAsync(ctl + 2, 0, sizeof(uint32_t), stream) == hipSuccess;
    };
    // pass 1 with ONE list: filter hits (tagged) and complex reads both go to the staged kernels
    hipLaunchKernelGGL(k_pe_topk_dual<NW>, dim3(g1), dim3(kBlock), 0, stream, view, codes2, offsets, err, n, sb,
                       max_mm, b, top_k, idx->d_mask_table, ranked, heap_n, stats, cplx_count, cplx_list, cplx_count,
                       cplx_list);
    for (uint32_t rd = 0; rd < kRounds; ++rd) {
      if (rd && !clear_round()) return fail(WALT_EHIP, "hipMemsetAsync failed");
      for (uint32_t seed = 0; seed < 3; ++seed) {  // seed by seed: what a seed found decides which probes the next one makes
        if (seed && !clear_queue()) return fail(WALT_EHIP, "hipMemsetAsync failed");
        hipLaunchKernelGGL((k_pe_stage<NW, false>), dim3(gs), dim3(kBlock), 0, stream, view, codes2, offsets, err, sb, max_mm, b,
                           idx->d_mask_table, stats, cplx_count, cplx_list, ps, lit_count, lit_list, fb_count, fb_list,
                           rd * w.ccap, rd + 1 == kRounds ? 1u : 0u, seed, top_k);
        verify();
      }
      push(cplx_count, cplx_list, fb_count, fb_list, rd * w.ccap);
    }
    // what the staged round could not hold: the list kernel, as before (it may add to the literal list)
    hipLaunchKernelGGL((k_pe_topk_list<NW, false>), dim3(g2), dim3(kBlock), 0, stream, view, codes2, offsets, err, sb,
                       max_mm, b, top_k, idx->d_mask_table, ranked, heap_n, stats, fb_count, fb_list, lit_count,
                       lit_list, 0u);
    // second round: the reads with a truly dangerous probe, sorted by the iteration of that probe; the complex
    // list's area is free again and takes what this round cannot hold
    launch_bin_deferred(lit_count, lit_list, lit_sorted, stream);
    if (!clear_round()) return fail(WALT_EHIP, "hipMemsetAsync failed");
    for (uint32_t seed = 0; seed < 3; ++seed) {
      if (seed && !clear_queue()) return fail(WALT_EHIP, "hipMemsetAsync failed");
      hipLaunchKernelGGL((k_pe_stage<NW, true>), dim3(gs), dim3(kBlock), 0, stream, view, codes2, offsets, err, sb, max_mm, b,
                         idx->d_mask_table, stats, lit_count, lit_sorted, ps, nullptr, nullptr, over_count, cplx_list, 0u, 1u, seed,
                         top_k);
      verify();
    }
    push(lit_count, lit_sorted, over_count, cplx_list, 0u);
    hipLaunchKernelGGL((k_pe_topk_list<NW, true>), dim3(g2), dim3(kBlock), 0, stream, view, codes2, offsets, err, sb,
                       max_mm, b, top_k, idx->d_mask_table, ranked, heap_n, stats, over_count, cplx_list, nullptr, nullptr, 0u);
    return WALT_OK;
  }
  hipLaunchKernelGGL(k_pe_topk_dual<NW>, dim3(g1), dim3(kBlock), 0, stream, view, codes2, offsets, err, n, sb,
                     max_mm, b, top_k, idx->d_mask_table, ranked, heap_n, stats, lit_count, lit_list, cplx_count,
                     cplx_list);
  hipLaunchKernelGGL((k_pe_topk_list<NW, false>), dim3(g2), dim3(kBlock), 0, stream, view, codes2, offsets, err, sb,
                     max_mm, b, top_k, idx->d_mask_table, ranked, heap_n, stats, cplx_count, cplx_list, lit_count,
                     lit_list, 0u);
  launch_bin_deferred(lit_count, lit_list, lit_sorted, stream);
  // literal list: 8-slot heaps, so that 64 reads share a wavefront instead of 15 at -k 50 (with thousands of
  // contigs a tenth of the reads is here); the few reads with more candidates overflow into the complex list's
  // area, which is free again, and are mapped with full heaps
  const uint32_t kSmallHeap = 8u | (idx->opt.pe_small_heaps ? 0x80000000u : 0u);  // option: force (tests)
  hipLaunchKernelGGL((k_pe_topk_list<NW, true>), dim3(g2), dim3(kBlock), 0, stream, view, codes2, offsets, err, sb,
                     max_mm, b, top_k, idx->d_mask_table, ranked, heap_n, stats, lit_count, lit_sorted, nullptr, nullptr, 0u,
                     kSmallHeap, over_count, cplx_list);
  hipLaunchKernelGGL((k_pe_topk_list<NW, true>), dim3(g2), dim3(kBlock), 0, stream, view, codes2, offsets, err, sb,
                     max_mm, b, top_k, idx->d_mask_table, ranked, heap_n, stats, over_count, cplx_list, nullptr, nullptr, 0u);
  return WALT_OK;
#endif
}

static int pe_streams(walt_index* idx) {
  if (idx->pe_start) return WALT_OK;
  for (int k = 0; k < 2; ++k) {
    for (int j = 0; j < 2; ++j) WALT_HIP(hipStreamCreateWithFlags(&idx->pe_stream[k][j], hipStreamNonBlocking));
    WALT_HIP(hipEventCreateWithFlags(&idx->pe_fork[k], hipEventDisableTiming));
    WALT_HIP(hipEventCreateWithFlags(&idx->pe_join[k], hipEventDisableTiming));
    WALT_HIP(hipEventCreateWithFlags(&idx->pe_done[k], hipEventDisableTiming));
  }
  WALT_HIP(hipEventCreateWithFlags(&idx->pe_start, hipEventDisableTiming));
  return WALT_OK;
}

// one pass (n <= pass capacity of the workspace) in pipeline slot `slot`; `stream` carries mate 1 and the merge
static int pe_chunk(walt_index* idx, const uint8_t* d_bases1, const uint64_t* d_off1, const uint8_t* d_bases2,
                    const uint64_t* d_off2, uint32_t n, int nw, uint32_t max_read_len, uint32_t max_mm, uint32_t b,
                    uint32_t top_k,
                    int frag_range, PairResult* d_out, unsigned long long* d_stats, const PeWorkspace& w,
                    uint32_t* pack_err, int slot, hipStream_t stream) {
  const uint8_t* bases[2] = {d_bases1, d_bases2};
  const uint64_t* offs[2] = {d_off1, d_off2};
  // err words of the slot's workspace: [64 + 32 m ..] deferral control block of mate m, [128] heavy-pair count
  // (per pass); pack_err[0..1] = invalid-read counters of the whole call
  WALT_HIP(hipMemsetAsync(w.err + 64, 0, 128 * sizeof(uint32_t), stream));
  // The two mates are independent until the merge; mate 2 runs on a second stream so that its
  // throughput-bound pass 1 overlaps mate 1's list kernels (a few slow reads, mostly idle CUs) and vice versa.
  // option pe_serial = 1 (profiling): both mates and every pass on one stream, so that a kernel's duration is its own
  const bool serial = idx->opt.pe_serial != 0;
  hipStream_t stream_b = serial ? stream : idx->pe_stream[slot][1];
  WALT_HIP(hipEventRecord(idx->pe_fork[slot], stream));
  WALT_HIP(hipStreamWaitEvent(stream_b, idx->pe_fork[slot], 0));
  hipStream_t user_stream = stream;
  for (int m = 0; m < 2; ++m) {
    stream = m ? stream_b : user_stream;
    // mate 1: C->T on _CT00/_CT01; mate 2: G->A on _GA10/_GA11 (paired.cpp:643,589-593)
    unsigned long long* st = w.shards[m];
    const uint32_t sb = m ? 2u : 0u;
    uint32_t* ctl = w.err + 64 + 32 * m;
    IndexView view = idx->view;  // this launch's copy: the limits lane_load_read enforces (the pass's share of the workspace)
    view.batch_max_len = max_read_len;
    view.batch_cap_bytes = (uint64_t)w.cap_reads * max_read_len;
    launch_ascii_to_2bit(bases[m], offs[m], n, w.codes2[m], view.batch_cap_bytes, pack_err, stream);
    int rc;
    switch (nw) {
      case 7: rc = launch_pe_topk<7>(idx, view, w.codes2[m], offs[m], pack_err, w.stride, n, sb, max_mm, b, top_k, w.heap_n[m], w.ranked[m], st, ctl, w.defer_list[m], w, m, stream); break;
      case 8: rc = launch_pe_topk<8>(idx, view, w.codes2[m], offs[m], pack_err, w.stride, n, sb, max_mm, b, top_k, w.heap_n[m], w.ranked[m], st, ctl, w.defer_list[m], w, m, stream); break;
#if WALT_SEEDPATTERN == 3  // patterns 5 / 7 stop at kMaxReadLen = 148 / 152 bases
      case 10: rc = launch_pe_topk<10>(idx, view, w.codes2[m], offs[m], pack_err, w.stride, n, sb, max_mm, b, top_k, w.heap_n[m], w.ranked[m], st, ctl, w.defer_list[m], w, m, stream); break;
      case 16: rc = launch_pe_topk<16>(idx, view, w.codes2[m], offs[m], pack_err, w.stride, n, sb, max_mm, b, top_k, w.heap_n[m], w.ranked[m], st, ctl, w.defer_list[m], w, m, stream); break;
      case 32: rc = launch_pe_topk<32>(idx, view, w.codes2[m], offs[m], pack_err, w.stride, n, sb, max_mm, b, top_k, w.heap_n[m], w.ranked[m], st, ctl, w.defer_list[m], w, m, stream); break;
      default: rc = launch_pe_topk<64>(idx, view, w.codes2[m], offs[m], pack_err, w.stride, n, sb, max_mm, b, top_k, w.heap_n[m], w.ranked[m], st, ctl, w.defer_list[m], w, m, stream); break;
#else
      default: rc = launch_pe_topk<10>(idx, view, w.codes2[m], offs[m], pack_err, w.stride, n, sb, max_mm, b, top_k, w.heap_n[m], w.ranked[m], st, ctl, w.defer_list[m], w, m, stream); break;
#endif
    }
    if (rc) return rc;
    launch_reduce_stats(w.shards[m], d_stats + 4 * m, stream);
  }
  stream = user_stream;
  WALT_HIP(hipEventRecord(idx->pe_join[slot], stream_b));
  WALT_HIP(hipStreamWaitEvent(stream, idx->pe_join[slot], 0));
  // both mates are mapped: mate 1's deferral list area is free and holds the heavy-pair list of the merge
  uint32_t* heavy_count = w.err + 128;
  uint32_t* heavy_list = w.defer_list[0];
  hipLaunchKernelGGL(k_pe_merge, dim3(grid_for(n)), dim3(kBlock), 0, stream, idx->view, w.ranked[0], w.heap_n[0],
                     w.ranked[1], w.heap_n[1], d_off1, d_off2, n, top_k, frag_range, max_mm, d_out, heavy_count,
                     heavy_list);
  hipLaunchKernelGGL(k_pe_merge_heavy, dim3(grid_for(n) < persistent_grid(idx) ? grid_for(n) : persistent_grid(idx)), dim3(kBlock),
                     (size_t)(kBlock / 64) * 2 * top_k * sizeof(uint4), stream,
                     idx->view, w.ranked[0], w.heap_n[0], w.ranked[1], w.heap_n[1], d_off1, d_off2, top_k, frag_range,
                     max_mm, d_out, heavy_count, heavy_list);
  WALT_HIP(hipGetLastError());
  return WALT_OK;
}

static int pe_check_args(walt_index* idx, uint32_t top_k, uint32_t max_read_len, int* nw) {
  if (!idx) return fail(WALT_EINVAL, "null index");
  if ((idx->strand_mask & WALT_STRANDS_ALL) != WALT_STRANDS_ALL)
    return fail(WALT_EINVAL, "paired-end mapping needs all four strand indexes resident");
  if (top_k < 2 || top_k > 300) return fail(WALT_EINVAL, "paired-end candidates must be in [2, 300]");  // walt.cpp:245-246
  *nw = nw_for_len(max_read_len);
  if (!*nw) return fail(WALT_EINVAL, "read length above 1024 is not supported");
  if (max_read_len > kMaxReadLen)
    return fail(WALT_EINVAL, "reads longer than " + std::to_string(kMaxReadLen) + " bases are outside the tables of seed pattern " +
                                 std::to_string(kPat) + " (seedpattern.hpp)");
  return WALT_OK;
}

}  // namespace walt

using namespace walt;

extern "C" {

// bytes a call needs under geometry g: a call of several passes keeps two of them in flight (two workspaces)
static size_t pe_bytes(const PeGeometry& g, uint32_t n, int nw, uint32_t top_k, uint32_t max_read_len, bool serial) {
  return (size_t)carve_pe(nullptr, g, nw, top_k, max_read_len).total_bytes * ((n > g.chunk && !serial) ? 2 : 1);
}
// the geometry of a call whose workspace has `have` bytes (0: unknown -- the least): roomy when that fits (or is forced)
static int pe_choose(const walt_options& opt, uint32_t n, int nw, uint32_t top_k, uint32_t max_read_len, size_t have, PeGeometry* out) {
  const PeGeometry g1 = pe_geometry(n, top_k, opt, true), g0 = pe_geometry(n, top_k, opt, false);
  const bool serial = opt.pe_serial != 0;
  const bool roomy = opt.pe_roomy == 1 || (opt.pe_roomy < 0 && have >= pe_bytes(g1, n, nw, top_k, max_read_len, serial));
  *out = roomy ? g1 : g0;
  if (pe_bytes(*out, n, nw, top_k, max_read_len, serial) > have)
    return fail(WALT_EINVAL, "walt_map_pe_batch_device: the workspace is smaller than this call needs under the index's options (" +
                                 std::to_string(pe_bytes(*out, n, nw, top_k, max_read_len, serial)) + " bytes; walt_pe_workspace_bytes_best)");
  return WALT_OK;
}

size_t walt_pe_workspace_bytes(uint32_t n, uint32_t max_read_len, uint32_t top_k) {
  int nw = nw_for_len(max_read_len);
  if (!nw) nw = 64;
  const walt_options defaults;
  return pe_bytes(pe_geometry(n, top_k, defaults, false), n, nw, top_k, max_read_len, false);
}

size_t walt_pe_workspace_bytes_best(walt_index* idx, uint32_t n, uint32_t max_read_len, uint32_t top_k) {
  if (!idx) return walt_pe_workspace_bytes(n, max_read_len, top_k);
  int nw = nw_for_len(max_read_len);
  if (!nw) nw = 64;
  const walt_options& opt = idx->opt;
  const bool serial = opt.pe_serial != 0;
  const size_t small = pe_bytes(pe_geometry(n, top_k, opt, false), n, nw, top_k, max_read_len, serial);
  const size_t roomy = pe_bytes(pe_geometry(n, top_k, opt, true), n, nw, top_k, max_read_len, serial);
  if (opt.pe_roomy == 0) return small;
  if (opt.pe_roomy == 1) return roomy;
  // roomy when idx's device has room for the roomy workspace and the batch's pair records beside what is allocated already
  size_t free_b = 0, total_b = 0;
  if (hipSetDevice(idx->device) != hipSuccess || hipMemGetInfo(&free_b, &total_b) != hipSuccess) return small;
  return free_b >= roomy + (uint64_t)n * sizeof(PairResult) + (2ull << 30) ? roomy : small;
}

int walt_map_pe_batch_device(walt_index* idx, const void* d_bases1, const void* d_offsets1, const void* d_bases2,
                             const void* d_offsets2, uint32_t n, uint32_t max_read_len, uint32_t max_mismatches,
                             uint32_t b, uint32_t top_k, int frag_range, void* d_out, void* d_stats,
                             void* d_workspace, size_t workspace_bytes, void* stream_) {
  int nw = 0;
  int rc = pe_check_args(idx, top_k, max_read_len, &nw);
  if (rc) return rc;
  if (n == 0) return WALT_OK;
  std::unique_lock<std::mutex> busy(idx->pe_busy, std::try_to_lock);
  if (!busy.owns_lock()) return fail(WALT_EINVAL, "walt_map_pe_batch: another paired-end call is running on this index (an index is not re-entrant)");
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  WALT_HIP(hipSetDevice(idx->device));
  PeGeometry geo;
  if ((rc = pe_choose(idx->opt, n, nw, top_k, max_read_len, workspace_bytes, &geo))) return rc;
  const uint32_t chunk = geo.chunk;
  if ((rc = pe_streams(idx))) return rc;
  // Passes alternate between two pipeline slots (own workspace and streams), so the latency-bound list
  // kernels and the merge of one pass run beside the throughput-bound pass 1 of the next.
  const bool serial = idx->opt.pe_serial != 0;
  const bool two = n > chunk && !serial;
  PeWorkspace w[2];
  w[0] = carve_pe(d_workspace, geo, nw, top_k, max_read_len);
  w[1] = two ? carve_pe(static_cast<uint8_t*>(d_workspace) + w[0].total_bytes, geo, nw, top_k, max_read_len) : w[0];
  WALT_HIP(hipMemsetAsync(w[0].err, 0, 128 * sizeof(uint32_t), stream));
  for (int k = 0; k < (two ? 2 : 1); ++k)
    for (int m = 0; m < 2; ++m) WALT_HIP(hipMemsetAsync(w[k].shards[m], 0, kStatShardBytes, stream));
  if (two) {
    WALT_HIP(hipEventRecord(idx->pe_start, stream));
    for (int k = 0; k < 2; ++k) WALT_HIP(hipStreamWaitEvent(idx->pe_stream[k][0], idx->pe_start, 0));
  }
  // work queued on the index's own streams must not outlive an error return (the caller frees its workspace)
  auto unwind = [&]() {
    for (int k = 0; k < 2; ++k)
      for (int j = 0; j < 2; ++j)
        if (idx->pe_stream[k][j]) (void)hipStreamSynchronize(idx->pe_stream[k][j]);
  };
  uint32_t pass = 0;
  for (uint32_t start = 0; start < n; start += chunk, ++pass) {
    uint32_t cnt = n - start < chunk ? n - start : chunk;
    const int slot = two ? (int)(pass & 1) : 0;
    rc = pe_chunk(idx, reinterpret_cast<const uint8_t*>(d_bases1), reinterpret_cast<const uint64_t*>(d_offsets1) + start,
                  reinterpret_cast<const uint8_t*>(d_bases2), reinterpret_cast<const uint64_t*>(d_offsets2) + start, cnt,
                  nw, max_read_len, max_mismatches, b, top_k, frag_range, reinterpret_cast<PairResult*>(d_out) + start,
                  reinterpret_cast<unsigned long long*>(d_stats), w[slot], w[0].err, slot,
                  two ? idx->pe_stream[slot][0] : stream);
    if (rc) { unwind(); return rc; }
  }
  if (two)
    for (int k = 0; k < 2; ++k) {
      if (hipEventRecord(idx->pe_done[k], idx->pe_stream[k][0]) != hipSuccess || hipStreamWaitEvent(stream, idx->pe_done[k], 0) != hipSuccess) {
        unwind();
        return fail(WALT_EHIP, "paired-end: joining the pipeline slots failed");
      }
    }
  return WALT_OK;
}

int walt_map_pe_batch(walt_index* idx, const char* bases1, const uint64_t* offsets1, const char* bases2,
                      const uint64_t* offsets2, uint32_t n, uint32_t max_mismatches, uint32_t b, uint32_t top_k,
                      int frag_range, walt_pair_result* out, walt_candidate* ranked1, uint32_t* ranked_n1,
                      walt_candidate* ranked2, uint32_t* ranked_n2, walt_batch_stats* stats) {
  if (!offsets1 || !offsets2 || (!out && n)) return fail(WALT_EINVAL, "walt_map_pe_batch: bad argument");
  if (stats) memset(stats, 0, 2 * sizeof(*stats));
  uint32_t max_len = 0;
  const uint64_t* offs[2] = {offsets1, offsets2};
  for (int m = 0; m < 2; ++m)
    for (uint32_t i = 0; i < n; ++i) {
      if (offs[m][i + 1] < offs[m][i]) return fail(WALT_EINVAL, "offsets not non-decreasing");
      uint64_t l = offs[m][i + 1] - offs[m][i];
      if (l > 1024) return fail(WALT_EINVAL, "read length above 1024 is not supported");
      if (l > max_len) max_len = (uint32_t)l;
    }
  int nw = 0;
  int rc = pe_check_args(idx, top_k, max_len, &nw);
  if (rc) return rc;
  if (n == 0) return WALT_OK;
  WALT_HIP(hipSetDevice(idx->device));
  const char* bases[2] = {bases1, bases2};
  void *d_bases[2] = {nullptr, nullptr}, *d_off[2] = {nullptr, nullptr}, *d_out = nullptr, *d_stats = nullptr, *d_ws = nullptr;
  hipError_t e = hipSuccess;
  for (int m = 0; m < 2 && e == hipSuccess; ++m) {
    const uint64_t nbytes = offs[m][n] - offs[m][0];
    const uint64_t* off_src = offs[m];
    std::vector<uint64_t> rel;
    if (offs[m][0] != 0) {  // a slice of a larger batch (several devices share one)
      rel.resize((size_t)n + 1);
      for (uint32_t i = 0; i <= n; ++i) rel[i] = offs[m][i] - offs[m][0];
      off_src = rel.data();
    }
    if ((e = host_api_buffer(idx, m, nbytes + 16, &d_bases[m])) != hipSuccess) break;
    if ((e = host_api_buffer(idx, 2 + m, ((size_t)n + 1) * sizeof(uint64_t), &d_off[m])) != hipSuccess) break;
    if ((e = hipMemcpy(d_bases[m], bases[m] + offs[m][0], nbytes, hipMemcpyHostToDevice)) != hipSuccess) break;
    e = hipMemcpy(d_off[m], off_src, ((size_t)n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice);
  }
  if (e == hipSuccess) e = host_api_buffer(idx, 4, (size_t)n * sizeof(walt_pair_result), &d_out);
  if (e == hipSuccess) e = host_api_buffer(idx, 5, 2 * sizeof(walt_batch_stats), &d_stats);
  // the same decision the device form's callers make (walt_pe_workspace_bytes_best): the larger passes when the device has
  // the room.  One pass at a time here (the ranked lists are copied out between passes): one workspace.
  walt_options one_slot = idx->opt;
  one_slot.pe_serial = 1;
  PeGeometry geo = pe_geometry(n, top_k, one_slot, false);
  {
    const PeGeometry g1 = pe_geometry(n, top_k, one_slot, true);
    const size_t roomy_bytes = carve_pe(nullptr, g1, nw, top_k, max_len).total_bytes;
    size_t free_b = 0, total_b = 0;
    const bool have = idx->host_api_cap[6] >= roomy_bytes;
    if (one_slot.pe_roomy == 1 || (one_slot.pe_roomy < 0 && (have || (hipMemGetInfo(&free_b, &total_b) == hipSuccess &&
                                                                   free_b >= roomy_bytes + (2ull << 30)))))
      geo = g1;
  }
  const uint32_t chunk = geo.chunk;
  const size_t ws_bytes = carve_pe(nullptr, geo, nw, top_k, max_len).total_bytes;
  if (e == hipSuccess) e = host_api_buffer(idx, 6, ws_bytes, &d_ws);
  if (e == hipSuccess) e = hipMemset(d_stats, 0, 2 * sizeof(walt_batch_stats));
  if (e != hipSuccess) return fail(WALT_EHIP, std::string("paired-end upload failed: ") + hipGetErrorString(e));
  PeWorkspace w = carve_pe(d_ws, geo, nw, top_k, max_len);
  if ((rc = pe_streams(idx))) return rc;
  e = hipMemset(w.err, 0, 128 * sizeof(uint32_t));
  for (int m = 0; m < 2 && e == hipSuccess; ++m) e = hipMemset(w.shards[m], 0, kStatShardBytes);
  if (e != hipSuccess) return fail(WALT_EHIP, std::string("workspace setup failed: ") + hipGetErrorString(e));
  for (uint32_t start = 0; start < n && !rc; start += chunk) {
    uint32_t cnt = n - start < chunk ? n - start : chunk;
    rc = pe_chunk(idx, reinterpret_cast<const uint8_t*>(d_bases[0]), reinterpret_cast<const uint64_t*>(d_off[0]) + start,
                  reinterpret_cast<const uint8_t*>(d_bases[1]), reinterpret_cast<const uint64_t*>(d_off[1]) + start, cnt, nw,
                  max_len, max_mismatches, b, top_k, frag_range, reinterpret_cast<PairResult*>(d_out) + start,
                  reinterpret_cast<unsigned long long*>(d_stats), w, w.err, 0, nullptr);
    if (rc) break;
    if (hipDeviceSynchronize() != hipSuccess) { rc = fail(WALT_EHIP, "paired-end kernels failed"); break; }
    walt_candidate* rk[2] = {ranked1, ranked2};
    uint32_t* rn[2] = {ranked_n1, ranked_n2};
    for (int m = 0; m < 2 && !rc; ++m) {
      if (rk[m] && hipMemcpy(rk[m] + (size_t)start * top_k, w.ranked[m], (size_t)cnt * top_k * sizeof(walt_candidate),
                             hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail(WALT_EHIP, "download of the ranked lists failed");
      if (!rc && rn[m] && hipMemcpy(rn[m] + start, w.heap_n[m], (size_t)cnt * 4, hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail(WALT_EHIP, "download of the ranked-list lengths failed");
    }
  }
  if (!rc) rc = check_read_errors(d_ws, nullptr);
  if (!rc) {
    if (hipMemcpy(out, d_out, (size_t)n * sizeof(walt_pair_result), hipMemcpyDeviceToHost) != hipSuccess)
      rc = fail(WALT_EHIP, "download failed");
    if (!rc && stats && hipMemcpy(stats, d_stats, 2 * sizeof(walt_batch_stats), hipMemcpyDeviceToHost) != hipSuccess)
      rc = fail(WALT_EHIP, "download of the statistics failed");
  }
  return rc;
}

}  // extern "C"
