// map_pe.hip -- paired-end seed-and-extend, top-k and pair merge on MI355X.
//
// Replaces, per batch: the mate / strand loops around PairEndMapping
// (reference paired.cpp:642-672, body 106-201), the per-read
// std::priority_queue top-k (paired.hpp:51-74) including its libstdc++ tie
// order, the heap drain (paired.cpp:685-692) and the pair search + single-mate
// fallback of MergePairedEndResults (paired.cpp:474-545), which the reference
// runs serially on the host.
//
// Kernels:
//   k_pe_topk   one read per lane (lookup as in map_se.hip); candidates with
//               mismatch <= max_mm are pushed IN CANDIDATE ORDER into the
//               read's heap in HBM; large regions are verified by the whole
//               wave and pushed by the owner lane in lane order.
//   k_pe_drain  pops every heap (-> ascending array in place, std::pop_heap
//               leaves the popped top at the end) and writes the ranked list in
//               pop order.
//   k_pe_merge  one pair per lane, pair_merge() of core.h.
#include <string.h>

#include "map_common.h"

namespace walt {

constexpr uint32_t kPeChunk = 1u << 21;  // pairs processed per workspace pass

// One read per lane.  LITERAL as in map_se.hip: pass 1 defers reads that hit a
// BAD bucket, pass 2 maps them from scratch (their heap restarts empty).
template <int NW, bool LITERAL>
__device__ __forceinline__ void pe_process(const IndexView& iv, BlockShared& sh, const uint32_t* si,
                                           const uint32_t* __restrict__ codes2, const uint64_t* __restrict__ offsets,
                                           uint32_t* __restrict__ err, uint32_t r,
                                           bool valid, uint32_t strand_base, uint32_t max_mm, uint32_t b,
                                           uint32_t top_k, HeapEnt* __restrict__ heaps,
                                           uint32_t* __restrict__ heap_n, uint32_t* __restrict__ defer_count,
                                           uint32_t* __restrict__ defer_list, uint32_t& n_probe,
                                           uint32_t& n_verified, uint32_t& n_big, uint32_t& len_out) {
  const uint32_t n_chrom = iv.n_chrom;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t ga = strand_base >> 1, Bd = iv.dir_bits;
  LaneRead<NW> lr;
  {
    uint64_t o = 0, oe = 0;
    if (valid) { o = offsets[r]; oe = offsets[r + 1]; }
    lane_load_read<NW>(lr, codes2, offsets[0], o, oe, valid, ga, err);
  }
  len_out = lr.len;
  bool mappable = valid && lr.len >= kMinReadLen;
  bool deferred = false;
  uint32_t defer_iter = 0;
  HeapEnt* heap = heaps + (uint64_t)(valid ? r : 0) * top_k;
  uint32_t hsize = 0;

  for (uint32_t fi = 0; fi < 2; ++fi) {
    const StrandView& sv = iv.s[strand_base + fi];
#pragma unroll 1
    for (uint32_t seed_i = 0; seed_i < 3; ++seed_i) {
      // paired.cpp:133-141: stop once the heap is full of exact (seed >= 1) or
      // one-mismatch (seed >= 2) candidates; top only decreases, so per-seed
      // predicates equal the reference's `break`.
      const bool full = hsize >= top_k;
      const uint32_t top_mm = hsize ? heap_mm(heap[0]) : 0xFFFFFFFFu;
      bool act = mappable && !(full && top_mm == 0 && seed_i) && !(full && top_mm == 1 && seed_i >= 2);
      Lookup lk;
      lk.npos = 0;
      lk.reg = empty_region();
      if (act) {
        uint32_t care[kCareWords];
        uint32_t slot, span;
        seed_query<NW>(lr.rd, lr.repeats, seed_i, ga, Bd, sh.pcode4, care, slot, span);
        if (!LITERAL && bloom_maybe(sh.bloom[fi], bloom_key_of_care(care))) {
          deferred = true;
          mappable = false;
          defer_iter = fi * 3 + seed_i;
        } else {
          seed_lookup_ex(iv, sv, care, slot, span, lr.repeats, lk, !LITERAL);
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
            if (verify_candidate<NW>(sv, si, n_chrom, pos, seed_i, lr.len, lr.rd, mk, gp, mm)) {
              ++n_verified;
              if (mm <= max_mm) {  // paired.cpp:192-195
                HeapEnt e; e.pos = gp; e.mms = mm | (fi << 31);
                topk_push(heap, hsize, top_k, e);
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
        for (uint32_t base = 0; base < o_size; base += 64) {
          uint32_t k = base + lane;
          uint32_t mm = 0xFFFFFFFFu, gp = 0;
          if (k < o_size) {
            uint32_t pos = sv.ent[o_l + k].pos, m;
            if (verify_candidate<NW>(sv, si, n_chrom, pos, seed_i, o_len, o_rd, o_mk, gp, m)) {
              mm = m;
              ++n_verified;
            }
          }
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
              topk_push(heap, hsize, top_k, e);
            }
          }
        }
        if ((int)lane == owner) ++n_big;
      }
    }
  }
  if (!LITERAL && deferred) {
    defer_list[atomicAdd(defer_count, 1u)] = r <= kDeferMask ? (r | (defer_iter << kDeferShift)) : r;
  } else if (valid) {
    heap_n[r] = hsize;
  }
}

__device__ __forceinline__ void pe_flush(uint32_t shortv, uint32_t n_probe, uint32_t n_verified, uint32_t n_big,
                                         unsigned long long* __restrict__ shards) {
  block_flush_stats(shortv, n_probe, n_verified, n_big, shards);
}

template <int NW>
__global__ __launch_bounds__(kBlock) void k_pe_topk(IndexView iv, const uint32_t* __restrict__ codes2,
                                                     const uint64_t* __restrict__ offsets,
                                                     uint32_t* __restrict__ err, uint32_t n,
                                                     uint32_t strand_base,
                                                     uint32_t max_mm, uint32_t b, uint32_t top_k,
                                                     const uint32_t* __restrict__ mask_table,
                                                     HeapEnt* __restrict__ heaps, uint32_t* __restrict__ heap_n,
                                                     unsigned long long* __restrict__ stats,
                                                     uint32_t* __restrict__ defer_count,
                                                     uint32_t* __restrict__ defer_list) {
  __shared__ BlockShared sh;
  const uint32_t* si = block_prologue(sh, iv, mask_table, strand_base);
  uint32_t n_probe = 0, n_verified = 0, n_big = 0, shortv = 0;
  // each block walks its own contiguous slice of the batch (consecutive 256-read
  // chunks share pages: a strided assignment made every load a TLB miss)
  const uint64_t chunks = ((uint64_t)n + blockDim.x - 1) / blockDim.x;
  const uint64_t per_block = (chunks + gridDim.x - 1) / gridDim.x;
  const uint64_t c_lo = (uint64_t)blockIdx.x * per_block;
  const uint64_t c_hi = c_lo + per_block < chunks ? c_lo + per_block : chunks;
  for (uint64_t c = c_lo; c < c_hi; ++c) {
    const uint64_t r64 = c * blockDim.x + threadIdx.x;
    const bool valid = r64 < n;
    const uint32_t r = valid ? (uint32_t)r64 : 0;
    uint32_t len;
    pe_process<NW, false>(iv, sh, si, codes2, offsets, err, r, valid, strand_base, max_mm, b,
                          top_k, heaps, heap_n, defer_count, defer_list, n_probe, n_verified, n_big, len);
    // paired.cpp:112-115: too_short once per strand pass
    shortv += (valid && len < kMinReadLen) ? 2u : 0u;
  }
  pe_flush(shortv, n_probe, n_verified, n_big, stats);
}

template <int NW>
__global__ __launch_bounds__(kBlock) void k_pe_topk_literal(IndexView iv, const uint32_t* __restrict__ codes2,
                                                             const uint64_t* __restrict__ offsets,
                                                             uint32_t* __restrict__ err, uint32_t strand_base,
                                                             uint32_t max_mm,
                                                             uint32_t b, uint32_t top_k,
                                                             const uint32_t* __restrict__ mask_table,
                                                             HeapEnt* __restrict__ heaps,
                                                             uint32_t* __restrict__ heap_n,
                                                             unsigned long long* __restrict__ stats,
                                                             const uint32_t* __restrict__ defer_count,
                                                             const uint32_t* __restrict__ defer_list) {
  __shared__ BlockShared sh;
  const uint32_t* si = block_prologue(sh, iv, mask_table, strand_base);
  const uint32_t count = *defer_count;
  uint32_t n_probe = 0, n_verified = 0, n_big = 0;
  for (uint32_t base = blockIdx.x * blockDim.x; base < count; base += gridDim.x * blockDim.x) {
    const uint32_t i = base + threadIdx.x;
    const bool valid = i < count;
    const uint32_t r = valid ? defer_list[i] : 0;
    uint32_t len;
    pe_process<NW, true>(iv, sh, si, codes2, offsets, err, r, valid, strand_base, max_mm, b, top_k,
                         heaps, heap_n, nullptr, nullptr, n_probe, n_verified, n_big, len);
  }
  pe_flush(0, n_probe, n_verified, n_big, stats);
}

// paired.cpp:685-692: pop everything; ranked[r][i] = i-th popped (descending mismatch).
__global__ void k_pe_drain(HeapEnt* __restrict__ heaps, const uint32_t* __restrict__ heap_n, uint32_t n,
                           uint32_t top_k, Candidate* __restrict__ ranked) {
  uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  HeapEnt* heap = heaps + (uint64_t)r * top_k;
  Candidate* out = ranked + (uint64_t)r * top_k;
  uint32_t hsize = heap_n[r];
  uint32_t i = 0;
  while (hsize) {
    HeapEnt e = heap_pop(heap, hsize);
    Candidate c; c.genome_pos = e.pos; c.strand = (e.mms >> 31) ? '-' : '+'; c.mismatch = heap_mm(e);
    out[i++] = c;
  }
}

// Pairs whose candidate lists span more than kLightCombos (i, j) combinations are
// left to k_pe_merge_heavy: one such lane would otherwise hold its whole wave for
// thousands of iterations (repeat families fill both lists to top_k).
constexpr uint32_t kLightCombos = 64;

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
    PairResult pr;
    pair_merge(ranked1 + (uint64_t)r * top_k, (int)a, ranked2 + (uint64_t)r * top_k, (int)b,
               (uint32_t)(off1[r + 1] - off1[r]), (uint32_t)(off2[r + 1] - off2[r]), starts, iv.n_chrom, frag_range,
               max_mm, pr);
    out[r] = pr;
  }
  const unsigned long long hv = __ballot(heavy);
  if (hv) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t base = 0;
    if (lane == (uint32_t)(__ffsll((long long)hv) - 1)) base = atomicAdd(heavy_count, (uint32_t)__popcll(hv));
    base = bcast(base, __ffsll((long long)hv) - 1);
    if (heavy) heavy_list[base + (uint32_t)__popcll(hv & ((1ull << lane) - 1ull))] = r;
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
  const bool fits = iv.n_chrom <= kLdsChroms;
  if (fits)
    for (uint32_t i = threadIdx.x; i <= iv.n_chrom; i += blockDim.x) s_start[i] = iv.start_index[i];
  __syncthreads();
  const uint32_t* starts = fits ? s_start : iv.start_index;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t waves_per_block = blockDim.x >> 6;
  const uint32_t count = *heavy_count;
  for (uint32_t h = blockIdx.x * waves_per_block + (threadIdx.x >> 6); h < count; h += gridDim.x * waves_per_block) {
    const uint32_t r = heavy_list[h];
    const Candidate* r1 = ranked1 + (uint64_t)r * top_k;
    const Candidate* r2 = ranked2 + (uint64_t)r * top_k;
    const uint32_t na = n1[r], nb = n2[r];
    const uint32_t len1 = (uint32_t)(off1[r + 1] - off1[r]), len2 = (uint32_t)(off2[r + 1] - off2[r]);
    const uint32_t total = na * nb;
    uint32_t min_mm = max_mm, best_times = 0;
    uint32_t best_hi = 0, best_lo = 0;  // best_pos = (pos1 << 32) + pos2
    int bi = -1, bj = -1;
    for (uint32_t base = 0; base < total; base += 64) {
      const uint32_t c = base + lane;
      bool ok = false;
      uint32_t mm = 0xFFFFFFFFu, p1 = 0, p2 = 0;
      int i = 0, j = 0;
      if (c < total) {
        i = (int)(na - 1 - c / nb);
        j = (int)(nb - 1 - c % nb);
        const Candidate A = r1[i], B = r2[j];
        p1 = A.genome_pos; p2 = B.genome_pos;
        if (A.strand != B.strand) {
          mm = A.mismatch + B.mismatch;
          if (mm <= min_mm) {
            const uint32_t c1 = chrom_id(starts, iv.n_chrom, p1), c2 = chrom_id(starts, iv.n_chrom, p2);
            if (c1 == c2) {
              uint32_t s1, e1, s2, e2;
              forward_pos(p1, A.strand, c1, len1, starts, s1, e1);
              forward_pos(p2, B.strand, c2, len2, starts, s2, e2);
              const int frag = A.strand == '+' ? (int)(e2 - s1) : (int)(e1 - s2);
              ok = frag > 0 && frag <= frag_range;
            }
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
    if (lane == 0) {
      PairResult pr;
      pair_finish(r1, (int)na, r2, (int)nb, len1, len2, starts, iv.n_chrom, max_mm, bi, bj, best_times, pr);
      out[r] = pr;
    }
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static inline uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

struct PeWorkspace {
  uint32_t* err;
  unsigned long long* shards[2];
  HeapEnt* heaps[2];
  uint32_t* heap_n[2];
  uint32_t* defer_list[2];
  uint32_t* codes2[2];
  Candidate* ranked[2];
  uint64_t stride;
  uint64_t total_bytes;
};

static PeWorkspace carve_pe(void* base, uint32_t chunk, int nw, uint32_t top_k, uint32_t max_read_len) {
  PeWorkspace w;
  uint8_t* p = reinterpret_cast<uint8_t*>(base);
  uint64_t off = 0;
  auto take = [&](uint64_t bytes) {
    uint8_t* q = p ? p + off : nullptr;
    off += align_up(bytes, 256);
    return q;
  };
  w.stride = align_up(chunk ? chunk : 1, 64);
  // [0..1] pack errors, [64 + 32 m ...] deferral control of mate m, [128] heavy-pair count of the merge
  w.err = reinterpret_cast<uint32_t*>(take(192 * sizeof(uint32_t)));
  for (int m = 0; m < 2; ++m) w.shards[m] = reinterpret_cast<unsigned long long*>(take(kStatShardBytes));
  for (int m = 0; m < 2; ++m) w.heaps[m] = reinterpret_cast<HeapEnt*>(take((uint64_t)chunk * top_k * sizeof(HeapEnt) + 64));
  for (int m = 0; m < 2; ++m) w.heap_n[m] = reinterpret_cast<uint32_t*>(take((uint64_t)chunk * 4 + 64));
  for (int m = 0; m < 2; ++m) w.defer_list[m] = reinterpret_cast<uint32_t*>(take(2 * w.stride * 4 + 64));
  for (int m = 0; m < 2; ++m) w.codes2[m] = reinterpret_cast<uint32_t*>(take(codes2_words((uint64_t)chunk * max_read_len) * 4 + 64));
  for (int m = 0; m < 2; ++m) w.ranked[m] = reinterpret_cast<Candidate*>(take((uint64_t)chunk * top_k * sizeof(Candidate) + 64));
  w.total_bytes = off;
  return w;
}

template <int NW>
static int launch_pe_topk(const walt_index* idx, const uint32_t* codes2, const uint64_t* offsets, uint32_t* err,
                          uint64_t stride, uint32_t n, uint32_t sb,
                          uint32_t max_mm, uint32_t b, uint32_t top_k, HeapEnt* heaps, uint32_t* heap_n,
                          unsigned long long* stats, uint32_t* defer_count, uint32_t* defer_list,
                          hipStream_t stream) {
  const unsigned g1 = grid_for(n) < kPersistentGrid ? grid_for(n) : kPersistentGrid;
  hipLaunchKernelGGL(k_pe_topk<NW>, dim3(g1), dim3(kBlock), 0, stream, idx->view, codes2, offsets, err, n,
                     sb, max_mm, b, top_k, idx->d_mask_table, heaps, heap_n, stats, defer_count, defer_list);
  uint32_t* defer_sorted = defer_list + stride;  // second half of the list area
  launch_bin_deferred(defer_count, defer_list, defer_sorted, stream);
  unsigned g2 = grid_for(n) < 1024u ? grid_for(n) : 1024u;
  hipLaunchKernelGGL(k_pe_topk_literal<NW>, dim3(g2), dim3(kBlock), 0, stream, idx->view, codes2, offsets, err, sb,
                     max_mm, b, top_k, idx->d_mask_table, heaps, heap_n, stats, defer_count, defer_sorted);
  return WALT_OK;
}

// one chunk (n <= chunk capacity of the workspace)
static int pe_chunk(walt_index* idx, const uint8_t* d_bases1, const uint64_t* d_off1, const uint8_t* d_bases2,
                    const uint64_t* d_off2, uint32_t n, int nw, uint32_t max_read_len, uint32_t max_mm, uint32_t b,
                    uint32_t top_k,
                    int frag_range, PairResult* d_out, unsigned long long* d_stats, const PeWorkspace& w,
                    hipStream_t stream) {
  const uint8_t* bases[2] = {d_bases1, d_bases2};
  const uint64_t* offs[2] = {d_off1, d_off2};
  // err words: [0..1] pack errors (kept across chunks), [64 + 32 m ..] deferral control block of mate m (per chunk)
  WALT_HIP(hipMemsetAsync(w.err + 64, 0, 128 * sizeof(uint32_t), stream));
  for (int m = 0; m < 2; ++m) {
    // mate 1: C->T on _CT00/_CT01; mate 2: G->A on _GA10/_GA11 (paired.cpp:643,589-593)
    unsigned long long* st = w.shards[m];
    const uint32_t sb = m ? 2u : 0u;
    uint32_t* ctl = w.err + 64 + 32 * m;
    launch_ascii_to_2bit(bases[m], offs[m], n, w.codes2[m], w.err, stream);
    int rc;
    switch (nw) {
      case 7: rc = launch_pe_topk<7>(idx, w.codes2[m], offs[m], w.err, w.stride, n, sb, max_mm, b, top_k, w.heaps[m], w.heap_n[m], st, ctl, w.defer_list[m], stream); break;
      case 8: rc = launch_pe_topk<8>(idx, w.codes2[m], offs[m], w.err, w.stride, n, sb, max_mm, b, top_k, w.heaps[m], w.heap_n[m], st, ctl, w.defer_list[m], stream); break;
      case 16: rc = launch_pe_topk<16>(idx, w.codes2[m], offs[m], w.err, w.stride, n, sb, max_mm, b, top_k, w.heaps[m], w.heap_n[m], st, ctl, w.defer_list[m], stream); break;
      case 32: rc = launch_pe_topk<32>(idx, w.codes2[m], offs[m], w.err, w.stride, n, sb, max_mm, b, top_k, w.heaps[m], w.heap_n[m], st, ctl, w.defer_list[m], stream); break;
      default: rc = launch_pe_topk<64>(idx, w.codes2[m], offs[m], w.err, w.stride, n, sb, max_mm, b, top_k, w.heaps[m], w.heap_n[m], st, ctl, w.defer_list[m], stream); break;
    }
    if (rc) return rc;
    launch_reduce_stats(w.shards[m], d_stats + 4 * m, stream);
    hipLaunchKernelGGL(k_pe_drain, dim3(grid_for(n)), dim3(kBlock), 0, stream, w.heaps[m], w.heap_n[m], n, top_k,
                       w.ranked[m]);
  }
  // both mates are mapped: mate 1's deferral list area is free and holds the heavy-pair list of the merge
  uint32_t* heavy_count = w.err + 128;
  uint32_t* heavy_list = w.defer_list[0];
  hipLaunchKernelGGL(k_pe_merge, dim3(grid_for(n)), dim3(kBlock), 0, stream, idx->view, w.ranked[0], w.heap_n[0],
                     w.ranked[1], w.heap_n[1], d_off1, d_off2, n, top_k, frag_range, max_mm, d_out, heavy_count,
                     heavy_list);
  hipLaunchKernelGGL(k_pe_merge_heavy, dim3(grid_for(n) < 2048u ? grid_for(n) : 2048u), dim3(kBlock), 0, stream,
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
  return WALT_OK;
}

}  // namespace walt

using namespace walt;

extern "C" {

size_t walt_pe_workspace_bytes(uint32_t n, uint32_t max_read_len, uint32_t top_k) {
  int nw = nw_for_len(max_read_len);
  if (!nw) nw = 64;
  uint32_t chunk = n < kPeChunk ? n : kPeChunk;
  return (size_t)carve_pe(nullptr, chunk, nw, top_k, max_read_len).total_bytes;
}

int walt_map_pe_batch_device(walt_index* idx, const void* d_bases1, const void* d_offsets1, const void* d_bases2,
                             const void* d_offsets2, uint32_t n, uint32_t max_read_len, uint32_t max_mismatches,
                             uint32_t b, uint32_t top_k, int frag_range, void* d_out, void* d_stats,
                             void* d_workspace, void* stream_) {
  int nw = 0;
  int rc = pe_check_args(idx, top_k, max_read_len, &nw);
  if (rc) return rc;
  if (n == 0) return WALT_OK;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  WALT_HIP(hipSetDevice(idx->device));
  const uint32_t chunk = n < kPeChunk ? n : kPeChunk;
  PeWorkspace w = carve_pe(d_workspace, chunk, nw, top_k, max_read_len);
  WALT_HIP(hipMemsetAsync(w.err, 0, 128 * sizeof(uint32_t), stream));
  for (int m = 0; m < 2; ++m) WALT_HIP(hipMemsetAsync(w.shards[m], 0, kStatShardBytes, stream));
  for (uint32_t start = 0; start < n; start += chunk) {
    uint32_t cnt = n - start < chunk ? n - start : chunk;
    rc = pe_chunk(idx, reinterpret_cast<const uint8_t*>(d_bases1), reinterpret_cast<const uint64_t*>(d_offsets1) + start,
                  reinterpret_cast<const uint8_t*>(d_bases2), reinterpret_cast<const uint64_t*>(d_offsets2) + start, cnt,
                  nw, max_read_len, max_mismatches, b, top_k, frag_range, reinterpret_cast<PairResult*>(d_out) + start,
                  reinterpret_cast<unsigned long long*>(d_stats), w, stream);
    if (rc) return rc;
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
  auto cleanup = [&]() {
    for (int m = 0; m < 2; ++m) { hipFree(d_bases[m]); hipFree(d_off[m]); }
    hipFree(d_out); hipFree(d_stats); hipFree(d_ws);
  };
  hipError_t e = hipSuccess;
  for (int m = 0; m < 2 && e == hipSuccess; ++m) {
    const uint64_t nbytes = offs[m][n] - offs[m][0];
    std::vector<uint64_t> rel(n + 1);
    for (uint32_t i = 0; i <= n; ++i) rel[i] = offs[m][i] - offs[m][0];
    if ((e = hipMalloc(&d_bases[m], nbytes + 16)) != hipSuccess) break;
    if ((e = hipMalloc(&d_off[m], (n + 1) * sizeof(uint64_t))) != hipSuccess) break;
    if ((e = hipMemcpy(d_bases[m], bases[m] + offs[m][0], nbytes, hipMemcpyHostToDevice)) != hipSuccess) break;
    e = hipMemcpy(d_off[m], rel.data(), (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice);
  }
  const uint32_t chunk = n < kPeChunk ? n : kPeChunk;
  const size_t ws_bytes = carve_pe(nullptr, chunk, nw, top_k, max_len).total_bytes;
  if (e == hipSuccess) e = hipMalloc(&d_out, (size_t)n * sizeof(walt_pair_result));
  if (e == hipSuccess) e = hipMalloc(&d_stats, 2 * sizeof(walt_batch_stats));
  if (e == hipSuccess) e = hipMalloc(&d_ws, ws_bytes);
  if (e == hipSuccess) e = hipMemset(d_stats, 0, 2 * sizeof(walt_batch_stats));
  if (e != hipSuccess) {
    cleanup();
    return fail(WALT_EHIP, std::string("paired-end upload failed: ") + hipGetErrorString(e));
  }
  PeWorkspace w = carve_pe(d_ws, chunk, nw, top_k, max_len);
  hipMemset(w.err, 0, 128 * sizeof(uint32_t));
  for (int m = 0; m < 2; ++m) hipMemset(w.shards[m], 0, kStatShardBytes);
  for (uint32_t start = 0; start < n && !rc; start += chunk) {
    uint32_t cnt = n - start < chunk ? n - start : chunk;
    rc = pe_chunk(idx, reinterpret_cast<const uint8_t*>(d_bases[0]), reinterpret_cast<const uint64_t*>(d_off[0]) + start,
                  reinterpret_cast<const uint8_t*>(d_bases[1]), reinterpret_cast<const uint64_t*>(d_off[1]) + start, cnt, nw,
                  max_len, max_mismatches, b, top_k, frag_range, reinterpret_cast<PairResult*>(d_out) + start,
                  reinterpret_cast<unsigned long long*>(d_stats), w, nullptr);
    if (rc) break;
    if (hipDeviceSynchronize() != hipSuccess) { rc = fail(WALT_EHIP, "paired-end kernels failed"); break; }
    walt_candidate* rk[2] = {ranked1, ranked2};
    uint32_t* rn[2] = {ranked_n1, ranked_n2};
    for (int m = 0; m < 2; ++m) {
      if (rk[m]) hipMemcpy(rk[m] + (size_t)start * top_k, w.ranked[m], (size_t)cnt * top_k * sizeof(walt_candidate), hipMemcpyDeviceToHost);
      if (rn[m]) hipMemcpy(rn[m] + start, w.heap_n[m], (size_t)cnt * 4, hipMemcpyDeviceToHost);
    }
  }
  if (!rc) rc = check_read_errors(d_ws, nullptr);
  if (!rc) {
    if (hipMemcpy(out, d_out, (size_t)n * sizeof(walt_pair_result), hipMemcpyDeviceToHost) != hipSuccess)
      rc = fail(WALT_EHIP, "download failed");
    if (!rc && stats) hipMemcpy(stats, d_stats, 2 * sizeof(walt_batch_stats), hipMemcpyDeviceToHost);
  }
  cleanup();
  return rc;
}

}  // extern "C"
