// map_common.h -- device-side pieces shared by the single-end and paired-end
// mapping kernels: block prologue (LDS staging), per-lane read record, candidate
// verification, wave-level reductions.
#ifndef WALT_AMD_MAP_COMMON_H_
#define WALT_AMD_MAP_COMMON_H_

#include "device_common.h"

namespace walt {

constexpr uint32_t kMaskTableWords = kPat * (kMaxRepeats - kMinRepeats + 1) * kMaskWords;  // 1170 for pattern 3
constexpr uint32_t kLdsChroms = 1023;  // start_index entries staged in LDS when they fit
constexpr unsigned kPersistentGrid = 256 * 8;  // blocks of the persistent mapping kernels (256 CUs x 8)
constexpr uint32_t kSmallRegion = 4;   // regions up to this size are verified by their own lane

struct BlockShared {
  uint32_t mask_table[kMaskTableWords];
  uint32_t start_index[kLdsChroms + 1];
  uint16_t pcode4[256];            // prefix code of 4 care chars: bits | len << 8 (core.h pcode_*)
};

// getChromID (reference.cpp:43-60) with a fixed number of steps: largest l with
// si[l] <= pos.  top_step = largest power of two <= n_chrom (wave-uniform).
__device__ __forceinline__ uint32_t chrom_id_steps(const uint32_t* si, uint32_t n_chrom, uint32_t top_step,
                                                   uint32_t pos) {
  uint32_t l = 0;
  for (uint32_t step = top_step; step; step >>= 1) {
    const uint32_t c = l + step;
    const uint32_t v = si[c <= n_chrom ? c : n_chrom];
    l = (c <= n_chrom && pos >= v) ? c : l;
  }
  return l;
}
__device__ __forceinline__ uint32_t top_step_of(uint32_t n_chrom) {
  return n_chrom ? 1u << (31 - __clz((int)n_chrom)) : 0u;
}

// Chromosome starts for ANY number of sequences (round 4; an assembly like hg38's analysis set has 3,366).  Up to
// kLdsChroms sequences the LDS array holds every start.  Beyond that it holds every 2^shift-th start (shift the
// smallest that fits) plus the genome's end: getChromID (reference.cpp:43-60: the largest l with start[l] <= pos) is a
// search over the sampled starts in LDS, then over the at most 2^shift starts between two samples -- for shift <= 2
// (up to 4,092 sequences) five neighbouring words of the device array fetched together, no search.  Before, every
// candidate of such an assembly paid a bisection of log2(n) DEPENDENT loads over the device array in every kernel
// (3,000 contigs: pass 1 18.9 against 10.7 ms, stage kernels 38 against 16, verifier 27 against 13).
struct ChromTab {
  uint32_t n_chrom, shift, m, top;  // m = sampled intervals = ceil(n_chrom / 2^shift) <= kLdsChroms, top = top_step_of(m)
};
__device__ __forceinline__ ChromTab chrom_tab_of(uint32_t n_chrom) {
  ChromTab t;
  t.n_chrom = n_chrom;
  uint32_t sh = 0;
  while (((n_chrom + (1u << sh) - 1u) >> sh) > kLdsChroms) ++sh;  // (uniform: scalar)
  t.shift = sh;
  t.m = (n_chrom + (1u << sh) - 1u) >> sh;
  t.top = top_step_of(t.m);
  return t;
}
// fills lds[0 .. m] (kLdsChroms + 1 words); the caller's barrier follows
__device__ __forceinline__ void chrom_tab_stage(uint32_t* lds, const uint32_t* __restrict__ gs, const ChromTab& t) {
  for (uint32_t i = threadIdx.x; i <= t.m; i += blockDim.x) {
    const uint32_t c = i << t.shift;
    lds[i] = gs[c < t.n_chrom ? c : t.n_chrom];
  }
}
// chr = getChromID(pos); c_lo / c_hi = its first base and the next chromosome's (lds: the staged samples, gs: the device array)
__device__ __forceinline__ void chrom_find(const uint32_t* lds, const uint32_t* __restrict__ gs, const ChromTab& t, uint32_t pos,
                                           uint32_t& chr, uint32_t& c_lo, uint32_t& c_hi) {
  uint32_t ci = chrom_id_steps(lds, t.m, t.top, pos);
  ci = ci < t.m ? ci : (t.m ? t.m - 1u : 0u);  // (a position at or beyond the genome's end: the last interval)
  if (t.shift == 0) {  // (uniform)
    chr = ci; c_lo = lds[ci]; c_hi = lds[ci + 1];
    return;
  }
  const uint32_t base = ci << t.shift;
  if (t.shift <= 2) {
    uint32_t w[5];
#pragma unroll
    for (uint32_t k = 0; k < 5; ++k) w[k] = gs[base + k < t.n_chrom ? base + k : t.n_chrom];  // independent loads, one or two lines
    uint32_t off = 0;
    c_lo = w[0]; c_hi = w[1];
#pragma unroll
    for (uint32_t k = 1; k < 4; ++k) {
      const bool take = k < (1u << t.shift) && base + k < t.n_chrom && pos >= w[k];
      off = take ? k : off;
      c_lo = take ? w[k] : c_lo;
      c_hi = take ? w[k + 1] : c_hi;
    }
    chr = base + off;
    return;
  }
  const uint32_t nsub = t.n_chrom - base < (1u << t.shift) ? t.n_chrom - base : (1u << t.shift);
  const uint32_t off = chrom_id_steps(gs + base, nsub, 1u << t.shift, pos);
  chr = base + off; c_lo = gs[base + off]; c_hi = gs[base + off + 1];
}
__device__ __forceinline__ void chrom_bounds(const uint32_t* lds, const uint32_t* __restrict__ gs, const ChromTab& t, uint32_t pos,
                                             uint32_t& c_lo, uint32_t& c_hi) {
  uint32_t chr;
  chrom_find(lds, gs, t, pos, chr, c_lo, c_hi);
}

// Stage the compare-mask table and the chromosome starts in LDS (ChromTab: all of them, or every 2^shift-th).
// Returns the LDS array; the lookups take it together with the device array (chrom_find).
__device__ __forceinline__ const uint32_t* block_prologue(BlockShared& sh, const IndexView& iv,
                                                          const uint32_t* __restrict__ mask_table,
                                                          uint32_t strand_base) {
  for (uint32_t i = threadIdx.x; i < kMaskTableWords; i += blockDim.x) sh.mask_table[i] = mask_table[i];
  for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) {  // 4 chars, first char in bits 7..6
    uint32_t bits = 0, len = 0;
    for (uint32_t k = 0; k < 4; ++k) {
      const uint32_t c = (i >> (6 - 2 * k)) & 3u, l = pcode_len(c, strand_base >> 1);
      bits = (bits << l) | pcode_bits(c, strand_base >> 1);
      len += l;
    }
    sh.pcode4[i] = (uint16_t)(bits | (len << 8));
  }
  chrom_tab_stage(sh.start_index, iv.start_index, chrom_tab_of(iv.n_chrom));  // every start, or every 2^shift-th (ChromTab)
  __syncthreads();
  return sh.start_index;
}

// LDS copy of the two strands' Bloom prefilters (core.h pre_hash); filled before block_prologue's barrier
struct PreFilter {
  uint32_t bits[2][kPreBits / 32];
};
__device__ __forceinline__ void prefilter_stage(PreFilter& pf, const IndexView& iv, uint32_t strand_base) {
  for (uint32_t i = threadIdx.x; i < kPreBits / 32; i += blockDim.x) {
    pf.bits[0][i] = iv.s[strand_base].pre[i];
    pf.bits[1][i] = iv.s[strand_base + 1].pre[i];
  }
}
__device__ __forceinline__ bool prefilter_hit(const PreFilter& pf, uint32_t fi, uint32_t key) {
  const uint32_t h = pre_hash(key);
  return (pf.bits[fi][h >> 5] >> (h & 31)) & 1u;
}

// Minimum over the 64 lanes, in every lane.  DPP moves inside the rows of 16 lanes (quad swaps, half-row and row
// mirrors: a few cycles each) and four readlanes across the rows, instead of six ds_bpermute round trips through
// the LDS crossbar: the wave-cooperative verification does one of these per 64 candidates, and with three waves per
// SIMD its dependent chain was a visible part of a step.
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
  uint32_t o;
  o = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
  v = o < v ? o : v;
  o = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
  v = o < v ? o : v;
  o = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, true);  // row_half_mirror
  v = o < v ? o : v;
  o = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xF, 0xF, true);  // row_mirror
  v = o < v ? o : v;
  const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
  const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), r3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
  const uint32_t a = r0 < r1 ? r0 : r1, b = r2 < r3 ? r2 : r3;
  return a < b ? a : b;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
// ds_bpermute whose result is PINNED where it is written.  hipcc turns `c ? __shfl(a, l) : __shfl(b, l)` (and a
// shuffle whose value is only used under a per-lane condition) into a branch with the ds_bpermute inside it, i.e. run
// with the lanes that fail the condition switched off -- and a ds_bpermute reads nothing from a switched-off lane (seen
// in the ISA of k_se_stage's read hand-out, where it mapped reads with other reads' offsets).  The empty volatile asm
// uses the value at this point, so the shuffle stays in the block it is written in.
__device__ __forceinline__ uint32_t shfl_pin(uint32_t v, uint32_t src_lane) {
  uint32_t r = (uint32_t)__shfl((int)v, (int)src_lane);
  asm volatile("" : "+v"(r));
  return r;
}
__device__ __forceinline__ uint32_t shfl_up_pin(uint32_t v, uint32_t d) {
  uint32_t r = (uint32_t)__shfl_up((int)v, d);
  asm volatile("" : "+v"(r));
  return r;
}
// exclusive prefix sum over the lanes (and the total, in every lane)
__device__ __forceinline__ uint32_t wave_excl_scan_u32(uint32_t v, uint32_t lane, uint32_t& total) {
  uint32_t x = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t y = (uint32_t)__shfl_up((int)x, off);
    x += lane >= (uint32_t)off ? y : 0u;
  }
  total = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
  return x - v;
}
__device__ __forceinline__ uint32_t bcast(uint32_t v, int lane) {  // lane is wave-uniform
  return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}

template <int NW>
struct LaneRead {
  uint32_t len;
  uint32_t repeats;   // seed_pattern_repeats == seed_len (mapping.cpp:235-239)
  uint32_t rd[NW];
};

// ---------------------------------------------------------------------------
// Read front-end.  k_ascii_to_2bit streams the concatenated ASCII reads once into
// a dense 2-bit array (base i of the byte stream at bits 2(i%16) of word i/16;
// coalesced, ~28 VALU per 4 bytes, validity of every byte checked there).  Each
// mapping lane then loads the NW+1 words that hold its read, aligns them with
// funnel shifts and converts C->T / G->A by a bit trick (mapping.cpp:142-164).
// The care string of a seed shift and its directory range are computed in
// registers (seed_query) -- the record index_core.h pack_read() specifies,
// without ever storing it.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t convert_word(uint32_t x, uint32_t ga) {
  const uint32_t lo = x & 0x55555555u;
  // C->T: 01 -> 11 (hi |= lo).  G->A: 10 -> 00 (hi &= lo).
  return ga ? (x & (0x55555555u | (lo << 1))) : (x | (lo << 1));
}

// 4 ASCII bases in a word -> 4 two-bit codes in bits 0..7 and a 4-bit "not ACGT" mask.
// code = ((c >> 1) ^ (c >> 2)) & 3 maps A,C,G,T to 0,1,2,3; validity by rebuilding the
// letter from the code (0x41 + 2 lo + 6 hi + 11 (lo & hi)) and comparing.
__device__ __forceinline__ void ascii4_to_codes(uint32_t w, uint32_t& codes8, uint32_t& bad4) {
  const uint32_t x = ((w >> 1) ^ (w >> 2)) & 0x03030303u;
  uint32_t y = x | (x >> 6);
  y |= y >> 12;
  codes8 = y & 0xFFu;
  const uint32_t lo = x & 0x01010101u, hi = (x >> 1) & 0x01010101u, both = lo & hi;
  const uint32_t expect = 0x41414141u + (lo << 1) + (hi << 2) + (hi << 1) + (both << 3) + (both << 1) + both;
  const uint32_t z = expect ^ w;
  uint32_t nz = (((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z) & 0x80808080u;  // bit 7 of each differing byte
  nz >>= 7;
  uint32_t t = nz | (nz >> 7);
  t |= t >> 14;
  bad4 = t & 0xFu;
}

// `valid` lanes load the read at byte offset o (length len <= 16 NW) of the dense array
template <int NW>
__device__ __forceinline__ void lane_load_read(LaneRead<NW>& lr, const uint32_t* __restrict__ codes2, uint64_t o0,
                                               uint64_t o, uint64_t oe, bool valid, uint32_t ga,
                                               uint32_t* __restrict__ err, const IndexView& iv) {
  uint64_t len64 = valid ? oe - o : 0;
  const uint64_t rel = valid ? o - o0 : 0;  // dense array starts at the first read of the batch
  // longer than the caller's max_read_len (the workspace is sized by it), or beyond what k_ascii_to_2bit converted
  if (len64 > 16ull * NW || len64 > iv.batch_max_len || rel + len64 > iv.batch_cap_bytes) { atomicAdd(err + 1, 1u); len64 = 0; }
  lr.len = (uint32_t)len64;
  lr.repeats = lr.len >= kMinReadLen ? seed_repeats(lr.len) : 0;
  const uint32_t* p = codes2 + (rel >> 4);
  const uint32_t sh = 2 * (uint32_t)(rel & 15);
  // 16-byte loads (the address is only 4-byte aligned, which global loads allow): a quarter of the per-lane
  // L1 accesses of word loads.  A group is fetched when the read reaches into it; it may run up to
  // three words past the read's last word, which the array's slack covers (codes2_words).
  uint32_t raw[NW + 1];
#pragma unroll
  for (int w = 0; w <= NW; w += 4) {
    constexpr int kAll = NW + 1;
    const int cnt = kAll - w < 4 ? kAll - w : 4;
    uint32_t q[4] = {0, 0, 0, 0};
    if (16u * w < lr.len + 16u) __builtin_memcpy(q, p + w, 4 * cnt);
#pragma unroll
    for (int j = 0; j < cnt; ++j) raw[w + j] = q[j];
  }
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    uint32_t v = funnel_r(raw[w], raw[w + 1], sh);
    const uint32_t nb = lr.len > 16u * w ? lr.len - 16u * w : 0u;
    v = nb >= 16 ? v : (nb ? v & ((1u << (2 * nb)) - 1u) : 0u);
    lr.rd[w] = convert_word(v, ga);
  }
}

void launch_ascii_to_2bit(const uint8_t* d_bases, const uint64_t* d_offsets, uint32_t n, uint32_t* d_codes2,
                          uint64_t cap_bytes, uint32_t* d_err, hipStream_t stream);
inline uint64_t codes2_words(uint64_t total_bytes) { return total_bytes / 16 + 10; }  // >= 8 words of slack behind the last word (zeroed by k_ascii_to_2bit)

// Care string (chars at read offsets seed_i + 1 + 3 i, MSB first) of a seed shift
// and its directory range, from the packed read in registers.  The prefix code is
// accumulated four characters at a time through the LDS table pcode4; characters
// beyond seed_len are zero, whose code bits are zeros, which is exactly the zero
// padding dir_range() applies to short seeds.
// number of care characters (of at most kMaxRepeats repeats) whose read offset, counted from the first
// care character, lies below `bases`
constexpr int care_chars_within(int bases) {
  int n = 0;
  while (n < (int)(kMaxRepeats * kCareW) && (int)(care_pos((uint32_t)n) - care_pos(0)) < bases) ++n;
  return n;
}

template <int NW>
__device__ __forceinline__ void seed_query(const uint32_t* rd, uint32_t seed_len, uint32_t seed_i, uint32_t ga,
                                           uint32_t Bd, const uint16_t* pcode4, uint32_t* care, uint32_t& slot,
                                           uint32_t& span) {
  constexpr int NS = NW < 10 ? NW : 10;                       // 50 care chars reach base 3*49 + 3 = 150
  constexpr int NC = care_chars_within(16 * NS);              // care characters whose base lies in those words
  uint32_t shd[NS];
  const uint32_t sh = 2 * (seed_i + care_pos(0));             // the read shifted to the first care character
#pragma unroll
  for (int w = 0; w < NS; ++w) shd[w] = funnel_r(rd[w], w + 1 < NW ? rd[w + 1] : 0u, sh);
#pragma unroll
  for (int w = 0; w < (int)kCareWords; ++w) care[w] = 0;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int q = (int)(care_pos(i) - care_pos(0));  // compile-time offset in the shifted read
    care[i >> 4] |= ((shd[q >> 4] >> (2 * (q & 15))) & 3u) << (30 - 2 * (i & 15));
  }
  const uint32_t care_len = care_len_of(seed_len);
#pragma unroll
  for (int w = 0; w < (int)kCareWords; ++w) {  // zero the characters at and beyond the seed (core.h care_len_of)
    const uint32_t keep = care_len > 16u * w ? care_len - 16u * w : 0u;
    care[w] = keep >= 16 ? care[w] : (keep ? care[w] & ~(0xFFFFFFFFu >> (2 * keep)) : 0u);
  }
  // prefix code of the first 32 characters (>= 32 bits), 4 characters per table lookup
  uint64_t acc = 0;
  uint32_t nb = 0;
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const uint32_t e = pcode4[(care[g >> 2] >> (24 - 8 * (g & 3))) & 0xFFu];
    acc = (acc << (e >> 8)) | (e & 0xFFu);
    nb += e >> 8;
  }
  const uint32_t v_lo = (uint32_t)(acc >> (nb - Bd));
  // code bits the seed itself carries: 2 per character minus the one-bit letters (T after C->T, A after G->A)
  const uint32_t k = care_len < 32 ? care_len : 32;
  uint32_t shorts = 0;
#pragma unroll
  for (int w = 0; w < 2; ++w) {
    const uint32_t x = care[w];
    uint32_t one = ga ? (~x & (~x >> 1)) : (x & (x >> 1));  // bit 2j set: char j is the one-bit letter
    one &= 0x55555555u;
    const uint32_t keep = k > 16u * w ? k - 16u * w : 0u;
    one = keep >= 16 ? one : (keep ? one & ~(0xFFFFFFFFu >> (2 * keep)) : 0u);
    shorts += __popc(one);
  }
  const uint32_t nb_seed = 2 * k - shorts;
  span = seed_len ? (nb_seed < Bd ? 1u << (Bd - nb_seed) : 1u) : 0u;
  slot = seed_len ? dir_top(Bd) - (nb_seed < Bd ? (v_lo >> (Bd - nb_seed)) << (Bd - nb_seed) : v_lo) : 0u;
}

template <int NW>
__device__ __forceinline__ void make_masks(uint32_t* mk, const uint32_t* mask_table_lds, uint32_t seed_i,
                                           uint32_t repeats, uint32_t len) {
#pragma unroll
  for (int w = 0; w < NW; ++w) mk[w] = compare_mask_word(mask_table_lds, seed_i, repeats, len, (uint32_t)w);
}

// One candidate: index slot j of strand sv for a read of length len at seed
// shift seed_i.  Edge filters of mapping.cpp:280-286; returns false when the
// candidate is skipped.  gp_out = genome_pos - seed_i.
template <int NW>
__device__ __forceinline__ bool verify_candidate(const StrandView& sv, const uint32_t* si, const uint32_t* __restrict__ gs,
                                                 uint32_t n_chrom, uint32_t slot_pos, uint32_t seed_i, uint32_t len,
                                                 const uint32_t* rd, const uint32_t* mk, uint32_t& gp_out,
                                                 uint32_t& mm_out) {
  uint32_t c_lo, c_hi;
  chrom_bounds(si, gs, chrom_tab_of(n_chrom), slot_pos, c_lo, c_hi);
  if (slot_pos - c_lo < seed_i) return false;
  uint32_t gp = slot_pos - seed_i;
  if (gp + len >= c_hi) return false;
  gp_out = gp;
  mm_out = count_mismatch<NW>(sv.g2, gp, rd, mk);
  return true;
}

// Batch statistics.  Per-wave atomics on the four counters of walt_batch_stats
// serialise at one L2 line (3 M same-address atomics cost ~8 ms per 50 M reads),
// so each block reduces its counters in LDS and adds them to one of kStatShards
// shards (one 128-byte line each) in the workspace; reduce_stats() folds the
// shards into the caller's walt_batch_stats at the end of the call.
constexpr uint32_t kStatShards = 256;
constexpr uint32_t kStatShardWords = 16;  // u64 words per shard = 128 B
constexpr uint64_t kStatShardBytes = (uint64_t)kStatShards * kStatShardWords * 8;

__device__ __forceinline__ void block_flush_stats(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3,
                                                  unsigned long long* __restrict__ shards) {
  __shared__ uint32_t red[kBlock / 64][4];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  a0 = wave_sum_u32(a0); a1 = wave_sum_u32(a1); a2 = wave_sum_u32(a2); a3 = wave_sum_u32(a3);
  if (lane == 0) { red[wave][0] = a0; red[wave][1] = a1; red[wave][2] = a2; red[wave][3] = a3; }
  __syncthreads();
  if (threadIdx.x < 4) {
    uint32_t t = 0;
    for (uint32_t w = 0; w < blockDim.x / 64; ++w) t += red[w][threadIdx.x];
    if (t) atomicAdd(&shards[(uint64_t)(blockIdx.x % kStatShards) * kStatShardWords + threadIdx.x], (unsigned long long)t);
  }
}

#if defined(WALT_DIAG)
// diagnostic build: event counters of the staged kernels (walt_profile_stage_stamps returns them behind the phase sums)
static __device__ unsigned long long g_diag_ctr[16];
#define WALT_DIAG_COUNT(i, v) do { const unsigned long long v_ = (unsigned long long)(v); if ((g_diag_twice >> 16) != 0 && (threadIdx.x & 63u) == 0) atomicAdd(&g_diag_ctr[(i)], v_); } while (0)
// ... and "do it twice" switches (walt_profile_stage_stamps, bits 8 and up of `on`): what a part of the kernel costs is
// the time the run gains when that part runs a second time with the same inputs (the results stay valid)
static __device__ uint32_t g_diag_twice;  // (bit 16: the event counters count -- millions of same-address atomics, 30 ms per step)
#define WALT_DIAG_TWICE(bit) ((g_diag_twice >> (bit)) & 1u)
#else
#define WALT_DIAG_COUNT(i, v)
#define WALT_DIAG_TWICE(bit) 0u
#endif

// ---- slot probes of the seed-major kernels (map_se.hip se_process_dual, map_pe.hip pe_process_dual) ----
struct SlotProbe {
  uint32_t lo, ne;      // slot range [lo, lo+ne) of the directory lookup (ne == 0: nothing)
  Ent e[kScan];         // its first entries (independent loads)
  bool inl;             // e[0] came inline with the slot table record (single-entry slot; lo is not known)
};

// directory pair of a probe: dir[slot] (start) and dir[slot - span] (end).  span is 1
// for every seed of >= dir_bits code bits, so the pair is ONE 8-byte load of
// dir[slot-1 .. slot]; short seeds (span > 1) fetch the far end separately.
// With a slot table (core.h StrandView::tab) the pair comes from the slot's 12-byte record, which holds the
// entry itself when the slot has exactly one: then lo = 0, hi = 1 and p.e[0] is already there.
__device__ __forceinline__ void probe_issue(const StrandView& sv, bool need, uint32_t slot, uint32_t span,
                                            SlotProbe& p, uint32_t& hi) {
  hi = p.lo = 0;
  p.inl = false;
  p.e[0].key_hi = p.e[0].key_lo = p.e[0].pos = 0;
  if (need) {  // idle lanes issue no load (see verify_nobranch)
    if (sv.tab != nullptr && span == 1) {
      uint32_t w[3];
      __builtin_memcpy(w, sv.tab + 3ull * (uint32_t)(slot - 1u), 12);
      if (w[2] == kTabMulti) {
        p.lo = w[0];
        hi = w[0] + w[1];
      } else {
        p.e[0].key_hi = w[0]; p.e[0].key_lo = w[1]; p.e[0].pos = w[2];
        p.inl = true;
        hi = 1;
      }
    } else {
      const uint32_t* q = sv.dir + (uint32_t)(slot - 1u);  // slot in [1, 2^Bd]; 2^32 is held as 0 (core.h dir_top)
      uint32_t pair[2];
      __builtin_memcpy(pair, q, 8);
      hi = pair[0];
      p.lo = pair[1];
      if (span != 1) hi = sv.dir[(uint32_t)(slot - span)];
    }
  }
}
__device__ __forceinline__ void probe_entries(const StrandView& sv, SlotProbe& p) {
  if (p.ne && !p.inl) p.e[0] = sv.ent[p.lo];
#pragma unroll
  for (uint32_t j = 1; j < kScan; ++j) {
    Ent z; z.key_hi = z.key_lo = z.pos = 0;
    p.e[j] = z;
    if (j < p.ne) p.e[j] = sv.ent[p.lo + j];
  }
}
// kernel instances for reads of up to 16 NW bases: can a seed exceed the 44 care characters of hash + key?
// (pattern 3: NW > 8, reads above 134 bases; pattern 5: NW >= 8, from 119 bases; pattern 7: every instance, from 90)
template <int NW>
constexpr bool long_seed_nw() {
  uint32_t len = 16u * NW < kMaxReadLen ? 16u * NW : kMaxReadLen;
  uint32_t r = (len - kPat + 1) / kPat;
  r = r < kMaxRepeats ? r : kMaxRepeats;
  return r * kCareW > kKeyWeight + kKeyChars;
}

// region + leading candidate positions from a probed slot (core.h seed_lookup_ex, scan branch)
// LONG_SEED: instances for reads whose seeds can exceed the 32 key characters (long_seed_nw)
// tail_check: set when a single key-equal candidate still has to pass the care characters behind the key
// (>= 44); the caller tests them on the genome words its verification loads anyway (tail_care_ok).
// unresolved (patterns 5 / 7, whose tail characters span more than the two-word window of lit_region_small): a
// key-equal range of several slots is returned as it is and flagged -- pass 1 hands such a read to the heavy pass
// instead of running lit_region's chain of dependent loads beside 63 reads that have none
template <bool LONG_SEED>
__device__ __forceinline__ void probe_resolve(const StrandView& sv, const SlotProbe& p, const uint32_t* care,
                                              uint32_t seed_len, Lookup& out, bool& tail_check, bool* unresolved = nullptr) {
  tail_check = false;
  if (unresolved) *unresolved = false;
  out.npos = 0;
  out.reg = empty_region();
  if (p.ne == 0) return;
  const uint32_t n = seed_len > kKeyWeight ? seed_len - kKeyWeight : 0u;
  const uint32_t nk = n < kKeyChars ? n : kKeyChars;
  const uint64_t M = key_mask(nk);
  const uint64_t T = target_key(care) & M;
  uint32_t a, u;
  if (p.ne <= kScanMax) {
    uint32_t n_lt = 0, n_eq = 0;
#pragma unroll
    for (uint32_t j = 0; j < kScan; ++j) {
      const uint64_t k = ent_key(p.e[j]) & M;
      n_lt += (j < p.ne && k < T) ? 1u : 0u;
      n_eq += (j < p.ne && k == T) ? 1u : 0u;
    }
#pragma unroll
    for (uint32_t i = 0; i < kLookupPos; ++i) {
      uint32_t q = 0;
#pragma unroll
      for (uint32_t j = 0; j < kScan; ++j) q = (n_lt + i == j) ? p.e[j].pos : q;
      out.pos[i] = q;
    }
    if (p.ne > kScan) slot_scan_more(sv, p.lo, p.ne, T, M, n_lt, n_eq, out.pos);  // one round trip per 4 more entries
    if (n_eq == 0) return;
    a = p.lo + n_lt;
    u = a + n_eq - 1;
    out.npos = n_eq < kLookupPos ? n_eq : kLookupPos;
  } else {
    if (sv.fen[0] != nullptr ? !slot_fence_search(sv, p.lo, p.lo + p.ne, T, M, a, u)
                             : !slot_kary_search(sv, p.lo, p.lo + p.ne, T, M, a, u)) return;
  }
  if (n > kKeyChars) {
    const uint32_t size = u - a + 1;
    if (LONG_SEED && size == 1 && out.npos == 1) {
      // IndexRegion on one slot (mapping.cpp:206-211): it survives iff every remaining care char matches
      tail_check = true;
      out.reg.l = a; out.reg.u = a;
    } else if (kPat == 3 && LONG_SEED && size <= kLookupPos && out.npos == size) {
      out.reg = lit_region_small(sv, care, seed_len, a, size, out.pos, out.npos);
    } else if (kPat != 3 && unresolved != nullptr) {
      *unresolved = true;
      out.npos = 0;
      out.reg.l = a; out.reg.u = u;
    } else {
      out.npos = 0;
      out.reg = lit_region(sv, care, kKeyWeight + kKeyChars, seed_len, a, u);
    }
    return;
  }
  out.reg.l = a; out.reg.u = u;
}

// One fence round (core.h fence_plan) of the four searches of a probe -- lower and upper bound on both strands -- in
// lock-step: the A pivots of all of them are loaded together, then the B pivots.  While the two searches of a strand
// still share their range (and quarter) the second one needs no loads of its own; the wavefront skips them when that
// holds for all of its lanes (a uniform branch: the loads in it are the last before the values are needed anyway).
__device__ __forceinline__ void fence_round_dual(const StrandView& svp, const StrandView& svm, KaryState* ks, uint64_t T,
                                                 uint64_t M, uint32_t safe_p, uint32_t safe_m) {
  // Round 4: (1) at most ONE set of pivot keys is live at a time -- while the two searches of a strand share their
  // range the second one's counts are taken from the first one's keys (the words it would load are the same), and its
  // own loads are issued only inside the wave-uniform branch that needs them; (2) a search that is finished (or never
  // started) reads entry 0 of its strand, the SAME address in every idle lane -- one broadcast access that stays in
  // the L1.  It used to read entry `safe` = its slot's first entry: a separate L1 access per idle lane and load, and for
  // a lane whose slot is EMPTY some other slot's entry, i.e. a random HBM line fetched for nothing.  The loads stay
  // unconditional: a load under a per-lane branch is waited for at the end of that branch, and eight of them in a row
  // made a sub-round eight round trips (tried: 12.1 -> 15.4 ms).  Same pivots, same counts, same ranges.
  (void)safe_p; (void)safe_m;
  FencePlan p1[2], p2[2];
  bool same[2];
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const StrandView& sv = f ? svm : svp;
    p1[f] = fence_plan(sv, ks[f].x1, ks[f].y1);
    p2[f] = fence_plan(sv, ks[f].x2, ks[f].y2);
    same[f] = ks[f].x1 == ks[f].x2 && ks[f].y1 == ks[f].y2;
  }
  uint32_t q1[2], q2[2];
  {
    uint64_t a1[2][4];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (uint32_t j = 0; j < 4; ++j) {
        a1[f][j] = fence_load(fence_ptr(f ? svm : svp, p1[f], 4 * j + 3, 0u));
      }
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      q1[f] = fence_count4(p1[f], a1[f], 3, 4, 4, T, M, true);
      q2[f] = fence_count4(p2[f], a1[f], 3, 4, 4, T, M, false);  // (right when same[f]: p2 == p1 and the keys are these)
    }
  }
  if (__ballot((!same[0] && p2[0].m) || (!same[1] && p2[1].m))) {
    uint64_t a2[2][4];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (uint32_t j = 0; j < 4; ++j) {
        FencePlan pz = p2[f];
        pz.m = same[f] ? 0u : pz.m;  // (its counts come from the first search's keys)
        a2[f][j] = fence_load(fence_ptr(f ? svm : svp, pz, 4 * j + 3, 0u));
      }
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const uint32_t own = fence_count4(p2[f], a2[f], 3, 4, 4, T, M, false);
      q2[f] = same[f] ? q2[f] : own;
    }
  }
  uint32_t c1[2], c2[2];
  {
    uint64_t b1[2][4];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
#pragma unroll
      for (uint32_t j = 0; j < 3; ++j) {
        b1[f][j] = fence_load(fence_ptr(f ? svm : svp, p1[f], 4 * q1[f] + j, 0u));
      }
      b1[f][3] = 0;
    }
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      c1[f] = fence_count4(p1[f], b1[f], 4 * q1[f], 1, 3, T, M, true);
      c2[f] = fence_count4(p2[f], b1[f], 4 * q2[f], 1, 3, T, M, false);  // (right when same[f] and q1[f] == q2[f])
    }
  }
  if (__ballot(((!same[0] || q1[0] != q2[0]) && p2[0].m) || ((!same[1] || q1[1] != q2[1]) && p2[1].m))) {
    uint64_t b2[2][4];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const bool mine = !same[f] || q1[f] != q2[f];
#pragma unroll
      for (uint32_t j = 0; j < 3; ++j) {
        FencePlan pz = p2[f];
        pz.m = mine ? pz.m : 0u;
        b2[f][j] = fence_load(fence_ptr(f ? svm : svp, pz, 4 * q2[f] + j, 0u));
      }
      b2[f][3] = 0;
    }
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const uint32_t own = fence_count4(p2[f], b2[f], 4 * q2[f], 1, 3, T, M, false);
      c2[f] = (same[f] && q1[f] == q2[f]) ? c2[f] : own;
    }
  }
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    fence_narrow(p1[f], 4 * q1[f] + c1[f], ks[f].x1, ks[f].y1);
    fence_narrow(p2[f], 4 * q2[f] + c2[f], ks[f].x2, ks[f].y2);
  }
}

// The heavy kernels' form of probe_resolve for BOTH strands of a probe.  Nearly every wavefront of a heavy pass
// holds lanes with long slots on either strand, and two searches one after the other cost the wavefront the
// sum of their dependent rounds; here
//   1. slots of more than kScan entries on either strand are searched TOGETHER (core.h kary_round: eight
//      independent loads per strand and round; rounds = the longer of the two searches, not their sum),
//   2. the positions of regions of up to kLookupPos slots found that way are fetched in one round for both strands
//      (the small-region verification and lit_region_small want them in registers),
//   3. each strand finishes as probe_resolve does (tail characters of long seeds).
// Same regions as probe_resolve: the equal range of a key in a sorted slot does not depend on how it is searched.
// DEFER (the staged kernels, reads above 134 bases): a key-equal range of more than `defer_min` slots that lies inside
// dense candidate windows is NOT narrowed here by the care characters behind the key (>= 44; lit_region: a
// LowerBound / UpperBound bisection per character, each step an entry load and a dependent genome load) -- the whole
// range becomes the region, defer_x is set, and the verifier tests those characters on the records it streams
// anyway (map_items.h item_stream).  Dense ranges hold no chromosome-end entry
// (device_index.hip k_win_break), so inside them the index is sorted on the real characters and IndexRegion's
// result IS the set of candidates whose characters equal the read's.
template <bool LONG_SEED, bool DEFER = false>
__device__ __forceinline__ void probe_resolve_dual(const StrandView& svp, const StrandView& svm, const SlotProbe& pp,
                                                   const SlotProbe& pm, const uint32_t* care, uint32_t seed_len,
                                                   Lookup& lp, Lookup& lm, bool& tail_p, bool& tail_m,
                                                   bool* defer_p = nullptr, bool* defer_m = nullptr, uint32_t defer_min = 0,
                                                   bool win_ok = false, uint32_t multi_max = 0) {
  const uint32_t n = seed_len > kKeyWeight ? seed_len - kKeyWeight : 0u;
  const uint32_t nk = n < kKeyChars ? n : kKeyChars;
  const uint64_t M = key_mask(nk);
  const uint64_t T = target_key(care) & M;
  uint32_t a[2] = {0, 0}, u[2] = {0, 0};
  bool found[2] = {false, false};
  uint32_t npos[2] = {0, 0};
  uint32_t pos[2][kLookupPos];
  // ---- 1. [a, u] of both strands
  KaryState ks[2];
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const SlotProbe& p = f ? pm : pp;
#pragma unroll
    for (uint32_t i = 0; i < kLookupPos; ++i) pos[f][i] = 0;
    kary_init(ks[f], p.lo, p.lo);  // idle
    if (p.ne == 0) continue;
    if (p.ne <= kScan) {
      uint32_t n_lt = 0, n_eq = 0;
#pragma unroll
      for (uint32_t j = 0; j < kScan; ++j) {
        const uint64_t k = ent_key(p.e[j]) & M;
        n_lt += (j < p.ne && k < T) ? 1u : 0u;
        n_eq += (j < p.ne && k == T) ? 1u : 0u;
      }
#pragma unroll
      for (uint32_t i = 0; i < kLookupPos; ++i) {
        uint32_t q = 0;
#pragma unroll
        for (uint32_t j = 0; j < kScan; ++j) q = (n_lt + i == j) ? p.e[j].pos : q;
        pos[f][i] = q;
      }
      found[f] = n_eq != 0;
      a[f] = p.lo + n_lt;
      u[f] = a[f] + n_eq - 1;
      npos[f] = n_eq < kLookupPos ? n_eq : kLookupPos;
    } else {
      kary_init(ks[f], p.lo, p.lo + p.ne);
    }
  }
  if (svp.fen[0] != nullptr && svm.fen[0] != nullptr) {  // uniform
#if defined(WALT_DIAG)
    if (WALT_DIAG_TWICE(0)) {
      KaryState k2[2] = {ks[0], ks[1]};
      while (kary_busy(k2[0]) || kary_busy(k2[1])) fence_round_dual(svp, svm, k2, T, M, pp.lo, pm.lo);
      if (k2[0].x1 == 0xFFFFFFF0u) ks[0] = k2[0];  // (keeps the copy alive)
    }
#endif
    while (kary_busy(ks[0]) || kary_busy(ks[1])) {
      WALT_DIAG_COUNT(0, 1);                                                        // fence rounds run by a wavefront
      WALT_DIAG_COUNT(1, __popcll(__ballot(kary_busy(ks[0]) || kary_busy(ks[1]))));  // ... and the lanes that needed them
      WALT_DIAG_COUNT(6, __popcll(__ballot(kary_busy(ks[0]))) + __popcll(__ballot(kary_busy(ks[1]))));  // searches busy
      fence_round_dual(svp, svm, ks, T, M, pp.lo, pm.lo);
    }
  } else {
    while (kary_busy(ks[0]) || kary_busy(ks[1])) {
      kary_round(svp, ks[0], T, M, pp.lo);
      kary_round(svm, ks[1], T, M, pm.lo);
    }
  }
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const SlotProbe& p = f ? pm : pp;
    if (p.ne > kScan) found[f] = kary_result(ks[f], a[f], u[f]);
  }
  // ---- 2. positions of short regions the search found (one round, both strands; idle lanes read their slot's first entry)
#pragma unroll 1
  for (uint32_t twice = 0; twice <= WALT_DIAG_TWICE(1); ++twice) {
    uint32_t v[2][kLookupPos];
    bool want[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const SlotProbe& p = f ? pm : pp;
      const StrandView& sv = f ? svm : svp;
      const uint32_t size = found[f] ? u[f] - a[f] + 1 : 0u;
      want[f] = p.ne > kScan && size != 0 && size <= kLookupPos;
#pragma unroll
      // (idle lanes all read entry 0: one broadcast access; the slot's first entry, as it was, is a random line when the slot is empty)
      for (uint32_t i = 0; i < kLookupPos; ++i) v[f][i] = sv.ent[(want[f] && i < size) ? a[f] + i : 0u].pos;
    }
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      if (want[f]) {
#pragma unroll
        for (uint32_t i = 0; i < kLookupPos; ++i) pos[f][i] = v[f][i];
        npos[f] = u[f] - a[f] + 1;
      }
    }
  }
  // ---- 3. per strand
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const StrandView& sv = f ? svm : svp;
    Lookup& out = f ? lm : lp;
    bool& tail_check = f ? tail_m : tail_p;
    tail_check = false;
    out.npos = 0;
    out.reg = empty_region();
#pragma unroll
    for (uint32_t i = 0; i < kLookupPos; ++i) out.pos[i] = pos[f][i];
    bool deferred_here = false;
    if constexpr (DEFER && LONG_SEED) {  // (every lane issues dense_range's four loads: no load under a divergent branch)
      const uint32_t sz = found[f] ? u[f] - a[f] + 1 : 0u;
      const DenseRange d = dense_range(sv, a[f], sz, n > kKeyChars && sz > defer_min && win_ok);
      deferred_here = d.hi > d.lo;
      *(f ? defer_m : defer_p) = deferred_here;
    }
    if (!found[f]) continue;
    out.npos = npos[f];
    if (n > kKeyChars) {
      const uint32_t size = u[f] - a[f] + 1;
      if (deferred_here) {
        out.npos = 0;
        out.reg.l = a[f]; out.reg.u = u[f];
      } else if (LONG_SEED && size == 1 && out.npos == 1) {
        tail_check = true;  // IndexRegion on one slot (mapping.cpp:206-211)
        out.reg.l = a[f]; out.reg.u = a[f];
      } else if (kPat == 3 && LONG_SEED && size <= kLookupPos && out.npos == size) {
        out.reg = lit_region_small(sv, care, seed_len, a[f], size, out.pos, out.npos);
      } else if (kPat != 3 && LONG_SEED && size <= multi_max) {
        // patterns 5 / 7: a short key-equal range goes to the caller as it is, every candidate owing its tail characters
        // (tail_check with a region of several slots).  The caller verifies all of them side by side and keeps those
        // whose tail characters equal the read's -- IndexRegion's result when the range is sorted on them, which holds
        // when every candidate passes the edge filters of mapping.cpp:280-286 (its tail characters then lie inside its
        // chromosome, where makedb compared real characters); otherwise the caller falls back to lit_region.
        tail_check = true;
        out.reg.l = a[f]; out.reg.u = u[f];
      } else {
        out.npos = 0;
        out.reg = lit_region(sv, care, kKeyWeight + kKeyChars, seed_len, a[f], u[f]);
      }
      continue;
    }
    out.reg.l = a[f]; out.reg.u = u[f];
  }
}

// care chars [44, seed_len) of the slot at slot_pos against the read's (ent_char semantics: a position at or
// beyond the genome end matches nothing).  The two words read here lie inside the window that
// verify_nobranch loads for the same candidate, so this costs no extra memory round trip.
__device__ __forceinline__ bool tail_care_ok(const StrandView& sv, uint32_t slot_pos, const uint32_t* care,
                                             uint32_t seed_len) {
  const uint64_t q0 = (uint64_t)slot_pos + care_pos(kKeyWeight + kKeyChars);
  const uint64_t w = q0 >> 4;
  const uint64_t win = (uint64_t)sv.g2[w] | ((uint64_t)sv.g2[w + 1] << 32);
  bool ok = true;
  for (uint32_t p = kKeyWeight + kKeyChars; p < seed_len; ++p) {
    const uint64_t q = (uint64_t)slot_pos + care_pos(p);
    const uint32_t c = (uint32_t)((win >> (2 * (uint32_t)(q - (w << 4)))) & 3u);
    ok = ok && q < sv.genome_len && c == care_char(care, p);
  }
  return ok;
}

// branch-free candidate check: the genome window is always loaded (from position
// 0 when the edge filters of mapping.cpp:280-286 reject the candidate)
template <int NW>
__device__ __forceinline__ void verify_nobranch(const StrandView& sv, const BlockShared& sh, const uint32_t* si,
                                                const uint32_t* __restrict__ gs, uint32_t n_chrom, uint32_t top_step, bool active, uint32_t slot_pos,
                                                uint32_t seed_i, uint32_t len, const uint32_t* rd,
                                                const uint32_t* mk, bool& ok, uint32_t& gp, uint32_t& mm) {
  // chromosome starts from LDS when they fit (ds_read), else from HBM; uniform branch
  uint32_t c_lo, c_hi;
  (void)top_step; (void)si;
  chrom_bounds(sh.start_index, gs, chrom_tab_of(n_chrom), slot_pos, c_lo, c_hi);
  const uint32_t g = slot_pos - seed_i;
  ok = active && (slot_pos - c_lo >= seed_i) && (g + len < c_hi);
  gp = ok ? g : 0u;
  mm = 0;
  // only lanes with a candidate touch memory: the mapping kernels are bound by the number of per-lane
  // accesses the L1 (TCP) processes, not by instruction issue, so an idle lane's dummy load is not free
  if (ok) mm = count_mismatch<NW>(sv.g2, gp, rd, mk);
}

// mismatches under the compare masks and at the seed's care characters >= 44 on NW + 1 window words in registers
template <int NW, int W = 0>
__device__ __forceinline__ void count_mismatch_regs_tail(const uint32_t* g, uint32_t sh, const uint32_t* rd, const uint32_t* mask,
                                                         uint32_t seed_i, uint32_t cut, uint32_t& mm, uint32_t& tmm) {
  if constexpr (W < NW) {
    const uint32_t x = funnel_r(g[W], g[W + 1], sh) ^ rd[W];
    const uint32_t d = x | (x >> 1);
    mm += __popc(d & mask[W]);
    tmm += __popc(d & tail_care_mask_word<W>(seed_i, cut));
    count_mismatch_regs_tail<NW, W + 1>(g, sh, rd, mask, seed_i, cut, mm, tmm);
  }
}

// the same with the seed's care characters >= 44 tested on the window the count loads (patterns 5 / 7; pattern 3 has
// tail_care_ok's two words): tail_ok = they all equal the read's
template <int NW>
__device__ __forceinline__ void verify_nobranch_tail(const StrandView& sv, const BlockShared& sh, const uint32_t* si,
                                                     const uint32_t* __restrict__ gs, uint32_t n_chrom, uint32_t top_step, bool active, uint32_t slot_pos,
                                                     uint32_t seed_i, uint32_t len, const uint32_t* rd,
                                                     const uint32_t* mk, uint32_t cut, bool& ok, uint32_t& gp, uint32_t& mm,
                                                     bool& tail_ok) {
  uint32_t c_lo, c_hi;
  (void)top_step; (void)si;
  chrom_bounds(sh.start_index, gs, chrom_tab_of(n_chrom), slot_pos, c_lo, c_hi);
  const uint32_t g = slot_pos - seed_i;
  ok = active && (slot_pos - c_lo >= seed_i) && (g + len < c_hi);
  gp = ok ? g : 0u;
  mm = 0;
  uint32_t tmm = 0;
  if (ok) mm = count_mismatch_tail<NW>(sv.g2, gp, rd, mk, seed_i, cut, tmm);  // (a candidate inside its chromosome: every care character lies in the genome)
  tail_ok = tmm == 0;
}

// ---- candidates of a wave-cooperative region: G groups of 64 consecutive index slots per step ----
// Lane `lane` takes slots l + base + 64 u + lane, u < G.  Two rounds of INDEPENDENT loads per step (a
// listed / owner read has its whole wave to itself, nothing else hides the latency):
//   1. the dense record {pos, window} (32 contiguous bytes per candidate, 48 for reads above 110 bases) of the
//      slots inside the region's dense range (core.h dense_range: four loads per region), or the index
//      entry's pos where there is none,
//   2. for the latter the genome window (one scattered 128-byte line per candidate).
// mm[u] = 0xFFFFFFFF where the slot is beyond the region or fails the edge filters of mapping.cpp:280-286.
template <int NW>
__device__ __forceinline__ bool win_usable(const StrandView& sv, uint32_t len) {
  return NW <= 10 && sv.wbits != nullptr && len <= (NW <= 7 ? kWinMaxLen1 : kWinMaxLen2);
}
template <int NW, int G>
__device__ __forceinline__ void coop_verify_groups(const StrandView& sv, const uint32_t* si, const uint32_t* __restrict__ gs,
                                                   uint32_t n_chrom, uint32_t l, uint32_t size, uint32_t base, uint32_t seed_i,
                                                   uint32_t len, const uint32_t* rd, const uint32_t* mk,
                                                   uint32_t lane, const DenseRange& dr, uint32_t* gp, uint32_t* mm) {
  const ChromTab ct = chrom_tab_of(n_chrom);
  if constexpr (NW <= 10) {
    // EVERY lane issues EVERY load of a round (a lane without work reads the first words of g2: one broadcast
    // line).  A load under `if (lane has work)` sits in a divergent branch whose results are copied out before
    // the branch ends, i.e. the compiler waits for it there -- the G groups then took G round trips, not one.
    bool dense[G];
    uint4 ra[G], rc[G], re[G];
    uint32_t epos[G], pos[G], g[G];
    bool ok[G];
    const uint4* const idle = reinterpret_cast<const uint4*>(sv.g2);
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const uint32_t k = base + 64 * u + lane;
      const uint32_t slot = l + (k < size ? k : size - 1);
      dense[u] = k < size && slot >= dr.lo && slot < dr.hi;
      const uint64_t rec = dr.rec + (slot - dr.lo);
      const uint4* rp = dense[u] ? reinterpret_cast<const uint4*>(sv.win) + 2 * rec : idle;
      ra[u] = rp[0];
      rc[u] = rp[1];
      if constexpr (NW > 7) re[u] = *(dense[u] ? reinterpret_cast<const uint4*>(sv.win2) + rec : idle);
      const uint32_t* pp = dense[u] ? sv.g2 : &sv.ent[slot].pos;
      epos[u] = *pp;
    }
    bool any_gather = false;
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const uint32_t k = base + 64 * u + lane;
      pos[u] = dense[u] ? ra[u].x : epos[u];
      uint32_t c_lo, c_hi;
      chrom_bounds(si, gs, ct, pos[u], c_lo, c_hi);  // fixed trip count: the G chains interleave
      g[u] = pos[u] - seed_i;
      ok[u] = k < size && (pos[u] - c_lo >= seed_i) && (g[u] + len < c_hi);  // mapping.cpp:280-286
      gp[u] = ok[u] ? g[u] : 0u;
      any_gather = any_gather || (ok[u] && !dense[u]);
    }
    uint32_t gw[G][NW + 1];
#pragma unroll
    for (int u = 0; u < G; ++u)
#pragma unroll
      for (int w = 0; w <= NW; ++w) gw[u][w] = 0;
    if (__ballot(any_gather)) {  // wave-uniform: a step inside the dense range has no second round
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const uint32_t* gwp = (ok[u] && !dense[u]) ? sv.g2 + (g[u] >> 4) : sv.g2;
#pragma unroll
        for (int w = 0; w <= NW; w += 4) {
          constexpr int kAll = NW + 1;
          const int cnt = kAll - w < 4 ? kAll - w : 4;
          uint32_t q[4] = {0, 0, 0, 0};
          __builtin_memcpy(q, gwp + w, 4 * cnt);
#pragma unroll
          for (int t = 0; t < cnt; ++t) gw[u][w + t] = q[t];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < G; ++u) {
      uint32_t wv[NW + 1];
      const uint32_t first[11] = {ra[u].y, ra[u].z, ra[u].w, rc[u].x, rc[u].y, rc[u].z, rc[u].w,
                                  NW > 7 ? re[u].x : 0u, NW > 7 ? re[u].y : 0u, NW > 7 ? re[u].z : 0u, NW > 7 ? re[u].w : 0u};
#pragma unroll
      for (int w = 0; w <= NW; ++w) wv[w] = dense[u] ? (w < 11 ? first[w] : 0u) : gw[u][w];
      const uint32_t shv = dense[u] ? 2 * (kWinLead - seed_i) : 2 * (g[u] & 15u);
      const uint32_t m = count_mismatch_regs<NW>(wv, shv, rd, mk);
      mm[u] = ok[u] ? m : 0xFFFFFFFFu;
    }
  } else {
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const uint32_t k = base + 64 * u + lane;
      const uint32_t pos = sv.ent[l + (k < size ? k : size - 1)].pos;
      uint32_t c_lo, c_hi;
      chrom_bounds(si, gs, ct, pos, c_lo, c_hi);
      const uint32_t g = pos - seed_i;
      const bool ok = k < size && (pos - c_lo >= seed_i) && (g + len < c_hi);
      gp[u] = ok ? g : 0u;
      mm[u] = 0xFFFFFFFFu;
      if (ok) mm[u] = count_mismatch<NW>(sv.g2, g, rd, mk);
    }
  }
}

// Append `value` to a device list for the lanes with `take` set: ONE atomic per wavefront (an assembly of
// thousands of contigs defers a fifth of the reads, and ten million same-address atomics cost milliseconds).
// All 64 lanes must call this.
__device__ __forceinline__ void wave_append(bool take, uint32_t value, uint32_t* __restrict__ count,
                                            uint32_t* __restrict__ list) {
  const unsigned long long m = __ballot(take);
  if (!m) return;
  const uint32_t lane = threadIdx.x & 63;
  const int leader = (int)__ffsll((long long)m) - 1;
  uint32_t base = 0;
  if ((int)lane == leader) base = atomicAdd(count, (uint32_t)__popcll(m));
  base = bcast(base, leader);
  if (take) list[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = value;
}

// The same through a per-wavefront LDS buffer: entries collect over many calls and reach the device list kWaveBuf at a
// time.  wave_append costs one atomic on ONE address per call, and the device serves about a hundred million of
// those per second -- pass 1 of a repeat-rich genome hands a read in six to the heavy list, so that nearly every one
// of its 780,000 wavefront iterations made one: 7.8 ms of queueing in an 11 ms kernel (measured round 3: a second
// list filled the same way took the kernel from 11.6 to 16.3 ms).  Buffered, the same appends are a few thousand
// atomics.  `n` is wave-uniform; all 64 lanes call; wavelist_flush once more before the kernel ends.
constexpr uint32_t kWaveBuf = 256;
struct WaveList {
  uint32_t* buf;  // this wavefront's `cap` words of LDS (cap >= 64)
  uint32_t n;
  uint32_t cap;
};
__device__ __forceinline__ void wavelist_flush(WaveList& w, uint32_t* __restrict__ count, uint32_t* __restrict__ list) {
  const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)w.n);
  if (!n) return;
  const uint32_t lane = threadIdx.x & 63;
  uint32_t base = 0;
  if (lane == 0) base = atomicAdd(count, n);
  base = bcast(base, 0);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the buffer was written by other lanes of this wavefront
  __builtin_amdgcn_wave_barrier();
  for (uint32_t i = lane; i < n; i += 64) list[base + i] = w.buf[i];
  __builtin_amdgcn_wave_barrier();
  w.n = 0;
}
__device__ __forceinline__ void wavelist_append(WaveList& w, bool take, uint32_t value, uint32_t* __restrict__ count,
                                                uint32_t* __restrict__ list) {
  const unsigned long long m = __ballot(take);
  if (!m) return;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t k = (uint32_t)__popcll(m);
  if (w.n + k > w.cap) wavelist_flush(w, count, list);
  if (take) w.buf[w.n + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = value;
  w.n += k;
}

// Deferred reads are tagged with the (strand, seed) iteration of their first BAD
// probe and grouped by it before the literal pass, so that the lanes of a
// literal-pass wave run their long searches in the same iteration instead of
// one after another.  Entry = read | iteration << kDeferShift.
constexpr uint32_t kDeferShift = 28;
constexpr uint32_t kDeferMask = (1u << kDeferShift) - 1;
void launch_bin_deferred(uint32_t* d_ctl /*32 zeroed words: [0] = count*/, const uint32_t* d_list, uint32_t* d_sorted,
                         hipStream_t stream, const uint32_t* d_first = nullptr /*device word: first entry of the share (nullptr: 0)*/);

void launch_reduce_stats(unsigned long long* d_shards, unsigned long long* d_stats, hipStream_t stream);

int check_read_errors(const void* d_workspace, hipStream_t stream);

}  // namespace walt
#endif
