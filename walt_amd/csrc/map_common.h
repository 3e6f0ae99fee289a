// map_common.h -- device-side pieces shared by the single-end and paired-end
// mapping kernels: block prologue (LDS staging), per-lane read record, candidate
// verification, wave-level reductions.
#ifndef WALT_AMD_MAP_COMMON_H_
#define WALT_AMD_MAP_COMMON_H_

#include "device_common.h"

namespace walt {

constexpr uint32_t kMaskTableWords = 3 * (kMaxRepeats - kMinRepeats + 1) * kMaskWords;  // 1170
constexpr uint32_t kLdsChroms = 1023;  // start_index entries staged in LDS when they fit
constexpr unsigned kPersistentGrid = 256 * 8;  // blocks of the persistent mapping kernels (256 CUs x 8)
constexpr uint32_t kSmallRegion = 4;   // regions up to this size are verified by their own lane

struct BlockShared {
  uint32_t mask_table[kMaskTableWords];
  uint32_t start_index[kLdsChroms + 1];
  uint32_t bloom[2][kBloomWords];  // BAD-bucket filters of the two strands this launch maps against
};

// Stage the compare-mask table and the chromosome starts in LDS.  Returns the
// pointer the lanes use for start_index lookups (LDS when it fits, else HBM).
__device__ __forceinline__ const uint32_t* block_prologue(BlockShared& sh, const IndexView& iv,
                                                          const uint32_t* __restrict__ mask_table,
                                                          uint32_t strand_base) {
  for (uint32_t i = threadIdx.x; i < kMaskTableWords; i += blockDim.x) sh.mask_table[i] = mask_table[i];
  for (uint32_t fi = 0; fi < 2; ++fi) {
    const uint32_t* __restrict__ bl = iv.s[strand_base + fi].bloom;
    for (uint32_t i = threadIdx.x; i < kBloomWords; i += blockDim.x) sh.bloom[fi][i] = bl[i];
  }
  const bool fits = iv.n_chrom <= kLdsChroms;
  if (fits)
    for (uint32_t i = threadIdx.x; i <= iv.n_chrom; i += blockDim.x) sh.start_index[i] = iv.start_index[i];
  __syncthreads();
  return fits ? sh.start_index : iv.start_index;
}

template <int NW>
struct LaneRead {
  uint32_t len;
  uint32_t repeats;   // seed_pattern_repeats == seed_len (mapping.cpp:235-239)
  uint32_t rd[NW];
};

template <int NW>
__device__ __forceinline__ void load_lane_read(LaneRead<NW>& lr, const uint32_t* __restrict__ packed,
                                               uint64_t stride, uint32_t r, bool valid) {
  lr.len = valid ? packed[r] : 0;
  lr.repeats = lr.len >= kMinReadLen ? seed_repeats(lr.len) : 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) lr.rd[w] = valid ? packed[(uint64_t)(1 + w) * stride + r] : 0;
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
__device__ __forceinline__ bool verify_candidate(const StrandView& sv, const uint32_t* si, uint32_t n_chrom,
                                                 uint32_t slot_pos, uint32_t seed_i, uint32_t len,
                                                 const uint32_t* rd, const uint32_t* mk, uint32_t& gp_out,
                                                 uint32_t& mm_out) {
  uint32_t chr = chrom_id(si, n_chrom, slot_pos);
  if (slot_pos - si[chr] < seed_i) return false;
  uint32_t gp = slot_pos - seed_i;
  if (gp + len >= si[chr + 1]) return false;
  gp_out = gp;
  mm_out = count_mismatch<NW>(sv.g2, gp, rd, mk);
  return true;
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    uint32_t o = __shfl_xor(v, off);
    v = o < v ? o : v;
  }
  return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ uint32_t bcast(uint32_t v, int lane) {  // lane is wave-uniform
  return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
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

// Deferred reads are tagged with the (strand, seed) iteration of their first BAD
// probe and grouped by it before the literal pass, so that the lanes of a
// literal-pass wave run their long searches in the same iteration instead of
// one after another.  Entry = read | iteration << kDeferShift.
constexpr uint32_t kDeferShift = 28;
constexpr uint32_t kDeferMask = (1u << kDeferShift) - 1;
void launch_bin_deferred(uint32_t* d_ctl /*32 zeroed words: [0] = count*/, const uint32_t* d_list, uint32_t* d_sorted,
                         hipStream_t stream);

void launch_reduce_stats(unsigned long long* d_shards, unsigned long long* d_stats, hipStream_t stream);

// Packing: ASCII reads -> packed records (index_core.h pack_read).  Defined in
// map_se.hip; err[0] counts reads with a non-ACGT base, err[1] reads longer
// than 16*nw.
void launch_pack_reads(const uint8_t* d_bases, const uint64_t* d_offsets, uint32_t n, uint32_t ga, uint32_t Bd,
                       uint32_t nw, uint32_t* d_packed, uint64_t stride, uint32_t* d_err, hipStream_t stream);
int check_pack_errors(const void* d_workspace, hipStream_t stream);

}  // namespace walt
#endif
