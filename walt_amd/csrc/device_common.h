// device_common.h -- the walt_index handle and HIP helpers shared by the .hip
// translation units.
#ifndef WALT_AMD_DEVICE_COMMON_H_
#define WALT_AMD_DEVICE_COMMON_H_

#include <hip/hip_runtime.h>

#include <mutex>
#include <string>
#include <vector>

#include "../../include/walt_amd.h"
#include "host_common.h"
#include "index_core.h"

#define WALT_HIP(expr)                                                                       \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return walt::fail(WALT_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));       \
  } while (0)

// Tuning values and test hooks of the mapping calls (walt_index_set_option; names in options.h).  The mapping calls
// read NO environment variable: an index maps the same way whatever the process environment holds.  None of these
// values can make a call need MORE workspace than walt_se_workspace_bytes / walt_pe_workspace_bytes promise.
struct walt_options {
  // single-end
  int se_pipe = 1;            // the staged heavy pass in two halves on two streams
  long long se_heavy_chunk = 0;  // reads per chunk of the heavy list (0: the default; test hook: several chunks on a small batch)
  int se_stage_blocks = 0;    // blocks per compute unit of the stage launches (0: fill the device)
  int se_verify_blocks = 0;   // ... of the dense verifier launches (0: its occupancy)
  int se_stagger = 0;         // the second half of the heavy pass starts one look-up stage behind the first
  int se_lit_ablate = 0;      // measurement only (results wrong when set)
  int se_lit_side = 1;        // the literal pass on a side stream beside the end of the heavy pass (2: pass 1's share right after pass 1)
  int se_lit_staged = 0;      // the deferred reads through staged rounds with the reference's search on instead of the strand-major kernel
  long long se_defer_min = -1;   // long seeds: key-equal ranges of more slots go to the verifier (-1: default, 0: never)
  int se_stage_occ = 0;       // wavefronts per SIMD the stage kernel is built for (0: default)
  int se_carry = 1;           // pass 1 hands its state to the staged rounds (0: they start over at seed 0)
  int se_heavy_mono = 0;      // the one-kernel heavy pass instead of the staged rounds (comparison)
  long long grid = 0;         // blocks of the persistent mapping kernels (0: 8 per compute unit)
  // paired-end
  int pe_mode = 0;            // 0: staged path, 1: list kernels only
  long long pe_chunk = 0;     // pairs per pass (0: default; test hook)
  long long pe_rounds = 0;    // rounds a staged list is taken in (0: default)
  long long pe_stage_cap = 0; // reads of a staged round (0: default; test hook)
  int pe_small_heaps = 0;     // force the 8-slot heaps + overflow list of long literal lists
  int pe_serial = 0;          // mates and pipeline slots one after the other (profiling)
  int pe_push_wide = 0;       // 4-byte heap entries in the push kernel whatever -m is (A/B; 0: 2-byte entries when -m <= 15)
  long long pe_defer_min = -1;
  int pe_lit_fuse = 1;        // the literal round's three seed shifts in one launch when its list is short (0: seed by seed; A/B)
  int pe_roomy = -1;          // -1: decided once per index from the device's free memory
};

struct walt_index {
  int device = 0;
  int n_cu = 256;              // compute units of the device (hipDeviceProp_t::multiProcessorCount)
  walt_options opt;
  walt::IndexHead head;
  std::vector<uint32_t> start_index;  // n_chrom + 1
  walt::IndexView view;               // device pointers
  std::vector<void*> allocs;          // everything to hipFree
  uint32_t* d_mask_table = nullptr;   // compare_mask_table() on the device
  uint64_t device_bytes = 0;
  uint64_t bad_buckets[4] = {0, 0, 0, 0};
  uint64_t outliers[4] = {0, 0, 0, 0};
  uint32_t* brk[4] = {nullptr, nullptr, nullptr, nullptr};  // index slots of the chromosome-end entries (k_make_ent), for build_windows
  uint32_t n_brk[4] = {0, 0, 0, 0};
  uint32_t window_records[4] = {0, 0, 0, 0};  // index slots with a dense candidate window (core.h StrandView::win)
  uint64_t window_eligible[4] = {0, 0, 0, 0};  // index slots in runs that qualify for one (more than the records when the budget ended first)
  unsigned strand_mask = 0;
  // measurement hooks (walt_profile_enable / walt_profile_last)
  bool profile = false;
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};  // before pack, before map, after map
  bool ev_valid = false;
  // walt_profile_detail: events at the boundaries between the kernel groups of the last single-end call (at most
  // kDetailEvents - 1 intervals; kind of the interval that ENDS at event i: 0 pass 1, 1 heavy stages, 2 verifier,
  // 3 literal pass incl. its sort)
  static constexpr int kDetailEvents = 96;
  hipEvent_t ev_detail[kDetailEvents] = {};
  unsigned char ev_kind[kDetailEvents] = {};
  int n_detail = 0;
  std::mutex se_busy;  // one single-end call at a time per index (the streams and events below are the call's)
  hipStream_t se_side = nullptr;  // the literal pass beside the end of the heavy pass (created on first use)
  hipEvent_t se_fork = nullptr, se_join = nullptr;
  // single-end staged heavy pass in two halves (map_se.hip launch_map_se): the second half's stream and the events
  // that order the halves: [0] pass 1 done, [1] the first half's last look-up stage done, [2] second half done
  hipStream_t se_pipe = nullptr;
  hipEvent_t se_pipe_ev[3] = {nullptr, nullptr, nullptr};
  // paired-end (created on first use): two pipeline slots, each with a stream for mate 1 + merge (A, unused in
  // slot 0 of a single-pass call: the caller's stream plays that role) and one for mate 2 (B)
  std::mutex pe_busy;  // one paired-end call at a time per index
  hipStream_t pe_stream[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
  hipEvent_t pe_fork[2] = {nullptr, nullptr}, pe_join[2] = {nullptr, nullptr}, pe_done[2] = {nullptr, nullptr};
  hipEvent_t pe_start = nullptr;
  // device buffers of the host-buffer entry points (walt_map_se_batch / walt_map_pe_batch), kept between calls and
  // grown on demand: the reference's driver calls once per -N batch (mapping.cpp:479-517), and a hipMalloc /
  // hipFree pair per buffer and call costs milliseconds each.  One call at a time per index.
  void* host_api_buf[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t host_api_cap[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

namespace walt {
// grow-only device buffer `slot` of idx (freed by walt_index_close)
inline hipError_t host_api_buffer(walt_index* idx, int slot, size_t bytes, void** out) {
  if (idx->host_api_cap[slot] < bytes) {
    if (idx->host_api_buf[slot]) (void)hipFree(idx->host_api_buf[slot]);
    idx->host_api_buf[slot] = nullptr;
    idx->host_api_cap[slot] = 0;
    const size_t want = bytes + bytes / 8 + 256;
    const hipError_t e = hipMalloc(&idx->host_api_buf[slot], want);
    if (e != hipSuccess) return e;
    idx->host_api_cap[slot] = want;
  }
  *out = idx->host_api_buf[slot];
  return hipSuccess;
}
}  // namespace walt

namespace walt {

constexpr int kBlock = 256;
constexpr uint32_t kG2PadWords = 96;  // slack behind the packed genome for window loads

inline unsigned grid_for(uint64_t n, int block = kBlock) { return (unsigned)((n + block - 1) / block); }
unsigned persistent_grid(const walt_index* idx);  // blocks of the persistent mapping kernels: 8 per compute unit (option grid)

// words per packed read for a maximum read length (template instances 7/8/10/16/32/64)
inline int nw_for_len(uint32_t max_len) {
  if (max_len <= 112) return 7;  // 100 bp reads: 7 words, genome window of exactly two 16-byte loads
  if (max_len <= 128) return 8;
  if (max_len <= 160) return 10;  // 150 bp reads: 10 words instead of 16 keeps three waves per SIMD
  if (max_len <= 256) return 16;
  if (max_len <= 512) return 32;
  if (max_len <= 1024) return 64;
  return 0;
}

constexpr uint32_t kHashSpan = care_pos(kKeyWeight - 1) - care_pos(0);
constexpr int kHashWin = (int)(kHashSpan / 16) + 2;
// the 12 hashed characters (indexed positions keep them inside the genome, reference.cpp:202-203), MSB first
__device__ __forceinline__ uint32_t hash_at_dev(const uint32_t* __restrict__ g2, uint32_t pos) {
  const uint64_t first = (uint64_t)pos + care_pos(0);
  const uint32_t* w = g2 + (first >> 4);
  const uint32_t sh = 2 * (uint32_t)(first & 15);
  uint32_t raw[kHashWin + 1];
#pragma unroll
  for (int i = 0; i <= kHashWin; ++i) raw[i] = w[i];
  uint32_t h = 0;
#pragma unroll
  for (uint32_t p = 0; p < kKeyWeight; ++p) {
    const uint32_t off = care_pos(p) - care_pos(0);
    const uint32_t word = funnel_r(raw[off >> 4], raw[(off >> 4) + 1], sh);
    h = (h << 2) | ((word >> (2 * (off & 15))) & 3u);
  }
  return h;
}

// Care characters [P0, P1) (at most 32) behind genome position pos as makedb's comparator ranks them
// (reference.cpp:271-288): 0 when the character lies at or beyond `room` = chromosome end - pos, else 1 / 2 / 3
// for the strand's three letters in order; one window of genome words, characters at compile-time offsets.
template <uint32_t P0, uint32_t P1>
__device__ __forceinline__ unsigned long long marked_chars_dev(const uint32_t* __restrict__ g2, uint32_t pos,
                                                               uint32_t room) {
  static_assert(P1 > P0 && P1 - P0 <= 32, "at most 32 characters");
  constexpr uint32_t span = care_pos(P1 - 1) - care_pos(P0);
  constexpr int kWin = (int)(span / 16) + 2;
  const uint64_t first = (uint64_t)pos + care_pos(P0);
  const uint32_t* w = g2 + (first >> 4);  // g2 carries kG2PadWords of slack behind the genome
  const uint32_t sh = 2 * (uint32_t)(first & 15);
  uint32_t raw[kWin + 1];
#pragma unroll
  for (int i = 0; i <= kWin; ++i) raw[i] = w[i];
  unsigned long long k = 0;
#pragma unroll
  for (uint32_t p = P0; p < P1; ++p) {
    const uint32_t off = care_pos(p) - care_pos(P0);
    const uint32_t word = funnel_r(raw[off >> 4], raw[(off >> 4) + 1], sh);
    const uint32_t c = (word >> (2 * (off & 15))) & 3u;
    const uint32_t v = care_pos(p) >= room ? 0u : (c == 0 ? 1u : (c == 3 ? 3u : 2u));
    k = (k << 2) | v;
  }
  return k;
}

// Upload + derived-structure build for one strand from DEVICE-resident raw
// arrays (genome bytes, counter, index).  Takes ownership of nothing; the raw
// index/bytes may be freed by the caller afterwards; counter is copied.
int build_strand_device(walt_index* idx, int strand, const uint8_t* d_bytes, const uint32_t* d_counter,
                        const uint32_t* d_index, uint32_t index_size, hipStream_t stream);
int alloc_strand_g2(walt_index* idx, uint32_t** g2_out, hipStream_t stream);
int finish_strand_device(walt_index* idx, int strand, uint32_t* g2, const uint32_t* d_counter,
                         const uint32_t* d_index, uint32_t index_size, hipStream_t stream);
int new_index(int device, const IndexHead& head, int dir_bits, int n_strands, walt_index** out);
int finish_index_device(walt_index* idx);  // start_index, mask table
int choose_dir_bits(uint64_t max_index_size, int requested, int n_strands, uint64_t device_bytes);

}  // namespace walt
#endif
