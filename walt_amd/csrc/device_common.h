// device_common.h -- the walt_index handle and HIP helpers shared by the .hip
// translation units.
#ifndef WALT_AMD_DEVICE_COMMON_H_
#define WALT_AMD_DEVICE_COMMON_H_

#include <hip/hip_runtime.h>

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

struct walt_index {
  int device = 0;
  walt::IndexHead head;
  std::vector<uint32_t> start_index;  // n_chrom + 1
  walt::IndexView view;               // device pointers
  std::vector<void*> allocs;          // everything to hipFree
  uint32_t* d_mask_table = nullptr;   // compare_mask_table() on the device
  uint64_t device_bytes = 0;
  uint64_t bad_buckets[4] = {0, 0, 0, 0};
  uint64_t outliers[4] = {0, 0, 0, 0};
  unsigned strand_mask = 0;
  // measurement hooks (walt_profile_enable / walt_profile_last)
  bool profile = false;
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};  // before pack, before map, after map
  bool ev_valid = false;
  // paired-end (created on first use): two pipeline slots, each with a stream for mate 1 + merge (A, unused in
  // slot 0 of a single-pass call: the caller's stream plays that role) and one for mate 2 (B)
  hipStream_t pe_stream[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
  hipEvent_t pe_fork[2] = {nullptr, nullptr}, pe_join[2] = {nullptr, nullptr}, pe_done[2] = {nullptr, nullptr};
  hipEvent_t pe_start = nullptr;
};

namespace walt {

constexpr int kBlock = 256;
constexpr uint32_t kG2PadWords = 96;  // slack behind the packed genome for window loads

inline unsigned grid_for(uint64_t n, int block = kBlock) { return (unsigned)((n + block - 1) / block); }

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
