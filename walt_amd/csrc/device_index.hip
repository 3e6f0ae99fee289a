// device_index.hip -- makes a WALT .dbindex resident in HBM and builds the
// derived structures the mapping kernels use (layout: core.h header comment).
//
// Replaces ReadIndexHeadInfo + the per-batch, per-strand ReadIndex of the
// reference (reference.cpp:324-351,381-417; call sites mapping.cpp:437,492,
// paired.cpp:583,661): all selected strands are loaded ONCE and stay resident.
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstring>
#include <thread>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <stddef.h>

#include "device_common.h"

namespace walt {

// ---------------------------------------------------------------------------
// kernels (one element per thread; all streaming/coalesced on the outputs)
// ---------------------------------------------------------------------------

// genome bytes -> 2 bits/base.  err[0] counts bytes outside the strand alphabet.
__global__ void k_pack_genome(const uint8_t* __restrict__ bytes, uint32_t len, uint32_t ga,
                              uint32_t* __restrict__ g2, uint32_t nwords, uint32_t* __restrict__ err) {
  uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= nwords) return;
  uint64_t b0 = (uint64_t)w * 16;
  uint32_t v = 0, bad = 0;
  if (b0 + 16 <= len) {
    const uint4 q = *reinterpret_cast<const uint4*>(bytes + b0);  // hipMalloc base is 256-B aligned
    uint32_t qs[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        uint8_t c = (uint8_t)(qs[i] >> (8 * k));
        uint32_t code = base_code(c);
        bad += (code > 3) || (code == (ga ? 2u : 1u));
        v |= (code & 3u) << (2 * (4 * i + k));
      }
    }
  } else {
    for (uint32_t k = 0; k < 16; ++k) {
      if (b0 + k < len) {
        uint32_t code = base_code(bytes[b0 + k]);
        bad += (code > 3) || (code == (ga ? 2u : 1u));
        v |= (code & 3u) << (2 * k);
      }
    }
  }
  g2[w] = v;
  if (bad) atomicAdd(err, bad);
}

// Device forms of index_core.h make_ent / ent_prefix: the genome words that hold the characters are loaded
// once (these builders are bound by the number of per-lane loads, like the mapping kernels) and aligned to
// the first character by funnel shifts, so every character sits at a compile-time offset.
constexpr uint32_t kKeySpan = care_pos(kKeyWeight + kKeyChars - 1) - care_pos(kKeyWeight);  // bases between care chars 12 and 43
constexpr int kKeyWin = (int)(kKeySpan / 16) + 2;
__device__ __forceinline__ Ent make_ent_dev(const uint32_t* __restrict__ g2, uint32_t genome_len, uint32_t pos,
                                            bool& touches_end) {
  const uint64_t first = (uint64_t)pos + care_pos(kKeyWeight);
  const uint32_t* w = g2 + (first >> 4);  // g2 carries kG2PadWords of slack behind the genome
  const uint32_t sh = 2 * (uint32_t)(first & 15);
  uint32_t raw[kKeyWin + 1];
#pragma unroll
  for (int i = 0; i <= kKeyWin; ++i) raw[i] = w[i];
  uint64_t key = 0;
  touches_end = false;
#pragma unroll
  for (uint32_t p = kKeyWeight; p < kKeyWeight + kKeyChars; ++p) {
    const uint32_t off = care_pos(p) - care_pos(kKeyWeight);  // compile-time
    const uint32_t word = funnel_r(raw[off >> 4], raw[(off >> 4) + 1], sh);
    uint32_t c = (word >> (2 * (off & 15))) & 3u;
    if (first + off >= genome_len) { c = 0; touches_end = true; }
    key = (key << 2) | c;
  }
  Ent e;
  e.key_hi = (uint32_t)(key >> 32);
  e.key_lo = (uint32_t)key;
  e.pos = pos;
  return e;
}
__device__ __forceinline__ uint32_t ent_prefix_dev(const uint32_t* __restrict__ g2, const Ent& e, uint32_t ga,
                                                   uint32_t Bd) {
  const uint32_t h = hash_at_dev(g2, e.pos);
  const uint64_t key = ent_key(e);
  uint64_t acc = 0;
  uint32_t nb = 0;
  for (uint32_t i = 0; i < kKeyWeight + kKeyChars && nb < Bd; ++i) {
    const uint32_t c = i < kKeyWeight ? (h >> (2 * (kKeyWeight - 1 - i))) & 3u
                                      : (uint32_t)((key >> (2 * (kKeyWeight + kKeyChars - 1 - i))) & 3u);
    const uint32_t l = pcode_len(c, ga);
    acc = (acc << l) | pcode_bits(c, ga);
    nb += l;
  }
  return (uint32_t)(acc >> (nb - Bd));
}

// index[] -> Ent {key, pos}; entries whose care characters < 44 run over their
// chromosome's end are appended to the outlier list (core.h struct Outlier).
__global__ void k_make_ent(const uint32_t* __restrict__ g2, uint32_t genome_len,
                           const uint32_t* __restrict__ index, uint32_t n, Ent* __restrict__ ent,
                           const uint32_t* __restrict__ start, uint32_t n_chrom, Outlier* __restrict__ outl,
                           uint32_t outl_cap, uint32_t* __restrict__ err, uint32_t* __restrict__ brk, uint32_t brk_cap) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  uint32_t pos = index[j];
  if ((uint64_t)pos + kMinSeedLen > genome_len) {  // not a position makedb can emit (reference.cpp:202-203)
    atomicAdd(err + 1, 1u);
    Ent z; z.key_hi = 0; z.key_lo = 0; z.pos = 0;
    ent[j] = z;
    return;
  }
  bool touches;
  Ent e = make_ent_dev(g2, genome_len, pos, touches);
  ent[j] = e;
  const uint32_t chr = chrom_id(start, n_chrom, pos);
  const uint32_t room = start[chr + 1] - pos;
  // run breakers (core.h kTailBreakRoom): slots whose care characters -- ANY of the kNumCare a seed can have -- run
  // over their chromosome's end.  build_windows ends the dense runs at them, so that a range of slots that is dense
  // holds only entries sorted on real characters (care characters >= 44 narrowed by the verifier)
  if (room <= kTailBreakRoom) {
    const uint32_t k = atomicAdd(err + 4, 1u);
    if (k < brk_cap) brk[k] = j;
  }
  if (room <= care_pos(kKeyWeight + kKeyChars - 1)) {
    const uint32_t k = atomicAdd(err + 3, 1u);
    if (k < outl_cap) {
      Outlier o; o.h = hash_at(g2, pos); o.q = first_beyond(room); o.key_hi = e.key_hi; o.key_lo = e.key_lo;
      outl[k] = o;
    }
  }
}

// Adjacent keys of one bucket out of order: explained when one of the two is an
// outlier whose beyond-the-end characters start at or before the first character
// in which the two differ; anything else marks the whole bucket BAD.
__global__ void k_mark_unsorted(const uint32_t* __restrict__ g2, const Ent* __restrict__ ent, uint32_t n,
                                const uint32_t* __restrict__ start, uint32_t n_chrom,
                                uint32_t* __restrict__ bad) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x + 1;
  if (j >= n) return;
  Ent a = ent[j - 1], b = ent[j];
  if (ent_key(a) > ent_key(b)) {
    uint32_t ha = hash_at(g2, a.pos), hb = hash_at(g2, b.pos);
    if (ha != hb) return;
    const uint64_t x = ent_key(a) ^ ent_key(b);
    const uint32_t first_diff = kKeyWeight + (uint32_t)(__clzll((long long)x) >> 1);  // care char index
    bool explained = false;
    const Ent two[2] = {a, b};
    for (int t = 0; t < 2; ++t) {
      const uint32_t chr = chrom_id(start, n_chrom, two[t].pos);
      const uint32_t q = first_beyond(start[chr + 1] - two[t].pos);
      if (q <= first_diff) explained = true;
    }
    if (!explained) atomicOr(&bad[ha >> 5], 1u << (ha & 31));
  }
}

// the reference's counter[] must be the bucket of every entry: cnt[h] <= j < cnt[h+1]
__global__ void k_check_buckets(const uint32_t* __restrict__ g2, const Ent* __restrict__ ent, uint32_t n,
                                const uint32_t* __restrict__ cnt, uint32_t* __restrict__ err) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  uint32_t h = hash_at_dev(g2, ent[j].pos);
  if (!(cnt[h] <= j && j < cnt[h + 1])) atomicAdd(err + 2, 1u);
}

// Reversed directory, step 1: dir[S - v] = smallest index slot whose code prefix
// is v (only run starts need to write; atomicMin because BAD buckets are not
// monotone).  Step 2 (host driver): inclusive running minimum over dir[0..S].
__global__ void k_fill_u32(uint32_t* __restrict__ p, uint64_t n, uint32_t v) {  // grid-stride: n may exceed 2^32
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    p[i] = v;
}
// slot table (core.h StrandView::tab) from the finished directory: record t belongs to slot t + 1
// Fence keys (core.h StrandView::fen): thread i copies the key of entry 16 i to level 1, and to the levels above
// when i is a multiple of 16, 256, 4096.
__global__ void k_make_fences(const Ent* __restrict__ ent, uint32_t index_size, uint32_t* __restrict__ f1,
                              uint32_t* __restrict__ f2, uint32_t* __restrict__ f3, uint32_t* __restrict__ f4) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i * 16 >= index_size) return;
  const Ent e = ent[i * 16];
  f1[2 * i] = e.key_hi; f1[2 * i + 1] = e.key_lo;
  if ((i & 15u) == 0) { f2[2 * (i >> 4)] = e.key_hi; f2[2 * (i >> 4) + 1] = e.key_lo; }
  if ((i & 255u) == 0) { f3[2 * (i >> 8)] = e.key_hi; f3[2 * (i >> 8) + 1] = e.key_lo; }
  if ((i & 4095u) == 0) { f4[2 * (i >> 12)] = e.key_hi; f4[2 * (i >> 12) + 1] = e.key_lo; }
}

__global__ void k_make_table(const uint32_t* __restrict__ dir, const Ent* __restrict__ ent, uint64_t slots,
                             uint32_t* __restrict__ tab) {
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < slots; t += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t hi = dir[t], lo = dir[t + 1];
    const uint32_t ne = hi > lo ? hi - lo : 0u;
    uint32_t w0 = lo, w1 = ne, w2 = kTabMulti;
    if (ne == 1) {
      const Ent e = ent[lo];
      w0 = e.key_hi; w1 = e.key_lo; w2 = e.pos;
    }
    tab[3 * t] = w0; tab[3 * t + 1] = w1; tab[3 * t + 2] = w2;
  }
}
// Directory (core.h StrandView::dir), reversed: dir[S - v] = R[v] = 1 + the largest index slot whose code prefix is
// below v (0 when there is none).  Step 1: every entry that ends a run of equal prefixes u writes its index + 1 at R's
// position u + 1 (atomicMax: an entry out of place -- a chromosome-end entry, a BAD bucket -- must not lower it).
// Step 2: running maximum over rising v, i.e. over FALLING array index (k_rmax_*).
__global__ void k_dir_scatter(const uint32_t* __restrict__ g2, const Ent* __restrict__ ent, uint32_t n, uint32_t ga,
                              uint32_t Bd, uint32_t* __restrict__ dir) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const uint32_t v = ent_prefix_dev(g2, ent[j], ga, Bd);
  // the successor's prefix comes from the neighbouring lane (the last lane of a wave computes it)
  uint32_t nv = __shfl_down(v, 1);
  if (((threadIdx.x & 63) == 63 || j + 1 == n) && j + 1 < n) nv = ent_prefix_dev(g2, ent[j + 1], ga, Bd);
  if (j + 1 == n || nv != v) atomicMax(&dir[(1ull << Bd) - 1ull - v], j + 1);
}
// running maximum from the array's end towards its start, in segments of kRmaxSeg elements:
// 1. the maximum of every segment, 2. seg[b] := max of the segments behind b, 3. the scan inside each segment
constexpr uint32_t kRmaxSeg = 1u << 16;
__global__ void k_rmax_reduce(const uint32_t* __restrict__ p, uint64_t total, uint32_t* __restrict__ seg) {
  const uint64_t base = (uint64_t)blockIdx.x * kRmaxSeg;
  uint32_t m = 0;
  for (uint64_t i = base + threadIdx.x; i < base + kRmaxSeg && i < total; i += blockDim.x) m = p[i] > m ? p[i] : m;
  __shared__ uint32_t red[kBlock];
  red[threadIdx.x] = m;
  __syncthreads();
  for (uint32_t s = kBlock / 2; s; s >>= 1) {
    if (threadIdx.x < s && red[threadIdx.x + s] > red[threadIdx.x]) red[threadIdx.x] = red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) seg[blockIdx.x] = red[0];
}
__global__ void k_rmax_carry(uint32_t* __restrict__ seg, uint32_t n_seg) {  // one thread: a few ten thousand segments
  uint32_t run = 0;
  for (uint32_t b = n_seg; b-- > 0;) {
    const uint32_t own = seg[b];
    seg[b] = run;
    run = own > run ? own : run;
  }
}
__global__ void k_rmax_apply(uint32_t* __restrict__ p, uint64_t total, const uint32_t* __restrict__ seg) {
  // thread t owns the kRmaxSeg / kBlock consecutive elements [lo, hi) of the block's segment
  constexpr uint32_t kPer = kRmaxSeg / kBlock;
  const uint64_t base = (uint64_t)blockIdx.x * kRmaxSeg + (uint64_t)threadIdx.x * kPer;
  uint32_t m = 0;
  for (uint32_t k = 0; k < kPer; ++k) {
    const uint64_t i = base + k;
    if (i < total && p[i] > m) m = p[i];
  }
  __shared__ uint32_t part[kBlock];
  part[threadIdx.x] = m;
  __syncthreads();
  if (threadIdx.x == 0) {  // suffix maxima over the threads' parts (256 of them)
    uint32_t run = seg[blockIdx.x];
    for (uint32_t t = kBlock; t-- > 0;) {
      const uint32_t own = part[t];
      part[t] = run;
      run = own > run ? own : run;
    }
  }
  __syncthreads();
  uint32_t run = part[threadIdx.x];
  for (uint32_t k = kPer; k-- > 0;) {
    const uint64_t i = base + k;
    if (i < total) {
      const uint32_t own = p[i];
      run = own > run ? own : run;
      p[i] = run;
    }
  }
}

__global__ void k_popcount(const uint32_t* __restrict__ words, uint32_t n, unsigned long long* __restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t c = i < n ? __popc(words[i]) : 0;
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, (unsigned long long)c);
}

// ---------------------------------------------------------------------------
// host drivers
// ---------------------------------------------------------------------------
template <typename T>
static int dev_alloc(walt_index* idx, T** p, uint64_t count) {
  void* q = nullptr;
  uint64_t bytes = count * sizeof(T);
  if (bytes == 0) bytes = sizeof(T);
  hipError_t e = hipMalloc(&q, bytes);
  if (e != hipSuccess) return fail(WALT_ENOMEM, std::string("hipMalloc failed: ") + hipGetErrorString(e));
  idx->allocs.push_back(q);
  idx->device_bytes += bytes;
  *p = reinterpret_cast<T*>(q);
  return WALT_OK;
}

int choose_dir_bits(uint64_t max_index_size, int requested, int n_strands, uint64_t device_bytes) {
  if (requested >= 0) return requested > (int)kMaxDirBits ? (int)kMaxDirBits : requested < (int)kMinDirBits ? (int)kMinDirBits : requested;
  // smallest Bd with index_size / 2^Bd <= 2 entries per directory slot (slots of
  // up to kScan entries are searched with independent loads, core.h) ...
  int B = (int)kMinDirBits;
  while (B < 31 && max_index_size > (2ull << B)) ++B;
  // ... and one bit more (2^32 slots, 17 GB per strand, 0.72 entries per slot at hg19 scale) when the device
  // has the room: half of the slots a non-matching probe lands in are then empty and cost no entry line
  // (single-end pass 1: 12.9 -> 11.5 ms; paired-end: 875 -> 958 M pairs/s).  Room = all strands resident plus
  // 30 GB for batches, and the GPU builder's peak while it sorts the last strand (24 bytes of keys and
  // positions per entry, 4 of slack) beside the strands already finished.
  if (B == 31 && max_index_size > (1ull << 31)) {
    const uint64_t strand = 12ull * max_index_size + (4ull << 32) + max_index_size / 3 + (80ull << 20) +
                            8ull * max_index_size / 15;  // entries, directory, genome and bitmaps, small tables, fence keys
    // (all four strands = paired-end use: the roomy paired-end workspace and its batch want ~80 GB, map_pe.hip pe_roomy;
    // 2^31 slots cost the paired-end pass 1 a per cent, the larger passes and single rounds win ten)
    const uint64_t resident = (uint64_t)n_strands * strand + (n_strands >= 4 ? 90ull << 30 : 30ull << 30);
    const uint64_t build_peak = (uint64_t)(n_strands - 1) * strand + 28ull * max_index_size;
    if (resident <= device_bytes && build_peak <= device_bytes) B = 32;
  }
  return B;
}

int alloc_strand_g2(walt_index* idx, uint32_t** g2_out, hipStream_t stream) {
  const uint32_t nwords = (idx->head.genome_len + 15) / 16;
  int rc = dev_alloc(idx, g2_out, (uint64_t)nwords + kG2PadWords);
  if (rc) return rc;
  WALT_HIP(hipMemsetAsync(*g2_out + nwords, 0, kG2PadWords * sizeof(uint32_t), stream));
  return WALT_OK;
}

// Derived structures of one strand from its packed genome (already owned by
// idx) and the DEVICE-resident counter[] / index[] arrays.
int finish_strand_device(walt_index* idx, int strand, uint32_t* g2, const uint32_t* d_counter,
                         const uint32_t* d_index, uint32_t index_size, hipStream_t stream) {
  const uint32_t genome_len = idx->head.genome_len;
  const uint32_t ga = strand >= 2 ? 1u : 0u;
  const uint32_t Bd = idx->view.dir_bits;
  const uint64_t slots = 1ull << Bd;
  StrandView& sv = idx->view.s[strand];
  uint32_t *cnt = nullptr, *bad = nullptr, *dir = nullptr, *err = nullptr;
  uint64_t* bloom = nullptr;
  uint32_t* pre = nullptr;
  uint32_t bloom_blocks = 0;
  Ent* ent = nullptr;
  int rc;
  if ((rc = dev_alloc(idx, &cnt, (uint64_t)kNumBuckets + 1))) return rc;
  if ((rc = dev_alloc(idx, &bad, kNumBuckets / 32))) return rc;
  if ((rc = dev_alloc(idx, &dir, (uint64_t)slots + 1))) return rc;
  if ((rc = dev_alloc(idx, &ent, (uint64_t)index_size + 1))) return rc;
  struct Scoped {  // temporaries of this function: freed on every return path
    void* p = nullptr;
    ~Scoped() { if (p) hipFree(p); }
  } err_buf, tmp_buf, cnt_buf;
  WALT_HIP(hipMalloc(&err_buf.p, 8 * sizeof(uint32_t)));
  err = reinterpret_cast<uint32_t*>(err_buf.p);
  WALT_HIP(hipMemsetAsync(err, 0, 8 * sizeof(uint32_t), stream));
  WALT_HIP(hipMemsetAsync(bad, 0, kNumBuckets / 8, stream));
  WALT_HIP(hipMemcpyAsync(cnt, d_counter, ((uint64_t)kNumBuckets + 1) * 4, hipMemcpyDeviceToDevice, stream));
  const uint32_t n_chrom = (uint32_t)idx->head.lengths.size();
  const uint32_t outl_cap = n_chrom * 100 + 16;  // <= 94 positions per chromosome have room in [37, 130]
  Outlier* outl = nullptr;
  if ((rc = dev_alloc(idx, &outl, outl_cap))) return rc;
  const uint32_t brk_cap = n_chrom * (kTailBreakRoom + 1) + 16;  // at most one indexed position per base of a chromosome's last kTailBreakRoom
  uint32_t* brk = nullptr;
  if ((rc = dev_alloc(idx, &brk, brk_cap))) return rc;
  if (index_size) {
    hipLaunchKernelGGL(k_make_ent, dim3(grid_for(index_size)), dim3(kBlock), 0, stream, g2, genome_len, d_index,
                       index_size, ent, idx->view.start_index, n_chrom, outl, outl_cap, err, brk, brk_cap);
  }
  uint32_t herr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  WALT_HIP(hipMemcpyAsync(herr, err, sizeof(herr), hipMemcpyDeviceToHost, stream));
  WALT_HIP(hipStreamSynchronize(stream));
  if (herr[1]) {
    return fail(WALT_EFORMAT, "strand index " + std::to_string(strand) + ": " + std::to_string(herr[1]) +
                                  " index positions beyond the genome");
  }
  // fence keys of the long-slot search (WALT_AMD_FENCE=0: none, the k-ary search over the entries is used; A/B runs)
  uint32_t* fen[kFenceLevels] = {nullptr, nullptr, nullptr, nullptr};
  {
    const char* e = getenv("WALT_AMD_FENCE");
    if (index_size && !(e && atoi(e) == 0)) {
      // best effort, like the dense windows: the long-slot search works without them (core.h slot_kary_search), so a
      // device that holds the index with little room to spare opens it as before the fences existed
      bool have = true;
      const size_t n_before = idx->allocs.size();
      const uint64_t bytes_before = idx->device_bytes;
      for (uint32_t k = 0; k < kFenceLevels && have; ++k) {
        const uint64_t n_k = (((uint64_t)index_size - 1) >> (4 * (k + 1))) + 1;
        have = dev_alloc(idx, &fen[k], 2 * n_k + 32) == WALT_OK;  // (+ slack: a clamped pivot never reads beyond, a whole line may)
      }
      if (have) {
        hipLaunchKernelGGL(k_make_fences, dim3(grid_for(((uint64_t)index_size + 15) / 16)), dim3(kBlock), 0, stream, ent, index_size,
                           fen[0], fen[1], fen[2], fen[3]);
      } else {
        (void)hipGetLastError();
        while (idx->allocs.size() > n_before) { (void)hipFree(idx->allocs.back()); idx->allocs.pop_back(); }
        idx->device_bytes = bytes_before;
        for (uint32_t k = 0; k < kFenceLevels; ++k) fen[k] = nullptr;
        if (getenv("WALT_AMD_VERBOSE")) fprintf(stderr, "[walt_amd index: strand %d: no fence keys (device memory)]\n", strand);
      }
    }
  }
  if (index_size) {
    hipLaunchKernelGGL(k_check_buckets, dim3(grid_for(index_size)), dim3(kBlock), 0, stream, g2, ent, index_size,
                       cnt, err);
    if (index_size > 1)
      hipLaunchKernelGGL(k_mark_unsorted, dim3(grid_for(index_size - 1)), dim3(kBlock), 0, stream, g2, ent,
                         index_size, idx->view.start_index, n_chrom, bad);
  }
  // reversed directory: fill with 0, scatter run ends, running maximum towards the array's start
  hipLaunchKernelGGL(k_fill_u32, dim3(1u << 16), dim3(kBlock), 0, stream, dir, (uint64_t)slots + 1, 0u);
  if (index_size)
    hipLaunchKernelGGL(k_dir_scatter, dim3(grid_for(index_size)), dim3(kBlock), 0, stream, g2, ent, index_size, ga,
                       Bd, dir);
  {
    const uint64_t total = slots + 1;
    const uint32_t n_seg = (uint32_t)((total + kRmaxSeg - 1) / kRmaxSeg);
    WALT_HIP(hipMalloc(&tmp_buf.p, (size_t)n_seg * 4 + 16));
    uint32_t* seg = reinterpret_cast<uint32_t*>(tmp_buf.p);
    hipLaunchKernelGGL(k_rmax_reduce, dim3(n_seg), dim3(kBlock), 0, stream, dir, total, seg);
    hipLaunchKernelGGL(k_rmax_carry, dim3(1), dim3(1), 0, stream, seg, n_seg);
    hipLaunchKernelGGL(k_rmax_apply, dim3(n_seg), dim3(kBlock), 0, stream, dir, total, seg);
    WALT_HIP(hipStreamSynchronize(stream));
  }
  // Slot table (opt-in, WALT_AMD_TABLE=1): measured at hg19 scale with 2^32 slots it takes pass 1 from 11.45
  // to 10.9 ms (+5 % reads/s) for 51.5 GB more per strand -- not enough to be the default; it needs the device
  // to keep >= 48 GB free afterwards (the other strand's builder temporaries and the batches).
  uint32_t* tab = nullptr;
  {
    const char* e = getenv("WALT_AMD_TABLE");
    const uint64_t tab_bytes = 12ull * slots;
    bool want = e && atoi(e) != 0;
    if (want && Bd >= 30) {
      size_t free_b = 0, total_b = 0;
      WALT_HIP(hipMemGetInfo(&free_b, &total_b));
      want = free_b >= tab_bytes + (48ull << 30);
    }
    if (want && index_size) {
      if ((rc = dev_alloc(idx, &tab, 3ull * slots))) return rc;
      hipLaunchKernelGGL(k_make_table, dim3(1u << 16), dim3(kBlock), 0, stream, dir, ent, slots, tab);
      WALT_HIP(hipStreamSynchronize(stream));
      WALT_HIP(hipGetLastError());
    }
  }
  WALT_HIP(hipMalloc(&cnt_buf.p, sizeof(unsigned long long)));
  unsigned long long* d_cnt64 = reinterpret_cast<unsigned long long*>(cnt_buf.p);
  WALT_HIP(hipMemsetAsync(d_cnt64, 0, sizeof(unsigned long long), stream));
  hipLaunchKernelGGL(k_popcount, dim3(grid_for(kNumBuckets / 32)), dim3(kBlock), 0, stream, bad, kNumBuckets / 32,
                     d_cnt64);
  unsigned long long nbad = 0;
  WALT_HIP(hipMemcpyAsync(&nbad, d_cnt64, sizeof(nbad), hipMemcpyDeviceToHost, stream));
  WALT_HIP(hipMemcpyAsync(herr, err, sizeof(herr), hipMemcpyDeviceToHost, stream));
  WALT_HIP(hipStreamSynchronize(stream));
  WALT_HIP(hipGetLastError());
  if (herr[2])
    return fail(WALT_EFORMAT, "strand index " + std::to_string(strand) + ": " + std::to_string(herr[2]) +
                                  " index entries are not in the bucket of their hash");
  // outliers: sort by bucket on the host (a few per chromosome end); Bloom filter of the
  // (bucket, char 12, char 13) keys a dangerous probe can have (core.h)
  const uint32_t n_outl = herr[3] < outl_cap ? herr[3] : outl_cap;
  if (herr[3] > outl_cap) return fail(WALT_EFORMAT, "more chromosome-end entries than a makedb index can hold");
  {
    std::vector<Outlier> ho(n_outl);
    if (n_outl) WALT_HIP(hipMemcpy(ho.data(), outl, n_outl * sizeof(Outlier), hipMemcpyDeviceToHost));
    std::sort(ho.begin(), ho.end(), [](const Outlier& x, const Outlier& y) {
      if (x.h != y.h) return x.h < y.h;
      if (x.q != y.q) return x.q < y.q;
      if (x.key_hi != y.key_hi) return x.key_hi < y.key_hi;
      return x.key_lo < y.key_lo;
    });
    // the level table (core.h StrandView::olev): which outliers a probe shares its characters with, by look-up
    std::vector<OlevEnt> lev;
    std::vector<uint32_t> collided;
    // (the genome's last words on the host: which buckets hold an entry that reads beyond the genome's end)
    std::vector<uint32_t> tail_words;
    uint64_t tail_base = 0;
    {
      const uint32_t reach = care_pos(kKeyWeight + kKeyChars - 1) + 16;
      tail_base = genome_len > reach ? ((uint64_t)(genome_len - reach) & ~15ull) : 0;
      const uint64_t w0 = tail_base >> 4, w1 = ((uint64_t)genome_len + 15) / 16 + 4;  // (+ slack words: g2 carries kG2PadWords)
      tail_words.resize((size_t)(w1 - w0));
      WALT_HIP(hipMemcpy(tail_words.data(), g2 + w0, tail_words.size() * 4, hipMemcpyDeviceToHost));
    }
    const uint32_t ents = build_outlier_levels(ho.data(), n_outl, lev, collided,
                                               beyond_genome_buckets(tail_words.data(), tail_base, genome_len));
    for (uint32_t hb : collided) {  // two keys, one fingerprint: their buckets are searched literally (never seen)
      uint32_t word = 0;
      WALT_HIP(hipMemcpy(&word, bad + (hb >> 5), 4, hipMemcpyDeviceToHost));
      if (!((word >> (hb & 31)) & 1u)) ++nbad;
      word |= 1u << (hb & 31);
      WALT_HIP(hipMemcpy(bad + (hb >> 5), &word, 4, hipMemcpyHostToDevice));
    }
    OlevEnt* d_lev = nullptr;
    if ((rc = dev_alloc(idx, &d_lev, (uint64_t)ents))) return rc;
    WALT_HIP(hipMemcpy(d_lev, lev.data(), (size_t)ents * sizeof(OlevEnt), hipMemcpyHostToDevice));
    sv.olev = d_lev;
    sv.olev_mask = ents - 1;
    // filter keys: see core.h; the filter is sized by their number
    std::vector<uint32_t> keys;
    std::vector<std::pair<uint32_t, uint32_t>> keys2;  // (16-character key: block and prefilter, 20-character key: bits)
    for (const Outlier& o : ho) {
      const uint32_t own = o.key_hi >> (32 - 2 * kBloomChars);  // care characters 12..15, MSB first
      if (o.q >= kKeyWeight + kBloomChars2) {  // second level (core.h bloom_key2): its characters 12..19 are real
        const uint32_t k16 = bloom_key(o.h, own);
        keys2.push_back(std::make_pair(k16, bloom_key2(k16, (o.key_hi >> (32 - 2 * kBloomChars2)) & 0xFFu)));
        continue;
      }
      // characters from q on take every value (q == 12 + j: the low 2 * (kBloomChars - j) bits are free)
      const uint32_t real = o.q <= kKeyWeight ? 0u : (o.q - kKeyWeight < kBloomChars ? o.q - kKeyWeight : kBloomChars);
      const uint32_t free_bits = 2 * (kBloomChars - real);
      const uint32_t fixed = free_bits >= 8 ? 0u : (own >> free_bits) << free_bits;
      // of the values the free characters can take, only those whose FIRST free character (index q) does not
      // exceed the outlier's real byte there can belong to a dangerous probe (core.h probe_is_dangerous)
      const uint32_t xq = real < kBloomChars ? (o.key_hi >> (30 - 2 * real)) & 3u : 0u;
      for (uint32_t v = 0; v < (1u << free_bits); ++v) {
        if (free_bits && (v >> (free_bits - 2)) > xq) continue;
        keys.push_back(bloom_key(o.h, fixed | v));
      }
    }
    if (nbad) {  // buckets with unexplained disorder: every probe into them is dangerous
      std::vector<uint32_t> bm(kNumBuckets / 32);
      WALT_HIP(hipMemcpy(bm.data(), bad, kNumBuckets / 8, hipMemcpyDeviceToHost));
      for (uint32_t w = 0; w < kNumBuckets / 32; ++w)
        for (uint32_t bit = 0; bit < 32; ++bit)
          if ((bm[w] >> bit) & 1u)
            for (uint32_t v = 0; v < (1u << (2 * kBloomChars)); ++v) keys.push_back(bloom_key(w * 32 + bit, v));
    }
    bloom_blocks = bloom_blocks_for(keys.size() + keys2.size());
    std::vector<uint64_t> hb(bloom_blocks, 0);
    for (uint32_t k : keys) bloom_insert(hb.data(), bloom_blocks - 1, k);
    for (const auto& k : keys2) hb[bloom_block(k.first, bloom_blocks - 1)] |= bloom_bits(k.second);
    if (getenv("WALT_AMD_VERBOSE")) {
      uint64_t bits = 0;
      for (uint64_t w : hb) bits += (uint64_t)__builtin_popcountll(w);
      fprintf(stderr, "[walt_amd index: strand %d: %u outliers, %zu filter keys, %u Bloom blocks, %.2f %% of its bits set]\n",
              strand, n_outl, keys.size() + keys2.size(), bloom_blocks, 100.0 * (double)bits / (64.0 * bloom_blocks));
    }
    if ((rc = dev_alloc(idx, &bloom, (uint64_t)bloom_blocks))) return rc;
    std::vector<uint32_t> hp(kPreBits / 32, 0);
    for (uint32_t k : keys) hp[pre_hash(k) >> 5] |= 1u << (pre_hash(k) & 31);
    for (const auto& k : keys2) hp[pre_hash(k.first) >> 5] |= 1u << (pre_hash(k.first) & 31);
    if ((rc = dev_alloc(idx, &pre, (uint64_t)kPreBits / 32))) return rc;
    WALT_HIP(hipMemcpy(pre, hp.data(), kPreBits / 8, hipMemcpyHostToDevice));
    if (n_outl) WALT_HIP(hipMemcpy(outl, ho.data(), n_outl * sizeof(Outlier), hipMemcpyHostToDevice));
    WALT_HIP(hipMemcpy(bloom, hb.data(), (size_t)bloom_blocks * 8, hipMemcpyHostToDevice));
    std::vector<uint32_t> od;
    const uint32_t pairs = build_outlier_dir(ho.data(), n_outl, od);
    uint32_t* d_od = nullptr;
    if ((rc = dev_alloc(idx, &d_od, (uint64_t)od.size()))) return rc;
    WALT_HIP(hipMemcpy(d_od, od.data(), od.size() * 4, hipMemcpyHostToDevice));
    sv.outl_dir = d_od;
    sv.outl_dir_mask = pairs - 1;
  }
  idx->bad_buckets[strand] = nbad;
  idx->outliers[strand] = n_outl;
  if (herr[4] > brk_cap) return fail(WALT_EFORMAT, "more chromosome-end entries than a makedb index can hold");
  idx->brk[strand] = brk;
  idx->n_brk[strand] = herr[4];
  sv.outl = outl; sv.n_outl = n_outl;
  sv.g2 = g2; sv.cnt = cnt; sv.bad = bad; sv.dir = dir; sv.ent = ent;
  sv.index_size = index_size; sv.genome_len = genome_len; sv.ga = ga; sv.bloom = bloom; sv.bloom_mask = bloom_blocks - 1; sv.pre = pre; sv.tab = tab;
  sv.wbits = nullptr; sv.wrank = nullptr; sv.win = nullptr; sv.win2 = nullptr; sv.wcap = 0;  // build_windows, once every strand is resident
  for (uint32_t k = 0; k < kFenceLevels; ++k) sv.fen[k] = fen[k];
  idx->strand_mask |= 1u << strand;
  return WALT_OK;
}

int build_strand_device(walt_index* idx, int strand, const uint8_t* d_bytes, const uint32_t* d_counter,
                        const uint32_t* d_index, uint32_t index_size, hipStream_t stream) {
  const uint32_t genome_len = idx->head.genome_len;
  const uint32_t ga = strand >= 2 ? 1u : 0u;
  const uint32_t nwords = (genome_len + 15) / 16;
  uint32_t *g2 = nullptr, *err = nullptr;
  int rc = alloc_strand_g2(idx, &g2, stream);
  if (rc) return rc;
  WALT_HIP(hipMalloc(reinterpret_cast<void**>(&err), sizeof(uint32_t)));
  WALT_HIP(hipMemsetAsync(err, 0, sizeof(uint32_t), stream));
  if (nwords)
    hipLaunchKernelGGL(k_pack_genome, dim3(grid_for(nwords)), dim3(kBlock), 0, stream, d_bytes, genome_len, ga, g2,
                       nwords, err);
  uint32_t herr = 0;
  WALT_HIP(hipMemcpyAsync(&herr, err, sizeof(herr), hipMemcpyDeviceToHost, stream));
  WALT_HIP(hipStreamSynchronize(stream));
  hipFree(err);
  if (herr)
    return fail(WALT_EFORMAT, "strand index " + std::to_string(strand) + ": " + std::to_string(herr) +
                                  " genome bytes outside the converted alphabet");
  return finish_strand_device(idx, strand, g2, d_counter, d_index, index_size, stream);
}

// ---------------------------------------------------------------------------
// Dense candidate windows (core.h StrandView::wbits / wrank / win / win2)
// ---------------------------------------------------------------------------
// eq bit j = slots j and j + 1 belong to the same run: same bucket and same first kWinKeyChars key characters, i.e. a
// 100-base read whose region holds one holds the other (a chromosome-end entry out of order only changes which
// slots are chosen, never what a record holds).  One wavefront per 64-slot word.
__global__ void k_win_eq(const uint32_t* __restrict__ g2, const Ent* __restrict__ ent, uint32_t index_size,
                         unsigned long long* __restrict__ eq) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool same = false;
  if (t + 1 < index_size) {
    const Ent a = ent[t], z = ent[t + 1];
    if (((ent_key(a) ^ ent_key(z)) >> (64 - 2 * kWinKeyChars)) == 0) same = hash_at_dev(g2, a.pos) == hash_at_dev(g2, z.pos);
  }
  const unsigned long long m = __ballot(same);
  if ((threadIdx.x & 63) == 0 && t < index_size) eq[t >> 6] = m;
}
// eq bits j - 1 and j of every breaker slot j cleared: the slot is a run of its own
__global__ void k_win_break(const uint32_t* __restrict__ brk, uint32_t n, unsigned long long* __restrict__ eq) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const uint32_t j = brk[t];
  atomicAnd(&eq[j >> 6], ~(1ull << (j & 63u)));
  if (j) atomicAnd(&eq[(j - 1) >> 6], ~(1ull << ((j - 1) & 63u)));
}
// bitmap of the slots with dense records: every slot of a run of at least kWinMinRun slots.  The run of slot j
// reaches L slots down and R slots up, each counted to 16 on the eq bits around j: one thread per 64-slot word.
__global__ void k_win_bits(const unsigned long long* __restrict__ eq, uint32_t nw, unsigned long long* __restrict__ bits) {
  static_assert(kWinMinRun >= 2 && kWinMinRun <= 33, "run lengths are counted on 16 bits either side");
  const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= nw) return;
  const unsigned long long prev = w ? eq[w - 1] : 0ull, cur = eq[w], next = w + 1 < nw ? eq[w + 1] : 0ull;
  unsigned long long out = 0;
  for (uint32_t b = 0; b < 64; ++b) {
    const unsigned long long up = b ? (cur >> b) | (next << (64 - b)) : cur;        // bit 0: eq of slot j
    const unsigned long long dn = b >= 16 ? cur >> (b - 16) : (cur << (16 - b)) | (prev >> (48 + b));  // bit 15: eq of slot j - 1
    const uint32_t nu = (uint32_t)(~up) & 0xFFFFu, nd = (uint32_t)(~dn) & 0xFFFFu;
    const uint32_t R = nu ? (uint32_t)__ffs((int)nu) - 1u : 16u;
    const uint32_t L = nd ? (uint32_t)__clz((int)(nd << 16)) : 16u;
    if (L + R + 1 >= kWinMinRun) out |= 1ull << b;
  }
  bits[w] = out;
}
__global__ void k_win_popc(const unsigned long long* __restrict__ bits, uint32_t nw, uint32_t* __restrict__ cnt) {
  const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w < nw) cnt[w] = (uint32_t)__popcll(bits[w]);
}
// one thread per index slot: {pos, 112 bases from pos - kWinLead} into win, the next 64 bases into win2.  The bases
// are the g2 bits count_mismatch reads for a candidate at pos - seed_i (g2 carries kG2PadWords of slack);
// bases in front of the genome's first (pos < kWinLead) are never compared and are stored as 0.
__global__ void k_win_fill(const uint32_t* __restrict__ g2, const Ent* __restrict__ ent, uint32_t index_size,
                           const unsigned long long* __restrict__ bits, const uint32_t* __restrict__ rank, uint32_t cap,
                           uint32_t* __restrict__ win, uint32_t* __restrict__ win2) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= index_size) return;
  const unsigned long long wbits = bits[t >> 6];
  if (!((wbits >> (t & 63u)) & 1ull)) return;
  const uint32_t rec = rank[t >> 6] + (uint32_t)__popcll(wbits & ((1ull << (t & 63u)) - 1ull));
  if (rec >= cap) return;
  const uint32_t pos = ent[t].pos;
  constexpr int kAll = (int)(kWinWords + kWinWords2);
  uint32_t w[kAll];
  const int64_t start = (int64_t)pos - (int64_t)kWinLead;
#pragma unroll
  for (int i = 0; i < kAll; ++i) {
    const int64_t q = start + 16 * i;
    if (q >= 0) {
      const uint64_t x = (uint64_t)q >> 4;
      w[i] = funnel_r(g2[x], g2[x + 1], 2 * (uint32_t)(q & 15));
    } else {
      w[i] = g2[0] << (2 * (uint32_t)(-q));  // only i == 0: 16 > kWinLead
    }
  }
  uint4 a, c, e;
  a.x = pos; a.y = w[0]; a.z = w[1]; a.w = w[2];
  c.x = w[3]; c.y = w[4]; c.z = w[5]; c.w = w[6];
  e.x = w[7]; e.y = w[8]; e.z = w[9]; e.w = w[10];
  reinterpret_cast<uint4*>(win)[2ull * rec] = a;
  reinterpret_cast<uint4*>(win)[2ull * rec + 1] = c;
  reinterpret_cast<uint4*>(win2)[rec] = e;
}

// Windows for every resident strand, within what the device can spare: WALT_AMD_WIN_GB (default 16) per strand,
// and never more than the free memory minus a reserve for the batches (40 GB, or WALT_AMD_WIN_RESERVE_GB).
// WALT_AMD_WIN=0 switches them off (A/B measurements).
static int build_windows(walt_index* idx) {
  static_assert(kWinWords + kWinWords2 == 11 && kWinLead < 16, "record layout of k_win_fill");
  if (const char* e = getenv("WALT_AMD_WIN")) if (atoi(e) == 0) return WALT_OK;
  int n_strands = 0;
  for (int s = 0; s < 4; ++s) n_strands += (idx->strand_mask >> s) & 1u;
  if (!n_strands) return WALT_OK;
  double cap_gb = 16.0, reserve_gb = 40.0;
  if (const char* e = getenv("WALT_AMD_WIN_GB")) cap_gb = atof(e);
  if (const char* e = getenv("WALT_AMD_WIN_RESERVE_GB")) reserve_gb = atof(e);
  hipStream_t stream = nullptr;
  for (int s = 0; s < 4; ++s) {
    if (!((idx->strand_mask >> s) & 1u)) continue;
    StrandView& sv = idx->view.s[s];
    const uint32_t nw = (uint32_t)(((uint64_t)sv.index_size + 63) >> 6);
    if (nw == 0) { --n_strands; continue; }
    size_t free_b = 0, total_b = 0;
    WALT_HIP(hipMemGetInfo(&free_b, &total_b));
    // a small index (tests, bacterial genomes) needs no reserve: its batches are small too
    const double reserve = std::min(reserve_gb * 1e9, 0.25 * (double)total_b);
    double budget = ((double)free_b - reserve - 24.0 * nw) / n_strands;  // bitmap, ranks and the builder's temporaries
    budget = std::min(budget, cap_gb * 1e9);
    --n_strands;
    const uint64_t rec_bytes = 4ull * (kWinWords + 1 + kWinWords2);
    if (budget < 1024.0 * rec_bytes) continue;
    // record numbers must fit 32 bits: the mapping kernels pass them between lanes as one word
    const uint32_t cap_recs = (uint32_t)std::min<double>(budget / (double)rec_bytes, 4.0e9);
    // The windows are an optional accelerator: when memory for them cannot be had after all (the budget came from
    // hipMemGetInfo a moment ago; another process or a caching allocator may have taken it since) the strand simply
    // has none and its large regions are verified from ent[] + g2[] -- same results, slower.
    struct Scoped {  // temporaries: freed on every path
      void* p = nullptr;
      ~Scoped() { if (p) (void)hipFree(p); }
    } flag_buf, cnt_buf, tmp_buf, bits_buf, rank_buf, win_buf, win2_buf;
    auto give_up = [&](const char* what) {
      if (getenv("WALT_AMD_VERBOSE")) fprintf(stderr, "[walt_amd index: strand %d: no dense candidate windows (%s)]\n", s, what);
      (void)hipGetLastError();
    };
    if (hipMalloc(&bits_buf.p, ((uint64_t)nw + 1) * 8) != hipSuccess || hipMalloc(&rank_buf.p, ((uint64_t)nw + 1) * 4) != hipSuccess ||
        hipMalloc(&flag_buf.p, ((uint64_t)nw + 1) * 8) != hipSuccess) { give_up("bitmap allocation failed"); continue; }
    unsigned long long* bits = reinterpret_cast<unsigned long long*>(bits_buf.p);
    unsigned long long* flag = reinterpret_cast<unsigned long long*>(flag_buf.p);
    uint32_t* rank = reinterpret_cast<uint32_t*>(rank_buf.p);
    hipLaunchKernelGGL(k_win_eq, dim3((unsigned)(((uint64_t)nw * 64 + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, sv.g2, sv.ent,
                       sv.index_size, flag);
    if (idx->n_brk[s])  // no run crosses a chromosome-end entry (k_make_ent)
      hipLaunchKernelGGL(k_win_break, dim3(grid_for(idx->n_brk[s])), dim3(kBlock), 0, stream, idx->brk[s], idx->n_brk[s], flag);
    hipLaunchKernelGGL(k_win_bits, dim3(grid_for(nw)), dim3(kBlock), 0, stream, flag, nw, bits);
    WALT_HIP(hipStreamSynchronize(stream));
    (void)hipFree(flag_buf.p);
    flag_buf.p = nullptr;
    if (hipMalloc(&cnt_buf.p, ((uint64_t)nw + 1) * 4) != hipSuccess) { give_up("temporary allocation failed"); continue; }
    uint32_t* cnt = reinterpret_cast<uint32_t*>(cnt_buf.p);
    hipLaunchKernelGGL(k_win_popc, dim3(grid_for(nw)), dim3(kBlock), 0, stream, bits, nw, cnt);
    size_t tmp_bytes = 0;
    WALT_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, cnt, rank, 0u, (size_t)nw, rocprim::plus<uint32_t>(), stream));
    if (hipMalloc(&tmp_buf.p, tmp_bytes ? tmp_bytes : 16) != hipSuccess) { give_up("temporary allocation failed"); continue; }
    WALT_HIP(rocprim::exclusive_scan(tmp_buf.p, tmp_bytes, cnt, rank, 0u, (size_t)nw, rocprim::plus<uint32_t>(), stream));
    uint32_t last_rank = 0, last_cnt = 0;
    WALT_HIP(hipMemcpyAsync(&last_rank, rank + (nw - 1), 4, hipMemcpyDeviceToHost, stream));
    WALT_HIP(hipMemcpyAsync(&last_cnt, cnt + (nw - 1), 4, hipMemcpyDeviceToHost, stream));
    WALT_HIP(hipStreamSynchronize(stream));
    const uint64_t want = (uint64_t)last_rank + last_cnt;
    const uint32_t n_recs = (uint32_t)std::min<uint64_t>(want, cap_recs);
    idx->window_eligible[s] = want;
    if (getenv("WALT_AMD_VERBOSE"))
      fprintf(stderr, "[walt_amd index: strand %d: %llu of %u index slots (%.2f %%) lie in runs with dense candidate windows; records for "
              "%u of them, %.2f GB]\n", s, (unsigned long long)want, sv.index_size, 100.0 * (double)want / (double)sv.index_size, n_recs,
              (double)n_recs * rec_bytes / 1e9);
    if (n_recs == 0) continue;
    if (hipMalloc(&win_buf.p, ((uint64_t)n_recs * (kWinWords + 1) + 16) * 4) != hipSuccess ||
        hipMalloc(&win2_buf.p, ((uint64_t)n_recs * kWinWords2 + 16) * 4) != hipSuccess) { give_up("record allocation failed"); continue; }
    uint32_t* win = reinterpret_cast<uint32_t*>(win_buf.p);
    uint32_t* win2 = reinterpret_cast<uint32_t*>(win2_buf.p);
    hipLaunchKernelGGL(k_win_fill, dim3((unsigned)(((uint64_t)sv.index_size + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, sv.g2,
                       sv.ent, sv.index_size, bits, rank, n_recs, win, win2);
    WALT_HIP(hipStreamSynchronize(stream));
    WALT_HIP(hipGetLastError());
    // the strand keeps them: hand the four arrays over to the index
    for (Scoped* b : {&bits_buf, &rank_buf, &win_buf, &win2_buf}) { idx->allocs.push_back(b->p); b->p = nullptr; }
    idx->device_bytes += ((uint64_t)nw + 1) * 12 + ((uint64_t)n_recs * (kWinWords + 1 + kWinWords2) + 32) * 4;
    idx->window_records[s] = n_recs;
    sv.wbits = bits; sv.wrank = rank; sv.win = win; sv.win2 = win2; sv.wcap = n_recs;
  }
  return WALT_OK;
}

int finish_index_device(walt_index* idx) {
  int rc;
  if ((rc = build_windows(idx))) return rc;
  const std::vector<uint32_t>& mt = compare_mask_table();
  if ((rc = dev_alloc(idx, &idx->d_mask_table, mt.size()))) return rc;
  WALT_HIP(hipMemcpy(idx->d_mask_table, mt.data(), mt.size() * 4, hipMemcpyHostToDevice));
  {  // edge bitmap (core.h kEdgeBlockShift): the blocks within kEdgeMargin of a chromosome boundary
    std::vector<uint32_t> bits(kEdgeWords, 0u);
    for (uint32_t b : idx->start_index) {
      const uint64_t lo = b > kEdgeMargin ? (uint64_t)b - kEdgeMargin : 0ull, hi = (uint64_t)b + kEdgeMargin;
      for (uint64_t blk = lo >> kEdgeBlockShift; blk <= (hi >> kEdgeBlockShift) && blk < 32ull * kEdgeWords; ++blk)
        bits[blk >> 5] |= 1u << (blk & 31u);
    }
    uint32_t* d_edge = nullptr;
    if ((rc = dev_alloc(idx, &d_edge, kEdgeWords))) return rc;
    WALT_HIP(hipMemcpy(d_edge, bits.data(), kEdgeWords * 4, hipMemcpyHostToDevice));
    idx->view.edge_bits = d_edge;
  }
  return WALT_OK;
}

int new_index(int device, const IndexHead& head, int dir_bits, int n_strands, walt_index** out) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(WALT_EHIP, "no HIP device available (the walt_amd hot path has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(WALT_EINVAL, "device ordinal out of range");
  WALT_HIP(hipSetDevice(device));
  walt_index* idx = new walt_index();
  idx->device = device;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) idx->n_cu = prop.multiProcessorCount;
  }
  idx->head = head;
  memset(&idx->view, 0, sizeof(idx->view));
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) total_b = 0;
  idx->view.dir_bits = (uint32_t)choose_dir_bits(head.max_index_size, dir_bits, n_strands < 1 ? 1 : n_strands, total_b);
  idx->view.dir_slots = dir_top(idx->view.dir_bits);
  // chromosome starts are needed by the strand builders (outlier detection)
  const uint32_t n = (uint32_t)idx->head.lengths.size();
  idx->start_index.assign(n + 1, 0);
  for (uint32_t i = 0; i < n; ++i) idx->start_index[i + 1] = idx->start_index[i] + idx->head.lengths[i];
  uint32_t* d_start = nullptr;
  int rc = dev_alloc(idx, &d_start, n + 1);
  if (!rc && hipMemcpy(d_start, idx->start_index.data(), (n + 1) * 4, hipMemcpyHostToDevice) != hipSuccess)
    rc = fail(WALT_EHIP, "upload of chromosome starts failed");
  if (rc) {
    std::string keep = walt_last_error();
    walt_index_close(idx);
    set_error(keep);
    return rc;
  }
  idx->view.start_index = d_start;
  idx->view.n_chrom = n;
  *out = idx;
  return WALT_OK;
}

// validate a host counter[] before it is trusted on the device
static int check_counter(const uint32_t* counter, uint32_t index_size) {
  if (counter[0] != 0 || counter[kNumBuckets] != index_size) return fail(WALT_EFORMAT, "counter[] ends do not match index_size");
  for (uint32_t i = 0; i < kNumBuckets; ++i)
    if (counter[i] > counter[i + 1]) return fail(WALT_EFORMAT, "counter[] is not non-decreasing");
  return WALT_OK;
}

static int upload_strand(walt_index* idx, int strand, const uint8_t* genome, const uint32_t* counter,
                         const uint32_t* index, uint32_t index_size) {
  int rc = check_counter(counter, index_size);
  if (rc) return rc;
  const uint32_t genome_len = idx->head.genome_len;
  uint8_t* d_bytes = nullptr;
  uint32_t *d_counter = nullptr, *d_index = nullptr;
  auto cleanup = [&]() {
    if (d_bytes) hipFree(d_bytes);
    if (d_counter) hipFree(d_counter);
    if (d_index) hipFree(d_index);
  };
  hipError_t e;
  if ((e = hipMalloc(reinterpret_cast<void**>(&d_bytes), (uint64_t)genome_len + 16)) != hipSuccess ||
      (e = hipMalloc(reinterpret_cast<void**>(&d_counter), ((uint64_t)kNumBuckets + 1) * 4)) != hipSuccess ||
      (e = hipMalloc(reinterpret_cast<void**>(&d_index), ((uint64_t)index_size + 1) * 4)) != hipSuccess) {
    cleanup();
    return fail(WALT_ENOMEM, std::string("hipMalloc failed: ") + hipGetErrorString(e));
  }
  if ((e = hipMemcpy(d_bytes, genome, genome_len, hipMemcpyHostToDevice)) != hipSuccess ||
      (e = hipMemcpy(d_counter, counter, ((uint64_t)kNumBuckets + 1) * 4, hipMemcpyHostToDevice)) != hipSuccess ||
      (e = hipMemcpy(d_index, index, (uint64_t)index_size * 4, hipMemcpyHostToDevice)) != hipSuccess) {
    cleanup();
    return fail(WALT_EHIP, std::string("hipMemcpy failed: ") + hipGetErrorString(e));
  }
  rc = build_strand_device(idx, strand, d_bytes, d_counter, d_index, index_size, nullptr);
  cleanup();
  return rc;
}


// ---------------------------------------------------------------------------
// strand file -> HBM without a host copy of the 15 GB arrays: reader threads
// pread 32 MiB pieces into page-locked buffers and queue each piece on their own
// stream (ReadIndex, reference.cpp:324-351, freads the whole file into vectors)
// ---------------------------------------------------------------------------
struct StreamRing {
  static constexpr int kThreads = 16, kDepth = 2;
  static constexpr size_t kPiece = 16u << 20;
  char* buf = nullptr;  // kThreads * kDepth * kPiece, page-locked
  hipStream_t stream[kThreads] = {};
  int device = 0;
  int init(int dev) {
    device = dev;
    if (hipHostMalloc(reinterpret_cast<void**>(&buf), (size_t)kThreads * kDepth * kPiece, hipHostMallocDefault) != hipSuccess)
      return fail(WALT_ENOMEM, "hipHostMalloc failed (index staging)");
    for (int t = 0; t < kThreads; ++t)
      if (hipStreamCreateWithFlags(&stream[t], hipStreamNonBlocking) != hipSuccess) return fail(WALT_EHIP, "hipStreamCreate failed");
    return WALT_OK;
  }
  ~StreamRing() {
    for (int t = 0; t < kThreads; ++t)
      if (stream[t]) hipStreamDestroy(stream[t]);
    if (buf) hipHostFree(buf);
  }
  // file[off, off + bytes) -> d_dst[0, bytes)
  int copy(int fd, uint64_t off, void* d_dst, uint64_t bytes) {
    const uint64_t pieces = (bytes + kPiece - 1) / kPiece;
    std::atomic<int> err(0);
    auto worker = [&](int t) {
      if (hipSetDevice(device) != hipSuccess) { err = WALT_EHIP; return; }
      hipEvent_t done[kDepth] = {};
      for (int d = 0; d < kDepth; ++d)
        if (hipEventCreateWithFlags(&done[d], hipEventDisableTiming) != hipSuccess) err = WALT_EHIP;
      bool used[kDepth] = {};
      int slot = 0;
      for (uint64_t p = t; p < pieces && !err; p += kThreads, slot = (slot + 1) % kDepth) {
        char* b = buf + ((size_t)t * kDepth + slot) * kPiece;
        if (used[slot] && hipEventSynchronize(done[slot]) != hipSuccess) { err = WALT_EHIP; break; }
        const uint64_t at = p * kPiece, len = std::min<uint64_t>(kPiece, bytes - at);
        uint64_t got = 0;
        while (got < len) {
          ssize_t r = pread(fd, b + got, len - got, (off_t)(off + at + got));
          if (r <= 0) { err = WALT_EFORMAT; break; }
          got += (uint64_t)r;
        }
        if (err) break;
        if (hipMemcpyAsync(static_cast<char*>(d_dst) + at, b, len, hipMemcpyHostToDevice, stream[t]) != hipSuccess ||
            hipEventRecord(done[slot], stream[t]) != hipSuccess) { err = WALT_EHIP; break; }
        used[slot] = true;
      }
      if (hipStreamSynchronize(stream[t]) != hipSuccess && !err) err = WALT_EHIP;
      for (int d = 0; d < kDepth; ++d)
        if (done[d]) hipEventDestroy(done[d]);
    };
    std::vector<std::thread> th;
    for (int t = 0; t < kThreads; ++t) th.emplace_back(worker, t);
    for (auto& x : th) x.join();
    if (err == WALT_EFORMAT) return fail(WALT_EFORMAT, "read file error (strand index): file too short");
    if (err) return fail(WALT_EHIP, "streaming upload failed");
    return WALT_OK;
  }
};

static double wall_s() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static int stream_strand_file(walt_index* idx, int strand, const std::string& path, StreamRing& ring) {
  const bool verbose = getenv("WALT_AMD_VERBOSE") != nullptr;
  double t0 = wall_s();
  int fd = ::open(path.c_str(), O_RDONLY);
  if (fd < 0) return fail(WALT_EIO, "cannot open input file " + path);
  const uint32_t genome_len = idx->head.genome_len;
  char strand_byte = 0;
  uint32_t sizes[2] = {0, 0};  // counter_size, index_size (reference.cpp:311-316)
  bool ok = pread(fd, &strand_byte, 1, 0) == 1 && pread(fd, sizes, 8, (off_t)1 + genome_len) == 8;
  if (ok && sizes[0] != kNumBuckets) ok = false;
  if (!ok) { ::close(fd); return fail(WALT_EFORMAT, "read file error (strand index) " + path); }
  if (strand_byte != ((strand & 1) ? '-' : '+')) { ::close(fd); return fail(WALT_EFORMAT, "strand byte mismatch in " + path); }
  const uint32_t index_size = sizes[1];
  const uint64_t counter_bytes = ((uint64_t)kNumBuckets + 1) * 4;
  const uint64_t off_counter = (uint64_t)1 + genome_len + 8, off_index = off_counter + counter_bytes;
  std::vector<uint32_t> counter((size_t)kNumBuckets + 1);
  {
    uint64_t got = 0;
    while (got < counter_bytes) {
      ssize_t r = pread(fd, reinterpret_cast<char*>(counter.data()) + got, counter_bytes - got, (off_t)(off_counter + got));
      if (r <= 0) break;
      got += (uint64_t)r;
    }
    if (got != counter_bytes) { ::close(fd); return fail(WALT_EFORMAT, "read file error (strand index) " + path); }
  }
  int rc = check_counter(counter.data(), index_size);
  if (rc) { ::close(fd); return rc; }
  uint8_t* d_bytes = nullptr;
  uint32_t *d_counter = nullptr, *d_index = nullptr;
  auto cleanup = [&]() {
    if (d_bytes) hipFree(d_bytes);
    if (d_counter) hipFree(d_counter);
    if (d_index) hipFree(d_index);
    ::close(fd);
  };
  hipError_t e;
  if ((e = hipMalloc(reinterpret_cast<void**>(&d_bytes), (uint64_t)genome_len + 16)) != hipSuccess ||
      (e = hipMalloc(reinterpret_cast<void**>(&d_counter), counter_bytes)) != hipSuccess ||
      (e = hipMalloc(reinterpret_cast<void**>(&d_index), ((uint64_t)index_size + 1) * 4)) != hipSuccess) {
    cleanup();
    return fail(WALT_ENOMEM, std::string("hipMalloc failed: ") + hipGetErrorString(e));
  }
  if ((e = hipMemcpy(d_counter, counter.data(), counter_bytes, hipMemcpyHostToDevice)) != hipSuccess) {
    cleanup();
    return fail(WALT_EHIP, std::string("hipMemcpy failed: ") + hipGetErrorString(e));
  }
  rc = ring.copy(fd, 1, d_bytes, genome_len);
  if (!rc) rc = ring.copy(fd, off_index, d_index, (uint64_t)index_size * 4);
  double t1 = wall_s();
  if (!rc) rc = build_strand_device(idx, strand, d_bytes, d_counter, d_index, index_size, nullptr);
  if (!rc && hipDeviceSynchronize() != hipSuccess) rc = fail(WALT_EHIP, "strand build failed");
  if (verbose)
    fprintf(stderr, "[walt_amd index: %s  %.1f GB streamed in %.2f s, derived structures %.2f s]\n", path.c_str(),
            (genome_len + 4.0 * index_size) / 1e9, t1 - t0, wall_s() - t1);
  cleanup();
  return rc;
}

}  // namespace walt

using namespace walt;

extern "C" {

int walt_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int walt_host_alloc(size_t bytes, void** out) {
  if (!out) return fail(WALT_EINVAL, "walt_host_alloc: bad argument");
  *out = nullptr;
  hipError_t e = hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault);
  if (e != hipSuccess) return fail(WALT_ENOMEM, std::string("hipHostMalloc failed: ") + hipGetErrorString(e));
  return WALT_OK;
}

void walt_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

int walt_index_open(const char* dbindex_path, int device, unsigned strand_mask, int dir_bits, walt_index** out) {
  if (!dbindex_path || !out || !(strand_mask & 15u)) return fail(WALT_EINVAL, "walt_index_open: bad argument");
  *out = nullptr;
  IndexHead head;
  int rc = read_index_head(dbindex_path, head);
  if (rc) return rc;
  walt_index* idx = nullptr;
  if ((rc = new_index(device, head, dir_bits, __builtin_popcount(strand_mask & 15u), &idx))) return rc;
  static const char* sfx[4] = {"_CT00", "_CT01", "_GA10", "_GA11"};
  {
    StreamRing ring;
    rc = ring.init(device);
    for (int s = 0; s < 4 && !rc; ++s) {
      if (!(strand_mask & (1u << s))) continue;
      rc = stream_strand_file(idx, s, std::string(dbindex_path) + sfx[s], ring);
    }
  }
  if (!rc) rc = finish_index_device(idx);
  if (rc) {
    std::string keep = walt_last_error();
    walt_index_close(idx);
    set_error(keep);
    return rc;
  }
  *out = idx;
  return WALT_OK;
}

int walt_index_from_host(uint32_t n_chrom, const uint32_t* chrom_len, const char* const* chrom_names,
                         const uint8_t* const genome[4], const uint32_t* const counter[4],
                         const uint32_t* const index[4], const uint32_t index_size[4], int device, int dir_bits,
                         walt_index** out) {
  if (!out || !chrom_len || !n_chrom) return fail(WALT_EINVAL, "walt_index_from_host: bad argument");
  *out = nullptr;
  IndexHead head;
  uint64_t total = 0;
  for (uint32_t i = 0; i < n_chrom; ++i) {
    head.lengths.push_back(chrom_len[i]);
    head.names.push_back(chrom_names && chrom_names[i] ? chrom_names[i] : ("chr" + std::to_string(i)));
    total += chrom_len[i];
  }
  // positions and position + read length are 32-bit (like the reference's uint32_t); 0xFFFFFFFF marks a slot-table record
  if (total >= (1ull << 32) - 256) return fail(WALT_EINVAL, "genome longer than 2^32 - 256 bases");
  head.genome_len = (uint32_t)total;
  for (int s = 0; s < 4; ++s)
    if (genome[s] && index_size[s] > head.max_index_size) head.max_index_size = index_size[s];
  walt_index* idx = nullptr;
  int n_strands = 0;
  for (int s = 0; s < 4; ++s) n_strands += genome[s] ? 1 : 0;
  int rc = new_index(device, head, dir_bits, n_strands, &idx);
  if (rc) return rc;
  for (int s = 0; s < 4 && !rc; ++s) {
    if (!genome[s]) continue;
    if (!counter[s] || (!index[s] && index_size[s])) rc = fail(WALT_EINVAL, "strand arrays missing");
    if (!rc) rc = upload_strand(idx, s, genome[s], counter[s], index[s], index_size[s]);
  }
  if (!rc) rc = finish_index_device(idx);
  if (rc) {
    std::string keep = walt_last_error();
    walt_index_close(idx);
    set_error(keep);
    return rc;
  }
  *out = idx;
  return WALT_OK;
}

void walt_index_close(walt_index* idx) {
  if (!idx) return;
  hipSetDevice(idx->device);
  for (void* p : idx->allocs) hipFree(p);
  for (void* p : idx->host_api_buf)
    if (p) hipFree(p);
  for (int i = 0; i < 3; ++i)
    if (idx->ev[i]) hipEventDestroy(idx->ev[i]);
  for (int i = 0; i < walt_index::kDetailEvents; ++i)
    if (idx->ev_detail[i]) hipEventDestroy(idx->ev_detail[i]);
  for (int k = 0; k < 2; ++k) {
    if (idx->pe_fork[k]) hipEventDestroy(idx->pe_fork[k]);
    if (idx->pe_join[k]) hipEventDestroy(idx->pe_join[k]);
    if (idx->pe_done[k]) hipEventDestroy(idx->pe_done[k]);
    for (int j = 0; j < 2; ++j)
      if (idx->pe_stream[k][j]) hipStreamDestroy(idx->pe_stream[k][j]);
  }
  if (idx->pe_start) hipEventDestroy(idx->pe_start);
  if (idx->se_fork) hipEventDestroy(idx->se_fork);
  if (idx->se_join) hipEventDestroy(idx->se_join);
  if (idx->se_side) hipStreamDestroy(idx->se_side);
  for (hipEvent_t e : idx->se_pipe_ev) if (e) hipEventDestroy(e);
  if (idx->se_pipe) hipStreamDestroy(idx->se_pipe);
  delete idx;
}

uint32_t walt_index_n_chrom(const walt_index* idx) { return idx ? (uint32_t)idx->head.lengths.size() : 0; }
uint32_t walt_index_chrom_len(const walt_index* idx, uint32_t i) {
  return idx && i < idx->head.lengths.size() ? idx->head.lengths[i] : 0;
}
const char* walt_index_chrom_name(const walt_index* idx, uint32_t i) {
  return idx && i < idx->head.names.size() ? idx->head.names[i].c_str() : "";
}
uint64_t walt_index_genome_len(const walt_index* idx) { return idx ? idx->head.genome_len : 0; }
uint64_t walt_index_device_bytes(const walt_index* idx) { return idx ? idx->device_bytes : 0; }
int walt_index_dir_bits(const walt_index* idx) { return idx ? (int)idx->view.dir_bits : -1; }
uint64_t walt_index_bad_buckets(const walt_index* idx, int strand) {
  return idx && strand >= 0 && strand < 4 ? idx->bad_buckets[strand] : 0;
}
// ---- options (include/walt_amd.h): the table of names; every value changes a call's schedule, never its results
namespace {
struct OptionName {
  const char* name;
  int kind;  // 0: int field, 1: long long field
  size_t offset;
  long long lo, hi;
};
#define WALT_OPT_I(f, lo, hi) {#f, 0, offsetof(walt_options, f), lo, hi}
#define WALT_OPT_L(f, lo, hi) {#f, 1, offsetof(walt_options, f), lo, hi}
const OptionName kOptions[] = {
    WALT_OPT_I(se_pipe, 0, 1),        WALT_OPT_L(se_heavy_chunk, 0, 1ll << 28), WALT_OPT_I(se_lit_staged, 0, 1), WALT_OPT_I(se_stage_blocks, 0, 16), WALT_OPT_I(se_verify_blocks, 0, 16), WALT_OPT_I(se_stagger, 0, 1), WALT_OPT_I(se_lit_side, 0, 2), WALT_OPT_I(se_lit_ablate, 0, 7),
    WALT_OPT_L(se_defer_min, -1, 1ll << 30), WALT_OPT_I(se_stage_occ, 0, 4),   WALT_OPT_I(se_carry, 0, 1),
    WALT_OPT_I(se_heavy_mono, 0, 1),  WALT_OPT_L(grid, 0, 1ll << 20),           WALT_OPT_I(pe_mode, 0, 1),
    WALT_OPT_L(pe_chunk, 0, 1ll << 28), WALT_OPT_L(pe_rounds, 0, 4),            WALT_OPT_L(pe_stage_cap, 0, 1ll << 28),
    WALT_OPT_I(pe_small_heaps, 0, 1), WALT_OPT_I(pe_serial, 0, 1), WALT_OPT_I(pe_push_wide, 0, 1),              WALT_OPT_L(pe_defer_min, -1, 1ll << 30),
    WALT_OPT_I(pe_roomy, -1, 1),      WALT_OPT_I(pe_lit_fuse, 0, 1),
};
#undef WALT_OPT_I
#undef WALT_OPT_L
const OptionName* find_option(const char* name) {
  if (!name) return nullptr;
  for (const OptionName& o : kOptions)
    if (!strcmp(o.name, name)) return &o;
  return nullptr;
}
}  // namespace

int walt_index_set_option(walt_index* idx, const char* name, long long value) {
  if (!idx) return fail(WALT_EINVAL, "null index");
  const OptionName* o = find_option(name);
  if (!o) return fail(WALT_EINVAL, std::string("walt_index_set_option: unknown option '") + (name ? name : "(null)") + "'");
  if (value < o->lo || value > o->hi)
    return fail(WALT_EINVAL, std::string("walt_index_set_option: ") + name + " = " + std::to_string(value) + " is outside [" +
                                 std::to_string(o->lo) + ", " + std::to_string(o->hi) + "]");
  char* base = reinterpret_cast<char*>(&idx->opt) + o->offset;
  if (o->kind == 0) *reinterpret_cast<int*>(base) = (int)value;
  else *reinterpret_cast<long long*>(base) = value;
  return WALT_OK;
}
int walt_index_get_option(const walt_index* idx, const char* name, long long* value) {
  if (!idx || !value) return fail(WALT_EINVAL, "walt_index_get_option: bad argument");
  const OptionName* o = find_option(name);
  if (!o) return fail(WALT_EINVAL, std::string("walt_index_get_option: unknown option '") + (name ? name : "(null)") + "'");
  const char* base = reinterpret_cast<const char*>(&idx->opt) + o->offset;
  *value = o->kind == 0 ? (long long)*reinterpret_cast<const int*>(base) : *reinterpret_cast<const long long*>(base);
  return WALT_OK;
}

uint64_t walt_index_outliers(const walt_index* idx, int strand) {
  return idx && strand >= 0 && strand < 4 ? idx->outliers[strand] : 0;
}
uint64_t walt_index_window_eligible(const walt_index* idx, int strand) {
  return idx && strand >= 0 && strand < 4 ? idx->window_eligible[strand] : 0;
}
uint64_t walt_index_window_entries(const walt_index* idx, int strand) {
  return idx && strand >= 0 && strand < 4 ? idx->window_records[strand] : 0;
}

}  // extern "C"
