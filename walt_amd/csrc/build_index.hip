// build_index.hip -- makedb on the GPU: builds the four strand indexes of a
// genome that is already resident in HBM (ASCII ACGT, N-free), in WALT's
// .dbindex semantics, and can export / write them as .dbindex files.
//
// Restates BuildIndex (reference makedb.cpp:46-85): ReverseComplementGenome
// (reference.cpp:131-146), C2T/G2A (148-162), CountBucketSize incl. the
// >= 500000 bucket erase (192-229), HashToBucket (231-256) and
// SortHashTableBucket with the chromosome-end rule of its comparator (258-300).
// The per-bucket std::sort is replaced by two stable LSD radix passes over
// (bucket, care chars 12..59 with 0 = "beyond the chromosome end"); the result
// is the same order except among entries whose 60 care characters are all
// equal, where the reference keeps whatever std::sort leaves and this builder
// keeps ascending position.  For genomes without such
// ties the files are byte-identical to the reference makedb's.
//
// rocPRIM (device select / scan / radix sort) is used here: this is the
// offline builder, not the per-read hot path.
#include <stdio.h>
#include <string.h>

#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "device_common.h"

namespace walt {

// strand genome: optional per-chromosome reverse complement, then conversion
__global__ void k_strand_genome(const uint8_t* __restrict__ ascii, const uint32_t* __restrict__ start,
                                uint32_t n_chrom, uint32_t genome_len, uint32_t rev, uint32_t ga,
                                uint32_t* __restrict__ g2, uint32_t nwords, uint32_t* __restrict__ err) {
  uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= nwords) return;
  uint32_t v = 0, bad = 0;
  uint32_t p0 = w * 16;
  uint32_t chr = chrom_id(start, n_chrom, p0);
  for (uint32_t k = 0; k < 16; ++k) {
    uint32_t p = p0 + k;
    if (p >= genome_len) break;
    while (p >= start[chr + 1]) ++chr;
    uint32_t src = rev ? start[chr] + (start[chr + 1] - 1 - p) : p;
    uint32_t code = base_code(ascii[src]);
    if (code > 3) { ++bad; code = 0; }
    if (rev) code = 3 - code;
    code = convert_code(code, ga);
    v |= code << (2 * k);
  }
  g2[w] = v;
  if (bad) atomicAdd(err, bad);
}

// position j is indexed iff its chromosome has >= 36 bases and j < chrom_end - 36
// (CountBucketSize, reference.cpp:199-203)
__device__ __forceinline__ bool indexed_position(const uint32_t* start, uint32_t n_chrom, uint32_t j) {
  uint32_t chr = chrom_id(start, n_chrom, j);
  uint32_t len = start[chr + 1] - start[chr];
  return len >= kMinSeedLen && j < start[chr + 1] - kMinSeedLen;
}

__global__ void k_count_buckets(const uint32_t* __restrict__ g2, const uint32_t* __restrict__ start,
                                uint32_t n_chrom, uint32_t genome_len, uint32_t* __restrict__ hist) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= genome_len) return;
  if (!indexed_position(start, n_chrom, j)) return;
  atomicAdd(&hist[hash_at_dev(g2, j)], 1u);
}

// erase buckets >= 500000 (reference.cpp:211-218): hist -> 0, erased bit set
__global__ void k_erase_large(uint32_t* __restrict__ hist, uint32_t* __restrict__ erased,
                              uint32_t* __restrict__ n_erased) {
  uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= kNumBuckets) return;
  if (hist[h] >= kEraseBucket) {
    hist[h] = 0;
    atomicOr(&erased[h >> 5], 1u << (h & 31));
    atomicAdd(n_erased, 1u);
  }
}

struct KeepPosition {
  const uint32_t* g2;
  const uint32_t* start;
  const uint32_t* erased;
  uint32_t n_chrom;
  __device__ bool operator()(const uint32_t& j) const {
    if (!indexed_position(start, n_chrom, j)) return false;
    uint32_t h = hash_at_dev(g2, j);
    return !((erased[h >> 5] >> (h & 31)) & 1u);
  }
};

// sort keys.  Care char q of position p: 0 when p + care_pos(q) runs over the
// end of p's chromosome (SortHashTableBucketCMP treats that as smallest,
// reference.cpp:271-276), else 1/2/3 for the three letters of the strand.
__device__ __forceinline__ uint32_t marked_char(const uint32_t* g2, uint32_t p, uint32_t room, uint32_t q) {
  uint32_t cp = care_pos(q);
  if (cp >= room) return 0;
  uint32_t c = g2_code(g2, (uint64_t)p + cp);
  return c == 0 ? 1u : c == 3 ? 3u : 2u;
}
// pass A: care chars [Q_LO, Q_HI), at most 32 of them (28..59 for pattern 3; pattern 7's 28..79 take two passes)
template <uint32_t Q_LO, uint32_t Q_HI>
__global__ void k_keys_low(const uint32_t* __restrict__ g2, const uint32_t* __restrict__ start, uint32_t n_chrom,
                           const uint32_t* __restrict__ pos, uint32_t n, unsigned long long* __restrict__ keys) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t p = pos[i];
  uint32_t chr = chrom_id(start, n_chrom, p);
  keys[i] = marked_chars_dev<Q_LO, Q_HI>(g2, p, start[chr + 1] - p);
}
// pass B: (bucket << 32) | care chars 12..27
__global__ void k_keys_high(const uint32_t* __restrict__ g2, const uint32_t* __restrict__ start, uint32_t n_chrom,
                            const uint32_t* __restrict__ pos, uint32_t n, unsigned long long* __restrict__ keys) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t p = pos[i];
  uint32_t chr = chrom_id(start, n_chrom, p);
  keys[i] = ((unsigned long long)hash_at_dev(g2, p) << 32) | marked_chars_dev<12, 28>(g2, p, start[chr + 1] - p);
}

__global__ void k_unpack_genome(const uint32_t* __restrict__ g2, uint32_t genome_len, uint8_t* __restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= genome_len) return;
  out[i] = "ACGT"[g2_code(g2, i)];
}
__global__ void k_ent_positions(const Ent* __restrict__ ent, uint32_t n, uint32_t* __restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = ent[i].pos;
}

struct DevBuf {  // scoped device allocation
  void* p = nullptr;
  ~DevBuf() { if (p) hipFree(p); }
  hipError_t alloc(uint64_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
  template <typename T> T* as() { return reinterpret_cast<T*>(p); }
};

#define WALT_HIPB(expr) WALT_HIP(expr)

static int build_one_strand(walt_index* idx, int strand, const uint8_t* d_ascii, const uint32_t* d_start,
                            hipStream_t stream) {
  const uint32_t genome_len = idx->head.genome_len;
  const uint32_t n_chrom = (uint32_t)idx->head.lengths.size();
  const uint32_t nwords = (genome_len + 15) / 16;
  const uint32_t rev = strand & 1, ga = strand >= 2;
  uint32_t* g2 = nullptr;
  int rc = alloc_strand_g2(idx, &g2, stream);
  if (rc) return rc;
  DevBuf err, hist, erased, counter;
  WALT_HIPB(err.alloc(16));
  WALT_HIPB(hist.alloc(((uint64_t)kNumBuckets + 1) * 4));
  WALT_HIPB(erased.alloc(kNumBuckets / 8));
  WALT_HIPB(counter.alloc(((uint64_t)kNumBuckets + 1) * 4));
  WALT_HIPB(hipMemsetAsync(err.p, 0, 16, stream));
  WALT_HIPB(hipMemsetAsync(hist.p, 0, ((uint64_t)kNumBuckets + 1) * 4, stream));
  WALT_HIPB(hipMemsetAsync(erased.p, 0, kNumBuckets / 8, stream));
  if (nwords)
    hipLaunchKernelGGL(k_strand_genome, dim3(grid_for(nwords)), dim3(kBlock), 0, stream, d_ascii, d_start, n_chrom,
                       genome_len, rev, (uint32_t)ga, g2, nwords, err.as<uint32_t>());
  if (genome_len)
    hipLaunchKernelGGL(k_count_buckets, dim3(grid_for(genome_len)), dim3(kBlock), 0, stream, g2, d_start, n_chrom,
                       genome_len, hist.as<uint32_t>());
  hipLaunchKernelGGL(k_erase_large, dim3(grid_for(kNumBuckets)), dim3(kBlock), 0, stream, hist.as<uint32_t>(),
                     erased.as<uint32_t>(), err.as<uint32_t>() + 1);
  // counter = exclusive prefix sum of the bucket sizes (the reference's += / shift, reference.cpp:221-228)
  {
    size_t tmp_bytes = 0;
    WALT_HIPB(rocprim::exclusive_scan(nullptr, tmp_bytes, hist.as<uint32_t>(), counter.as<uint32_t>(), 0u,
                                      (size_t)kNumBuckets + 1, rocprim::plus<uint32_t>(), stream));
    DevBuf tmp;
    WALT_HIPB(tmp.alloc(tmp_bytes));
    WALT_HIPB(rocprim::exclusive_scan(tmp.p, tmp_bytes, hist.as<uint32_t>(), counter.as<uint32_t>(), 0u,
                                      (size_t)kNumBuckets + 1, rocprim::plus<uint32_t>(), stream));
    WALT_HIPB(hipStreamSynchronize(stream));
  }
  uint32_t herr[4] = {0, 0, 0, 0};
  uint32_t index_size = 0;
  WALT_HIPB(hipMemcpy(herr, err.p, 16, hipMemcpyDeviceToHost));
  WALT_HIPB(hipMemcpy(&index_size, counter.as<uint32_t>() + kNumBuckets, 4, hipMemcpyDeviceToHost));
  if (herr[0]) return fail(WALT_EBASE, "genome contains " + std::to_string(herr[0]) + " non-ACGT bytes (resolve N before building)");
  if (herr[1]) fprintf(stderr, "[walt_amd makedb: erased %u buckets of size >= %u on strand %d]\n", herr[1], kEraseBucket, strand);

  // positions of the kept k-mers in ascending order == HashToBucket's fill order
  DevBuf pos_a, pos_b, key_a, key_b, n_sel;
  WALT_HIPB(pos_a.alloc((uint64_t)index_size * 4));
  WALT_HIPB(pos_b.alloc((uint64_t)index_size * 4));
  WALT_HIPB(key_a.alloc((uint64_t)index_size * 8));
  WALT_HIPB(key_b.alloc((uint64_t)index_size * 8));
  WALT_HIPB(n_sel.alloc(16));
  if (index_size) {
    KeepPosition keep{g2, d_start, erased.as<uint32_t>(), n_chrom};
    rocprim::counting_iterator<uint32_t> all(0);
    size_t tmp_bytes = 0;
    WALT_HIPB(rocprim::select(nullptr, tmp_bytes, all, pos_a.as<uint32_t>(), n_sel.as<size_t>(), (size_t)genome_len,
                              keep, stream));
    DevBuf tmp;
    WALT_HIPB(tmp.alloc(tmp_bytes));
    WALT_HIPB(rocprim::select(tmp.p, tmp_bytes, all, pos_a.as<uint32_t>(), n_sel.as<size_t>(), (size_t)genome_len,
                              keep, stream));
    size_t got = 0;
    WALT_HIPB(hipMemcpyAsync(&got, n_sel.p, sizeof(size_t), hipMemcpyDeviceToHost, stream));
    WALT_HIPB(hipStreamSynchronize(stream));
    if (got != index_size) return fail(WALT_EHIP, "index builder: selected count does not match the bucket total");

    rocprim::double_buffer<unsigned long long> keys(key_a.as<unsigned long long>(), key_b.as<unsigned long long>());
    rocprim::double_buffer<uint32_t> vals(pos_a.as<uint32_t>(), pos_b.as<uint32_t>());
    size_t sort_bytes = 0;
    WALT_HIPB(rocprim::radix_sort_pairs(nullptr, sort_bytes, keys, vals, (size_t)index_size, 0u, 64u, stream));
    DevBuf sort_tmp;
    WALT_HIPB(sort_tmp.alloc(sort_bytes));
    // pass A (stable LSD passes): care chars 28 .. kNumCare-1, the least significant 32 first
    auto pass_a = [&](auto kernel, uint32_t nchars) -> hipError_t {
      hipLaunchKernelGGL(kernel, dim3(grid_for(index_size)), dim3(kBlock), 0, stream, g2, d_start, n_chrom,
                         vals.current(), index_size, keys.current());
      return rocprim::radix_sort_pairs(sort_tmp.p, sort_bytes, keys, vals, (size_t)index_size, 0u, 2u * nchars, stream);
    };
#if WALT_SEEDPATTERN == 7  // 80 care characters: 48..79, then 28..47
    WALT_HIPB(pass_a(k_keys_low<kNumCare - 32, kNumCare>, 32));
    WALT_HIPB(pass_a(k_keys_low<28, kNumCare - 32>, kNumCare - 32 - 28));
#else
    WALT_HIPB(pass_a(k_keys_low<28, kNumCare>, kNumCare - 28));
#endif
    // pass B (stable): bucket + first 16 care chars
    hipLaunchKernelGGL(k_keys_high, dim3(grid_for(index_size)), dim3(kBlock), 0, stream, g2, d_start, n_chrom,
                       vals.current(), index_size, keys.current());
    WALT_HIPB(rocprim::radix_sort_pairs(sort_tmp.p, sort_bytes, keys, vals, (size_t)index_size, 0u, 56u, stream));
    WALT_HIPB(hipStreamSynchronize(stream));
    WALT_HIPB(hipGetLastError());
    if (vals.current() != pos_a.as<uint32_t>()) std::swap(pos_a.p, pos_b.p);
  }
  // release the big temporaries before the derived structures are allocated
  hipFree(key_a.p); key_a.p = nullptr;
  hipFree(key_b.p); key_b.p = nullptr;
  hipFree(pos_b.p); pos_b.p = nullptr;
  return finish_strand_device(idx, strand, g2, counter.as<uint32_t>(), pos_a.as<uint32_t>(), index_size, stream);
}

}  // namespace walt

using namespace walt;

extern "C" {

int walt_index_build_device(const void* d_genome_ascii, uint32_t n_chrom, const uint32_t* chrom_len,
                            const char* const* chrom_names, int device, unsigned strand_mask, int dir_bits,
                            walt_index** out) {
  if (!d_genome_ascii || !out || !chrom_len || !n_chrom || !(strand_mask & 15u))
    return fail(WALT_EINVAL, "walt_index_build_device: bad argument");
  *out = nullptr;
  IndexHead head;
  uint64_t total = 0;
  for (uint32_t i = 0; i < n_chrom; ++i) {
    head.lengths.push_back(chrom_len[i]);
    head.names.push_back(chrom_names && chrom_names[i] ? chrom_names[i] : ("chr" + std::to_string(i + 1)));
    total += chrom_len[i];
  }
  if (total >= (1ull << 32) - 256) return fail(WALT_EINVAL, "genome longer than 2^32 bases");
  head.genome_len = (uint32_t)total;
  head.max_index_size = head.genome_len;  // upper bound, used to pick the directory depth
  walt_index* idx = nullptr;
  int rc = new_index(device, head, dir_bits, __builtin_popcount(strand_mask & 15u), &idx);
  if (rc) return rc;
  std::vector<uint32_t> start(n_chrom + 1, 0);
  for (uint32_t i = 0; i < n_chrom; ++i) start[i + 1] = start[i] + chrom_len[i];
  uint32_t* d_start = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_start), (n_chrom + 1) * 4);
  if (e == hipSuccess) e = hipMemcpy(d_start, start.data(), (n_chrom + 1) * 4, hipMemcpyHostToDevice);
  if (e != hipSuccess) rc = fail(WALT_EHIP, std::string("index builder: ") + hipGetErrorString(e));
  uint32_t max_index = 0;
  for (int s = 0; s < 4 && !rc; ++s) {
    if (!(strand_mask & (1u << s))) continue;
    rc = build_one_strand(idx, s, reinterpret_cast<const uint8_t*>(d_genome_ascii), d_start, nullptr);
    if (!rc && idx->view.s[s].index_size > max_index) max_index = idx->view.s[s].index_size;
  }
  if (d_start) hipFree(d_start);
  if (!rc) {
    idx->head.max_index_size = max_index;
    rc = finish_index_device(idx);
  }
  if (rc) {
    std::string keep = walt_last_error();
    walt_index_close(idx);
    set_error(keep);
    return rc;
  }
  *out = idx;
  return WALT_OK;
}

uint32_t walt_index_size(const walt_index* idx, int strand) {
  return idx && strand >= 0 && strand < 4 && (idx->strand_mask & (1u << strand)) ? idx->view.s[strand].index_size : 0;
}

int walt_index_export_strand(const walt_index* idx, int strand, uint8_t* genome_out, uint32_t* counter_out,
                             uint32_t* index_out) {
  if (!idx || strand < 0 || strand > 3 || !(idx->strand_mask & (1u << strand)))
    return fail(WALT_EINVAL, "walt_index_export_strand: strand not resident");
  WALT_HIP(hipSetDevice(idx->device));
  const StrandView& sv = idx->view.s[strand];
  if (genome_out && sv.genome_len) {
    DevBuf tmp;
    WALT_HIP(tmp.alloc(sv.genome_len));
    hipLaunchKernelGGL(k_unpack_genome, dim3(grid_for(sv.genome_len)), dim3(kBlock), 0, nullptr, sv.g2, sv.genome_len,
                       tmp.as<uint8_t>());
    WALT_HIP(hipMemcpy(genome_out, tmp.p, sv.genome_len, hipMemcpyDeviceToHost));
  }
  if (counter_out) WALT_HIP(hipMemcpy(counter_out, sv.cnt, ((uint64_t)kNumBuckets + 1) * 4, hipMemcpyDeviceToHost));
  if (index_out && sv.index_size) {
    DevBuf tmp;
    WALT_HIP(tmp.alloc((uint64_t)sv.index_size * 4));
    hipLaunchKernelGGL(k_ent_positions, dim3(grid_for(sv.index_size)), dim3(kBlock), 0, nullptr, sv.ent, sv.index_size,
                       tmp.as<uint32_t>());
    WALT_HIP(hipMemcpy(index_out, tmp.p, (uint64_t)sv.index_size * 4, hipMemcpyDeviceToHost));
  }
  return WALT_OK;
}

// Write the resident index as the five .dbindex files (WriteIndex /
// WriteIndexHeadInfo, reference.cpp:302-322, 353-379).  All four strands must
// be resident.
int walt_index_write(const walt_index* idx, const char* dbindex_path) {
  if (!idx || !dbindex_path) return fail(WALT_EINVAL, "walt_index_write: bad argument");
  if ((idx->strand_mask & 15u) == 0) return fail(WALT_EINVAL, "walt_index_write: no strand resident");
  static const char* sfx[4] = {"_CT00", "_CT01", "_GA10", "_GA11"};
  IndexHead head = idx->head;
  head.max_index_size = 0;
  for (int s = 0; s < 4; ++s) {
    if (!(idx->strand_mask & (1u << s))) continue;  // a C->T-only (or G->A-only) index writes its two strand files
    StrandFile sf;
    sf.strand = (s & 1) ? '-' : '+';
    sf.genome.resize(head.genome_len);
    sf.counter.resize((size_t)kNumBuckets + 1);
    sf.index.resize(idx->view.s[s].index_size);
    int rc = walt_index_export_strand(idx, s, sf.genome.data(), sf.counter.data(), sf.index.data());
    if (rc) return rc;
    if ((rc = write_strand_file(std::string(dbindex_path) + sfx[s], sf))) return rc;
    if (sf.index.size() > head.max_index_size) head.max_index_size = (uint32_t)sf.index.size();
  }
  return write_index_head(dbindex_path, head);
}

// makedb on the GPU: FASTA -> resident index -> the five .dbindex files
int walt_makedb_device(const char* fasta_path, const char* out_path, int device) {
  if (!fasta_path || !out_path) return fail(WALT_EINVAL, "null path");
  std::vector<std::string> names;
  std::vector<uint32_t> lengths;
  std::vector<uint8_t> seq;
  int rc = read_fasta_genome(fasta_path, names, lengths, seq);
  if (rc) return rc;
  if (lengths.empty()) return fail(WALT_EFORMAT, "no sequence in " + std::string(fasta_path));
  WALT_HIP(hipSetDevice(device));
  void* d_genome = nullptr;
  hipError_t e = hipMalloc(&d_genome, seq.size() + 16);
  if (e != hipSuccess) return fail(WALT_ENOMEM, std::string("hipMalloc failed: ") + hipGetErrorString(e));
  if ((e = hipMemcpy(d_genome, seq.data(), seq.size(), hipMemcpyHostToDevice)) != hipSuccess) {
    hipFree(d_genome);
    return fail(WALT_EHIP, std::string("hipMemcpy failed: ") + hipGetErrorString(e));
  }
  std::vector<uint8_t>().swap(seq);
  std::vector<const char*> cn;
  for (const std::string& s : names) cn.push_back(s.c_str());
  walt_index* idx = nullptr;
  rc = walt_index_build_device(d_genome, (uint32_t)lengths.size(), lengths.data(), cn.data(), device, WALT_STRANDS_ALL, -1,
                               &idx);
  hipFree(d_genome);
  if (rc) return rc;
  rc = walt_index_write(idx, out_path);
  walt_index_close(idx);
  return rc;
}

}  // extern "C"
