/* walt_amd.h -- C ABI of the MI355X-native WALT seed-and-extend hot path.
 *
 * This is the drop-in boundary for the body of WALT's two `#pragma omp parallel
 * for` loops (reference src/walt/mapping.cpp:494-499 and paired.cpp:664-669)
 * plus the serial pair-merge loop (paired.cpp:684-699).  Plain pointers and
 * sizes only; no C++/torch types.  Every function returns 0 on success or a
 * negative WALT_E* code; walt_last_error() gives the message.  The library
 * never calls exit().  All four strand indexes stay resident in HBM, so ONE
 * call covers both strand passes that the reference makes per batch
 * (mapping.cpp:491-500).
 *
 * Citations are file:line in smithlabcode/walt v1.0.
 */
#ifndef WALT_AMD_H_
#define WALT_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WALT_OK 0
#define WALT_EINVAL (-1)   /* bad argument */
#define WALT_EIO (-2)      /* file missing / short read (reference: FREAD_CHECK exit, util.hpp:62-69) */
#define WALT_EHIP (-3)     /* HIP runtime error / no device */
#define WALT_EBASE (-4)    /* non-ACGT base in a read (reference: getBits exit, util.hpp:117-119) */
#define WALT_ENOMEM (-5)
#define WALT_EFORMAT (-6)  /* malformed .dbindex */

/* strand_mask bits for walt_index_open: which strand files to make resident */
#define WALT_STRAND_CT00 1u
#define WALT_STRAND_CT01 2u
#define WALT_STRAND_GA10 4u
#define WALT_STRAND_GA11 8u
#define WALT_STRANDS_CT 3u   /* single-end default (mapping.cpp:443-445) */
#define WALT_STRANDS_GA 12u  /* single-end -A (mapping.cpp:446-449) */
#define WALT_STRANDS_ALL 15u /* paired-end (paired.cpp:589-593) */

/* BestMatch, mapping.hpp:39-52: 16 bytes, strand char at offset 8. */
typedef struct {
  uint32_t genome_pos;
  uint32_t times;
  char strand;
  char pad_[3]; /* written as 0 */
  uint32_t mismatch;
} walt_best_match;

/* CandidatePosition, paired.hpp:35-46: 12 bytes. */
typedef struct {
  uint32_t genome_pos;
  char strand;
  char pad_[3];
  uint32_t mismatch;
} walt_candidate;

/* What MergePairedEndResults (paired.cpp:474-545) derives for one pair. */
typedef struct {
  walt_best_match m1, m2; /* per-mate records the writers print (paired.cpp:515-569) */
  uint32_t best_times;    /* 0: no pair, 1: unique proper pair, >=2: ambiguous */
  int32_t frag_len;       /* `len` of OutputBestPairedResults when best_times==1, else 0 */
  int32_t best_i, best_j; /* indices into the ranked lists when best_times==1, else -1 */
  uint32_t pair_mm;       /* r1.mismatch + r2.mismatch of the reported pair */
  uint32_t pad_[3];
} walt_pair_result;

/* Work/statistics block returned by the batch calls (all counters are sums
 * over the batch; too_short counts one per strand pass like
 * stat.num_of_short_reads++ at mapping.cpp:230-233 / paired.cpp:112-115). */
typedef struct {
  uint64_t too_short;
  /* Diagnostic work counters of the kernels, NOT the reference's.  `probes` counts seed probes that found a
   * NON-EMPTY REGION (at least one index slot whose care characters equal the seed's) -- fewer than the probes into
   * non-empty buckets the reference's loop makes (mapping.cpp:268-274; bench.py's oracle counts those: ~4.0 per
   * read on the benchmark genome against ~1.4 here), although the kernels issue MORE look-ups than the reference
   * (a superset on the '-' strand; every probe of a staged paired-end read up to the exits it can prove).  A read
   * that moves on to the literal pass after part of its work is counted in both places. */
  uint64_t probes;     /* seed probes whose region is not empty */
  uint64_t candidates; /* candidates verified */
  uint64_t big_regions;/* regions handed to a wavefront (work items / cooperative verification) */
} walt_batch_stats;

typedef struct walt_index walt_index;

const char* walt_last_error(void);
int walt_device_count(void);

/* Seed pattern this library was compiled for.  The reference selects it at compile time
 * (-D SEEDPATTERN3 / 5 / 7, src/walt/Makefile:34, FAQ.md:5-13; tables in seedpattern.hpp); here
 * libwalt_amd.so is pattern 3 and libwalt_amd_sp5.so / libwalt_amd_sp7.so are the other two
 * (make PAT=5 / PAT=7).  An index is specific to the pattern of the makedb that wrote it.
 * walt_min_read_len() is MINIMALREADLEN (38 / 32 / 23; shorter reads count as too_short);
 * walt_max_read_len() is the longest read a batch may hold: 1024, or 148 / 152 for patterns 5 / 7,
 * beyond which the reference reads its seed tables out of bounds (mapping.cpp:238 caps the repeats
 * at 50, the tables hold 28 / 20). */
int walt_seed_pattern(void);
uint32_t walt_min_read_len(void);
uint32_t walt_max_read_len(void);

/* Page-locked host memory for the read / result buffers a caller hands to
 * walt_map_se_batch / walt_map_pe_batch (the reference keeps them in
 * std::vector<std::string>, mapping.cpp:462-464); transfers from such buffers
 * run at PCIe rate.  Ordinary pageable pointers are accepted by every call too. */
int walt_host_alloc(size_t bytes, void** out);
void walt_host_free(void* p);

/* ---- index ------------------------------------------------------------- */

/* Replaces ReadIndexHeadInfo + per-batch ReadIndex (reference.cpp:381-417,
 * 324-351; call sites mapping.cpp:437,492 and paired.cpp:583,661): reads
 * <path> and the selected <path>_CT00/_CT01/_GA10/_GA11 files once, uploads
 * them to `device` and builds the derived HBM structures (2-bit genome, entry
 * keys, directory).  dir_bits = directory prefix length in bits (24..32); < 0
 * picks it from the index size. */
int walt_index_open(const char* dbindex_path, int device, unsigned strand_mask, int dir_bits,
                    walt_index** out);

/* Same from host arrays laid out exactly like the .dbindex strand files
 * (reference.cpp:302-322): per strand s in {CT00,CT01,GA10,GA11} (NULL = not
 * present) genome bytes [genome_len], counter [4^12+1], index [index_size]. */
int walt_index_from_host(uint32_t n_chrom, const uint32_t* chrom_len, const char* const* chrom_names,
                         const uint8_t* const genome[4], const uint32_t* const counter[4],
                         const uint32_t* const index[4], const uint32_t index_size[4], int device,
                         int dir_bits, walt_index** out);

void walt_index_close(walt_index* idx);

/* Genome::num_of_chroms / length / name / start_index (reference.hpp:44-70). */
uint32_t walt_index_n_chrom(const walt_index* idx);
uint32_t walt_index_chrom_len(const walt_index* idx, uint32_t i);
const char* walt_index_chrom_name(const walt_index* idx, uint32_t i);
uint64_t walt_index_genome_len(const walt_index* idx);
uint64_t walt_index_device_bytes(const walt_index* idx);
int walt_index_dir_bits(const walt_index* idx);
/* diagnostics, per strand: buckets in which EVERY probe takes the literal search (0 for a
 * makedb-built index), and chromosome-end entries ("outliers") around which probes do */
uint64_t walt_index_bad_buckets(const walt_index* idx, int strand);
uint64_t walt_index_outliers(const walt_index* idx, int strand);
/* index entries whose genome window is also stored in slot order ("dense candidate windows": the regions of
 * thousands of candidates that repeats produce are then verified from contiguous memory) */
uint64_t walt_index_window_entries(const walt_index* idx, int strand);
/* index entries in runs that qualify for a dense window; larger than walt_index_window_entries when the memory budget
 * ended before the last run (those regions are verified from the scattered genome windows: same results, slower) */
uint64_t walt_index_window_eligible(const walt_index* idx, int strand);

/* ---- single-end: replaces the strand loop + omp loop over SingleEndMapping,
 *      mapping.cpp:486-500 / 224-316 ------------------------------------- */

/* Host buffers.  bases: concatenated sanitised reads (only ACGT, as
 * LoadReadsFromFastqFile leaves them, mapping.cpp:101-103); offsets[n+1].
 * out[n] is initialised by the callee to (0,0,'+',max_mm) (mapping.cpp:486-489)
 * and holds the state after the '+' and '-' passes. */
int walt_map_se_batch(walt_index* idx, const char* bases, const uint64_t* offsets, uint32_t n,
                      int ag_wildcard, uint32_t max_mismatches, uint32_t b, walt_best_match* out,
                      walt_batch_stats* stats);

/* Device-resident form (pointers are HBM addresses on idx's device; stream is a
 * hipStream_t or NULL).  Asynchronous: returns after enqueueing; d_stats
 * (walt_batch_stats, device memory) is accumulated into, not cleared.
 * d_workspace must hold walt_se_workspace_bytes(n, max_read_len) bytes; the call is told how many it holds
 * (workspace_bytes) and refuses a smaller one with WALT_EINVAL instead of writing beyond it.  No option of the
 * index (walt_index_set_option) makes a call need more than that.
 * One single-end call at a time per index (its side streams and events belong to the index): a second call that
 * arrives while one is being enqueued is refused with WALT_EINVAL; different indexes are independent. */
size_t walt_se_workspace_bytes(uint32_t n, uint32_t max_read_len);
int walt_map_se_batch_device(walt_index* idx, const void* d_bases, const void* d_offsets, uint32_t n,
                             uint32_t max_read_len, int ag_wildcard, uint32_t max_mismatches,
                             uint32_t b, void* d_out, void* d_stats, void* d_workspace,
                             size_t workspace_bytes, void* stream);

/* The device-resident calls are asynchronous, so invalid input cannot come back as their status.
 * walt_batch_check waits for `stream` and reports what the last call on `d_workspace` found:
 * WALT_EBASE (a read holds a non-ACGT base: the reference's getBits exits, util.hpp:117-119; its
 * record is left as initialised), WALT_EINVAL (a read longer than max_read_len, or a batch of more than
 * n x max_read_len bases -- the workspace is sized by that product: such reads are refused in the kernels, never
 * converted or read beyond the workspace, their records left as initialised), WALT_EHIP (internal: a work-item
 * queue overflowed -- never with a workspace of the promised size), else WALT_OK.
 * The host-buffer calls do this themselves. */
int walt_batch_check(const void* d_workspace, void* stream);

/* ---- paired-end: replaces PairEndMapping over both mates and strands
 *      (paired.cpp:642-672 / 106-201), the heap drain (685-692) and the pair
 *      search of MergePairedEndResults (474-545) ----------------------------- */

/* ranked1/ranked2 (optional, may be NULL): n*top_k candidates per mate in the
 * pop order of the reference's priority_queue, ranked_n1/2[n] their counts. */
int walt_map_pe_batch(walt_index* idx, const char* bases1, const uint64_t* offsets1,
                      const char* bases2, const uint64_t* offsets2, uint32_t n,
                      uint32_t max_mismatches, uint32_t b, uint32_t top_k, int frag_range,
                      walt_pair_result* out, walt_candidate* ranked1, uint32_t* ranked_n1,
                      walt_candidate* ranked2, uint32_t* ranked_n2, walt_batch_stats* stats /*[2]*/);

/* One paired-end call at a time per walt_index: the call's internal streams and events (mate 2 runs beside mate 1,
 * passes alternate between two pipeline slots) belong to the index.  Different indexes (devices) are independent.
 * Workspace: walt_pe_workspace_bytes is the LEAST a call needs (8 M-pair passes, staged lists in four rounds);
 * walt_pe_workspace_bytes_best is what it uses best on idx's device as it stands now -- 10 M-pair passes in one round
 * when the device has that much free memory -- and never less than the former.  The call takes the geometry the
 * workspace it is given has room for (workspace_bytes), so sizing and mapping cannot disagree, and nothing about the
 * choice is kept in the process: two indexes on two devices decide independently. */
size_t walt_pe_workspace_bytes(uint32_t n, uint32_t max_read_len, uint32_t top_k);
size_t walt_pe_workspace_bytes_best(walt_index* idx, uint32_t n, uint32_t max_read_len, uint32_t top_k);
int walt_map_pe_batch_device(walt_index* idx, const void* d_bases1, const void* d_offsets1,
                             const void* d_bases2, const void* d_offsets2, uint32_t n,
                             uint32_t max_read_len, uint32_t max_mismatches, uint32_t b,
                             uint32_t top_k, int frag_range, void* d_out, void* d_stats /*[2]*/,
                             void* d_workspace, size_t workspace_bytes, void* stream);

/* ---- options ---------------------------------------------------------------------------------
 * Tuning values and test hooks of the mapping calls, per index.  The mapping calls read NO environment
 * variable: an index maps the same way whatever the process environment holds (the library's only
 * environment inputs are read once, when an index is opened or built: WALT_AMD_WIN / _WIN_GB /
 * _WIN_RESERVE_GB (dense candidate windows: on/off, memory cap, memory left free), WALT_AMD_FENCE,
 * WALT_AMD_TABLE, WALT_AMD_VERBOSE; and WALT_AMD_RCCL, the library walt_comm_* binds).  Every option
 * changes the schedule of a call, never its results.  Names (value 0 / 1 unless said otherwise):
 *   se_pipe        1  staged heavy pass in two halves on two streams
 *   se_heavy_chunk 0  reads per chunk of the heavy list (0: default; a test hook for several chunks on a small batch)
 *   se_stage_blocks / se_verify_blocks 0  blocks per compute unit of the stage / dense verifier launches (0: fill the device)
 *   se_stagger     0  the second half of the heavy pass starts one look-up stage behind the first
 *   se_lit_side    1  literal pass on a side stream beside the end of the heavy pass (0: after it; 2: what pass 1
 *                     deferred beside the whole heavy pass, the staged rounds' deferrals beside its end)
 *   se_lit_staged  0  reads with a truly dangerous probe go through staged rounds with the reference's search on instead
 *   se_defer_min  -1  long seeds: key-equal ranges of more slots than this go to the verifier unnarrowed (-1: default 4, 0: never)
 *   se_stage_occ   0  wavefronts per SIMD the stage kernel is built for (0: chosen by read length and sequence count;
 *                     3 / 4 for reads of up to 128 bases, 2 / 3 up to 160)
 *   se_carry       1  pass 1 hands its state to the staged rounds (0: they start over at seed 0)
 *   se_heavy_mono  0  the one-kernel heavy pass instead of the staged rounds
 *   grid           0  blocks of the persistent kernels (0: 8 per compute unit)
 *   pe_mode        0  0: staged path, 1: list kernels only
 *   pe_chunk       0  pairs per pass (0: default)          pe_rounds    0  rounds of a staged list (0: default, 1, 2, 4)
 *   pe_stage_cap   0  staged reads per round (0: default)  pe_small_heaps 0  force the 8-slot heaps of long literal lists
 *   pe_serial      0  mates and passes on one stream (profiling)   pe_push_wide 0  4-byte heap entries in the push kernel (A/B)
 *   pe_defer_min  -1  as se_defer_min                      pe_roomy    -1  -1: by the workspace's size, 0 / 1: forced
 *   pe_lit_fuse    1  the literal round's three seed shifts in one launch when its list is short (0: seed by seed)
 * Set between calls, not during one.  WALT_EINVAL for an unknown name. */
int walt_index_set_option(walt_index* idx, const char* name, long long value);
int walt_index_get_option(const walt_index* idx, const char* name, long long* value);

/* ---- makedb-compatible index builders (reference.cpp:79-322,
 *      makedb.cpp:46-159) ---------------------------------------------------- */

/* Host builder: reads FASTA (file or directory of .fa), writes the five .dbindex
 * files.  Byte-identical to the reference makedb for N-free FASTA. */
int walt_makedb(const char* fasta_path, const char* out_dbindex_path, int threads);

/* The same files from the GPU builder (walt_index_build_device + walt_index_write): seconds
 * instead of hours at 3 Gbp.  Identical to the host builder except for the order of entries
 * whose 60 compared characters are all equal (ascending position here, whatever std::sort
 * leaves in the reference, reference.cpp:296-298); one N fill serves all four strands. */
int walt_makedb_device(const char* fasta_path, const char* out_dbindex_path, int device);

/* GPU builder: d_genome_ascii is the concatenated genome (upper-case ACGT, no N)
 * in HBM on `device`; builds the selected strand indexes there and leaves them
 * resident (BuildIndex, makedb.cpp:46-85).  Same result as the reference makedb
 * except for the order of entries whose 60 care characters are all equal
 * (std::sort leaves those in an unspecified order, reference.cpp:296-298). */
int walt_index_build_device(const void* d_genome_ascii, uint32_t n_chrom, const uint32_t* chrom_len,
                            const char* const* chrom_names, int device, unsigned strand_mask,
                            int dir_bits, walt_index** out);

/* HashTable::index_size of a resident strand (reference.hpp:84). */
uint32_t walt_index_size(const walt_index* idx, int strand);

/* Copy a resident strand back to host arrays laid out like the strand file
 * (reference.cpp:302-322): genome_out[genome_len] chars, counter_out[4^12+1],
 * index_out[index_size].  Any pointer may be NULL. */
int walt_index_export_strand(const walt_index* idx, int strand, uint8_t* genome_out,
                             uint32_t* counter_out, uint32_t* index_out);

/* WriteIndex x4 + WriteIndexHeadInfo (reference.cpp:302-322, 353-379) from the
 * resident index.  Writes the head file and the strand file of every resident
 * strand (a C->T-only index writes <path>, <path>_CT00 and <path>_CT01, which is
 * all the reference's single-end mode without -A reads, mapping.cpp:491-492). */
int walt_index_write(const walt_index* idx, const char* dbindex_path);

/* ---- multi-GPU: one process per GPU, the statistics block is the only exchange -----------------
 *
 * The reference is a single process; what it carries ACROSS reads is the statistics block that
 * ProcessSingledEndReads / ProcessPairedEndReads accumulate and print as <out>.mapstats
 * (StatSingleReads: total, unique, ambiguous, unmapped, too_short, mapping.hpp:94-100, updated at
 * mapping.cpp:318-327,504; StatPairedReads: 4 pair counters + one StatSingleReads per mate +
 * fragment_len_count[frag_range + 1], paired.hpp:96-105, updated at paired.cpp:519-547).  Reads shard
 * over the ranks in contiguous blocks, every rank holds an index replica, and at the end of the run
 * each rank hands its block to walt_stats_allreduce: ONE sum over RCCL (ncclAllReduce, uint64).
 *
 * walt_comm_unique_id: rank 0 makes the 128-byte id (ncclGetUniqueId) and the caller distributes it out
 * of band (a file, MPI, torch.distributed ...).  walt_comm_init: every rank joins with the same id
 * (ncclCommInitRank; collective, blocks until all ranks have called).  walt_stats_allreduce sums v[0..n)
 * over the ranks in place (host vector; collective, same n on every rank); with comm == NULL it is
 * the identity, which is what a single process needs.  bin/walt, one process driving several GPUs,
 * adds its per-device blocks on the host instead.
 * walt_comm_available: WALT_OK when librccl could be loaded in this process (no communication) -- a job can
 * agree on that BEFORE any rank enters the blocking calls (walt_comm_init and walt_stats_allreduce have no
 * timeout: a rank that never arrives leaves its peers waiting, as with ncclCommInitRank / ncclAllReduce
 * themselves). */
#define WALT_COMM_ID_BYTES 128
typedef struct walt_comm walt_comm;
int walt_comm_available(void);
int walt_comm_unique_id(void* id_out /* WALT_COMM_ID_BYTES */);
int walt_comm_init(int device, int rank, int world, const void* id, walt_comm** out);
int walt_stats_allreduce(walt_comm* comm, uint64_t* v, size_t n);
int walt_comm_rank(const walt_comm* comm);
int walt_comm_world(const walt_comm* comm);
void walt_comm_close(walt_comm* comm);

/* ---- measurement hooks (bench.py) ---------------------------------------- */

/* When enabled, the *_device batch calls record HIP events on their stream
 * around the read-packing and the mapping kernels; walt_profile_last waits for
 * the last call's events and returns the two durations in milliseconds. */
int walt_profile_enable(walt_index* idx, int on);
int walt_profile_last(walt_index* idx, float* pack_ms, float* map_ms);
/* The last single-end call's mapping time by kernel group, from events between the groups: out4 = milliseconds of
 * {pass 1, heavy stages, region verifier, literal pass incl. its sort}. */
int walt_profile_detail(walt_index* idx, float* out4);
/* Diagnostic BUILD only (make -C walt_amd/csrc diag: libwalt_amd_diag.so, environment WALT_AMD_STAMPS=1):
 * in-kernel s_memtime sums per phase of the single-end mapping kernel, cycles summed over waves; reading
 * clears them.  The product library returns WALT_EINVAL. */
int walt_profile_stamps(unsigned long long* out16);

#ifdef __cplusplus
}
#endif
#endif /* WALT_AMD_H_ */
