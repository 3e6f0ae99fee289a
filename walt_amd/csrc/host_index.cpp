// host_index.cpp -- host side of the index: literal seed tables, .dbindex file
// I/O, and a makedb-compatible builder.  No GPU code here.
//
// File formats follow WriteIndex / WriteIndexHeadInfo (reference.cpp:302-322,
// 353-379): little-endian, no magic.
#include <dirent.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <fstream>
#include <string>
#include <vector>

#include "../../include/walt_amd.h"
#include "host_common.h"

namespace walt {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
const char* last_error_cstr() { return g_err.c_str(); }

// ---------------------------------------------------------------------------
// Seed tables.  F2NOCAREDPOSITION rows are the ascending non-care offsets of
// the (010)* seed shifted by s, with the four literal deviations of
// seedpattern.hpp:439,451,454,455; rows carry 121/121/122 explicit values and
// zeros up to 150 slots.  tests/test_seedtab.py checks them against a golden
// dump of the reference header.
// ---------------------------------------------------------------------------
static uint32_t g_nocare[kPat][150];
static std::vector<uint32_t> g_mask_table;
static bool g_tables_ready = false;

static void init_tables() {
  if (g_tables_ready) return;
  // explicit values per row in the reference header: pattern 3: 121/121/122 of 150 (seedpattern.hpp:431-455),
  // pattern 5: 84+s of 90 (281-351), pattern 7: 60+s of 70 (82-222); C++ zero-fills the rest of a row
  for (uint32_t s = 0; s < kPat; ++s) {
    const int explicit_len = kPat == 3 ? (s == 2 ? 122 : 121) : (kPat == 5 ? 84 + (int)s : 60 + (int)s);
    int n = 0;
    for (uint32_t p = 0; n < explicit_len; ++p) {
      bool is_care = false;
      if (p >= s)
        for (uint32_t k = 0; k < kCareW; ++k) is_care = is_care || (p - s) % kPat == care_pos(k);
      if (!is_care) g_nocare[s][n++] = p;
    }
    for (; n < 150; ++n) g_nocare[s][n] = 0;
  }
  if (kPat == 3) {  // the four literal deviations of the pattern-3 table (patterns 5 and 7 follow the formula)
    g_nocare[0][118] = 178;
    g_nocare[2][47] = 60;
    g_nocare[2][95] = 141;
    g_nocare[2][115] = 171;
  }

  // compare masks: for (seed_i, repeats) the offsets F2NOCAREDPOSITION[seed_i][p],
  // p < kNoCareW*repeats + seed_i (mapping.cpp:290-298), one bit per base at bit 2k.
  const uint32_t nrep = kMaxRepeats - kMinRepeats + 1;
  g_mask_table.assign(kPat * nrep * kMaskWords, 0);
  for (uint32_t s = 0; s < kPat; ++s) {
    for (uint32_t rep = kMinRepeats; rep <= kMaxRepeats; ++rep) {
      uint32_t n_nocare = kNoCareW * rep + s;
      for (uint32_t p = 0; p < n_nocare; ++p) {
        uint32_t q = g_nocare[s][p];
        // The single-mask formulation needs every listed offset to be distinct
        // and below the tail start kPat*rep+s; true for all reachable prefixes
        // (checked here so a table change cannot silently break it).
        uint32_t& word = g_mask_table[mask_table_index(s, rep, q >> 4)];
        uint32_t bit = 1u << (2 * (q & 15));
        if (q >= kPat * rep + s || (word & bit)) {
          fprintf(stderr, "walt_amd: seed table invariant broken (s=%u rep=%u p=%u)\n", s, rep, p);
          abort();
        }
        word |= bit;
      }
    }
  }
  g_tables_ready = true;
}

const uint32_t* nocare_row(int seed_i) {
  init_tables();
  return g_nocare[seed_i];
}
const std::vector<uint32_t>& compare_mask_table() {
  init_tables();
  return g_mask_table;
}

// ---------------------------------------------------------------------------
// .dbindex I/O
// ---------------------------------------------------------------------------
static bool rd(FILE* f, void* p, size_t sz, size_t n) { return fread(p, sz, n, f) == n; }

int read_index_head(const std::string& path, IndexHead& head) {  // reference.cpp:381-417
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return fail(WALT_EIO, "cannot open input file " + path);
  uint32_t n = 0;
  bool ok = rd(f, &n, 4, 1);
  if (ok && n > (1u << 24)) ok = false;
  head.names.assign(ok ? n : 0, std::string());
  head.lengths.assign(ok ? n : 0, 0);
  for (uint32_t i = 0; ok && i < n; ++i) {
    uint32_t len = 0;
    char buf[256];
    ok = rd(f, &len, 4, 1) && len <= 255 && rd(f, buf, 1, len);
    if (ok) head.names[i].assign(buf, len);
  }
  ok = ok && rd(f, head.lengths.data(), 4, n) && rd(f, &head.genome_len, 4, 1) &&
       rd(f, &head.max_index_size, 4, 1);
  fclose(f);
  if (!ok) return fail(WALT_EFORMAT, "read file error (index head) " + path);
  uint64_t sum = 0;
  for (uint32_t l : head.lengths) sum += l;
  if (sum != head.genome_len) return fail(WALT_EFORMAT, "index head: chromosome lengths do not sum to genome length");
  // positions and position + read length are 32-bit here as in the reference; the top 256 values stay free
  if (head.genome_len >= 0xFFFFFF00u) return fail(WALT_EINVAL, "genome longer than 2^32 - 256 bases");
  return WALT_OK;
}

int write_index_head(const std::string& path, const IndexHead& head) {  // reference.cpp:353-379
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) return fail(WALT_EIO, "cannot open output file " + path);
  uint32_t n = (uint32_t)head.names.size();
  fwrite(&n, 4, 1, f);
  for (uint32_t i = 0; i < n; ++i) {
    uint32_t len = (uint32_t)head.names[i].size();
    if (len > 255) len = 255;
    fwrite(&len, 4, 1, f);
    fwrite(head.names[i].data(), 1, len, f);
  }
  fwrite(head.lengths.data(), 4, n, f);
  fwrite(&head.genome_len, 4, 1, f);
  fwrite(&head.max_index_size, 4, 1, f);
  fclose(f);
  return WALT_OK;
}

int read_strand_file(const std::string& path, uint32_t genome_len, StrandFile& sf) {  // reference.cpp:324-351
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return fail(WALT_EIO, "cannot open input file " + path);
  sf.genome.resize(genome_len);
  uint32_t counter_size = 0, index_size = 0;
  bool ok = rd(f, &sf.strand, 1, 1) && rd(f, sf.genome.data(), 1, genome_len) &&
            rd(f, &counter_size, 4, 1) && rd(f, &index_size, 4, 1);
  if (ok && counter_size != kNumBuckets) ok = false;
  if (ok) {
    sf.counter.resize((size_t)counter_size + 1);
    sf.index.resize(index_size);
    ok = rd(f, sf.counter.data(), 4, (size_t)counter_size + 1) && rd(f, sf.index.data(), 4, index_size);
  }
  fclose(f);
  if (!ok) return fail(WALT_EFORMAT, "read file error (strand index) " + path);
  return WALT_OK;
}

int write_strand_file(const std::string& path, const StrandFile& sf) {  // reference.cpp:302-322
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) return fail(WALT_EIO, "cannot open output file " + path);
  uint32_t counter_size = kNumBuckets, index_size = (uint32_t)sf.index.size();
  fwrite(&sf.strand, 1, 1, f);
  fwrite(sf.genome.data(), 1, sf.genome.size(), f);
  fwrite(&counter_size, 4, 1, f);
  fwrite(&index_size, 4, 1, f);
  fwrite(sf.counter.data(), 4, sf.counter.size(), f);
  fwrite(sf.index.data(), 4, sf.index.size(), f);
  bool ok = !ferror(f);
  fclose(f);
  return ok ? WALT_OK : fail(WALT_EIO, "write error " + path);
}

// ---------------------------------------------------------------------------
// makedb-compatible builder
// ---------------------------------------------------------------------------
struct HostGenome {
  std::vector<std::string> names;
  std::vector<uint32_t> lengths;
  std::vector<uint32_t> start;  // n + 1
  std::vector<uint8_t> seq;
};

static bool has_suffix(const std::string& name, const std::string& sfx) {  // is_valid_filename, smithlab_os.cpp:103-107
  return name.substr(name.find_last_of(".") + 1) == sfx;
}

static int list_chrom_files(const std::string& path, std::vector<std::string>& files) {  // reference.cpp:62-77
  DIR* d = opendir(path.c_str());
  if (!d) {
    files.push_back(path);
    return WALT_OK;
  }
  struct dirent* ent;
  while ((ent = readdir(d))) {
    if (has_suffix(ent->d_name, "fa")) files.push_back(path + "/" + ent->d_name);
  }
  closedir(d);
  if (files.empty()) return fail(WALT_EIO, "no valid files found in: " + path);
  return WALT_OK;
}

static char random_base() { return "ACGT"[rand() % 4]; }  // toACGT / getNT, util.hpp:90-104,156-163

// ReadGenome, reference.cpp:79-129 (FASTA parsing as read_fasta_file,
// smithlab_os.cpp:366-387: '>' lines start a sequence, every other line is
// appended whole).
static int read_genome(const std::vector<std::string>& files, HostGenome& g) {
  std::vector<std::string> seqs;
  g.names.clear();
  for (const std::string& fn : files) {
    std::ifstream in(fn.c_str());
    if (!in) return fail(WALT_EIO, "cannot open input file " + fn);
    std::string line;
    bool any = false;
    while (std::getline(in, line)) {
      if (!line.empty() && line[0] == '>') {
        std::string nm = line.substr(1);
        g.names.push_back(nm.substr(0, nm.find_first_of(" \t")));  // reference.cpp:94-95
        seqs.push_back(std::string());
        any = true;
      } else if (any) {
        seqs.back() += line;
      } else {
        return fail(WALT_EFORMAT, "FASTA does not start with '>': " + fn);
      }
    }
  }
  uint64_t total = 0;
  for (const std::string& s : seqs) total += s.size();
  if (total >= (1ull << 32)) return fail(WALT_EINVAL, "genome longer than 2^32 bases");
  size_t n = seqs.size();
  g.lengths.resize(n);
  g.start.resize(n + 1);
  g.seq.resize(total);
  uint32_t k = 0;
  for (size_t i = 0; i < n; ++i) {
    g.lengths[i] = (uint32_t)seqs[i].size();
    g.start[i] = k;
    for (char ch : seqs[i]) {
      char c = (char)toupper((unsigned char)ch);
      if (!(c == 'A' || c == 'C' || c == 'G' || c == 'T')) c = random_base();
      g.seq[k++] = (uint8_t)c;
    }
    std::string().swap(seqs[i]);
  }
  g.start[n] = k;
  return WALT_OK;
}

static void revcomp_genome(HostGenome& g) {  // reference.cpp:131-146
  for (size_t i = 0; i < g.lengths.size(); ++i) {
    uint8_t* a = g.seq.data() + g.start[i];
    uint32_t len = g.lengths[i];
    std::reverse(a, a + len);
    for (uint32_t j = 0; j < len; ++j) {
      uint8_t c = a[j];
      a[j] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A';
    }
  }
}

static inline uint32_t hash_bytes(const uint8_t* s) {  // getHashValue, util.hpp:175-182
  uint32_t h = 0;
  for (uint32_t i = 0; i < kKeyWeight; ++i) h = (h << 2) | base_code(s[care_pos(i)]);
  return h;
}

struct SortRec {
  uint64_t k0;   // care chars 12..43, 2 bits each, 0 = beyond the chromosome end
  uint64_t k1;   // care chars 44..75 (pattern 3: ..59, pattern 5: ..55)
  uint32_t k2;   // care chars 76..79 (pattern 7)
  uint32_t pos;
};
struct SortRecLess {  // SortHashTableBucketCMP, reference.cpp:258-288, on precomputed keys
  bool operator()(const SortRec& a, const SortRec& b) const {
    if (a.k0 != b.k0) return a.k0 < b.k0;
    if (a.k1 != b.k1) return a.k1 < b.k1;
    return a.k2 < b.k2;
  }
};

// One strand index: CountBucketSize, HashToBucket, SortHashTableBucket
// (reference.cpp:192-300).
static int build_strand(HostGenome& g, int indicator, int threads, StrandFile& sf) {
  if (indicator & 1) revcomp_genome(g);
  const bool ga = indicator >= 2;
  for (uint8_t& c : g.seq) {
    if (!ga && c == 'C') c = 'T';
    if (ga && c == 'G') c = 'A';
  }
  sf.strand = (indicator & 1) ? '-' : '+';
  sf.counter.assign((size_t)kNumBuckets + 1, 0);
  const size_t nchr = g.lengths.size();
  for (size_t i = 0; i < nchr; ++i) {
    if (g.lengths[i] < kMinSeedLen) continue;
    uint32_t end = g.start[i + 1] - kMinSeedLen;
    for (uint32_t j = g.start[i]; j < end; ++j) sf.counter[hash_bytes(&g.seq[j])]++;
  }
  std::vector<uint8_t> erased(kNumBuckets, 0);
  for (uint32_t i = 0; i < kNumBuckets; ++i) {
    if (sf.counter[i] >= kEraseBucket) {  // reference.cpp:211-218
      fprintf(stderr, "[NOTICE: ERASE THE BUCKET %u SINCE ITS SIZE IS %u]\n", i, sf.counter[i]);
      sf.counter[i] = 0;
      erased[i] = 1;
    }
  }
  uint64_t run = 0;
  for (uint32_t i = 0; i <= kNumBuckets; ++i) {  // exclusive prefix sum == the reference's shift dance
    uint64_t c = i < kNumBuckets ? sf.counter[i] : 0;
    sf.counter[i] = (uint32_t)run;
    run += c;
  }
  if (run >= (1ull << 32)) return fail(WALT_EINVAL, "index size overflow");
  sf.index.assign((size_t)run, 0);
  {
    std::vector<uint32_t> cursor(sf.counter.begin(), sf.counter.end() - 1);
    for (size_t i = 0; i < nchr; ++i) {
      if (g.lengths[i] < kMinSeedLen) continue;
      uint32_t end = g.start[i + 1] - kMinSeedLen;
      for (uint32_t j = g.start[i]; j < end; ++j) {
        uint32_t h = hash_bytes(&g.seq[j]);
        if (erased[h]) continue;
        sf.index[cursor[h]++] = j;
      }
    }
  }
  // bucket sort; std::sort on (key,pos) records makes the same comparisons and
  // moves as the reference's std::sort on positions with its comparator, so
  // the order of equal keys is identical (same libstdc++).
  const uint8_t* seq = g.seq.data();
  const uint32_t* start = g.start.data();
  const uint32_t n_chrom = (uint32_t)nchr;
  if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
  {
    std::vector<SortRec> recs;
#pragma omp for schedule(dynamic, 4096)
    for (int64_t b = 0; b < (int64_t)kNumBuckets; ++b) {
      uint32_t lo = sf.counter[b], hi = sf.counter[b + 1];
      if (hi - lo <= 1) continue;
      recs.resize(hi - lo);
      for (uint32_t j = lo; j < hi; ++j) {
        uint32_t p = sf.index[j];
        uint32_t chr = chrom_id(start, n_chrom, p);
        uint32_t room = start[chr + 1] - p;  // l1 / l2 of the comparator
        uint64_t k0 = 0, k1 = 0;
        uint32_t k2 = 0;
        for (uint32_t q = kKeyWeight; q < kNumCare; ++q) {
          uint32_t cp = care_pos(q);
          uint32_t v = cp >= room ? 0u : (seq[p + cp] == 'A' ? 1u : seq[p + cp] == 'T' ? 3u : 2u);
          if (q < kKeyWeight + 32) k0 = (k0 << 2) | v;
          else if (q < kKeyWeight + 64) k1 = (k1 << 2) | v;
          else k2 = (k2 << 2) | v;
        }
        recs[j - lo].k0 = k0;
        recs[j - lo].k1 = k1;
        recs[j - lo].k2 = k2;
        recs[j - lo].pos = p;
      }
      std::sort(recs.begin(), recs.end(), SortRecLess());
      for (uint32_t j = lo; j < hi; ++j) sf.index[j] = recs[j - lo].pos;
    }
  }
  sf.genome = g.seq;
  return WALT_OK;
}

int read_fasta_genome(const char* fasta_path, std::vector<std::string>& names, std::vector<uint32_t>& lengths,
                      std::vector<uint8_t>& seq) {
  if (!fasta_path) return fail(WALT_EINVAL, "null path");
  const char* seed_env = getenv("WALT_MAKEDB_SEED");
  srand(seed_env ? (unsigned)atoi(seed_env) : (unsigned)time(NULL));
  std::vector<std::string> files;
  int rc = list_chrom_files(fasta_path, files);
  if (rc) return rc;
  HostGenome g;
  if ((rc = read_genome(files, g))) return rc;
  names.swap(g.names);
  lengths.swap(g.lengths);
  seq.swap(g.seq);
  return WALT_OK;
}

}  // namespace walt

using namespace walt;

extern "C" {

const char* walt_last_error(void) { return last_error_cstr(); }

int walt_seed_pattern(void) { return (int)kPat; }
uint32_t walt_min_read_len(void) { return kMinReadLen; }
uint32_t walt_max_read_len(void) { return kMaxReadLen; }

// makedb main flow, makedb.cpp:128-159.  The reference seeds rand() from the
// clock (makedb.cpp:88) and re-reads the genome for each of the four indexes
// and once more for the head file; WALT_MAKEDB_SEED pins the seed instead.
int walt_makedb(const char* fasta_path, const char* out_path, int threads) {
  if (!fasta_path || !out_path) return fail(WALT_EINVAL, "null path");
  const char* seed_env = getenv("WALT_MAKEDB_SEED");
  srand(seed_env ? (unsigned)atoi(seed_env) : (unsigned)time(NULL));
  std::vector<std::string> files;
  int rc = list_chrom_files(fasta_path, files);
  if (rc) return rc;
  static const char* sfx[4] = {"_CT00", "_CT01", "_GA10", "_GA11"};
  uint32_t max_index = 0;
  IndexHead head;
  for (int ind = 0; ind < 4; ++ind) {
    HostGenome g;
    rc = read_genome(files, g);
    if (rc) return rc;
    StrandFile sf;
    rc = build_strand(g, ind, threads, sf);
    if (rc) return rc;
    rc = write_strand_file(std::string(out_path) + sfx[ind], sf);
    if (rc) return rc;
    if (sf.index.size() > max_index) max_index = (uint32_t)sf.index.size();
  }
  {
    HostGenome g;
    rc = read_genome(files, g);
    if (rc) return rc;
    head.names = g.names;
    head.lengths = g.lengths;
    head.genome_len = (uint32_t)g.seq.size();
    head.max_index_size = max_index;
  }
  return write_index_head(out_path, head);
}

}  // extern "C"
