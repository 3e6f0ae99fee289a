// walt_oracle.cpp -- CPU restatement of the WALT seed-and-extend hot path.
//
// *** TEST INFRASTRUCTURE ONLY ***  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load this.  The product (walt_amd/) never
// links, imports or calls anything under oracle/.
//
// Parity status: PINNED.  This restatement is checked (tests/test_oracle_*.py)
//   (i)  against the committed golden outputs in tests/golden/ that were
//        produced by the real reference binaries (oracle/_ref/walt, makedb,
//        built by oracle/Makefile.ref from /root/reference), and
//   (ii) when oracle/_ref/ is present, live against the reference binary on
//        seeded random inputs.
//
// Every function cites the reference file:line it restates (paths relative to
// the reference checkout, smithlabcode/walt v1.0).  Nothing here is copied
// from the reference; data structures are flat arrays instead of
// std::string/std::vector, and the seed tables are generated from the
// (010)* formula plus the literal deviations of the reference header.
//
// Build: g++ -O3 -fopenmp -shared -fPIC -o liboracle.so walt_oracle.cpp
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <queue>
#include <vector>

extern "C" {

// ---------------------------------------------------------------------------
// Seed pattern tables (src/walt/seedpattern.hpp), selected at compile time like the reference's
// -D SEEDPATTERN3 / 5 / 7 (src/walt/Makefile:34): -DORC_PAT=3 (default), 5 or 7.
//   pattern 3 (355-456): (0 1 0)*,         MINIMALREADLEN 38, MINIMALSEEDLEN 36, 60 care positions 1 + 3 i,
//                        no-care rows [3][150] with 121/121/122 explicit values
//   pattern 5 (226-352): (1 0 1 0 0)*,     32, 30, 56 care positions {0,2} + 5 r, rows [5][90], 84 + s explicit
//   pattern 7 (29-223):  (1 1 1 0 1 0 0)*, 23, 21, 80 care positions {0,1,2,4} + 7 r, rows [7][70], 60 + s explicit
//   F2SEEDKEYWEIGHT = 12 in all three.
//   F2NOCAREDPOSITION[s] = ascending positions that are not care positions of the seed shifted by s;
//   the rest of a row is 0 (C++ zero fill).  Patterns 5 and 7 follow the formula exactly; pattern 3 has
//   four literal deviations:
//     row 0 idx 118: 178 (formula 177)  seedpattern.hpp:439  (unreachable)
//     row 2 idx  47:  60 (formula  70)  seedpattern.hpp:451  (reachable)
//     row 2 idx  95: 141 (formula 142)  seedpattern.hpp:454  (reachable)
//     row 2 idx 115: 171 (formula 172)  seedpattern.hpp:455  (unreachable)
//   (tests/golden/seedpattern{3,5,7}.json are data dumps of the header's tables.)
// The reference caps the repeats at 50 for every pattern (mapping.cpp:238); with patterns 5 / 7 a read of
// more than 148 / 152 bases then indexes both tables past their ends (undefined behaviour).  The oracle
// keeps the cap and must not be given such reads.
// ---------------------------------------------------------------------------
#ifndef ORC_PAT
#define ORC_PAT 3
#endif
enum { ORC_KEYW = 12, ORC_PATLEN = ORC_PAT,
       ORC_CAREW = ORC_PAT == 3 ? 1 : (ORC_PAT == 5 ? 2 : 4), ORC_NOCAREW = ORC_PAT - ORC_CAREW,
       ORC_NCARE = ORC_PAT == 3 ? 60 : (ORC_PAT == 5 ? 56 : 80),
       ORC_NNOCARE = ORC_PAT == 3 ? 150 : (ORC_PAT == 5 ? 90 : 70),
       ORC_MINREAD = ORC_PAT == 3 ? 38 : (ORC_PAT == 5 ? 32 : 23),
       ORC_MINSEED = ORC_PAT == 3 ? 36 : (ORC_PAT == 5 ? 30 : 21),
       ORC_EXIT1 = ORC_PAT == 7 ? 4 : 2,  // mapping.cpp:253-262
       ORC_MAXREP = 50 };

static uint32_t g_care[ORC_NCARE];
static uint32_t g_nocare[ORC_PATLEN][ORC_NNOCARE];
static int g_tab_ready = 0;

static void orc_tables_init(void) {
  if (g_tab_ready) return;
  static const uint32_t off3[1] = {1}, off5[2] = {0, 2}, off7[4] = {0, 1, 2, 4};
  const uint32_t* off = ORC_PAT == 3 ? off3 : (ORC_PAT == 5 ? off5 : off7);
  for (int i = 0; i < ORC_NCARE; ++i) g_care[i] = (i / ORC_CAREW) * ORC_PATLEN + off[i % ORC_CAREW];
  for (int s = 0; s < ORC_PATLEN; ++s) {
    const int explicit_len = ORC_PAT == 3 ? (s == 2 ? 122 : 121) : (ORC_PAT == 5 ? 84 + s : 60 + s);
    int n = 0;
    for (uint32_t p = 0; n < explicit_len; ++p) {
      int is_care = 0;
      if (p >= (uint32_t)s)
        for (int k = 0; k < ORC_CAREW; ++k) is_care |= ((p - s) % ORC_PATLEN == off[k]);
      if (!is_care) g_nocare[s][n++] = p;
    }
    for (; n < ORC_NNOCARE; ++n) g_nocare[s][n] = 0;
  }
  if (ORC_PAT == 3) {
    g_nocare[0][118] = 178;
    g_nocare[2][47] = 60;
    g_nocare[2][95] = 141;
    g_nocare[2][115] = 171;
  }
  g_tab_ready = 1;
}

int orc_pattern(void) { return ORC_PAT; }

// Exposed so tests can compare the generated tables with the golden dump.
void orc_get_tables(uint32_t* care /*60 | 56 | 80*/, uint32_t* nocare /*3x150 | 5x90 | 7x70*/) {
  orc_tables_init();
  memcpy(care, g_care, sizeof(g_care));
  memcpy(nocare, g_nocare, sizeof(g_nocare));
}

// ---------------------------------------------------------------------------
// Index view: what ReadIndex (reference.cpp:324-351) leaves in Genome and
// HashTable (reference.hpp:44-92) for ONE strand file.
// Reads of genome bytes at or beyond genome_len (the reference indexes
// genome.sequence out of bounds there, mapping.cpp:172,188) are defined as 0,
// i.e. smaller than every base, matching what the reference observes in the
// zeroed slack behind its vector.
// ---------------------------------------------------------------------------
typedef struct {
  const uint8_t* genome;        // genome_len bytes, 'A','C','G','T' after C->T / G->A
  uint64_t genome_len;
  const uint32_t* counter;      // 4^12 + 1 bucket starts
  const uint32_t* index;        // index_size genome positions
  uint32_t index_size;
  const uint32_t* start_index;  // n_chrom + 1
  uint32_t n_chrom;
} orc_strand;

// BestMatch, mapping.hpp:39-52 (16 bytes, strand at offset 8).
typedef struct {
  uint32_t genome_pos;
  uint32_t times;
  char strand;
  char pad_[3];
  uint32_t mismatch;
} orc_best;

// CandidatePosition, paired.hpp:35-46 (12 bytes).
typedef struct {
  uint32_t genome_pos;
  char strand;
  char pad_[3];
  uint32_t mismatch;
} orc_cand;

// Work counters for SURVEY 8(d): probes P, binary-search steps S, verified
// candidates C (loop entries at mapping.cpp:288).
typedef struct {
  uint64_t probes;
  uint64_t steps;
  uint64_t cands;
  uint64_t too_short;
  uint64_t cands_big;  // of cands: those in regions of more than 16 slots
} orc_work;

// Optional per-read trace (bench.py: which sampled reads met large regions / the -b cap), accumulated over
// the strand passes.  region sizes are those IndexRegion returns (mapping.cpp:274), before the -b test.
typedef struct {
  uint32_t probes;      // probes into non-empty buckets
  uint32_t cands;       // candidates verified
  uint32_t max_region;  // largest region met, skipped ones included
  uint32_t over_b;      // regions skipped by the -b cap (mapping.cpp:275-277)
  uint32_t cands_big;   // of cands: those in regions of more than 16 slots (bench.py prices them as streamed records)
  uint32_t pad_[3];
} orc_trace;

static inline uint8_t gat(const orc_strand* x, uint64_t pos) {
  return pos < x->genome_len ? x->genome[pos] : 0;
}

// getBits, util.hpp:107-121.  Returns 4 for a non-ACGT char (the reference
// exits the process there; callers guarantee sanitised reads).
static inline uint32_t orc_bits(char c) {
  switch (c) {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    default: return 4;
  }
}

// getHashValue, util.hpp:175-182.
uint32_t orc_hash(const char* s) {
  orc_tables_init();
  uint32_t h = 0;
  for (int i = 0; i < ORC_KEYW; ++i) h = (h << 2) + (orc_bits(s[g_care[i]]) & 3);
  return h;
}

// getChromID, reference.cpp:43-60 (upper-mid binary search).
uint32_t orc_chrom_id(const uint32_t* start_index, uint32_t n_plus_1, uint32_t pos) {
  uint32_t l = 0, h = n_plus_1 - 1;
  while (l < h) {
    uint32_t m = (l + h + 1) / 2;
    if (pos >= start_index[m]) l = m; else h = m - 1;
  }
  return l;
}

// LowerBound, mapping.cpp:166-180.
static uint32_t orc_lower(const orc_strand* x, uint32_t low, uint32_t high, uint8_t chr,
                          uint32_t cmp_pos, uint64_t* steps) {
  while (low < high) {
    uint32_t mid = low + (high - low) / 2;
    ++*steps;
    uint8_t c = gat(x, (uint64_t)x->index[mid] + cmp_pos);
    if (c >= chr) high = mid; else low = mid + 1;
  }
  return low;
}

// UpperBound, mapping.cpp:182-196.
static uint32_t orc_upper(const orc_strand* x, uint32_t low, uint32_t high, uint8_t chr,
                          uint32_t cmp_pos, uint64_t* steps) {
  while (low < high) {
    uint32_t mid = low + (high - low + 1) / 2;
    ++*steps;
    uint8_t c = gat(x, (uint64_t)x->index[mid] + cmp_pos);
    if (c <= chr) low = mid; else high = mid - 1;
  }
  return low;
}

// IndexRegion, mapping.cpp:198-222.  In: half-open bucket [first, second).
// Out: inclusive [first, second], or the empty marker (1, 0).
void orc_index_region(const orc_strand* x, const char* seed, uint32_t seed_len,
                      uint32_t* first, uint32_t* second, uint64_t* steps) {
  orc_tables_init();
  uint32_t l = *first, u = *second - 1;
  for (uint32_t p = ORC_KEYW; p < seed_len; ++p) {
    uint32_t cp = g_care[p];
    uint8_t ch = (uint8_t)seed[cp];
    l = orc_lower(x, l, u, ch, cp, steps);
    u = orc_upper(x, l, u, ch, cp, steps);
    if (l == u && ch != gat(x, (uint64_t)x->index[l] + cp)) {
      *first = 1; *second = 0; return;
    }
  }
  if (l > u) { *first = 1; *second = 0; return; }
  *first = l; *second = u;
}

// Read conversion, mapping.cpp:142-164.
static void orc_convert(const char* in, uint32_t len, int ag, char* out) {
  for (uint32_t i = 0; i < len; ++i) {
    char c = in[i];
    if (ag) out[i] = (c == 'G') ? 'A' : c; else out[i] = (c == 'C') ? 'T' : c;
  }
  out[len] = 0;
}

// Seed geometry, mapping.cpp:235-239.
static inline void orc_seed_geom(uint32_t read_len, uint32_t* repeats, uint32_t* seed_len) {
  uint32_t r = (read_len - ORC_PATLEN + 1) / ORC_PATLEN;
  if (r > ORC_MAXREP) r = ORC_MAXREP;
  *repeats = r;
  *seed_len = r * ORC_CAREW;
}

// Mismatch count, mapping.cpp:288-304 (limit = best_match.mismatch there,
// cur_max_mismatches in paired.cpp:175-190).  The early exit is kept literally.
static uint32_t orc_count_mm(const orc_strand* x, const char* read, uint32_t read_len,
                             uint32_t genome_pos, uint32_t seed_i, uint32_t repeats,
                             uint32_t limit) {
  uint32_t mm = 0;
  uint32_t n_nocare = repeats * ORC_NOCAREW + seed_i;
  for (uint32_t p = 0; p < n_nocare && mm <= limit; ++p) {
    uint32_t q = g_nocare[seed_i][p];
    if (gat(x, (uint64_t)genome_pos + q) != (uint8_t)read[q]) ++mm;
  }
  for (uint32_t p = repeats * ORC_PATLEN + seed_i; p < read_len && mm <= limit; ++p) {
    if (gat(x, (uint64_t)genome_pos + p) != (uint8_t)read[p]) ++mm;
  }
  return mm;
}

// SingleEndMapping, mapping.cpp:224-316, one (read, strand) call.
void orc_se_map_read_trace(const orc_strand* x, const char* org_read, uint32_t read_len, char strand,
                           int ag_wildcard, uint32_t b, orc_best* best, orc_work* work, orc_trace* tr);
void orc_se_map_read(const orc_strand* x, const char* org_read, uint32_t read_len, char strand,
                     int ag_wildcard, uint32_t b, orc_best* best, orc_work* work) {
  orc_se_map_read_trace(x, org_read, read_len, strand, ag_wildcard, b, best, work, NULL);
}
void orc_se_map_read_trace(const orc_strand* x, const char* org_read, uint32_t read_len, char strand,
                           int ag_wildcard, uint32_t b, orc_best* best, orc_work* work, orc_trace* tr) {
  orc_tables_init();
  if (read_len < ORC_MINREAD) { ++work->too_short; return; }
  uint32_t repeats, seed_len;
  orc_seed_geom(read_len, &repeats, &seed_len);
  std::vector<char> buf(read_len + 8);  // zero slack: pattern 7 hashes up to offset 24 of a 23-base read (the reference exits in getBits there)
  char* read = buf.data();
  orc_convert(org_read, read_len, ag_wildcard, read);

  for (uint32_t seed_i = 0; seed_i < ORC_PATLEN; ++seed_i) {
    if (best->mismatch == 0 && seed_i) break;          // mapping.cpp:250-251
    if (best->mismatch == 1 && seed_i >= ORC_EXIT1) break;  // mapping.cpp:253-262
    const char* seed = read + seed_i;                  // read.substr(seed_i), 265
    uint32_t h = orc_hash(seed);
    uint32_t first = x->counter[h], second = x->counter[h + 1];
    if (first == second) continue;                     // 271-272
    ++work->probes;
    orc_index_region(x, seed, seed_len, &first, &second, &work->steps);
    if (tr) {
      ++tr->probes;
      if (second - first + 1 > tr->max_region) tr->max_region = second - first + 1;
      if (second - first + 1 > b) ++tr->over_b;
    }
    if (second - first + 1 > b) continue;              // 275-277 (u32 wrap on the empty marker)
    for (uint32_t j = first; j <= second; ++j) {
      uint32_t gp = x->index[j];
      uint32_t chr = orc_chrom_id(x->start_index, x->n_chrom + 1, gp);
      if (gp - x->start_index[chr] < seed_i) continue;             // 282-283
      gp -= seed_i;
      if (gp + read_len >= x->start_index[chr + 1]) continue;      // 285-286
      ++work->cands;
      if (second - first + 1 > 16) ++work->cands_big;
      if (tr) { ++tr->cands; if (second - first + 1 > 16) ++tr->cands_big; }
      uint32_t mm = orc_count_mm(x, read, read_len, gp, seed_i, repeats, best->mismatch);
      if (mm < best->mismatch) {                                   // 306-313
        best->genome_pos = gp; best->times = 1; best->strand = strand; best->mismatch = mm;
      } else if (best->mismatch == mm && best->genome_pos != gp) {
        best->genome_pos = gp; best->strand = strand; best->times++;
      }
    }
  }
}

// Batch form of the strand loop in ProcessSingledEndReads, mapping.cpp:486-500:
// results initialised to (0,0,'+',max_mm), then every read is mapped against
// strands[0] with '+' and strands[1] with '-'.
// reads: concatenated sanitised bases, offsets[n+1].
void orc_se_map_batch(const orc_strand* strands /*[2]*/, const char* bases,
                      const uint64_t* offsets, uint32_t n, int ag_wildcard,
                      uint32_t max_mm, uint32_t b, int threads, orc_best* out,
                      orc_work* work_out) {
  orc_tables_init();
  for (uint32_t j = 0; j < n; ++j) {
    out[j].genome_pos = 0; out[j].times = 0; out[j].strand = '+';
    out[j].pad_[0] = out[j].pad_[1] = out[j].pad_[2] = 0;
    out[j].mismatch = max_mm;
  }
  orc_work total = {0, 0, 0, 0, 0};
  if (threads < 1) threads = 1;
  for (int fi = 0; fi < 2; ++fi) {
    const orc_strand* x = &strands[fi];
    char strand = fi == 0 ? '+' : '-';
    uint64_t p = 0, s = 0, c = 0, t = 0, g = 0;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 256) reduction(+ : p, s, c, t, g)
    for (int64_t j = 0; j < (int64_t)n; ++j) {
      orc_work w = {0, 0, 0, 0, 0};
      orc_se_map_read(x, bases + offsets[j], (uint32_t)(offsets[j + 1] - offsets[j]), strand,
                      ag_wildcard, b, &out[j], &w);
      p += w.probes; s += w.steps; c += w.cands; t += w.too_short; g += w.cands_big;
    }
    total.probes += p; total.steps += s; total.cands += c; total.too_short += t; total.cands_big += g;
  }
  if (work_out) *work_out = total;
}

// One strand pass of the same loop (mapping.cpp:491-500 body for one fi): `out`
// carries the BestMatch state in and out, so a caller that can hold only one
// strand index in host memory at a time (as the reference does) calls this
// once with '+' and once with '-'.  orc_se_init gives the initial state.
void orc_se_init(orc_best* out, uint32_t n, uint32_t max_mm) {
  for (uint32_t j = 0; j < n; ++j) {
    out[j].genome_pos = 0; out[j].times = 0; out[j].strand = '+';
    out[j].pad_[0] = out[j].pad_[1] = out[j].pad_[2] = 0;
    out[j].mismatch = max_mm;
  }
}
void orc_se_map_strand_trace(const orc_strand* x, char strand, const char* bases, const uint64_t* offsets,
                             uint32_t n, int ag_wildcard, uint32_t b, int threads, orc_best* out,
                             orc_work* work_out, orc_trace* trace /* n entries, accumulated; may be NULL */);
void orc_se_map_strand(const orc_strand* x, char strand, const char* bases, const uint64_t* offsets,
                       uint32_t n, int ag_wildcard, uint32_t b, int threads, orc_best* out,
                       orc_work* work_out) {
  orc_se_map_strand_trace(x, strand, bases, offsets, n, ag_wildcard, b, threads, out, work_out, NULL);
}
void orc_se_map_strand_trace(const orc_strand* x, char strand, const char* bases, const uint64_t* offsets,
                             uint32_t n, int ag_wildcard, uint32_t b, int threads, orc_best* out,
                             orc_work* work_out, orc_trace* trace) {
  orc_tables_init();
  if (threads < 1) threads = 1;
  uint64_t p = 0, s = 0, c = 0, t = 0, g = 0;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 256) reduction(+ : p, s, c, t, g)
  for (int64_t j = 0; j < (int64_t)n; ++j) {
    orc_work w = {0, 0, 0, 0, 0};
    orc_se_map_read_trace(x, bases + offsets[j], (uint32_t)(offsets[j + 1] - offsets[j]), strand, ag_wildcard, b,
                          &out[j], &w, trace ? &trace[j] : NULL);
    p += w.probes; s += w.steps; c += w.cands; t += w.too_short; g += w.cands_big;
  }
  if (work_out) {
    work_out->probes += p; work_out->steps += s; work_out->cands += c; work_out->too_short += t; work_out->cands_big += g;
  }
}

// ---------------------------------------------------------------------------
// Paired-end.
// ---------------------------------------------------------------------------
struct OrcCandLess {  // CandidatePosition::operator<, paired.hpp:39-41
  bool operator()(const orc_cand& a, const orc_cand& b) const { return a.mismatch < b.mismatch; }
};
typedef std::priority_queue<orc_cand, std::vector<orc_cand>, OrcCandLess> orc_pq;

// TopCandidates, paired.hpp:51-74 (libstdc++ std::priority_queue semantics).
struct OrcTop {
  orc_pq pq;
  size_t max_size;
  bool Full() const { return pq.size() >= max_size; }
  void Push(const orc_cand& c) {
    if (pq.size() < max_size) pq.push(c);
    else if (c.mismatch < pq.top().mismatch) { pq.pop(); pq.push(c); }
  }
};

// PairEndMapping, paired.cpp:106-201, one (read, strand) call.
static void orc_pe_map_read(const orc_strand* x, const char* org_read, uint32_t read_len,
                            char strand, int ag_wildcard, uint32_t max_mm, uint32_t b,
                            OrcTop* top, orc_work* work) {
  if (read_len < ORC_MINREAD) { ++work->too_short; return; }
  uint32_t repeats, seed_len;
  orc_seed_geom(read_len, &repeats, &seed_len);
  std::vector<char> buf(read_len + 8);  // zero slack: pattern 7 hashes up to offset 24 of a 23-base read (the reference exits in getBits there)
  char* read = buf.data();
  orc_convert(org_read, read_len, ag_wildcard, read);

  uint32_t cur_max = max_mm;
  for (uint32_t seed_i = 0; seed_i < ORC_PATLEN; ++seed_i) {
    if (!top->pq.empty() && top->Full() && top->pq.top().mismatch == 0 && seed_i) break;      // 133-135
    if (!top->pq.empty() && top->Full() && top->pq.top().mismatch == 1 && seed_i >= ORC_EXIT1) break; // 137-149
    const char* seed = read + seed_i;
    uint32_t h = orc_hash(seed);
    uint32_t first = x->counter[h], second = x->counter[h + 1];
    if (first == second) continue;
    ++work->probes;
    orc_index_region(x, seed, seed_len, &first, &second, &work->steps);
    if (second - first + 1 > b) continue;
    for (uint32_t j = first; j <= second; ++j) {
      uint32_t gp = x->index[j];
      uint32_t chr = orc_chrom_id(x->start_index, x->n_chrom + 1, gp);
      if (gp - x->start_index[chr] < seed_i) continue;
      gp -= seed_i;
      if (gp + read_len >= x->start_index[chr + 1]) continue;
      ++work->cands;
      if (second - first + 1 > 16) ++work->cands_big;
      uint32_t mm = orc_count_mm(x, read, read_len, gp, seed_i, repeats, cur_max);
      if (mm > max_mm) continue;                                   // 192-194
      orc_cand c; c.genome_pos = gp; c.strand = strand; c.pad_[0] = c.pad_[1] = c.pad_[2] = 0;
      c.mismatch = mm;
      top->Push(c);                                                // 195
      if (top->Full()) cur_max = top->pq.top().mismatch;           // 196-198
    }
  }
}

// Per-mate top-k lists as ProcessPairedEndReads builds and drains them
// (paired.cpp:655-671 then 685-692): for each read, map against strands[0]
// ('+') and strands[1] ('-') into one heap, then pop everything: ranked[]
// holds the pop order (descending mismatch), ranked_n the count.
// ranked is n * top_k entries.
void orc_pe_topk_batch(const orc_strand* strands /*[2]*/, const char* bases,
                       const uint64_t* offsets, uint32_t n, int ag_wildcard, uint32_t max_mm,
                       uint32_t b, uint32_t top_k, int threads, orc_cand* ranked,
                       uint32_t* ranked_n, orc_work* work_out) {
  orc_tables_init();
  if (threads < 1) threads = 1;
  uint64_t p = 0, s = 0, c = 0, t = 0, g = 0;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 64) reduction(+ : p, s, c, t, g)
  for (int64_t j = 0; j < (int64_t)n; ++j) {
    OrcTop top; top.max_size = top_k;
    orc_work w = {0, 0, 0, 0, 0};
    for (int fi = 0; fi < 2; ++fi)
      orc_pe_map_read(&strands[fi], bases + offsets[j], (uint32_t)(offsets[j + 1] - offsets[j]),
                      fi == 0 ? '+' : '-', ag_wildcard, max_mm, b, &top, &w);
    uint32_t k = 0;
    while (!top.pq.empty()) { ranked[(uint64_t)j * top_k + k++] = top.pq.top(); top.pq.pop(); }
    ranked_n[j] = k;
    p += w.probes; s += w.steps; c += w.cands; t += w.too_short; g += w.cands_big;
  }
  if (work_out) { work_out->probes = p; work_out->steps = s; work_out->cands = c; work_out->too_short = t; work_out->cands_big = g; }
}

// ForwardChromPosition, paired.cpp:98-104.
static inline void orc_fwd_pos(uint32_t gp, char strand, uint32_t chr, uint32_t read_len,
                               const uint32_t* start_index, uint32_t* s, uint32_t* e) {
  uint32_t len = start_index[chr + 1] - start_index[chr];
  uint32_t v = gp - start_index[chr];
  *s = strand == '+' ? v : len - v - read_len;
  *e = *s + read_len;
}

// GetBestMatch4Single, paired.cpp:296-318.
static void orc_best4single(const orc_cand* r, int n, orc_best* best) {
  for (int i = n - 1; i >= 0; --i) {
    if (r[i].mismatch < best->mismatch) {
      best->genome_pos = r[i].genome_pos; best->times = 1; best->strand = r[i].strand;
      best->mismatch = r[i].mismatch;
    } else if (r[i].mismatch == best->mismatch) {
      if (best->genome_pos == r[i].genome_pos) continue;
      best->genome_pos = r[i].genome_pos; best->strand = r[i].strand; best->times++;
    } else {
      break;
    }
  }
}

// Result of the pair search, the tuple paired.cpp:515-569 derives.
typedef struct {
  orc_best m1, m2;       // per-mate records that the writers print
  uint32_t best_times;   // 0 unmapped pair, 1 unique (proper pair), >=2 ambiguous
  int32_t frag_len;      // fragment length when best_times == 1 (SAM TLEN / histogram), else 0
  int32_t best_i, best_j;// indices into ranked lists when best_times == 1, else -1
  uint32_t pair_mm;      // r1.mismatch + r2.mismatch for the best pair
} orc_pair;

// Fragment length of the reported proper pair: OutputBestPairedResults,
// paired.cpp:210-243 (the value returned as `len`).
static int orc_pair_len(const orc_cand* r1, const orc_cand* r2, uint32_t len1, uint32_t len2,
                        const uint32_t* start_index, uint32_t n_plus_1) {
  uint32_t c1 = orc_chrom_id(start_index, n_plus_1, r1->genome_pos);
  uint32_t c2 = orc_chrom_id(start_index, n_plus_1, r2->genome_pos);
  uint32_t s1, e1, s2, e2;
  orc_fwd_pos(r1->genome_pos, r1->strand, c1, len1, start_index, &s1, &e1);
  orc_fwd_pos(r2->genome_pos, r2->strand, c2, len2, start_index, &s2, &e2);
  uint32_t ov_s = s1 > s2 ? s1 : s2;
  uint32_t ov_e = e1 < e2 ? e1 : e2;
  uint32_t one_l = r1->strand == '+' ? s1 : (ov_e > s1 ? ov_e : s1);
  uint32_t one_r = r1->strand == '+' ? (ov_s < e1 ? ov_s : e1) : e1;
  uint32_t two_l = r1->strand == '+' ? (ov_e > s2 ? ov_e : s2) : s2;
  uint32_t two_r = r1->strand == '+' ? e2 : (ov_s < e2 ? ov_s : e2);
  return r1->strand == '+' ? (int)(two_r - one_l) : (int)(one_r - two_l);
}

// MergePairedEndResults pair search + fallback, paired.cpp:474-545.
void orc_pe_merge(const orc_cand* r1, int n1, const orc_cand* r2, int n2, uint32_t len1,
                  uint32_t len2, const uint32_t* start_index, uint32_t n_chrom, int frag_range,
                  uint32_t max_mm, orc_pair* out) {
  uint32_t n_plus_1 = n_chrom + 1;
  int bi = -1, bj = -1;
  uint32_t min_mm = max_mm;
  uint64_t best_pos = 0;
  uint32_t best_times = 0;
  for (int i = n1 - 1; i >= 0; --i) {
    for (int j = n2 - 1; j >= 0; --j) {
      if (r1[i].strand == r2[j].strand) continue;                       // 482-483
      uint32_t mm = r1[i].mismatch + r2[j].mismatch;
      if (mm > min_mm) break;                                           // 486-487
      uint32_t c1 = orc_chrom_id(start_index, n_plus_1, r1[i].genome_pos);
      uint32_t c2 = orc_chrom_id(start_index, n_plus_1, r2[j].genome_pos);
      if (c1 != c2) continue;                                           // 491-492
      uint32_t s1, e1, s2, e2;                                          // GetFragmentLength 320-331
      orc_fwd_pos(r1[i].genome_pos, r1[i].strand, c1, len1, start_index, &s1, &e1);
      orc_fwd_pos(r2[j].genome_pos, r2[j].strand, c2, len2, start_index, &s2, &e2);
      int frag = r1[i].strand == '+' ? (int)(e2 - s1) : (int)(e1 - s2);
      if (frag <= 0 || frag > frag_range) continue;                     // 496-497
      uint64_t cur = ((uint64_t)r1[i].genome_pos << 32) + r2[j].genome_pos;
      if (mm < min_mm) {
        bi = i; bj = j; best_times = 1; min_mm = mm; best_pos = cur;
      } else if (mm == min_mm && cur != best_pos) {
        bi = i; bj = j; best_times++;                                   // best_pos NOT updated, 507-510
      }
    }
  }
  orc_best init; init.genome_pos = 0; init.times = 0; init.strand = '+';
  init.pad_[0] = init.pad_[1] = init.pad_[2] = 0; init.mismatch = max_mm;
  out->m1 = init; out->m2 = init;
  out->best_times = best_times;
  out->frag_len = 0; out->best_i = -1; out->best_j = -1; out->pair_mm = 0;
  if (best_times == 1) {
    out->best_i = bi; out->best_j = bj;
    out->frag_len = orc_pair_len(&r1[bi], &r2[bj], len1, len2, start_index, n_plus_1);
    out->pair_mm = r1[bi].mismatch + r2[bj].mismatch;
    out->m1.genome_pos = r1[bi].genome_pos; out->m1.times = 1; out->m1.strand = r1[bi].strand;
    out->m1.mismatch = r1[bi].mismatch;
    out->m2.genome_pos = r2[bj].genome_pos; out->m2.times = 1; out->m2.strand = r2[bj].strand;
    out->m2.mismatch = r2[bj].mismatch;
  } else {
    orc_best4single(r1, n1, &out->m1);
    orc_best4single(r2, n2, &out->m2);
  }
}

void orc_pe_merge_batch(const orc_cand* ranked1, const uint32_t* n1, const orc_cand* ranked2,
                        const uint32_t* n2, uint32_t top_k, const uint64_t* off1,
                        const uint64_t* off2, uint32_t n, const uint32_t* start_index,
                        uint32_t n_chrom, int frag_range, uint32_t max_mm, orc_pair* out) {
  for (uint32_t j = 0; j < n; ++j)
    orc_pe_merge(ranked1 + (uint64_t)j * top_k, (int)n1[j], ranked2 + (uint64_t)j * top_k,
                 (int)n2[j], (uint32_t)(off1[j + 1] - off1[j]), (uint32_t)(off2[j + 1] - off2[j]),
                 start_index, n_chrom, frag_range, max_mm, &out[j]);
}

}  // extern "C"
