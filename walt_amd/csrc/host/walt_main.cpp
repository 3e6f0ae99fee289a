// walt_main.cpp -- host driver with WALT's command-line surface on top of the
// C ABI (include/walt_amd.h).  It restates, in its own code, the parts of the
// reference that sit AROUND the hot path so that a user can swap binaries:
//   option table ............ walt.cpp:130-166 (single-dash long names)
//   FASTQ batch loader ...... mapping.cpp:65-121 (srand(0) per batch, N -> rand()%4)
//   adaptor clipping -C ..... util.hpp:189-233
//   SAM / MR / mapstats ..... mapping.cpp:47-63,329-419; paired.cpp:52-77,210-294,333-435,515-569
// The mapping itself (mapping.cpp:486-500, paired.cpp:642-699) is one library call
// per batch.  Output is byte-identical to the reference binary's
// (tests/test_gpu_cli.py compares against the golden files).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#include <algorithm>
#include <fstream>
#include <iostream>
#include <numeric>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/walt_amd.h"

using std::string;
using std::vector;

static const uint32_t MAX_LINE_LENGTH = 1000;  // util.hpp:43
static const int MINIMALREADLEN = 38;          // seedpattern.hpp:359

static void die(const string& msg) { throw std::runtime_error(msg); }
static void check(int rc) { if (rc != WALT_OK) die(walt_last_error()); }

// ---------------------------------------------------------------- options
struct Options {
  string index_file, se_csv, pe1_csv, pe2_csv, out_csv, adaptor;
  bool sam = false, ambiguous = false, unmapped = false, ag = false, verbose = false;
  uint32_t max_mismatches = 6, batch_size = 10000000, b = 5000, top_k = 50;
  int frag_range = 1000, threads = 1, device = 0;
};

static bool is_opt(const string& a, const char* s, const char* l) { return a == string("-") + s || a == string("-") + l || a == string("--") + l; }

static Options parse(int argc, const char** argv) {
  Options o;
  for (int i = 1; i < argc; ++i) {
    string a = argv[i];
    auto val = [&]() -> string {
      if (i + 1 >= argc) die("missing value for " + a);
      return argv[++i];
    };
    if (is_opt(a, "i", "index")) o.index_file = val();
    else if (is_opt(a, "r", "reads")) o.se_csv = val();
    else if (is_opt(a, "1", "reads1")) o.pe1_csv = val();
    else if (is_opt(a, "2", "reads2")) o.pe2_csv = val();
    else if (is_opt(a, "o", "output")) o.out_csv = val();
    else if (is_opt(a, "m", "mismatch")) o.max_mismatches = (uint32_t)strtoul(val().c_str(), 0, 10);
    else if (is_opt(a, "N", "number")) o.batch_size = (uint32_t)strtoul(val().c_str(), 0, 10);
    else if (is_opt(a, "a", "ambiguous")) o.ambiguous = true;
    else if (is_opt(a, "u", "unmapped")) o.unmapped = true;
    else if (is_opt(a, "C", "clip")) o.adaptor = val();
    else if (is_opt(a, "A", "ag-wild")) o.ag = true;
    else if (is_opt(a, "b", "bucket")) o.b = (uint32_t)strtoul(val().c_str(), 0, 10);
    else if (is_opt(a, "k", "topk")) o.top_k = (uint32_t)strtoul(val().c_str(), 0, 10);
    else if (is_opt(a, "L", "fraglen")) o.frag_range = atoi(val().c_str());
    else if (a == "-sam" || a == "--sam") o.sam = true;
    else if (is_opt(a, "v", "verbose")) o.verbose = true;
    else if (is_opt(a, "t", "thread")) o.threads = atoi(val().c_str());
    else if (is_opt(a, "g", "gpu")) o.device = atoi(val().c_str());  // extension: device ordinal
    else die("unknown option " + a);
  }
  if (o.index_file.empty() || o.out_csv.empty()) die("options -i and -o are required");
  return o;
}

static vector<string> split_csv(const string& s) {
  vector<string> out;
  std::istringstream is(s);
  string tok;
  while (std::getline(is, tok, ',')) if (!tok.empty()) out.push_back(tok);
  return out;
}
static bool valid_suffix(const string& fn) {  // walt.cpp:58-64 (checks .fastq / .fq)
  auto ends = [&](const string& sfx) { return fn.size() >= sfx.size() && fn.compare(fn.size() - sfx.size(), sfx.size(), sfx) == 0; };
  return ends(".fastq") || ends(".fq");
}
static bool exists(const string& p) { struct stat st; return stat(p.c_str(), &st) == 0; }

// ---------------------------------------------------------------- adaptor clipping, util.hpp:189-233
static const size_t head_length = 14, sufficient_head_match = 11, min_overlap = 5;
static size_t similarity(const string& s, size_t pos, const string& adaptor) {
  const size_t lim = std::min(std::min(s.length() - pos, adaptor.length()), head_length);
  size_t count = 0;
  for (size_t i = 0; i < lim; ++i) count += (s[pos + i] == adaptor[i]);
  return count;
}
static size_t clip_adaptor_from_read(const string& adaptor, string& s) {
  size_t lim1 = s.length() - head_length + 1;
  for (size_t i = 0; i < lim1; ++i)
    if (similarity(s, i, adaptor) >= sufficient_head_match) { std::fill(s.begin() + i, s.end(), 'N'); return s.length() - i; }
  const size_t lim2 = s.length() - min_overlap + 1;
  for (size_t i = lim1; i < lim2; ++i)
    if (similarity(s, i, adaptor) >= s.length() - i - 1) { std::fill(s.begin() + i, s.end(), 'N'); return s.length() - i; }
  return 0;
}
static void extract_adaptors(const string& adaptor, string& t_ad, string& a_ad) {
  const size_t sep = adaptor.find_first_of(":");
  if (adaptor.find_last_of(":") != sep) die("ERROR: adaptor format \"T_adaptor[:A_adaptor]\"");
  if (sep == string::npos) t_ad = a_ad = adaptor;
  else { t_ad = adaptor.substr(0, sep); a_ad = adaptor.substr(sep + 1); }
}

// ---------------------------------------------------------------- FASTQ loader, mapping.cpp:65-121
struct Batch {
  vector<string> names, seqs, scores;
  uint32_t n = 0;
};
static char to_acgt(char c) {  // toACGT, util.hpp:156-163
  if (c == 'A' || c == 'C' || c == 'G' || c == 'T') return c;
  return "ACGT"[rand() % 4];
}
static void load_batch(FILE* fin, uint32_t n_per_batch, const string& adaptor, Batch& bt) {
  srand(0);
  char cline[MAX_LINE_LENGTH];
  string line;
  int line_code = 0;
  uint32_t line_count = 0;
  bt.n = 0;
  const uint64_t lim = (uint64_t)n_per_batch * 4;
  while (line_count < lim && fgets(cline, MAX_LINE_LENGTH, fin)) {
    cline[strlen(cline) - 1] = 0;
    line = cline;
    if (line.size() == 0) continue;
    if (bt.names.size() <= bt.n) { bt.names.resize(bt.n + 1); bt.seqs.resize(bt.n + 1); bt.scores.resize(bt.n + 1); }
    switch (line_code) {
      case 0: {
        size_t sp = line.find_first_of(' ');
        bt.names[bt.n] = sp == string::npos ? line.substr(1) : line.substr(1, sp - 1);
        break;
      }
      case 1: {
        if (!adaptor.empty()) clip_adaptor_from_read(adaptor, line);
        for (size_t i = 0; i < line.size(); ++i) line[i] = to_acgt(line[i]);
        bt.seqs[bt.n] = line;
        break;
      }
      case 2: break;
      case 3: bt.scores[bt.n] = line; bt.n++; break;
    }
    ++line_count;
    if (++line_code == 4) line_code = 0;
  }
}

// ---------------------------------------------------------------- genome info + helpers
struct GenomeInfo {
  vector<string> name;
  vector<uint32_t> length, start;
};
static GenomeInfo genome_of(const walt_index* idx) {
  GenomeInfo g;
  uint32_t n = walt_index_n_chrom(idx);
  g.start.assign(n + 1, 0);
  for (uint32_t i = 0; i < n; ++i) {
    g.name.push_back(walt_index_chrom_name(idx, i));
    g.length.push_back(walt_index_chrom_len(idx, i));
    g.start[i + 1] = g.start[i] + g.length[i];
  }
  return g;
}
static uint32_t chrom_id(const GenomeInfo& g, uint32_t pos) {  // getChromID, reference.cpp:43-60
  uint32_t l = 0, h = (uint32_t)g.start.size() - 1;
  while (l < h) {
    uint32_t m = (l + h + 1) / 2;
    if (pos >= g.start[m]) l = m; else h = m - 1;
  }
  return l;
}
static string rev_str(const string& s) { return string(s.rbegin(), s.rend()); }
static string revcomp(const string& s) {
  string r(s.rbegin(), s.rend());
  for (char& c : r) c = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c;
  return r;
}
static void sam_head(const GenomeInfo& g, FILE* f) {  // SAMHead, reference.cpp:430-440
  fprintf(f, "@HD\tVN:1.0\n");
  for (size_t i = 0; i < g.name.size(); ++i) fprintf(f, "@SQ\tSN:%s\tLN:%u\n", g.name[i].c_str(), g.length[i]);
  fprintf(f, "@PG\tID:WALT\tVN:%s\tCL:%s\n", "1.0", "walt");
}

struct SeStats {  // StatSingleReads, mapping.hpp:55-108
  uint32_t total = 0, unique = 0, ambiguous = 0, unmapped = 0, too_short = 0;
  FILE* famb = nullptr;
  FILE* funm = nullptr;
  bool out_amb = false, out_unm = false, sam = false;
  void open(bool a, bool u, const string& out, bool is_sam) {
    out_amb = a; out_unm = u; sam = is_sam;
    if (a && !sam && !(famb = fopen((out + "_ambiguous").c_str(), "w"))) die("cannot open input file " + out + "_ambiguous");
    if (u && !sam && !(funm = fopen((out + "_unmapped").c_str(), "w"))) die("cannot open input file " + out + "_unmapped");
  }
  void close() { if (famb) fclose(famb); if (funm) fclose(funm); famb = funm = nullptr; }
  void update(uint32_t times) {  // StatInfoUpdate, mapping.cpp:318-327
    ++total;
    if (times == 0) unmapped++; else if (times == 1) unique++; else ambiguous++;
  }
  string tostring(size_t n_tabs = 0) const {  // mapping.cpp:47-63
    string t;
    for (size_t i = 0; i < n_tabs; ++i) t += "    ";
    std::ostringstream oss;
    oss << t << "total_reads: " << total << std::endl
        << t << "mapped:" << std::endl
        << t << "    unique: " << unique << std::endl
        << t << "    percent_unique: " << (100.0 * unique) / total << std::endl
        << t << "    ambiguous: " << ambiguous << std::endl
        << t << "unmapped: " << unmapped << std::endl
        << t << "min_read_length: " << MINIMALREADLEN << std::endl
        << t << "too_short: " << too_short;
    return oss.str();
  }
};

// ---------------------------------------------------------------- single-end writers, mapping.cpp:329-419
static void out_mr_line(const walt_best_match& bm, const string& name, const string& seq, const string& score,
                        const GenomeInfo& g, bool ag, FILE* f) {
  uint32_t chr = chrom_id(g, bm.genome_pos);
  uint32_t start = bm.genome_pos - g.start[chr];
  if (bm.strand == '-') start = g.length[chr] - start - (uint32_t)seq.size();
  uint32_t end = start + (uint32_t)seq.size();
  char strand = bm.strand;
  if (ag) strand = bm.strand == '+' ? '-' : '+';
  fprintf(f, "%s\t%u\t%u\t%s\t%u\t%c\t%s\t%s\n", g.name[chr].c_str(), start, end, name.c_str(), bm.mismatch, strand,
          seq.c_str(), score.c_str());
}
static void out_single_results(const walt_best_match& bm, const string& name, const string& seq, const string& score,
                               const GenomeInfo& g, bool ag, SeStats& st, FILE* fout) {
  string s = seq, q = score;
  if (ag) { s = revcomp(s); q = rev_str(q); }
  if (bm.times == 0 && st.out_unm) fprintf(st.funm, "%s\t%s\t%s\n", name.c_str(), s.c_str(), q.c_str());
  else if (bm.times == 1) out_mr_line(bm, name, s, q, g, ag, fout);
  else if (bm.times >= 2 && st.out_amb) out_mr_line(bm, name, s, q, g, ag, st.famb);
}
static void out_single_sam(const walt_best_match& bm, const string& name, const string& seq, const string& score,
                           const GenomeInfo& g, SeStats& st, FILE* fout) {
  uint32_t chr = chrom_id(g, bm.genome_pos);
  uint32_t start = bm.genome_pos - g.start[chr];
  if (bm.strand == '-') start = g.length[chr] - start - (uint32_t)seq.size();
  string s = seq, q = score;
  if (bm.strand == '-') { s = revcomp(s); q = rev_str(q); }
  uint32_t len = (uint32_t)seq.size();
  int flag = (bm.times == 0 ? 0x4 : 0) + (bm.strand == '-' ? 0x10 : 0) + (bm.times >= 2 ? 0x100 : 0);
  if (bm.times == 0 && st.out_unm)
    fprintf(fout, "%s\t%d\t*\t0\t255\t*\t*\t0\t0\t%s\t%s\tNM:i:0\n", name.c_str(), flag, s.c_str(), q.c_str());
  else if (bm.times == 1 || (bm.times >= 2 && st.out_amb))
    fprintf(fout, "%s\t%d\t%s\t%u\t255\t%uM\t*\t0\t0\t%s\t%s\tNM:i:%u\n", name.c_str(), flag, g.name[chr].c_str(),
            start + 1, len, s.c_str(), q.c_str(), bm.mismatch);
}

static void pack_batch(const Batch& bt, string& bases, vector<uint64_t>& offsets) {
  bases.clear();
  offsets.assign(bt.n + 1, 0);
  for (uint32_t j = 0; j < bt.n; ++j) { bases += bt.seqs[j]; offsets[j + 1] = bases.size(); }
}

// ProcessSingledEndReads, mapping.cpp:421-526
static void process_se(const Options& o, const string& reads_file, const string& out_file) {
  walt_index* idx = nullptr;
  check(walt_index_open(o.index_file.c_str(), o.device, o.ag ? WALT_STRANDS_GA : WALT_STRANDS_CT, -1, &idx));
  GenomeInfo g = genome_of(idx);
  FILE* fin = fopen(reads_file.c_str(), "r");
  if (!fin) die("cannot open input file " + reads_file);
  FILE* fout = fopen(out_file.c_str(), "a");
  if (!fout) die("cannot open input file " + out_file);
  SeStats st;
  st.open(o.ambiguous, o.unmapped, out_file, o.sam);
  if (o.verbose) std::cerr << "input_file: " << reads_file << std::endl << "output_file: " << out_file << std::endl;
  if (o.sam) sam_head(g, fout);
  Batch bt;
  string bases;
  vector<uint64_t> offsets;
  vector<walt_best_match> res;
  for (;;) {
    load_batch(fin, o.batch_size, o.adaptor, bt);
    if (bt.n == 0) break;
    pack_batch(bt, bases, offsets);
    res.resize(bt.n);
    walt_batch_stats bs;
    check(walt_map_se_batch(idx, bases.data(), offsets.data(), bt.n, o.ag, o.max_mismatches, o.b, res.data(), &bs));
    st.too_short += (uint32_t)bs.too_short;
    for (uint32_t j = 0; j < bt.n; ++j) {
      st.update(res[j].times);
      if (!o.sam) out_single_results(res[j], bt.names[j], bt.seqs[j], bt.scores[j], g, o.ag, st, fout);
      else out_single_sam(res[j], bt.names[j], bt.seqs[j], bt.scores[j], g, st, fout);
    }
    if (bt.n < o.batch_size) break;
  }
  fclose(fin);
  fclose(fout);
  st.close();
  std::ofstream mapstats(out_file + ".mapstats", std::ios::app);
  mapstats << st.tostring() << std::endl;
  walt_index_close(idx);
}

// ---------------------------------------------------------------- paired-end writers
static void forward_pos(uint32_t gp, char strand, uint32_t chr, uint32_t read_len, const GenomeInfo& g, uint32_t& s,
                        uint32_t& e) {  // paired.cpp:98-104
  s = gp - g.start[chr];
  s = strand == '+' ? s : g.length[chr] - s - read_len;
  e = s + read_len;
}
// OutputBestPairedResults, paired.cpp:210-294
static int out_best_pair(const walt_candidate& r1, const walt_candidate& r2, int frag_range, const GenomeInfo& g,
                         const string& name, const string& seq1, const string& scr1, const string& seq2,
                         const string& scr2, bool sam, FILE* fout) {
  const uint32_t len1 = (uint32_t)seq1.size(), len2 = (uint32_t)seq2.size();
  string seq2r = revcomp(seq2), scr2r = rev_str(scr2);
  uint32_t c1 = chrom_id(g, r1.genome_pos), c2 = chrom_id(g, r2.genome_pos);
  uint32_t s1, s2, e1, e2;
  forward_pos(r1.genome_pos, r1.strand, c1, len1, g, s1, e1);
  forward_pos(r2.genome_pos, r2.strand, c2, len2, g, s2, e2);
  uint32_t ov_s = std::max(s1, s2), ov_e = std::min(e1, e2);
  const bool plus = r1.strand == '+';
  uint32_t one_l = plus ? s1 : std::max(ov_e, s1);
  uint32_t one_r = plus ? std::min(ov_s, e1) : e1;
  uint32_t two_l = plus ? std::max(ov_e, s2) : s2;
  uint32_t two_r = plus ? e2 : std::min(ov_s, e2);
  int len = plus ? (int)(two_r - one_l) : (int)(one_r - two_l);
  if (sam) return len;
  string seq(len, 'N'), scr(len, 'B');
  if (len > 0 && len <= frag_range) {
    uint32_t lim_one = one_r - one_l;
    std::copy(seq1.begin(), seq1.begin() + lim_one, seq.begin());
    std::copy(scr1.begin(), scr1.begin() + lim_one, scr.begin());
    uint32_t lim_two = two_r - two_l;
    std::copy(seq2r.end() - lim_two, seq2r.end(), seq.end() - lim_two);
    std::copy(scr2r.end() - lim_two, scr2r.end(), scr.end() - lim_two);
    if (ov_s < ov_e) {
      int info_one = (int)len1 - ((int)std::count(seq1.begin(), seq1.end(), 'N') + (int)r1.mismatch);
      int info_two = (int)len2 - ((int)std::count(seq2r.begin(), seq2r.end(), 'N') + (int)r2.mismatch);
      if (info_one >= info_two) {
        uint32_t a = plus ? (ov_s - s1) : (e1 - ov_e), b = plus ? (ov_e - s1) : (e1 - ov_s);
        std::copy(seq1.begin() + a, seq1.begin() + b, seq.begin() + lim_one);
        std::copy(scr1.begin() + a, scr1.begin() + b, scr.begin() + lim_one);
      } else {
        uint32_t a = plus ? (ov_s - s2) : (e2 - ov_e), b = plus ? (ov_e - s2) : (e2 - ov_s);
        std::copy(seq2r.begin() + a, seq2r.begin() + b, seq.begin() + lim_one);
        std::copy(scr2r.begin() + a, scr2r.begin() + b, scr.begin() + lim_one);
      }
    }
  }
  uint32_t start_pos = plus ? s1 : s2;
  fprintf(fout, "%s\t%u\t%u\tFRAG:%s\t%u\t%c\t%s\t%s\n", g.name[c1].c_str(), start_pos, start_pos + len, name.c_str(),
          r1.mismatch + r2.mismatch, r1.strand, seq.c_str(), scr.c_str());
  return len;
}
static int sam_flag(bool paired_mapped, bool unmapped, bool next_unmapped, bool rev, bool next_rev, bool first,
                    bool secondary) {  // GetSAMFLAG, paired.cpp:80-95
  return 0x1 + (paired_mapped ? 0x2 : 0) + (unmapped ? 0x4 : 0) + (next_unmapped ? 0x8 : 0) + (rev ? 0x10 : 0) +
         (next_rev ? 0x20 : 0) + (first ? 0x40 : 0x80) + (secondary ? 0x100 : 0);
}
// OutputPairedSAM, paired.cpp:333-435
static void out_paired_sam(const walt_best_match& b1, const walt_best_match& b2, const GenomeInfo& g,
                           const string& name, const string& seq1, const string& scr1, const string& seq2,
                           const string& scr2, int len, int flag_1, int flag_2, bool out_amb, bool out_unm,
                           FILE* fout) {
  uint32_t c1 = chrom_id(g, b1.genome_pos), c2 = chrom_id(g, b2.genome_pos);
  uint32_t s1, s2, e1, e2;
  forward_pos(b1.genome_pos, b1.strand, c1, (uint32_t)seq1.size(), g, s1, e1);
  forward_pos(b2.genome_pos, b2.strand, c2, (uint32_t)seq2.size(), g, s2, e2);
  uint32_t mm1 = b1.mismatch, mm2 = b2.mismatch;
  if (b1.times == 0) { s1 = 0; mm1 = 0; } else s1 += 1;
  if (b2.times == 0) { s2 = 0; mm2 = 0; } else s2 += 1;
  int len1 = b1.strand == '+' ? len : -len;
  int len2 = b2.strand == '+' ? len : -len;
  string rn1 = "=", rn2 = "=";
  if (!(flag_1 & 0x2)) {
    rn1 = b1.times == 0 ? "*" : g.name[c1];
    rn2 = b2.times == 0 ? "*" : g.name[c2];
  }
  string q1 = seq1, q2 = seq2, k1 = scr1, k2 = scr2;
  if (b1.strand == '-') { q1 = revcomp(q1); k1 = rev_str(k1); }
  if (b2.strand == '-') { q2 = revcomp(q2); k2 = rev_str(k2); }
  const uint32_t rl1 = (uint32_t)seq1.size(), rl2 = (uint32_t)seq2.size();
  if (b1.times == 0 && out_unm)
    fprintf(fout, "%s\t%d\t*\t%u\t255\t*\t%s\t%u\t%d\t%s\t%s\tNM:i:%u\n", name.c_str(), flag_1, s1, rn2.c_str(), s2, len1,
            q1.c_str(), k1.c_str(), mm1);
  else if (b1.times == 1 || (b1.times >= 2 && out_amb))
    fprintf(fout, "%s\t%d\t%s\t%u\t255\t%uM\t%s\t%u\t%d\t%s\t%s\tNM:i:%u\n", name.c_str(), flag_1, g.name[c1].c_str(), s1,
            rl1, rn2.c_str(), s2, len1, q1.c_str(), k1.c_str(), mm1);
  if (b2.times == 0 && out_unm)
    fprintf(fout, "%s\t%d\t*\t%u\t255\t*\t%s\t%u\t%d\t%s\t%s\tNM:i:%u\n", name.c_str(), flag_2, s2, rn1.c_str(), s1, len2,
            q2.c_str(), k2.c_str(), mm2);
  else if (b2.times == 1 || (b2.times >= 2 && out_amb))
    fprintf(fout, "%s\t%d\t%s\t%u\t255\t%uM\t%s\t%u\t%d\t%s\t%s\tNM:i:%u\n", name.c_str(), flag_2, g.name[c2].c_str(), s2,
            rl2, rn1.c_str(), s1, len2, q2.c_str(), k2.c_str(), mm2);
}

// ProcessPairedEndReads, paired.cpp:572-713
static void process_pe(const Options& o, const string& f1, const string& f2, const string& out_file) {
  walt_index* idx = nullptr;
  check(walt_index_open(o.index_file.c_str(), o.device, WALT_STRANDS_ALL, -1, &idx));
  GenomeInfo g = genome_of(idx);
  FILE* fin[2] = {fopen(f1.c_str(), "r"), fopen(f2.c_str(), "r")};
  if (!fin[0]) die("cannot open input file " + f1);
  if (!fin[1]) die("cannot open input file " + f2);
  string adaptors[2];
  extract_adaptors(o.adaptor, adaptors[0], adaptors[1]);
  FILE* fout = fopen(out_file.c_str(), "a");
  if (!fout) die("cannot open input file " + out_file);
  SeStats st1, st2;  // StatPairedReads, paired.hpp:78-106
  st1.open(o.ambiguous, o.unmapped, out_file + "_1", o.sam);
  st2.open(o.ambiguous, o.unmapped, out_file + "_2", o.sam);
  uint32_t total_pairs = 0, unique_pairs = 0, ambiguous_pairs = 0, unmapped_pairs = 0;
  vector<uint32_t> frag_count(o.frag_range + 1, 0);
  fprintf(stderr, "[MAPPING PAIRED-END READS FROM THE FOLLOWING TWO FILES]\n   %s (AND)\n   %s\n", f1.c_str(), f2.c_str());
  fprintf(stderr, "[OUTPUT MAPPING RESULTS TO %s]\n", out_file.c_str());
  if (o.sam) sam_head(g, fout);
  Batch bt[2];
  string bases[2];
  vector<uint64_t> offs[2];
  vector<walt_pair_result> pr;
  vector<walt_candidate> rk[2];
  vector<uint32_t> rn[2];
  for (;;) {
    load_batch(fin[0], o.batch_size, adaptors[0], bt[0]);
    if (bt[0].n) load_batch(fin[1], o.batch_size, adaptors[1], bt[1]); else bt[1].n = 0;
    if (bt[0].n && bt[0].n != bt[1].n) {
      fprintf(stderr, "The number of reads in paired-end files should be the same.\n");
      exit(EXIT_FAILURE);
    }
    if (bt[0].n == 0) break;
    const uint32_t n = bt[0].n;
    total_pairs += n;
    for (int m = 0; m < 2; ++m) { pack_batch(bt[m], bases[m], offs[m]); rk[m].resize((size_t)n * o.top_k); rn[m].resize(n); }
    pr.resize(n);
    walt_batch_stats bs[2];
    check(walt_map_pe_batch(idx, bases[0].data(), offs[0].data(), bases[1].data(), offs[1].data(), n, o.max_mismatches,
                            o.b, o.top_k, o.frag_range, pr.data(), rk[0].data(), rn[0].data(), rk[1].data(),
                            rn[1].data(), bs));
    st1.too_short += (uint32_t)bs[0].too_short;
    st2.too_short += (uint32_t)bs[1].too_short;
    for (uint32_t j = 0; j < n; ++j) {  // MergePairedEndResults tail, paired.cpp:515-569
      const walt_pair_result& p = pr[j];
      const string &name = bt[0].names[j], &q1 = bt[0].seqs[j], &k1 = bt[0].scores[j], &q2 = bt[1].seqs[j],
                   &k2 = bt[1].scores[j];
      walt_best_match bm1 = {0, 0, '+', {0, 0, 0}, o.max_mismatches}, bm2 = bm1;
      bool is_paired = false;
      int len = 0;
      if (p.best_times == 1) {
        unique_pairs++;
        const walt_candidate& r1 = rk[0][(size_t)j * o.top_k + p.best_i];
        const walt_candidate& r2 = rk[1][(size_t)j * o.top_k + p.best_j];
        len = out_best_pair(r1, r2, o.frag_range, g, name, q1, k1, q2, k2, o.sam, fout);
        frag_count[len]++;
        if (o.sam) { is_paired = true; bm1 = p.m1; bm2 = p.m2; }
      } else {
        if (p.best_times >= 2) ambiguous_pairs++; else unmapped_pairs++;
        bm1 = p.m1; bm2 = p.m2;
        st1.update(bm1.times);
        st2.update(bm2.times);
        if (!o.sam) {
          out_single_results(bm1, name, q1, k1, g, false, st1, fout);
          out_single_results(bm2, name, q2, k2, g, true, st2, fout);
        }
      }
      if (o.sam) {
        int fl1 = sam_flag(is_paired, bm1.times == 0, bm2.times == 0, bm1.strand == '-', bm2.strand == '-', true, bm1.times >= 2);
        int fl2 = sam_flag(is_paired, bm2.times == 0, bm1.times == 0, bm2.strand == '-', bm1.strand == '-', false, bm2.times >= 2);
        out_paired_sam(bm1, bm2, g, name, q1, k1, q2, k2, len, fl1, fl2, o.ambiguous, o.unmapped, fout);
      }
    }
    if (n < o.batch_size) break;
  }
  fclose(fin[0]); fclose(fin[1]); fclose(fout);
  st1.close(); st2.close();
  std::ofstream mapstats(out_file + ".mapstats", std::ios::app);  // StatPairedReads::tostring, paired.cpp:52-77
  std::ostringstream oss;
  oss << "pairs:" << std::endl
      << "    total_read_pairs: " << total_pairs << std::endl
      << "    mapped:" << std::endl
      << "        unique: " << unique_pairs << std::endl
      << "        percent_unique: " << (100.0 * unique_pairs) / total_pairs << std::endl
      << "        ambiguous: " << ambiguous_pairs << std::endl
      << "    unmapped: " << unmapped_pairs << std::endl
      << "mate1:" << std::endl << st1.tostring(1) << std::endl
      << "mate2:" << std::endl << st2.tostring(1) << std::endl;
  oss << "frag_len_distribution:" << std::endl;
  double total = 0.0;
  for (size_t i = 0; i < frag_count.size(); ++i) {
    oss << "    " << i << ": " << frag_count[i] << std::endl;
    total += (i * frag_count[i]);
  }
  oss << "frag_len_mean: " << total / std::accumulate(frag_count.begin(), frag_count.end(), 0.0);
  mapstats << oss.str() << std::endl;
  walt_index_close(idx);
}

int main(int argc, const char** argv) {
  try {
    if (argc == 1) {
      fprintf(stderr, "Usage: walt -i <index> -r <reads> | -1 <reads1> -2 <reads2> -o <out> [-m -N -a -u -C -A -b -k -L -sam -v -t -g]\n");
      return EXIT_SUCCESS;
    }
    Options o = parse(argc, argv);
    static const char* sfx[5] = {"", "_CT00", "_CT01", "_GA10", "_GA11"};  // validate_index_file, walt.cpp:67-85
    for (const char* s : sfx) if (!exists(o.index_file + s)) die("index file missing: " + o.index_file + s);
    vector<string> se = split_csv(o.se_csv), p1 = split_csv(o.pe1_csv), p2 = split_csv(o.pe2_csv);
    for (auto& f : se) if (!valid_suffix(f)) die("read file invalid suffix: " + f);
    if (p1.size() != p2.size()) die("unequal number of end1 and end2 files");
    for (auto& f : p1) if (!valid_suffix(f)) die("read file invalid suffix: " + f);
    for (auto& f : p2) if (!valid_suffix(f)) die("read file invalid suffix: " + f);
    vector<string> outs = split_csv(o.out_csv);
    if (outs.size() != 1 && outs.size() != se.size() + p1.size()) die("wrong number of output files: " + o.out_csv);
    if (outs.size() == 1) outs.assign(se.size() + p1.size(), outs[0]);
    for (auto& f : outs) { std::ofstream out(f); std::ofstream stat(f + ".mapstats"); }  // walt.cpp:230-233
    if (o.batch_size > 100000000) die("batch size may not exceed100000000");
    if (o.top_k < 2 || o.top_k > 300) die("paired-end candidates must be in [2, 300]");
    size_t k = 0;
    for (auto& f : se) process_se(o, f, outs[k++]);
    for (size_t i = 0; i < p1.size(); ++i) process_pe(o, p1[i], p2[i], outs[k++]);
  } catch (const std::exception& e) {
    std::cerr << e.what() << std::endl;
    return EXIT_FAILURE;
  }
  return EXIT_SUCCESS;
}
