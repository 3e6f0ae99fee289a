// walt_main.cpp -- host driver with WALT's command-line surface on top of the
// C ABI (include/walt_amd.h).  It restates, in its own code, the parts of the
// reference that sit AROUND the hot path so that a user can swap binaries:
//   option table ............ walt.cpp:130-166 (single-dash long names)
//   FASTQ batch loader ...... mapping.cpp:65-121 (srand(0) per batch, N -> rand()%4)   [hostio.h]
//   adaptor clipping -C ..... util.hpp:189-233                                          [hostio.h]
//   SAM / MR / mapstats ..... mapping.cpp:47-63,329-419; paired.cpp:52-77,210-294,333-435,515-569
// The mapping itself (mapping.cpp:486-500, paired.cpp:642-699) is one library call
// per batch.  Ingest and output formatting run on -t host threads (default: all),
// each thread owning a contiguous range of the batch, so the files come out in
// read order and byte-identical to the reference binary's
// (tests/test_gpu_cli.py compares against the golden files).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#include <algorithm>
#include <exception>
#include <fstream>
#include <iostream>
#include <numeric>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>

#include "../../../include/walt_amd.h"
#include "hostio.h"

using std::string;
using std::vector;
using hostio::Batch;
using hostio::OutFile;
using hostio::Sink;
using hostio::View;

static const uint32_t MAX_LINE_LENGTH = 1000;  // util.hpp:43
#define MINIMALREADLEN walt_min_read_len()       // seedpattern.hpp:359 / 230 / 33 (38 / 32 / 23 by seed pattern)

static void die(const string& msg) { throw std::runtime_error(msg); }
static void check(int rc) { if (rc != WALT_OK) die(walt_last_error()); }

// ---------------------------------------------------------------- options
struct Options {
  string index_file, se_csv, pe1_csv, pe2_csv, out_csv, adaptor;
  bool sam = false, ambiguous = false, unmapped = false, ag = false, verbose = false, pbat = false;
  uint32_t max_mismatches = 6, batch_size = 10000000, b = 5000, top_k = 50;
  int frag_range = 1000, threads = 0;
  std::vector<int> devices;  // -g 0,1,...: every listed GPU holds an index replica and maps a contiguous share of each batch
  std::vector<std::pair<string, long long>> tune;  // -X name=value: walt_index_set_option on every replica (schedule only, never results)
};

static bool is_opt(const string& a, const char* s, const char* l) { return a == string("-") + s || a == string("-") + l || a == string("--") + l; }

static vector<string> split_csv(const string& s) {
  vector<string> out;
  std::istringstream is(s);
  string tok;
  while (std::getline(is, tok, ',')) if (!tok.empty()) out.push_back(tok);
  return out;
}
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
    else if (is_opt(a, "P", "pbat")) o.pbat = true;  // README.md:64,100-104; no code in the reference snapshot
    else if (is_opt(a, "b", "bucket")) o.b = (uint32_t)strtoul(val().c_str(), 0, 10);
    else if (is_opt(a, "k", "topk")) o.top_k = (uint32_t)strtoul(val().c_str(), 0, 10);
    else if (is_opt(a, "L", "fraglen")) o.frag_range = atoi(val().c_str());
    else if (a == "-sam" || a == "--sam") o.sam = true;
    else if (is_opt(a, "v", "verbose")) o.verbose = true;
    else if (is_opt(a, "t", "thread")) o.threads = atoi(val().c_str());
    else if (is_opt(a, "g", "gpu")) {  // extension: device ordinal(s), comma separated
      o.devices.clear();
      for (const string& t : split_csv(val())) o.devices.push_back(atoi(t.c_str()));
    }
    else if (a == "-X") {  // extension: an option of the mapping library (include/walt_amd.h, walt_index_set_option)
      const string kv = val();
      const size_t eq = kv.find('=');
      if (eq == string::npos || eq == 0) die("-X wants name=value");
      o.tune.emplace_back(kv.substr(0, eq), atoll(kv.c_str() + eq + 1));
    }
    else die("unknown option " + a);
  }
  if (o.index_file.empty() || o.out_csv.empty()) die("options -i and -o are required");
  if (o.devices.empty()) o.devices.push_back(0);
  if (o.pbat && !o.se_csv.empty()) o.ag = true;  // single-end PBAT reads are A-rich: same as -A
  return o;
}

static bool valid_suffix(const string& fn) {  // walt.cpp:58-64 (checks .fastq / .fq)
  auto ends = [&](const string& sfx) { return fn.size() >= sfx.size() && fn.compare(fn.size() - sfx.size(), sfx.size(), sfx) == 0; };
  return ends(".fastq") || ends(".fq");
}
static bool exists(const string& p) { struct stat st; return stat(p.c_str(), &st) == 0; }

// -C "T_adaptor[:A_adaptor]" (walt.cpp:150-152): one adaptor serves both mates, two are split at the colon
struct AdaptorPair {
  string mate1, mate2;
};
static AdaptorPair split_adaptor_option(const string& opt) {
  const size_t colons = (size_t)std::count(opt.begin(), opt.end(), ':');
  if (colons > 1) die("ERROR: adaptor format \"T_adaptor[:A_adaptor]\"");
  AdaptorPair ap;
  const size_t cut = colons ? opt.find(':') : opt.size();
  ap.mate1 = opt.substr(0, cut);
  ap.mate2 = colons ? opt.substr(cut + 1) : ap.mate1;
  return ap;
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
static string sam_head(const GenomeInfo& g) {  // SAMHead, reference.cpp:430-440
  std::ostringstream os;
  os << "@HD\tVN:1.0\n";
  for (size_t i = 0; i < g.name.size(); ++i) os << "@SQ\tSN:" << g.name[i] << "\tLN:" << g.length[i] << "\n";
  os << "@PG\tID:WALT\tVN:1.0\tCL:walt\n";
  return os.str();
}

// sinks of one host thread: main output, then the _ambiguous / _unmapped side files of mate 1 and mate 2
enum { kMain = 0, kAmb1 = 1, kUnm1 = 2, kAmb2 = 3, kUnm2 = 4, kSinks = 5 };

struct SeCounts {  // StatSingleReads, mapping.hpp:55-108
  uint32_t total = 0, unique = 0, ambiguous = 0, unmapped = 0, too_short = 0;
  void update(uint32_t times) {  // StatInfoUpdate, mapping.cpp:318-327
    ++total;
    if (times == 0) unmapped++; else if (times == 1) unique++; else ambiguous++;
  }
  void add(const SeCounts& o) {
    total += o.total; unique += o.unique; ambiguous += o.ambiguous; unmapped += o.unmapped; too_short += o.too_short;
  }
  // the block StatSingleReads prints into <out>.mapstats (mapping.cpp:47-63): `depth` levels of four spaces in front
  // of every line, no newline behind the last; percent_unique in the stream's default %g form (0 reads: "-nan")
  void put_block(Sink& f, int depth) const {
    auto line = [&](int extra, const char* key) {
      for (int i = 0; i < 4 * (depth + extra); ++i) f.ch(' ');
      f.lit(key);
    };
    char pct[40];
    snprintf(pct, sizeof pct, "%g", (100.0 * unique) / total);
    line(0, "total_reads: "); f.u32(total); f.ch('\n');
    line(0, "mapped:\n");
    line(1, "unique: "); f.u32(unique); f.ch('\n');
    line(1, "percent_unique: "); f.lit(pct); f.ch('\n');
    line(1, "ambiguous: "); f.u32(ambiguous); f.ch('\n');
    line(0, "unmapped: "); f.u32(unmapped); f.ch('\n');
    line(0, "min_read_length: "); f.u32(MINIMALREADLEN); f.ch('\n');
    line(0, "too_short: "); f.u32(too_short);
  }
};
struct SideFiles {  // the _ambiguous / _unmapped files of StatSingleReads (mapping.hpp:75-87)
  OutFile amb, unm;
  bool out_amb = false, out_unm = false;
  void open(bool a, bool u, const string& out, bool sam) {
    out_amb = a; out_unm = u;
    if (a && !sam && !amb.open_trunc(out + "_ambiguous")) die("cannot open input file " + out + "_ambiguous");
    if (u && !sam && !unm.open_trunc(out + "_unmapped")) die("cannot open input file " + out + "_unmapped");
  }
  void close() { amb.close(); unm.close(); }
};

// ---------------------------------------------------------------- single-end writers, mapping.cpp:329-419
static void out_mr_line(const walt_best_match& bm, View name, View seq, View score, bool flip, const GenomeInfo& g,
                        bool ag, Sink& f) {
  uint32_t chr = chrom_id(g, bm.genome_pos);
  uint32_t start = bm.genome_pos - g.start[chr];
  if (bm.strand == '-') start = g.length[chr] - start - seq.len;
  uint32_t end = start + seq.len;
  char strand = bm.strand;
  if (ag) strand = bm.strand == '+' ? '-' : '+';
  f.put(g.name[chr]); f.ch('\t'); f.u32(start); f.ch('\t'); f.u32(end); f.ch('\t'); f.put(name); f.ch('\t');
  f.u32(bm.mismatch); f.ch('\t'); f.ch(strand); f.ch('\t');
  if (flip) { f.revcomp(seq); f.ch('\t'); f.rev(score); } else { f.put(seq); f.ch('\t'); f.put(score); }
  f.ch('\n');
}
// OutputSingleResults, mapping.cpp:329-380: A-rich reads are printed reverse-complemented
static void out_single_results(const walt_best_match& bm, View name, View seq, View score, const GenomeInfo& g, bool ag,
                               bool out_amb, bool out_unm, Sink& fout, Sink& famb, Sink& funm) {
  if (bm.times == 0 && out_unm) {
    funm.put(name); funm.ch('\t');
    if (ag) { funm.revcomp(seq); funm.ch('\t'); funm.rev(score); } else { funm.put(seq); funm.ch('\t'); funm.put(score); }
    funm.ch('\n');
  } else if (bm.times == 1) {
    out_mr_line(bm, name, seq, score, ag, g, ag, fout);
  } else if (bm.times >= 2 && out_amb) {
    out_mr_line(bm, name, seq, score, ag, g, ag, famb);
  }
}
static void put_seq_qual(Sink& f, View seq, View score, bool flip) {
  if (flip) { f.revcomp(seq); f.ch('\t'); f.rev(score); } else { f.put(seq); f.ch('\t'); f.put(score); }
}
// OutputSingleSAM, mapping.cpp:382-419
static void out_single_sam(const walt_best_match& bm, View name, View seq, View score, const GenomeInfo& g,
                           bool out_amb, bool out_unm, Sink& f) {
  uint32_t chr = chrom_id(g, bm.genome_pos);
  uint32_t start = bm.genome_pos - g.start[chr];
  if (bm.strand == '-') start = g.length[chr] - start - seq.len;
  const bool flip = bm.strand == '-';
  int flag = (bm.times == 0 ? 0x4 : 0) + (bm.strand == '-' ? 0x10 : 0) + (bm.times >= 2 ? 0x100 : 0);
  if (bm.times == 0 && out_unm) {
    f.put(name); f.ch('\t'); f.i32(flag); f.lit("\t*\t0\t255\t*\t*\t0\t0\t");
    put_seq_qual(f, seq, score, flip);
    f.lit("\tNM:i:0\n");
  } else if (bm.times == 1 || (bm.times >= 2 && out_amb)) {
    f.put(name); f.ch('\t'); f.i32(flag); f.ch('\t'); f.put(g.name[chr]); f.ch('\t'); f.u32(start + 1);
    f.lit("\t255\t"); f.u32(seq.len); f.lit("M\t*\t0\t0\t");
    put_seq_qual(f, seq, score, flip);
    f.lit("\tNM:i:"); f.u32(bm.mismatch); f.ch('\n');
  }
}

static double g_t_main = 0;  // start of main (timeline under -v)
static int host_threads(const Options& o) { return o.threads > 0 ? o.threads : hostio::effective_cpus(); }
static double now_s() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

// ---------------------------------------------------------------- the GPUs of a run (-g)
// Every device holds a full index replica (opened side by side); a batch is cut into contiguous shares AFTER the
// loader has replaced its non-ACGT characters (the draws depend on the -N batch, not on the share: mapping.cpp:73),
// each share is mapped by its device from the same host arrays, and the records land in read order.  The
// statistics of the shares are added on the host -- this is one process; across processes the same sum is
// walt_stats_allreduce (include/walt_amd.h).
struct DeviceSet {
  vector<int> ids;
  vector<walt_index*> idx;
  void open(const Options& o, unsigned strands) {
    ids = o.devices;
    idx.assign(ids.size(), nullptr);
    vector<string> err(ids.size());
    vector<std::thread> th;
    for (size_t d = 0; d < ids.size(); ++d)
      th.emplace_back([&, d]() {
        if (walt_index_open(o.index_file.c_str(), ids[d], strands, -1, &idx[d]) != WALT_OK) err[d] = walt_last_error();
      });
    for (auto& t : th) t.join();
    for (const string& e : err) if (!e.empty()) { close(); die(e); }
    for (walt_index* i : idx)
      for (const auto& kv : o.tune)
        if (walt_index_set_option(i, kv.first.c_str(), kv.second) != WALT_OK) { const string e = walt_last_error(); close(); die(e); }
  }
  void close() {
    for (walt_index*& i : idx) { if (i) walt_index_close(i); i = nullptr; }
  }
  size_t size() const { return ids.size(); }
  void share(uint32_t n, size_t d, uint32_t& lo, uint32_t& hi) const {
    lo = (uint32_t)((uint64_t)n * d / ids.size());
    hi = (uint32_t)((uint64_t)n * (d + 1) / ids.size());
  }
  // fn(d, lo, hi) -> status, run for every device at once; the first failure is reported
  template <class F>
  void for_each_share(uint32_t n, F fn) const {
    if (ids.size() == 1) { uint32_t lo, hi; share(n, 0, lo, hi); if (fn(0, lo, hi) != WALT_OK) die(walt_last_error()); return; }
    vector<string> err(ids.size());
    vector<std::thread> th;
    for (size_t d = 0; d < ids.size(); ++d)
      th.emplace_back([&, d]() {
        uint32_t lo, hi;
        share(n, d, lo, hi);
        if (hi > lo && fn(d, lo, hi) != WALT_OK) err[d] = walt_last_error();
      });
    for (auto& t : th) t.join();
    for (const string& e : err) if (!e.empty()) die(e);
  }
};

// The next batch is read while the current one is mapped and written: the loader runs on its own thread with a
// share of the host threads (its N draws use rand(), which nothing else in the process calls meanwhile).
struct Prefetch {
  std::thread th;
  std::exception_ptr err;
  ~Prefetch() { if (th.joinable()) th.join(); }  // an error elsewhere unwinds past a running helper
  template <class F>
  void start(F fn) {
    err = nullptr;
    th = std::thread([this, fn]() { try { fn(); } catch (...) { err = std::current_exception(); } });
  }
  void wait() {
    if (th.joinable()) th.join();
    if (err) std::rethrow_exception(err);
  }
};

// -v: what this box's host side can move at all, measured where the run's output went -- T threads copying between private
// buffers (what formatting a line into a buffer costs at least) and T threads storing their buffers with pwrite at
// prefix-summed offsets into ONE scratch file beside the output (what OutFile does; one inode lock).  The run's format and
// write stages are to be read against these two figures (bench.py's end-to-end leg carries them).
static void host_ceiling(const string& out_file, int T) {
  const size_t per = 64u << 20;  // 64 MiB per thread and round
  std::vector<std::vector<char>> a((size_t)T), b((size_t)T);
#pragma omp parallel for num_threads(T) schedule(static)
  for (int t = 0; t < T; ++t) { a[(size_t)t].assign(per, (char)('A' + t)); b[(size_t)t].assign(per, 'x'); }
  double t0 = now_s();
  const int rounds = 4;
  for (int r = 0; r < rounds; ++r) {
#pragma omp parallel for num_threads(T) schedule(static)
    for (int t = 0; t < T; ++t) memcpy(b[(size_t)t].data(), a[(size_t)t].data(), per);
  }
  const double copy_gbs = (double)rounds * T * per / (now_s() - t0) / 1e9;
  const string tmp = out_file + ".ceiling.tmp";
  double write_gbs = 0.0;
  const int fd = ::open(tmp.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0600);
  if (fd >= 0) {
    t0 = now_s();
    for (int r = 0; r < rounds; ++r) {
#pragma omp parallel for num_threads(T) schedule(static)
      for (int t = 0; t < T; ++t) {
        size_t done = 0;
        while (done < per) {
          const ssize_t w = ::pwrite(fd, b[(size_t)t].data() + done, per - done, (off_t)(((size_t)r * T + (size_t)t) * per + done));
          if (w <= 0) break;
          done += (size_t)w;
        }
      }
    }
    write_gbs = (double)rounds * T * per / (now_s() - t0) / 1e9;
    ::close(fd);
    ::unlink(tmp.c_str());
  }
  fprintf(stderr, "[walt_amd host ceiling: memcpy %.1f GB/s on %d threads, pwrite into one file %.1f GB/s]\n", copy_gbs, T, write_gbs);
}

// The last read file is done, every output file closed, the indexes released: leave without unwinding.  What the
// destructors and the runtime's exit handlers would do -- hand back gigabytes of line buffers, unmap the read file,
// unload the HIP runtime -- took 0.6 s of a 2 s run, and the operating system does it anyway.
// Not when something else in the process counts on the exit handlers -- a preloaded tool that writes its trace there
// (LD_PRELOAD, rocprofv3's ROCP_TOOL_LIBRARIES), or WALT_AMD_FAST_EXIT=0 -- then main returns as usual.  A late failure
// of the standard streams still changes the exit status.
static void leave_now(bool verbose) {
  if (verbose) fprintf(stderr, "[walt_amd: %.2f s in main]\n", now_s() - g_t_main);
  const char* fe = getenv("WALT_AMD_FAST_EXIT");
  const char* pre = getenv("LD_PRELOAD");
  const char* tool = getenv("ROCP_TOOL_LIBRARIES");
  if ((fe && atoi(fe) == 0) || (pre && *pre) || (tool && *tool)) return;
  std::cout.flush();
  std::cerr.flush();
  const bool bad = fflush(stdout) != 0 || fflush(stderr) != 0 || ferror(stdout) || ferror(stderr) || !std::cout || !std::cerr;
  _exit(bad ? EXIT_FAILURE : EXIT_SUCCESS);
}

// ProcessSingledEndReads, mapping.cpp:421-526
static void process_se(const Options& o, const string& reads_file, const string& out_file, bool last_file) {
  const int T = host_threads(o);
  const int T_bg = std::max(1, T / 4);  // the loader's share while a batch is being formatted
  double t0 = now_s();
  // the first batch is read WHILE the index is opened (seconds of file reads and kernels that leave most host cores
  // idle); the loader's N draws are the only rand() calls of the process, as before
  hostio::FastqReader rd;
  Batch bt[2];
  Prefetch first;
  double t_open_reads = 0;
  first.start([&]() {
    const double t_o0 = now_s();
    rd.open(reads_file, std::max(1, T / 2));
    t_open_reads = now_s() - t_o0;
    rd.load(o.batch_size, o.adaptor, bt[0]);
  });
  DeviceSet dev;
  dev.open(o, o.ag ? WALT_STRANDS_GA : WALT_STRANDS_CT);
  double t_index = now_s() - t0, t_load = 0, t_map = 0, t_out = 0, t_write = 0;
  GenomeInfo g = genome_of(dev.idx[0]);
  OutFile fout;
  if (!fout.open_append(out_file)) die("cannot open input file " + out_file);
  SideFiles side;
  side.open(o.ambiguous, o.unmapped, out_file, o.sam);
  SeCounts st;
  if (o.verbose) std::cerr << "input_file: " << reads_file << std::endl << "output_file: " << out_file << std::endl;
  if (o.sam) { string h = sam_head(g); fout.write(h.data(), h.size()); }
  walt_best_match* res = nullptr;
  size_t res_cap = 0;
  vector<Sink> sinks((size_t)T * kSinks);
  vector<SeCounts> acc(T);
  // Only the ingest runs ahead.  Storing a batch's lines from a helper thread while the next batch is formatted was
  // tried and lost: both are memory copies on the same cores (40 M reads: format 0.67 -> 0.96 s, and 1.3 s more
  // waiting for the writer).
  Prefetch pre;
  // The last batch: the idle buffer's page-locked memory is handed back while the batch is formatted, the batch's own
  // while its lines are written (unlocking 2.6 GB at the end of the run was 0.5 s of a 2 s run).
  Prefetch unlock_idle, unlock_last;
  t0 = now_s();
  first.wait();  // what is still left of the first batch's ingest
  rd.threads = T;
  t_load += now_s() - t0;
  for (int cur = 0;; cur ^= 1) {
    Batch& b = bt[cur];
    if (b.n == 0) break;
    const uint32_t n = b.n;
    const bool more = n == o.batch_size;  // mapping.cpp:515-516: a short batch is the last one
    if (more) {
      rd.threads = T_bg;
      pre.start([&, cur]() { rd.load(o.batch_size, o.adaptor, bt[cur ^ 1]); });
    }
    if (res_cap < n) {
      walt_host_free(res);
      res = nullptr;
      res_cap = n + n / 8;
      check(walt_host_alloc(res_cap * sizeof(walt_best_match), (void**)&res));
    }
    t0 = now_s();
    vector<uint64_t> short_of(dev.size(), 0);
    dev.for_each_share(n, [&](size_t d, uint32_t lo, uint32_t hi) {
      walt_batch_stats bs;
      const int rc = walt_map_se_batch(dev.idx[d], b.bases, b.offsets + lo, hi - lo, o.ag, o.max_mismatches, o.b, res + lo, &bs);
      short_of[d] = bs.too_short;
      return rc;
    });
    for (uint64_t v : short_of) st.too_short += (uint32_t)v;
    t_map += now_s() - t0;
    if (!more) unlock_idle.start([&, cur]() { bt[cur ^ 1].release(); });
    t0 = now_s();
#pragma omp parallel for schedule(static, 1) num_threads(T)
    for (int t = 0; t < T; ++t) {
      Sink* s = &sinks[(size_t)t * kSinks];
      for (int k = 0; k < kSinks; ++k) s[k].clear();
      SeCounts c;
      const uint32_t lo = (uint32_t)((uint64_t)n * t / T), hi = (uint32_t)((uint64_t)n * (t + 1) / T);
      // one allocation per thread and batch instead of doubling: name + 2 x read + fixed fields per line
      s[kMain].grow((b.offsets[hi] - b.offsets[lo]) * 2 + (size_t)(hi - lo) * 96);
      for (uint32_t j = lo; j < hi; ++j) {
        c.update(res[j].times);
        if (!o.sam) out_single_results(res[j], b.name(j), b.seq(j), b.score(j), g, o.ag, side.out_amb, side.out_unm,
                                       s[kMain], s[kAmb1], s[kUnm1]);
        else out_single_sam(res[j], b.name(j), b.seq(j), b.score(j), g, side.out_amb, side.out_unm, s[kMain]);
      }
      acc[t] = c;
    }
    for (int t = 0; t < T; ++t) st.add(acc[t]);
    t_out += now_s() - t0;
    if (!more) unlock_last.start([&, cur]() { bt[cur].release(); });  // (the lines are in the sinks; names and qualities are views into the file)
    t0 = now_s();
    fout.write_sinks(sinks, kSinks, kMain, T);
    side.amb.write_sinks(sinks, kSinks, kAmb1, T);
    side.unm.write_sinks(sinks, kSinks, kUnm1, T);
    t_write += now_s() - t0;
    if (!more) { unlock_idle.wait(); unlock_last.wait(); break; }
    t0 = now_s();
    pre.wait();  // what is still left of the next batch's ingest
    t_load += now_s() - t0;
  }
  if (o.verbose)
    fprintf(stderr, "[walt_amd ingest: line scan %.2f s, views %.2f s, buffers %.2f s, copy %.2f s, N draws %.2f s]\n",
            rd.t_scan, rd.t_views, rd.t_alloc, rd.t_copy, rd.t_rng);
  rd.close();
  fout.close();
  side.close();
  walt_host_free(res);
  {
    Sink ms;
    st.put_block(ms, 0);
    ms.ch('\n');
    OutFile mf;
    if (!mf.open_append(out_file + ".mapstats")) die("cannot open input file " + out_file + ".mapstats");
    mf.write(ms.p, ms.n);
    mf.close();
  }
  const double t_c0 = now_s();
  dev.close();
  if (o.verbose)
    fprintf(stderr, "[walt_amd: %d host threads, %zu GPU(s); index %.2f s, ingest not hidden behind the previous batch %.2f s, map %.2f s, "
            "format %.2f s, write %.2f s; opening the reads %.2f s, closing the index %.2f s, since main %.2f s]\n", T, dev.size(), t_index,
            t_load, t_map, t_out, t_write, t_open_reads, now_s() - t_c0, now_s() - g_t_main);
  if (o.verbose && last_file && getenv("WALT_AMD_HOST_CEILING")) host_ceiling(out_file, T);
  if (last_file) leave_now(o.verbose);
}

// ---------------------------------------------------------------- paired-end writers
static void forward_pos(uint32_t gp, char strand, uint32_t chr, uint32_t read_len, const GenomeInfo& g, uint32_t& s,
                        uint32_t& e) {  // paired.cpp:98-104
  s = gp - g.start[chr];
  s = strand == '+' ? s : g.length[chr] - s - read_len;
  e = s + read_len;
}
// A uniquely paired fragment (OutputBestPairedResults, paired.cpp:210-294): its length, and in MR mode the FRAG line.
// Both mates are laid over the fragment, which is written in mate 1's reading direction: position x of the line is
// base x of mate 1 and base j2(x) of the reverse-complemented mate 2.  Where only one mate covers x its base is
// printed; where both do, the base of the mate with more informative positions (length minus Ns minus mismatches;
// mate 1 on a tie); where neither does (mates that do not meet), 'N' with quality 'B'.
static inline char comp_base(char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c; }
static uint32_t count_n(View v) {
  uint32_t n = 0;
  for (uint32_t i = 0; i < v.len; ++i) n += v.p[i] == 'N';
  return n;
}
static int put_fragment(const walt_candidate& r1, const walt_candidate& r2, int frag_range, const GenomeInfo& g, View name,
                        View seq1, View scr1, View seq2, View scr2, bool length_only, Sink& fout) {
  const uint32_t c1 = chrom_id(g, r1.genome_pos), c2 = chrom_id(g, r2.genome_pos);
  uint32_t s1, s2, e1, e2;
  forward_pos(r1.genome_pos, r1.strand, c1, seq1.len, g, s1, e1);
  forward_pos(r2.genome_pos, r2.strand, c2, seq2.len, g, s2, e2);
  const bool fwd = r1.strand == '+';
  // the fragment runs from mate 1's 5' end to mate 2's: [s1, e2) read upwards, or [s2, e1) read downwards
  const int len = fwd ? (int)(e2 - s1) : (int)(e1 - s2);
  if (length_only) return len;
  const bool lay = len > 0 && len <= frag_range;
  const int info1 = (int)seq1.len - (int)(count_n(seq1) + r1.mismatch);
  const int info2 = (int)seq2.len - (int)(count_n(seq2) + r2.mismatch);
  const bool mate1_wins = info1 >= info2;
  // index into the reverse-complemented mate 2 of line position x: (s1 + x) - s2 upwards, e2 - 1 - (e1 - 1 - x) downwards
  const int64_t shift = fwd ? (int64_t)s1 - (int64_t)s2 : (int64_t)e2 - (int64_t)e1;
  const uint32_t start_pos = fwd ? s1 : s2;
  fout.put(g.name[c1]); fout.ch('\t'); fout.u32(start_pos); fout.ch('\t'); fout.u32(start_pos + len);
  fout.lit("\tFRAG:"); fout.put(name); fout.ch('\t'); fout.u32(r1.mismatch + r2.mismatch); fout.ch('\t');
  fout.ch(r1.strand); fout.ch('\t');
  for (int pass = 0; pass < 2; ++pass) {  // bases, then qualities
    char* out = fout.grow((size_t)len);
    for (int x = 0; x < len; ++x) {
      const int64_t j2 = (int64_t)x + shift;
      const bool in1 = lay && (uint32_t)x < seq1.len, in2 = lay && j2 >= 0 && j2 < (int64_t)seq2.len;
      char ch = pass ? 'B' : 'N';
      if (in1 && (!in2 || mate1_wins)) {
        ch = pass ? scr1.p[x] : seq1.p[x];
      } else if (in2) {
        const uint32_t k = seq2.len - 1 - (uint32_t)j2;  // the reverse complement is never materialised
        ch = pass ? scr2.p[k] : comp_base(seq2.p[k]);
      }
      out[x] = ch;
    }
    fout.n += (size_t)len;
    fout.ch(pass ? '\n' : '\t');
  }
  return len;
}
static int sam_flag(bool paired_mapped, bool unmapped, bool next_unmapped, bool rev, bool next_rev, bool first,
                    bool secondary) {  // GetSAMFLAG, paired.cpp:80-95
  return 0x1 + (paired_mapped ? 0x2 : 0) + (unmapped ? 0x4 : 0) + (next_unmapped ? 0x8 : 0) + (rev ? 0x10 : 0) +
         (next_rev ? 0x20 : 0) + (first ? 0x40 : 0x80) + (secondary ? 0x100 : 0);
}
static void sam_mate_line(Sink& f, View name, int flag, bool mapped, const string& chrom, uint32_t pos, uint32_t read_len,
                          const string& rnext, uint32_t pnext, int tlen, View seq, View score, bool flip, uint32_t mm) {
  f.put(name); f.ch('\t'); f.i32(flag); f.ch('\t');
  if (mapped) { f.put(chrom); f.ch('\t'); f.u32(pos); f.lit("\t255\t"); f.u32(read_len); f.lit("M\t"); }
  else { f.lit("*\t"); f.u32(pos); f.lit("\t255\t*\t"); }
  f.put(rnext); f.ch('\t'); f.u32(pnext); f.ch('\t'); f.i32(tlen); f.ch('\t');
  put_seq_qual(f, seq, score, flip);
  f.lit("\tNM:i:"); f.u32(mm); f.ch('\n');
}
// OutputPairedSAM, paired.cpp:333-435
static void out_paired_sam(const walt_best_match& b1, const walt_best_match& b2, const GenomeInfo& g, View name,
                           View seq1, View scr1, View seq2, View scr2, int len, int flag_1, int flag_2, bool out_amb,
                           bool out_unm, bool second_first, Sink& fout) {
  uint32_t c1 = chrom_id(g, b1.genome_pos), c2 = chrom_id(g, b2.genome_pos);
  uint32_t s1, s2, e1, e2;
  forward_pos(b1.genome_pos, b1.strand, c1, seq1.len, g, s1, e1);
  forward_pos(b2.genome_pos, b2.strand, c2, seq2.len, g, s2, e2);
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
  auto first = [&]() {
    if (b1.times == 0 && out_unm)
      sam_mate_line(fout, name, flag_1, false, g.name[c1], s1, seq1.len, rn2, s2, len1, seq1, scr1, b1.strand == '-', mm1);
    else if (b1.times == 1 || (b1.times >= 2 && out_amb))
      sam_mate_line(fout, name, flag_1, true, g.name[c1], s1, seq1.len, rn2, s2, len1, seq1, scr1, b1.strand == '-', mm1);
  };
  auto second = [&]() {
    if (b2.times == 0 && out_unm)
      sam_mate_line(fout, name, flag_2, false, g.name[c2], s2, seq2.len, rn1, s1, len2, seq2, scr2, b2.strand == '-', mm2);
    else if (b2.times == 1 || (b2.times >= 2 && out_amb))
      sam_mate_line(fout, name, flag_2, true, g.name[c2], s2, seq2.len, rn1, s1, len2, seq2, scr2, b2.strand == '-', mm2);
  };
  if (second_first) { second(); first(); } else { first(); second(); }
}

struct PeAcc {
  SeCounts st1, st2;
  uint32_t unique_pairs = 0, ambiguous_pairs = 0, unmapped_pairs = 0;
  vector<uint32_t> frag_count;
};

// ProcessPairedEndReads, paired.cpp:572-713.
// -P (PBAT): the snapshot of the reference has no code for it ("parity unpinned"); it is DEFINED
// here by equivalence to a run the reference can do: mate 1 is the A-rich read, so the pair is mapped with the
// mate files exchanged (mate 2 against the C->T indexes, mate 1 against the G->A indexes), and the output is
// then put back in the user's order: mate 1's record / line / _1 side files / mapstats block first, FLAG
// 0x40 on mate 1 and 0x80 on mate 2, QNAME from the -1 file.
static void process_pe(const Options& o, const string& file1, const string& file2, const string& out_file, bool last_file) {
  const bool pbat = o.pbat;
  const string& f1 = pbat ? file2 : file1;  // slot 0: the T-rich mate, mapped on _CT00/_CT01
  const string& f2 = pbat ? file1 : file2;  // slot 1: the A-rich mate, mapped on _GA10/_GA11
  const int name_slot = pbat ? 1 : 0;       // paired.cpp:694 prints the -1 file's name for both records
  const int T = host_threads(o);
  const int T_bg = std::max(1, T / 4);
  double t0 = now_s();
  DeviceSet dev;
  dev.open(o, WALT_STRANDS_ALL);
  double t_index = now_s() - t0, t_load = 0, t_map = 0, t_out = 0;
  GenomeInfo g = genome_of(dev.idx[0]);
  hostio::FastqReader rd[2];
  rd[0].open(f1, T);
  rd[1].open(f2, T);
  const AdaptorPair ap = split_adaptor_option(o.adaptor);
  const string adaptors[2] = {ap.mate1, ap.mate2};
  OutFile fout;
  if (!fout.open_append(out_file)) die("cannot open input file " + out_file);
  SideFiles side1, side2;  // StatPairedReads, paired.hpp:78-106
  side1.open(o.ambiguous, o.unmapped, out_file + "_1", o.sam);
  side2.open(o.ambiguous, o.unmapped, out_file + "_2", o.sam);
  SeCounts st1, st2;
  uint32_t total_pairs = 0, unique_pairs = 0, ambiguous_pairs = 0, unmapped_pairs = 0;
  vector<uint32_t> frag_count(o.frag_range + 1, 0);
  fprintf(stderr, "[MAPPING PAIRED-END READS FROM THE FOLLOWING TWO FILES]\n   %s (AND)\n   %s\n", file1.c_str(), file2.c_str());
  fprintf(stderr, "[OUTPUT MAPPING RESULTS TO %s]\n", out_file.c_str());
  if (o.sam) { string h = sam_head(g); fout.write(h.data(), h.size()); }
  Batch bts[2][2];  // [buffer][mate]: one pair of batches is mapped and written while the next is read
  walt_pair_result* pr = nullptr;
  size_t pr_cap = 0;
  vector<Sink> sinks((size_t)T * kSinks);
  vector<PeAcc> acc(T);
  Prefetch pre, unlock_idle;  // (the idle buffers of the last batch: see process_se)
  auto load_pair = [&](Batch* b) {  // mate 1's file, then mate 2's: each load starts its own srand(0) sequence (paired.cpp:648)
    rd[0].load(o.batch_size, adaptors[0], b[0]);
    if (b[0].n) rd[1].load(o.batch_size, adaptors[1], b[1]); else b[1].n = 0;
  };
  t0 = now_s();
  load_pair(bts[0]);
  t_load += now_s() - t0;
  for (int cur = 0;; cur ^= 1) {
    Batch* bt = bts[cur];
    if (bt[0].n && bt[0].n != bt[1].n) {
      fprintf(stderr, "The number of reads in paired-end files should be the same.\n");
      exit(EXIT_FAILURE);
    }
    if (bt[0].n == 0) break;
    const uint32_t n = bt[0].n;
    const bool more = n == o.batch_size;
    if (more) {
      rd[0].threads = rd[1].threads = T_bg;
      pre.start([&, cur]() { load_pair(bts[cur ^ 1]); });
    }
    total_pairs += n;
    if (pr_cap < n) {
      walt_host_free(pr);
      pr = nullptr;
      pr_cap = n + n / 8;
      check(walt_host_alloc(pr_cap * sizeof(walt_pair_result), (void**)&pr));
    }
    t0 = now_s();
    // the best pair's two candidates come back inside walt_pair_result (m1/m2), so the ranked lists stay on the GPU
    vector<uint64_t> short1(dev.size(), 0), short2(dev.size(), 0);
    dev.for_each_share(n, [&](size_t d, uint32_t lo, uint32_t hi) {
      walt_batch_stats bs[2];
      const int rc = walt_map_pe_batch(dev.idx[d], bt[0].bases, bt[0].offsets + lo, bt[1].bases, bt[1].offsets + lo, hi - lo,
                                       o.max_mismatches, o.b, o.top_k, o.frag_range, pr + lo, nullptr, nullptr, nullptr, nullptr, bs);
      short1[d] = bs[0].too_short;
      short2[d] = bs[1].too_short;
      return rc;
    });
    for (size_t d = 0; d < dev.size(); ++d) { st1.too_short += (uint32_t)short1[d]; st2.too_short += (uint32_t)short2[d]; }
    t_map += now_s() - t0;
    if (!more) unlock_idle.start([&, cur]() { bts[cur ^ 1][0].release(); bts[cur ^ 1][1].release(); });
    t0 = now_s();
#pragma omp parallel for schedule(static, 1) num_threads(T)
    for (int t = 0; t < T; ++t) {
      Sink* s = &sinks[(size_t)t * kSinks];
      for (int k = 0; k < kSinks; ++k) s[k].clear();
      PeAcc a;
      a.frag_count.assign(o.frag_range + 1, 0);
      const uint32_t lo = (uint32_t)((uint64_t)n * t / T), hi = (uint32_t)((uint64_t)n * (t + 1) / T);
      for (uint32_t j = lo; j < hi; ++j) {  // MergePairedEndResults tail, paired.cpp:515-569
        const walt_pair_result& p = pr[j];
        const View name = bt[name_slot].name(j), q1 = bt[0].seq(j), k1 = bt[0].score(j), q2 = bt[1].seq(j), k2 = bt[1].score(j);
        walt_best_match bm1 = {0, 0, '+', {0, 0, 0}, o.max_mismatches}, bm2 = bm1;
        bool is_paired = false;
        int len = 0;
        if (p.best_times == 1) {
          a.unique_pairs++;
          walt_candidate r1 = {p.m1.genome_pos, p.m1.strand, {0, 0, 0}, p.m1.mismatch};
          walt_candidate r2 = {p.m2.genome_pos, p.m2.strand, {0, 0, 0}, p.m2.mismatch};
          len = put_fragment(r1, r2, o.frag_range, g, name, q1, k1, q2, k2, o.sam, s[kMain]);
          a.frag_count[len]++;
          if (o.sam) { is_paired = true; bm1 = p.m1; bm2 = p.m2; }
        } else {
          if (p.best_times >= 2) a.ambiguous_pairs++; else a.unmapped_pairs++;
          bm1 = p.m1; bm2 = p.m2;
          a.st1.update(bm1.times);
          a.st2.update(bm2.times);
          if (!o.sam && !pbat) {
            out_single_results(bm1, name, q1, k1, g, false, side1.out_amb, side1.out_unm, s[kMain], s[kAmb1], s[kUnm1]);
            out_single_results(bm2, name, q2, k2, g, true, side2.out_amb, side2.out_unm, s[kMain], s[kAmb2], s[kUnm2]);
          } else if (!o.sam) {  // the user's mate 1 sits in slot 1
            out_single_results(bm2, name, q2, k2, g, true, side1.out_amb, side1.out_unm, s[kMain], s[kAmb1], s[kUnm1]);
            out_single_results(bm1, name, q1, k1, g, false, side2.out_amb, side2.out_unm, s[kMain], s[kAmb2], s[kUnm2]);
          }
        }
        if (o.sam) {
          int fl1 = sam_flag(is_paired, bm1.times == 0, bm2.times == 0, bm1.strand == '-', bm2.strand == '-', !pbat, bm1.times >= 2);
          int fl2 = sam_flag(is_paired, bm2.times == 0, bm1.times == 0, bm2.strand == '-', bm1.strand == '-', pbat, bm2.times >= 2);
          out_paired_sam(bm1, bm2, g, name, q1, k1, q2, k2, len, fl1, fl2, o.ambiguous, o.unmapped, pbat, s[kMain]);
        }
      }
      acc[t] = std::move(a);
    }
    for (int t = 0; t < T; ++t) {
      st1.add(acc[t].st1);
      st2.add(acc[t].st2);
      unique_pairs += acc[t].unique_pairs; ambiguous_pairs += acc[t].ambiguous_pairs; unmapped_pairs += acc[t].unmapped_pairs;
      for (size_t i = 0; i < frag_count.size(); ++i) frag_count[i] += acc[t].frag_count[i];
    }
    fout.write_sinks(sinks, kSinks, kMain, T);
    side1.amb.write_sinks(sinks, kSinks, kAmb1, T);
    side1.unm.write_sinks(sinks, kSinks, kUnm1, T);
    side2.amb.write_sinks(sinks, kSinks, kAmb2, T);
    side2.unm.write_sinks(sinks, kSinks, kUnm2, T);
    t_out += now_s() - t0;
    if (!more) { unlock_idle.wait(); break; }
    t0 = now_s();
    pre.wait();
    t_load += now_s() - t0;
  }
  rd[0].close(); rd[1].close();
  fout.close();
  side1.close(); side2.close();
  walt_host_free(pr);
  {  // the block StatPairedReads prints (paired.cpp:52-77): pair counters, one mate block each, the fragment histogram
    Sink ms;
    char num[48];
    ms.lit("pairs:\n    total_read_pairs: "); ms.u32(total_pairs);
    ms.lit("\n    mapped:\n        unique: "); ms.u32(unique_pairs);
    snprintf(num, sizeof num, "%g", (100.0 * unique_pairs) / total_pairs);
    ms.lit("\n        percent_unique: "); ms.lit(num);
    ms.lit("\n        ambiguous: "); ms.u32(ambiguous_pairs);
    ms.lit("\n    unmapped: "); ms.u32(unmapped_pairs);
    ms.lit("\nmate1:\n"); (pbat ? st2 : st1).put_block(ms, 1);
    ms.lit("\nmate2:\n"); (pbat ? st1 : st2).put_block(ms, 1);
    ms.lit("\nfrag_len_distribution:\n");
    double weighted = 0.0, pairs_in_hist = 0.0;
    for (size_t i = 0; i < frag_count.size(); ++i) {
      ms.lit("    "); ms.u32((uint32_t)i); ms.lit(": "); ms.u32(frag_count[i]); ms.ch('\n');
      weighted += (double)i * frag_count[i];
      pairs_in_hist += frag_count[i];
    }
    snprintf(num, sizeof num, "%g", weighted / pairs_in_hist);
    ms.lit("frag_len_mean: "); ms.lit(num); ms.ch('\n');
    OutFile mf;
    if (!mf.open_append(out_file + ".mapstats")) die("cannot open input file " + out_file + ".mapstats");
    mf.write(ms.p, ms.n);
    mf.close();
  }
  dev.close();
  if (o.verbose)
    fprintf(stderr, "[walt_amd: %d host threads, %zu GPU(s); index %.2f s, ingest not hidden behind the previous batch %.2f s, map %.2f s, "
            "output %.2f s]\n", T, dev.size(), t_index, t_load, t_map, t_out);
  if (last_file) leave_now(o.verbose);
}

int main(int argc, const char** argv) {
  g_t_main = now_s();
  try {
    if (argc == 1) {
      fprintf(stderr, "Usage: walt -i <index> -r <reads> | -1 <reads1> -2 <reads2> -o <out> [-m -N -a -u -C -A -P -b -k -L -sam -v -t -g <gpu>[,<gpu>...]]\n");
      return EXIT_SUCCESS;
    }
    Options o = parse(argc, argv);
    setenv("GPU_MAX_HW_QUEUES", "8", 0);  // paired-end keeps several streams busy; read by the HIP runtime at start-up
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
    const size_t n_files = se.size() + p1.size();
    for (auto& f : se) { process_se(o, f, outs[k], k + 1 == n_files); ++k; }
    for (size_t i = 0; i < p1.size(); ++i) { process_pe(o, p1[i], p2[i], outs[k], k + 1 == n_files); ++k; }
    if (o.verbose) fprintf(stderr, "[walt_amd: %.2f s in main]\n", now_s() - g_t_main);
  } catch (const std::exception& e) {
    std::cerr << e.what() << std::endl;
    return EXIT_FAILURE;
  }
  return EXIT_SUCCESS;
}
