// hostio.h -- multi-threaded FASTQ ingest and output assembly for the walt
// driver.  The reference loads and prints serially (mapping.cpp:65-121,
// 329-419); at GPU mapping rates that is the whole run time, so the driver
// splits both sides over host threads while keeping every byte the reference
// would produce:
//   * batch boundaries, blank-line skipping, "drop the last character of every
//     fgets line" and name cutting follow LoadReadsFromFastqFile line by line;
//   * the N -> rand()%4 draws (util.hpp:156-163, srand(0) per batch,
//     mapping.cpp:73) are applied by ONE thread in file order after the
//     parallel copy has listed the characters that need one;
//   * a line of 999+ bytes, which fgets would split, makes the SAME scanner run over the
//     rest of the file as one chunk on one thread with the split rule applied; a file
//     that cannot be mmapped (a pipe) is read into memory first and scanned there.
#ifndef WALT_AMD_HOSTIO_H_
#define WALT_AMD_HOSTIO_H_
#include <fcntl.h>
#include <omp.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/walt_amd.h"

// Read / offset buffers come from the library's page-locked allocator; the CPU
// test harness (tests/hostio_harness.cpp) substitutes malloc to run without a GPU.
#ifndef HOSTIO_ALLOC
#define HOSTIO_ALLOC(bytes, out) walt_host_alloc((bytes), (out))
#define HOSTIO_FREE(p) walt_host_free(p)
#define HOSTIO_ALLOC_ERROR() walt_last_error()
#endif

namespace hostio {

static const uint32_t kMaxLine = 1000;  // MAX_LINE_LENGTH, util.hpp:43

struct View {
  const char* p;
  uint32_t len;
};

// ---------------------------------------------------------------- output buffers
struct Sink {
  char* p = nullptr;
  size_t n = 0, cap = 0;
  Sink() {}
  Sink(const Sink&) = delete;
  Sink& operator=(const Sink&) = delete;
  ~Sink() { free(p); }
  void clear() { n = 0; }
  char* grow(size_t add) {
    if (n + add > cap) {
      size_t nc = std::max(cap * 2, n + add + 4096);
      char* q = static_cast<char*>(realloc(p, nc));
      if (!q) throw std::bad_alloc();
      p = q;
      cap = nc;
    }
    return p + n;
  }
  void put(const char* s, size_t len) { memcpy(grow(len), s, len); n += len; }
  void put(const std::string& s) { put(s.data(), s.size()); }
  void put(View v) { put(v.p, v.len); }
  void lit(const char* s) { put(s, strlen(s)); }
  void ch(char c) { *grow(1) = c; n += 1; }
  void u32(uint32_t v) {
    char tmp[12];
    int k = 0;
    do { tmp[k++] = (char)('0' + v % 10); v /= 10; } while (v);
    char* q = grow(k);
    for (int i = 0; i < k; ++i) q[i] = tmp[k - 1 - i];
    n += k;
  }
  void i32(int v) {
    if (v < 0) { ch('-'); u32((uint32_t)(-(int64_t)v)); } else u32((uint32_t)v);
  }
  // reversed / reverse-complemented copies (revcomp as in smithlab_utils, used at mapping.cpp:337,393)
  void rev(View v) {
    char* q = grow(v.len);
    for (uint32_t i = 0; i < v.len; ++i) q[i] = v.p[v.len - 1 - i];
    n += v.len;
  }
  void revcomp(View v) {
    char* q = grow(v.len);
    for (uint32_t i = 0; i < v.len; ++i) {
      char c = v.p[v.len - 1 - i];
      q[i] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c;
    }
    n += v.len;
  }
};

// An output file written at explicit offsets so that per-thread buffers can be
// stored concurrently and still land in read order.
struct OutFile {
  int fd = -1;
  off_t pos = 0;
  bool open_append(const std::string& path) {  // fopen(path, "a") of the reference (mapping.cpp:460)
    fd = ::open(path.c_str(), O_WRONLY | O_CREAT, 0666);
    if (fd < 0) return false;
    pos = lseek(fd, 0, SEEK_END);
    return true;
  }
  bool open_trunc(const std::string& path) {  // fopen(path, "w") (mapping.hpp:75-87)
    fd = ::open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
    pos = 0;
    return fd >= 0;
  }
  void close() { if (fd >= 0) ::close(fd); fd = -1; }
  static void pwrite_all(int fd, const char* p, size_t len, off_t at) {
    while (len) {
      ssize_t w = ::pwrite(fd, p, len, at);
      if (w < 0) throw std::runtime_error("write failed");
      p += w; len -= (size_t)w; at += w;
    }
  }
  void write(const char* p, size_t len) { pwrite_all(fd, p, len, pos); pos += (off_t)len; }
  // sinks[t * stride + which] for t in [0, T): stored in thread order.  (Growing the file and copying into a shared
  // mapping of the new part, to get around the inode lock pwrite calls on one file take turns on, was tried: the page
  // faults of a tmpfs mapping cost more than the lock -- 2.95 s against 1.41 s for 10 GB.)
  void write_sinks(std::vector<Sink>& sinks, size_t stride, size_t which, int threads) {
    if (fd < 0) return;
    const size_t T = sinks.size() / stride;
    std::vector<off_t> at(T + 1, pos);
    for (size_t t = 0; t < T; ++t) at[t + 1] = at[t] + (off_t)sinks[t * stride + which].n;
    bool failed = false;
#pragma omp parallel for schedule(static, 1) num_threads(threads)
    for (long t = 0; t < (long)T; ++t) {
      const Sink& s = sinks[t * stride + which];
      try { if (s.n) pwrite_all(fd, s.p, s.n, at[t]); } catch (...) { failed = true; }
    }
    if (failed) throw std::runtime_error("write failed");
    pos = at[T];
  }
};

// ---------------------------------------------------------------- read batches
struct Batch {
  uint32_t n = 0;
  const char* base = nullptr;            // the input file's bytes (mapped, or read into memory)
  std::vector<uint64_t> name_v, score_v, seq_v;  // (offset << 16) | length
  char* bases = nullptr;                 // sanitised sequences, packed; pinned host memory
  uint64_t* offsets = nullptr;           // n + 1
  size_t bases_cap = 0, off_cap = 0;
  Batch() {}
  Batch(const Batch&) = delete;
  Batch& operator=(const Batch&) = delete;
  ~Batch() { release(); }
  // give the page-locked buffers back (unlocking a gigabyte takes ~0.2 s: the driver does it on a helper thread while
  // the last batch is formatted and written, walt_main.cpp)
  void release() {
    HOSTIO_FREE(bases); HOSTIO_FREE(offsets);
    bases = nullptr; offsets = nullptr;
    bases_cap = off_cap = 0;
    n = 0;
  }
  static uint64_t pack(size_t off, size_t len) { return ((uint64_t)off << 16) | (uint64_t)len; }
  View name(uint32_t j) const { return View{base + (name_v[j] >> 16), (uint32_t)(name_v[j] & 0xFFFF)}; }
  View score(uint32_t j) const { return View{base + (score_v[j] >> 16), (uint32_t)(score_v[j] & 0xFFFF)}; }
  View seq(uint32_t j) const { return View{bases + offsets[j], (uint32_t)(offsets[j + 1] - offsets[j])}; }
  void reserve_reads(size_t reads) {
    if (name_v.size() < reads) { name_v.resize(reads); score_v.resize(reads); seq_v.resize(reads); }
    if (off_cap < reads + 1) {
      HOSTIO_FREE(offsets);
      offsets = nullptr;
      off_cap = reads + 1;
      if (HOSTIO_ALLOC(off_cap * sizeof(uint64_t), (void**)&offsets) != 0) throw std::runtime_error(HOSTIO_ALLOC_ERROR());
    }
  }
  void reserve_bases(size_t bytes) {
    if (bases_cap < bytes + 16) {
      HOSTIO_FREE(bases);
      bases = nullptr;
      bases_cap = bytes + bytes / 8 + 4096;
      if (HOSTIO_ALLOC(bases_cap, (void**)&bases) != 0) throw std::runtime_error(HOSTIO_ALLOC_ERROR());
    }
  }
};

// ---------------------------------------------------------------- adaptor clipping (-C)
// What util.hpp:189-218 decides, stated per start position: the read's tail from `at` on is taken for adaptor
// when, of the first min(tail, adaptor, 14) characters of the adaptor, at least 11 agree with it (tails of 14
// bases or more), or all but one of the whole tail do (tails of 13 down to 5 bases).  Returns the first such
// `at`, or the read's length when there is none; the caller writes 'N' from there on (the loader then draws
// random bases for them like for any other non-ACGT character).
// The reference computes its loop bounds as size_t differences, which wrap for reads of fewer than 13 bases and
// send it reading far outside the string (it crashes: tests/test_hostio_cpu.py pins both sides of that limit
// against the reference binary); such reads are left alone here.
static const uint32_t kAdaptorHead = 14, kAdaptorHeadHits = 11, kAdaptorMinTail = 5;
inline uint32_t adaptor_clip_point(View read, View adaptor) {
  if (read.len < kAdaptorHead - 1) return read.len;
  for (uint32_t at = 0; at + kAdaptorMinTail <= read.len; ++at) {
    const uint32_t tail = read.len - at;
    const uint32_t window = std::min(std::min(tail, adaptor.len), kAdaptorHead);
    const uint32_t need = tail >= kAdaptorHead ? kAdaptorHeadHits : tail - 1;
    if (window < need) continue;
    uint32_t hits = 0;
    for (uint32_t k = 0; k < window; ++k) hits += read.p[at + k] == adaptor.p[k] ? 1u : 0u;
    if (hits >= need) return at;
  }
  return read.len;
}

inline bool is_acgt(char c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }

// CPUs this process may actually use: the affinity mask, capped by the cgroup CPU
// quota (containers often expose every core of the host but grant far fewer;
// running more threads than the quota only gets them throttled).
inline int effective_cpus() {
  int n = omp_get_num_procs();
  long long quota = -1, period = 100000;
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota|max> <period>"
    char q[64];
    if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
    fclose(f);
  } else {
    FILE* fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r");
    FILE* fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
    if (fq && fp && fscanf(fq, "%lld", &quota) == 1 && fscanf(fp, "%lld", &period) == 1) {}
    if (fq) fclose(fq);
    if (fp) fclose(fp);
  }
  if (quota > 0 && period > 0) {
    int c = (int)((quota + period - 1) / period);
    if (c >= 1 && c < n) n = c;
  }
  return n < 1 ? 1 : n;
}

inline double now_s() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

struct FastqReader {
  double t_scan = 0, t_views = 0, t_alloc = 0, t_copy = 0, t_rng = 0;  // -v breakdown
  std::string path;
  const char* data = nullptr;
  size_t size = 0, pos = 0;
  bool mapped = false;
  std::vector<char> slurped;  // contents of an input that cannot be mmapped
  bool serial = false;        // a long line was met: one chunk, one thread from there on (kept for the tests)
  int threads = 1;

  void open(const std::string& p, int nthreads) {
    path = p;
    threads = std::max(1, nthreads);
    int fd = ::open(p.c_str(), O_RDONLY);
    if (fd < 0) throw std::runtime_error("cannot open input file " + p);
    struct stat st;
    if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0 && !getenv("WALT_AMD_SERIAL_IO")) {
      void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
      if (m != MAP_FAILED) {
        data = static_cast<const char*>(m);
        size = (size_t)st.st_size;
        mapped = true;
        madvise(m, size, MADV_SEQUENTIAL);
      }
    }
    if (!data) {  // pipe, empty file, or WALT_AMD_SERIAL_IO (tests): read it all, then scan the copy on one thread
      char buf[1 << 16];
      for (;;) {
        const ssize_t got = ::read(fd, buf, sizeof buf);
        if (got < 0) { ::close(fd); throw std::runtime_error("cannot read input file " + p); }
        if (got == 0) break;
        slurped.insert(slurped.end(), buf, buf + got);
      }
      data = slurped.data();
      size = slurped.size();
      serial = true;
    }
    ::close(fd);
  }
  void close() {
    if (mapped) munmap(const_cast<char*>(data), size);
    data = nullptr;
    mapped = false;
    slurped.clear();
  }

  // ---- one fgets(cline, 1000, f) call of the reference (mapping.cpp:81) starting at s: the bytes [s, e) it returns --
  // up to and including the next newline, but never more than 999 of them
  inline size_t line_end(size_t s) const {
    const size_t lim = std::min(size, s + (kMaxLine - 1));
    const void* nl = memchr(data + s, '\n', lim - s);
    return nl ? (size_t)(static_cast<const char*>(nl) - data) + 1 : lim;
  }
  // first line start at or behind a chunk boundary (only used while no line is long enough to be split)
  inline size_t first_line_at_or_after(size_t a, size_t batch_start) const {
    if (a <= batch_start) return batch_start;
    if (a >= size) return size;
    if (data[a - 1] == '\n') return a;
    const void* nl = memchr(data + a, '\n', size - a);
    return nl ? (size_t)(static_cast<const char*>(nl) - data) + 1 : size;
  }

  // LoadReadsFromFastqFile, mapping.cpp:65-121
  void load(uint32_t n_per_batch, const std::string& adaptor, Batch& bt) {
    const uint64_t lim = (uint64_t)n_per_batch * 4;
    const size_t start = pos;
    bt.n = 0;
    bt.base = data;
    if (start >= size || lim == 0) { finish_sequences(adaptor, bt); return; }
    double tm = now_s();
    std::vector<uint64_t> lines;  // non-empty lines that start in each chunk
    uint64_t total = 0;
    size_t covered = start, chunk = 1u << 20;
    for (;;) {
      // pass 1: count lines per chunk, in waves of chunks until `lim` lines or the end of the file
      const int T = serial ? 1 : threads;
      if (serial) chunk = size - start + 1;  // one chunk: the scan follows the split rule from the batch's first byte
      lines.clear();
      total = 0;
      covered = start;
      bool long_line = false;
      size_t wave = 1;  // 1, 2, 4, ... chunks per wave: a small -N must not scan far beyond its batch
      while (total < lim && covered < size) {
        const size_t first = lines.size();
        const size_t want = std::min<size_t>((size - covered + chunk - 1) / chunk, wave);
        wave = std::min<size_t>(wave * 2, (size_t)T * 4);
        lines.resize(first + want, 0);
#pragma omp parallel for schedule(dynamic, 1) num_threads(T) reduction(|| : long_line)
        for (long c = 0; c < (long)want; ++c) {
          const size_t a = start + (first + c) * chunk, b = std::min(a + chunk, size);
          size_t s = first_line_at_or_after(a, start);
          uint64_t cnt = 0;
          const uint64_t stop = (first + c == 0) ? lim : ~0ull;  // the batch's first chunk may stop at `lim` lines
          while (s < b && cnt < stop) {
            const size_t e = line_end(s);
            if (e - s >= kMaxLine - 1 && data[e - 1] != '\n') long_line = true;  // fgets filled its buffer: the line goes on
            cnt += (e - s) > 1;  // the piece minus its last character is non-empty (mapping.cpp:82-85)
            s = e;
          }
          lines[first + c] = cnt;
        }
        for (size_t c = first; c < first + want; ++c) total += lines[c];
        covered = std::min(size, start + (first + want) * chunk);
      }
      if (!long_line || serial) break;
      serial = true;  // chunk starts are not piece starts once a line is split: scan again as one chunk
    }
    t_scan += now_s() - tm;
    tm = now_s();
    const uint64_t use = std::min<uint64_t>(total, lim);
    bt.n = (uint32_t)(use / 4);  // a record counts when its 4th line has been read (mapping.cpp:110-113)
    bt.reserve_reads(bt.n + 1);
    std::vector<uint64_t> first_line(lines.size() + 1, 0);
    for (size_t c = 0; c < lines.size(); ++c) first_line[c + 1] = first_line[c] + lines[c];
    // pass 2: record name / sequence / quality views; line L belongs to read L/4, role L%4
    size_t next_pos = covered;
    const uint64_t n_lines_used = (uint64_t)bt.n * 4;
#pragma omp parallel for schedule(dynamic, 1) num_threads(serial ? 1 : threads)
    for (long c = 0; c < (long)lines.size(); ++c) {
      uint64_t L = first_line[c];
      if (L >= lim) continue;
      const size_t a = start + (size_t)c * chunk, b = std::min(a + chunk, size);
      size_t s = first_line_at_or_after(a, start);
      while (s < b) {
        const size_t e = line_end(s);
        const size_t len = e - s - 1;  // content after dropping the last character
        s = e;
        if (len == 0) continue;
        if (L < n_lines_used) {
          const uint32_t r = (uint32_t)(L >> 2);
          const size_t at = e - 1 - len;
          switch (L & 3) {
            case 0: {  // name: up to the first space, without the leading character (mapping.cpp:87-95)
              const void* sp = memchr(data + at, ' ', len);
              size_t cut = sp ? (size_t)(static_cast<const char*>(sp) - (data + at)) : len;
              if (sp && cut == 0) cut = len;  // substr(1, 0 - 1): the unsigned wrap keeps the whole rest
              bt.name_v[r] = Batch::pack(at + 1, cut - 1);
              break;
            }
            case 1: bt.seq_v[r] = Batch::pack(at, len); break;
            case 3: bt.score_v[r] = Batch::pack(at, len); break;
            default: break;
          }
        }
        ++L;
        if (L == lim) {
#pragma omp critical(walt_next_pos)
          next_pos = e;
          break;
        }
      }
    }
    pos = total >= lim ? next_pos : size;
    t_views += now_s() - tm;
    finish_sequences(adaptor, bt);
  }

  // sequences: copy into the packed buffer, clip, and give non-ACGT characters their rand()%4 in file order
  void finish_sequences(const std::string& adaptor, Batch& bt) {
    const uint32_t n = bt.n;
    bt.reserve_reads(n + 1);
    const int T = threads;
    const View ad{adaptor.data(), (uint32_t)adaptor.size()};
    std::vector<uint64_t> part(T + 1, 0);
    auto lo_of = [&](int t) { return (uint32_t)((uint64_t)n * t / T); };
#pragma omp parallel for schedule(static, 1) num_threads(T)
    for (int t = 0; t < T; ++t) {
      uint64_t sum = 0;
      for (uint32_t j = lo_of(t); j < lo_of(t + 1); ++j) sum += bt.seq_v[j] & 0xFFFF;
      part[t + 1] = sum;
    }
    for (int t = 0; t < T; ++t) part[t + 1] += part[t];
    double tm = now_s();
    bt.reserve_bases(part[T]);
    t_alloc += now_s() - tm;
    tm = now_s();
    std::vector<std::vector<uint64_t>> fix(T);
#pragma omp parallel for schedule(static, 1) num_threads(T)
    for (int t = 0; t < T; ++t) {
      uint64_t o = part[t];
      for (uint32_t j = lo_of(t); j < lo_of(t + 1); ++j) {
        const char* src = bt.base + (bt.seq_v[j] >> 16);
        const uint32_t len = (uint32_t)(bt.seq_v[j] & 0xFFFF);
        bt.offsets[j] = o;
        // -C: the adaptor part becomes 'N' (mapping.cpp:97-99), i.e. one more kind of character that needs a draw
        const uint32_t keep = ad.len ? adaptor_clip_point(View{src, len}, ad) : len;
        char* dst = bt.bases + o;
        for (uint32_t i = 0; i < len; ++i) {
          const char c = i < keep ? src[i] : 'N';
          dst[i] = c;
          if (!is_acgt(c)) fix[t].push_back(o + i);
        }
        o += len;
      }
    }
    bt.offsets[n] = part[T];
    t_copy += now_s() - tm;
    tm = now_s();
    srand(0);  // mapping.cpp:73
    for (int t = 0; t < T; ++t)
      for (uint64_t at : fix[t]) bt.bases[at] = "ACGT"[rand() % 4];  // toACGT, util.hpp:156-163
    t_rng += now_s() - tm;
  }
};

}  // namespace hostio
#endif  // WALT_AMD_HOSTIO_H_
