// hostio_harness.cpp -- CPU test harness for walt_amd/csrc/host/hostio.h (TEST
// INFRASTRUCTURE): runs the driver's FASTQ loader without a GPU (malloc instead
// of the page-locked allocator) and dumps every batch as text so that the tests
// can compare the multi-threaded scanner, its serial restatement and the Python
// restatement of LoadReadsFromFastqFile (tests/refio.py) on awkward inputs.
#include <stdlib.h>
static int hh_alloc(size_t bytes, void** out) { *out = malloc(bytes ? bytes : 1); return *out ? 0 : -5; }
#define HOSTIO_ALLOC(bytes, out) hh_alloc((bytes), (out))
#define HOSTIO_FREE(p) free(p)
#define HOSTIO_ALLOC_ERROR() "out of memory"
#include "../walt_amd/csrc/host/hostio.h"

// dump format: one line "#batch <n>" per batch, then "<name>\t<seq>\t<score>" per read
extern "C" int hio_dump(const char* fastq, uint32_t n_per_batch, const char* adaptor, int threads, int force_serial,
                        const char* out_path, int* used_serial) {
  try {
    hostio::FastqReader rd;
    if (force_serial) setenv("WALT_AMD_SERIAL_IO", "1", 1); else unsetenv("WALT_AMD_SERIAL_IO");
    rd.open(fastq, threads);
    hostio::Batch bt;
    hostio::Sink s;
    for (;;) {
      rd.load(n_per_batch, adaptor ? adaptor : "", bt);
      if (bt.n == 0) break;
      s.lit("#batch "); s.u32(bt.n); s.ch('\n');
      for (uint32_t j = 0; j < bt.n; ++j) {
        s.put(bt.name(j)); s.ch('\t'); s.put(bt.seq(j)); s.ch('\t'); s.put(bt.score(j)); s.ch('\n');
      }
      if (bt.n < n_per_batch) break;
    }
    if (used_serial) *used_serial = rd.serial ? 1 : 0;
    rd.close();
    hostio::OutFile f;
    if (!f.open_trunc(out_path)) return -2;
    f.write(s.p, s.n);
    f.close();
    return 0;
  } catch (const std::exception& e) {
    fprintf(stderr, "hio_dump: %s\n", e.what());
    return -1;
  }
}

// Sink number formatting and reverse complement against printf / a plain loop
extern "C" int hio_format_check(void) {
  hostio::Sink s;
  const uint32_t us[] = {0u, 7u, 10u, 99u, 100u, 4294967295u, 123456789u};
  const int is[] = {0, -1, 17, -2147483647 - 1, 2147483647, -500};
  std::string want;
  char tmp[64];
  for (uint32_t v : us) { s.u32(v); s.ch(' '); snprintf(tmp, sizeof tmp, "%u ", v); want += tmp; }
  for (int v : is) { s.i32(v); s.ch(' '); snprintf(tmp, sizeof tmp, "%d ", v); want += tmp; }
  const char* q = "ACGTNacgt";
  hostio::View v{q, 9};
  s.revcomp(v); s.ch(' '); s.rev(v);
  want += "tgcaNACGT tgcaNTGCA";
  return std::string(s.p, s.n) == want ? 0 : 1;
}
