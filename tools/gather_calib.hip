// gather_calib.hip -- diagnostic micro-benchmark (NOT part of the product):
// random 4-byte / 12-byte gathers over tables of increasing size, to calibrate
// (a) the achievable random-line rate of HBM3E on MI355X, (b) how many memory
// requests one random gather costs (FETCH_SIZE / TCC_EA0_RDREQ under rocprofv3),
// i.e. the TLB page-walk overhead for multi-GB tables.
//   hipcc -O3 --offload-arch=gfx950 -o gather_calib gather_calib.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}

// each thread does `chain` DEPENDENT gathers (like a binary search) x `iters` independent chains
template <int WORDS>
__global__ void k_gather(const uint32_t* __restrict__ tab, uint64_t n_elems, int iters, int chain,
                         uint32_t* __restrict__ out) {
  uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (int it = 0; it < iters; ++it) {
    uint64_t h = mix(tid * 0x9E3779B97F4A7C15ULL + it);
    for (int c = 0; c < chain; ++c) {
      uint64_t i = h % n_elems;
      uint32_t v = 0;
#pragma unroll
      for (int w = 0; w < WORDS; ++w) v += tab[i * WORDS + w];
      acc += v;
      h = mix(h + v);
    }
  }
  out[tid] = acc;
}

// one configuration, one line (bench.py runs this beside its own kernels: the device's random-line ceiling, measured in
// the same run):  gather_calib <table GB> <words per element: 1 | 3> <dependent gathers per chain>
static int one(double gb, int words, int chain) {
  const int threads = 256, blocks = 256 * 20 * 4;
  uint32_t* out;
  if (hipMalloc(&out, (size_t)threads * blocks * 4) != hipSuccess) return 1;
  const uint64_t bytes = (uint64_t)(gb * (1ull << 30));
  uint32_t* tab;
  if (hipMalloc(&tab, bytes) != hipSuccess) { printf("alloc %.1f GB failed\n", gb); return 1; }
  hipMemset(tab, 1, bytes);
  const uint64_t n_elems = bytes / (4 * words);
  const int iters = chain == 1 ? 64 : 8;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    if (words == 1) hipLaunchKernelGGL(k_gather<1>, dim3(blocks), dim3(threads), 0, 0, tab, n_elems, iters, chain, out);
    else hipLaunchKernelGGL(k_gather<3>, dim3(blocks), dim3(threads), 0, 0, tab, n_elems, iters, chain, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  const double n = (double)threads * blocks * iters * chain;
  printf("table %6.2f GB  elem %2d B  chain %d : %8.3f ms  %7.2f G gathers/s  (%.0f gathers)\n", gb, 4 * words, chain, best,
         n / best / 1e6, n);
  hipFree(tab); hipFree(out);
  return 0;
}

int main(int argc, char** argv) {
  if (argc == 4) return one(atof(argv[1]), atoi(argv[2]), atoi(argv[3]));
  double gbs[] = {0.25, 1, 4, 16, 40, 80};
  int threads = 256, blocks = 256 * 20 * 4;  // 5 waves/SIMD worth of lanes x4
  uint32_t* out;
  hipMalloc(&out, (size_t)threads * blocks * 4);
  for (double gb : gbs) {
    uint64_t bytes = (uint64_t)(gb * (1ull << 30));
    uint32_t* tab;
    if (hipMalloc(&tab, bytes) != hipSuccess) { printf("alloc %.1f GB failed\n", gb); continue; }
    hipMemset(tab, 1, bytes);
    for (int words : {1, 3}) {
      for (int chain : {1, 8}) {
        uint64_t n_elems = bytes / (4 * words);
        int iters = chain == 1 ? 64 : 8;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
          hipEventRecord(e0);
          if (words == 1) hipLaunchKernelGGL(k_gather<1>, dim3(blocks), dim3(threads), 0, 0, tab, n_elems, iters, chain, out);
          else hipLaunchKernelGGL(k_gather<3>, dim3(blocks), dim3(threads), 0, 0, tab, n_elems, iters, chain, out);
          hipEventRecord(e1);
          hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double n = (double)threads * blocks * iters * chain;
        printf("table %6.2f GB  elem %2d B  chain %d : %8.3f ms  %7.2f G gathers/s  (%.0f gathers)\n", gb, 4 * words,
               chain, ms, n / ms / 1e6, n);
      }
    }
    hipFree(tab);
  }
  return 0;
}
