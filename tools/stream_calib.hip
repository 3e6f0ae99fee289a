// stream_calib.hip -- diagnostic micro-benchmark (NOT part of the product): what the device delivers for the access
// pattern of the dense candidate windows (core.h StrandView::win): each wavefront repeatedly picks a random run of
// `run_bytes` in a large table and reads it in steps of G x 64 lanes x 32 bytes (two 16-byte loads per lane),
// waiting for a step's data before it issues the next (one dependent round trip per step, like the verification
// loop).  Varies the run length, G and the number of wavefronts per CU.
//   hipcc -O3 --offload-arch=gfx950 -o stream_calib stream_calib.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}

template <int G>
__global__ __launch_bounds__(256) void k_stream(const uint4* __restrict__ tab, uint64_t n_recs /* 32-byte records */,
                                                uint32_t runs_per_wave, uint32_t recs_per_run, uint32_t* __restrict__ out,
                                                uint32_t lds_pad) {
  extern __shared__ uint32_t pad[];  // occupancy control
  if (lds_pad && threadIdx.x == 0) pad[0] = 0;
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
  uint32_t acc = 0;
  for (uint32_t r = 0; r < runs_per_wave; ++r) {
    const uint64_t h = mix(wave * 0x9E3779B97F4A7C15ULL + r);
    const uint64_t base = (h % (n_recs - recs_per_run - 64 * G)) & ~15ull;  // 512-byte aligned start
    for (uint32_t s = 0; s < recs_per_run; s += 64 * G) {
      uint4 a[G], c[G];
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const uint64_t rec = base + s + 64 * u + lane;
        a[u] = tab[2 * rec];
        c[u] = tab[2 * rec + 1];
      }
#pragma unroll
      for (int u = 0; u < G; ++u) acc += a[u].x ^ a[u].w ^ c[u].y ^ c[u].z;
      acc = __shfl_xor(acc, 1);  // a dependency between steps, like the reduction
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
  const double gb = argc > 1 ? atof(argv[1]) : 7.0;
  const uint64_t bytes = (uint64_t)(gb * (1ull << 30));
  uint4* tab;
  if (hipMalloc(&tab, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(tab, 1, bytes);
  const uint64_t n_recs = bytes / 32;
  uint32_t* out;
  hipMalloc(&out, 256u * 256 * 64 * 4);
  const int blocks_per_cu[] = {2, 3, 4, 8};
  const uint32_t run_recs[] = {64, 256, 1024, 4096};
  for (int bpc : blocks_per_cu) {
    for (uint32_t rr : run_recs) {
      for (int G : {1, 2, 4, 8}) {
        const uint32_t lds = bpc >= 8 ? 0 : (160 * 1024 / bpc) - 1024;  // dynamic LDS that leaves room for `bpc` blocks
        const unsigned blocks = 256 * bpc;
        const uint64_t target_bytes = 40ull << 30;  // per launch
        const uint64_t waves = (uint64_t)blocks * 4;
        const uint32_t rr_eff = rr < 64u * G ? 64u * G : rr;
        uint32_t runs = (uint32_t)(target_bytes / (waves * rr_eff * 32ull));
        if (runs < 1) runs = 1;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
          hipEventRecord(e0);
          switch (G) {
            case 1: hipLaunchKernelGGL(k_stream<1>, dim3(blocks), dim3(256), lds, 0, tab, n_recs, runs, rr_eff, out, lds); break;
            case 2: hipLaunchKernelGGL(k_stream<2>, dim3(blocks), dim3(256), lds, 0, tab, n_recs, runs, rr_eff, out, lds); break;
            case 4: hipLaunchKernelGGL(k_stream<4>, dim3(blocks), dim3(256), lds, 0, tab, n_recs, runs, rr_eff, out, lds); break;
            default: hipLaunchKernelGGL(k_stream<8>, dim3(blocks), dim3(256), lds, 0, tab, n_recs, runs, rr_eff, out, lds); break;
          }
          hipEventRecord(e1);
          hipEventSynchronize(e1);
          hipEventElapsedTime(&ms, e0, e1);
        }
        const double total = (double)waves * runs * rr_eff * 32.0;
        printf("table %.0f GB  blocks/CU %d  run %5u recs (%6.1f KB)  G %d : %8.3f ms  %6.2f TB/s  %.2f us/step\n", gb, bpc, rr_eff,
               rr_eff * 32 / 1024.0, G, ms, total / ms / 1e9, ms * 1e3 / ((double)runs * rr_eff / (64.0 * G)));
      }
    }
  }
  return 0;
}
