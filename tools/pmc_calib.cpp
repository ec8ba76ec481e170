// bring-up: known access patterns for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 at the widths the codec uses
// (run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, one pass each; compare with the counts printed here)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
// every lane reads 4 bytes at a random 4-byte-aligned place of a 4 GiB buffer
__global__ void calib_gather4(const uint32_t* buf, uint64_t words, uint32_t* out, int rounds) {
  uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; uint32_t acc = 0;
  for (int r = 0; r < rounds; r++) { s = mix(s + 0x9E3779B97F4A7C15ULL); acc += buf[s % words]; }
  out[(uint64_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// every lane writes 4 bytes at a random place
__global__ void calib_scatter4(uint32_t* buf, uint64_t words, int rounds) {
  uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int r = 0; r < rounds; r++) { s = mix(s + 0x9E3779B97F4A7C15ULL); buf[s % words] = (uint32_t)s; }
}
// every lane writes one whole, aligned 32-byte sector / 64-byte line at a random place
__global__ void calib_scatter32(uint4* buf, uint64_t sectors, int rounds) {
  uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int r = 0; r < rounds; r++) { s = mix(s + 0x9E3779B97F4A7C15ULL); uint4 v = make_uint4((uint32_t)s, 1, 2, 3); uint4* p = buf + 2 * (s % sectors); p[0] = v; p[1] = v; }
}
__global__ void calib_scatter64(uint4* buf, uint64_t lines, int rounds) {
  uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int r = 0; r < rounds; r++) { s = mix(s + 0x9E3779B97F4A7C15ULL); uint4 v = make_uint4((uint32_t)s, 1, 2, 3); uint4* p = buf + 4 * (s % lines); p[0] = v; p[1] = v; p[2] = v; p[3] = v; }
}
// every lane reads 8 bytes, lanes consecutive (the coalesced frame reads): 512 B per wave instruction
__global__ void calib_stream8(const uint64_t* buf, uint64_t n, uint64_t* out) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; uint64_t acc = 0;
  for (; i < n; i += (uint64_t)gridDim.x * blockDim.x) acc += buf[i];
  out[(uint64_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ void calib_streamwrite8(uint64_t* buf, uint64_t n) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (uint64_t)gridDim.x * blockDim.x) buf[i] = i;
}
int main() {
  const uint64_t bytes = 4ull << 30;
  void* b; void* o;
  if (hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&o, 64 << 20) != hipSuccess) return 1;
  hipMemset(b, 1, bytes);
  const int blocks = 4096, threads = 256, rounds = 256;
  const uint64_t nreq = (uint64_t)blocks * threads * rounds;
  hipEvent_t e[5]; for (auto& x : e) hipEventCreate(&x);
  hipEventRecord(e[0], 0);
  hipLaunchKernelGGL(calib_gather4, dim3(blocks), dim3(threads), 0, 0, (const uint32_t*)b, bytes / 4, (uint32_t*)o, rounds);
  hipEventRecord(e[1], 0);
  hipLaunchKernelGGL(calib_scatter4, dim3(blocks), dim3(threads), 0, 0, (uint32_t*)b, bytes / 4, rounds);
  hipEventRecord(e[2], 0);
  hipLaunchKernelGGL(calib_stream8, dim3(blocks), dim3(threads), 0, 0, (const uint64_t*)b, bytes / 8, (uint64_t*)o);
  hipEventRecord(e[3], 0);
  hipLaunchKernelGGL(calib_streamwrite8, dim3(blocks), dim3(threads), 0, 0, (uint64_t*)b, bytes / 8);
  hipEventRecord(e[4], 0);
  hipDeviceSynchronize();
  float ms[4]; for (int i = 0; i < 4; i++) hipEventElapsedTime(&ms[i], e[i], e[i + 1]);
  std::printf("rates: random 4 B reads %.1f G/s, random 4 B writes %.1f G/s, stream read %.0f GB/s, stream write %.0f GB/s\n",
              (double)blocks * threads * rounds / ms[0] / 1e6, (double)blocks * threads * rounds / ms[1] / 1e6, bytes / ms[2] / 1e6, bytes / ms[3] / 1e6);
  {
    hipEventRecord(e[0], 0);
    hipLaunchKernelGGL(calib_scatter32, dim3(blocks), dim3(threads), 0, 0, (uint4*)b, bytes / 32, rounds / 4);
    hipEventRecord(e[1], 0);
    hipLaunchKernelGGL(calib_scatter64, dim3(blocks), dim3(threads), 0, 0, (uint4*)b, bytes / 64, rounds / 4);
    hipEventRecord(e[2], 0);
    hipDeviceSynchronize();
    float m0, m1; hipEventElapsedTime(&m0, e[0], e[1]); hipEventElapsedTime(&m1, e[1], e[2]);
    std::printf("random whole 32 B sectors %.1f G/s, random whole 64 B lines %.1f G/s (4 GiB region)\n",
                (double)blocks * threads * (rounds / 4) / m0 / 1e6, (double)blocks * threads * (rounds / 4) / m1 / 1e6);
  }
  // the same random patterns inside regions that fit the 256 MiB Infinity Cache / one XCD's 4 MiB L2
  for (uint64_t region : {128ull << 20, 2ull << 20}) {
    hipEventRecord(e[0], 0);
    hipLaunchKernelGGL(calib_gather4, dim3(blocks), dim3(threads), 0, 0, (const uint32_t*)b, region / 4, (uint32_t*)o, rounds);
    hipEventRecord(e[1], 0);
    hipLaunchKernelGGL(calib_scatter4, dim3(blocks), dim3(threads), 0, 0, (uint32_t*)b, region / 4, rounds);
    hipEventRecord(e[2], 0);
    hipDeviceSynchronize();
    float m0, m1; hipEventElapsedTime(&m0, e[0], e[1]); hipEventElapsedTime(&m1, e[1], e[2]);
    std::printf("region %4llu MiB: random 4 B reads %.1f G/s, random 4 B writes %.1f G/s\n", (unsigned long long)(region >> 20),
                (double)blocks * threads * rounds / m0 / 1e6, (double)blocks * threads * rounds / m1 / 1e6);
  }
  std::printf("calib_gather4: %llu lane requests of 4 B (%.3f GB useful)\n", (unsigned long long)nreq, nreq * 4 / 1e9);
  std::printf("calib_scatter4: %llu lane stores of 4 B (%.3f GB useful)\n", (unsigned long long)nreq, nreq * 4 / 1e9);
  std::printf("calib_stream8 / calib_streamwrite8: %.3f GB each\n", bytes / 1e9);
  return 0;
}
