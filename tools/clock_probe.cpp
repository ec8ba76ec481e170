// bring-up: shader clock seen by a single-wave kernel (idle device vs right after a busy one), and the cost of dependent LDS / ALU steps
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
__global__ void probe(unsigned long long* out, int iters) {
  __shared__ unsigned tab[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) tab[i] = (i * 37 + 11) & 1023;
  __syncthreads();
  unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  unsigned x = threadIdx.x & 1;
  for (int i = 0; i < iters; i++) x = x * 3 + 1;                 // dependent ALU
  unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  unsigned y = x & 1023;
  for (int i = 0; i < iters; i++) y = tab[y];                    // dependent LDS
  unsigned long long c2 = __builtin_readcyclecounter(), r2 = wall_clock64();
  unsigned long long z = x | 1;
  for (int i = 0; i < iters; i++) z = (z << (z & 7)) ^ (z >> ((z >> 3) & 15)) ^ 0x9E3779B97F4A7C15ull;   // dependent 64-bit variable shifts
  unsigned long long c3 = __builtin_readcyclecounter(), r3 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; out[2] = c2 - c1; out[3] = r2 - r1; out[4] = c3 - c2; out[5] = r3 - r2; out[6] = x + y + z; }
}
__global__ void busy(float* p, int n) { float a = p[threadIdx.x]; for (int i = 0; i < n; i++) a = a * 1.0001f + 0.5f; p[blockIdx.x * blockDim.x + threadIdx.x] = a; }
int main() {
  unsigned long long* d; hipMalloc(&d, 64); float* f; hipMalloc(&f, 4 << 20);
  int wc = 0; hipDeviceGetAttribute(&wc, hipDeviceAttributeWallClockRate, 0);
  std::printf("wall clock rate %d kHz\n", wc);
  const int iters = 20000;
  auto run = [&](const char* what) {
    unsigned long long h[8];
    auto t = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, iters); hipDeviceSynchronize();
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
    hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    std::printf("%-28s host %.3f ms | ALU %.1f cyc/step, LDS %.1f cyc/step, shift64 %.1f cyc/step | shader clock %.0f MHz (cycles / wall ticks x rate)\n", what, ms,
                (double)h[0] / iters, (double)h[2] / iters, (double)h[4] / iters, (double)(h[0] + h[2] + h[4]) / (double)(h[1] + h[3] + h[5]) * wc / 1e3);
  };
  run("first launch");
  run("second launch");
  std::this_thread::sleep_for(std::chrono::milliseconds(1500));
  run("after 1.5 s idle");
  for (int k = 0; k < 5; k++) run("back to back");
  hipLaunchKernelGGL(busy, dim3(4096), dim3(256), 0, 0, f, 2000000); hipDeviceSynchronize();
  run("right after a busy kernel");
  run("next");
  return 0;
}
