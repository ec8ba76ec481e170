// bring-up: host<->device copy rates on the GPU box (pageable, registered, pinned; one and two directions at once)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %d at line %d\n", (int)e, __LINE__); return 1; } } while (0)
int main() {
  const size_t N = 1ull << 30;
  void *d0, *d1; CK(hipMalloc(&d0, N)); CK(hipMalloc(&d1, N));
  char* h = (char*)std::malloc(N); std::memset(h, 1, N);
  char* h2 = (char*)std::malloc(N); std::memset(h2, 2, N);
  hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  for (int r = 0; r < 2; r++) {
    double t = now(); CK(hipMemcpy(d0, h, N, hipMemcpyHostToDevice)); std::printf("pageable H2D   %.1f GB/s\n", N / (now() - t) / 1e9);
    t = now(); CK(hipMemcpy(h2, d1, N, hipMemcpyDeviceToHost)); std::printf("pageable D2H   %.1f GB/s\n", N / (now() - t) / 1e9);
  }
  { double t = now(); std::thread a([&] { (void)hipSetDevice(0); (void)hipMemcpy(d0, h, N, hipMemcpyHostToDevice); }); CK(hipMemcpy(h2, d1, N, hipMemcpyDeviceToHost)); a.join();
    std::printf("pageable both  %.1f GB/s each way\n", N / (now() - t) / 1e9); }
  { double t = now(); CK(hipHostRegister(h, N, hipHostRegisterDefault)); double t1 = now(); std::printf("hipHostRegister %.1f GB/s (%.1f ms / GiB)\n", N / (t1 - t) / 1e9, (t1 - t) * 1e3);
    t = now(); CK(hipMemcpyAsync(d0, h, N, hipMemcpyHostToDevice, s0)); CK(hipStreamSynchronize(s0)); std::printf("registered H2D %.1f GB/s\n", N / (now() - t) / 1e9);
    t = now(); CK(hipMemcpyAsync(h, d0, N, hipMemcpyDeviceToHost, s0)); CK(hipStreamSynchronize(s0)); std::printf("registered D2H %.1f GB/s\n", N / (now() - t) / 1e9);
    t = now(); CK(hipHostUnregister(h)); std::printf("unregister %.1f ms\n", (now() - t) * 1e3); }
  { double t = now(); CK(hipHostRegister(h, N, hipHostRegisterDefault)); std::printf("hipHostRegister again %.1f ms\n", (now() - t) * 1e3); CK(hipHostUnregister(h)); }
  { const size_t C = 64 << 20; double t = now(); for (size_t o = 0; o < N; o += C) { CK(hipHostRegister(h + o, C, hipHostRegisterDefault)); } std::printf("register in 64 MiB pieces %.1f ms\n", (now() - t) * 1e3);
    for (size_t o = 0; o < N; o += C) CK(hipHostUnregister(h + o)); }
  char *p0, *p1; CK(hipHostMalloc((void**)&p0, N, hipHostMallocDefault)); CK(hipHostMalloc((void**)&p1, N, hipHostMallocDefault));
  { double t = now(); std::memcpy(p0, h, N); std::printf("memcpy to pinned, 1 thread %.1f GB/s\n", N / (now() - t) / 1e9); }
  for (int nt : {2, 4, 8}) { double t = now(); std::vector<std::thread> th; for (int i = 0; i < nt; i++) th.emplace_back([&, i] { std::memcpy(p0 + N / nt * i, h + N / nt * i, N / nt); }); for (auto& x : th) x.join();
    std::printf("memcpy to pinned, %d threads %.1f GB/s\n", nt, N / (now() - t) / 1e9); }
  { double t = now(); CK(hipMemcpyAsync(d0, p0, N, hipMemcpyHostToDevice, s0)); CK(hipStreamSynchronize(s0)); std::printf("pinned H2D %.1f GB/s\n", N / (now() - t) / 1e9);
    t = now(); CK(hipMemcpyAsync(p1, d1, N, hipMemcpyDeviceToHost, s1)); CK(hipStreamSynchronize(s1)); std::printf("pinned D2H %.1f GB/s\n", N / (now() - t) / 1e9);
    t = now(); CK(hipMemcpyAsync(d0, p0, N, hipMemcpyHostToDevice, s0)); CK(hipMemcpyAsync(p1, d1, N, hipMemcpyDeviceToHost, s1)); CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1));
    std::printf("pinned both %.1f GB/s each way\n", N / (now() - t) / 1e9); }
  return 0;
}
