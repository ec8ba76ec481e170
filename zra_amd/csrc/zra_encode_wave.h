// zra_amd — wave-level helpers of the match-finder kernels (zra_encode_mf.hip). wave = 64 lanes.
#pragma once
#include "zra_dev.h"

namespace zra_wave {
using namespace zra_dev;

// LDS accesses that must really happen, in program order (slots other lanes of the wave write): volatile, and typed as LDS. A volatile
// access through a GENERIC pointer compiles to a flat load with "s_waitcnt vmcnt(0)" — the vector-memory path, and a wait for every
// global load in flight besides — instead of a ds_read (round 4: that was what the duplicate detection of both match finders waited for).
typedef __attribute__((address_space(3))) u32 lds_u32_t;
typedef __attribute__((address_space(3))) u8 lds_u8_t;
__device__ __forceinline__ u32 lds_read32(const u32* p) { return *(const volatile lds_u32_t*)p; }
typedef __attribute__((address_space(3))) u64 lds_u64_t;
__device__ __forceinline__ u64 lds_read64(const u64* p) { return *(const volatile lds_u64_t*)p; }
typedef u32 v4u32_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4u32_t lds_v4u32_t;
__device__ __forceinline__ void lds_write128(u8* p, uint4 q) { v4u32_t v = {q.x, q.y, q.z, q.w}; *(lds_v4u32_t*)p = v; }
typedef __attribute__((address_space(3))) u64_u lds_u64u_t;          // 8 bytes at any LDS address (one ds_read_b64: the hardware splits it)
__device__ __forceinline__ u32 lds_read8(const u8* p) { return *(const volatile lds_u8_t*)p; }
__device__ __forceinline__ void lds_write8(u8* p, u32 v) { *(volatile lds_u8_t*)p = (u8)v; }

__device__ __forceinline__ u32 rfl(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ u32 bcast(u32 v, u32 l) { return (u32)__builtin_amdgcn_readlane((int)v, (int)l); }

// common prefix of src[a..] and src[b..] (b < a), a limited to end; 64 lanes x 8 bytes per round trip
__device__ __forceinline__ u32 wave_count_eq(const u8* src, u32 a, u32 b, u32 end, int lane) {
  u32 total = 0;
  for (;;) {
    const u32 off = total + (u32)lane * 8;
    u32 eq;                                     // equal leading bytes in this lane's 8-byte window
    bool stop;
    if (a + off + 8 <= end) {
      const u64 d = ld64(src + a + off) ^ ld64(src + b + off);
      eq = d ? ((u32)__builtin_ctzll(d) >> 3) : 8;
      stop = d != 0;
    } else {
      eq = 0;
      while (a + off + eq < end && src[a + off + eq] == src[b + off + eq]) eq++;
      stop = true;                              // reaches the block end (or mismatches) inside this window
    }
    const u64 m = __ballot(stop);
    if (m) { const u32 l = (u32)__builtin_ctzll(m); return total + 8 * l + bcast(eq, l); }
    total += 512;
  }
}
// backward extension: number of k >= 0 with ip-1-k >= anchor, m-1-k >= 0 and equal bytes
__device__ __forceinline__ u32 wave_count_back(const u8* src, u32 ip, u32 m, u32 anchor, int lane) {
  const u32 lim = min(ip - anchor, m);
  u32 total = 0;
  for (;;) {
    const u32 k = total + (u32)lane;
    const bool ok = k < lim && src[ip - 1 - k] == src[m - 1 - k];
    const u64 bad = ~__ballot(ok);
    if (bad) return total + (u32)__builtin_ctzll(bad);
    total += 64;
  }
}

// lane l of `old` := val (val and l wave-uniform); a compare + select — v_writelane would need M0 for the lane select on gfx9
__device__ __forceinline__ u32 wlane(u32 old, u32 val, u32 l) { return (threadIdx.x & 63u) == l ? val : old; }
__device__ __forceinline__ bool lane_in(u64 mask) { return __builtin_amdgcn_inverse_ballot_w64(mask); }
__device__ __forceinline__ u64 bit64(u32 i) { return 1ull << i; }

}  // namespace zra_wave
