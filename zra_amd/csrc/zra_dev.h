// zra_amd — MI355X-native chunked-zstd random-access engine.
// Device-side helpers shared by the gfx950 kernels (decode + encode). wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace zra_dev {

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t i32;

// zstd error codes surfaced through ZraStatus.zstd (reference zra.cpp:19-23, zstd_errors.h of 1.4.9)
enum : u32 {
  ZE_OK = 0, ZE_GENERIC = 1, ZE_PREFIX_UNKNOWN = 10, ZE_FRAMEPARAM_UNSUPPORTED = 14, ZE_WINDOW_TOO_LARGE = 16,
  ZE_CORRUPTION = 20, ZE_CHECKSUM_WRONG = 22, ZE_DICT_CORRUPTED = 30, ZE_DICT_WRONG = 32, ZE_PARAM_UNSUPPORTED = 40,
  ZE_DSTSIZE_TOOSMALL = 70, ZE_SRCSIZE_WRONG = 72,
};

constexpr int WAVE = 64;

__device__ __forceinline__ u32 hb32(u32 x) { return 31u - (u32)__builtin_clz(x); }  // x != 0
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// unaligned little-endian loads from global memory (gfx950 supports unaligned vector memory access)
typedef u16 __attribute__((aligned(1))) u16_u;
typedef u32 __attribute__((aligned(1))) u32_u;
typedef u64 __attribute__((aligned(1))) u64_u;
__device__ __forceinline__ u32 ld16(const u8* p) { return *(const u16_u*)p; }
__device__ __forceinline__ u32 ld24(const u8* p) { return (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16); }
__device__ __forceinline__ u32 ld32(const u8* p) { return *(const u32_u*)p; }
__device__ __forceinline__ u64 ld64(const u8* p) { return *(const u64_u*)p; }
// load 8 bytes at p without touching memory at or beyond `end` (missing bytes read as zero)
__device__ __forceinline__ u64 ld64_safe(const u8* p, const u8* end) {
  if (p + 8 <= end) return ld64(p);
  u64 v = 0;
  for (int i = 0; i < 8; i++) if (p + i < end) v |= (u64)p[i] << (8 * i);
  return v;
}
// LDS accesses typed as LDS (a generic pointer would compile to flat instructions): 8 bytes at any LDS address (one ds_read_b64, the
// hardware splits it), 16- and 8-byte stores to aligned addresses
typedef __attribute__((address_space(3))) u64_u zra_lds_u64u_t;
typedef u32 zra_v4u32_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) zra_v4u32_t zra_lds_v4u32_t;
typedef __attribute__((address_space(3))) u64 zra_lds_u64_t;
__device__ __forceinline__ void lds_st128(u8* p, uint4 q) { zra_v4u32_t v = {q.x, q.y, q.z, q.w}; *(zra_lds_v4u32_t*)p = v; }
__device__ __forceinline__ void lds_st64(u8* p, u64 v) { *(zra_lds_u64_t*)p = v; }
__device__ __forceinline__ void st32(u8* p, u32 v) { *(u32_u*)p = v; }
__device__ __forceinline__ void st64(u8* p, u64 v) { *(u64_u*)p = v; }
struct __attribute__((packed, aligned(1))) u128_u { u32 a, b, c, d; };
__device__ __forceinline__ void st128(u8* p, u32 a, u32 b, u32 c, u32 d) { u128_u v; v.a = a; v.b = b; v.c = c; v.d = d; *(u128_u*)p = v; }

// wave-level inclusive prefix sum of a u32 (64 lanes), DPP-free portable shuffle form
__device__ __forceinline__ u32 wave_incl_scan(u32 v) {
  int l = lane_id();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    u32 t = __shfl_up(v, d, 64);
    if (l >= d) v += t;
  }
  return v;
}
__device__ __forceinline__ u32 wave_sum(u32 v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}
__device__ __forceinline__ u32 wave_max(u32 v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) { u32 t = __shfl_xor(v, d, 64); v = t > v ? t : v; }
  return v;
}

// ---- backward bitstream reader (zstd "BIT_DStream" convention, RFC 8878 §4.1) over global memory.
// pos = number of unread bits. Bits below bit 0 read as zero; pos may go negative (= over-read).
struct BitR {
  const u8* base;   // first byte of the stream
  const u8* lim;    // one past the last readable byte of the enclosing buffer
  i32 pos;          // unread bits
  i32 wlo;          // bit index of window bit 0 (may be negative)
  u64 w;            // 64-bit window covering bits [wlo, wlo+64)
  __device__ __forceinline__ void reload() {
    i32 byteIdx = ((pos + 7) >> 3) - 8;
    if (byteIdx >= 0) { w = ld64_safe(base + byteIdx, lim); wlo = byteIdx * 8; }
    else {
      i32 sh = -byteIdx * 8;                      // 8..64 (64 only when pos <= 0)
      w = sh >= 64 ? 0 : (ld64_safe(base, lim) << sh);
      wlo = byteIdx * 8;
    }
  }
  // returns 0 ok / nonzero if the stream is malformed (empty or last byte zero)
  __device__ __forceinline__ int init(const u8* b, u32 n, const u8* bufLim) {
    base = b; lim = bufLim;
    if (n == 0) return 1;
    u32 last = b[n - 1];
    if (last == 0) return 1;
    pos = (i32)(n - 1) * 8 + (i32)hb32(last);
    reload();
    return 0;
  }
  __device__ __forceinline__ void ensure(int nb) { if (pos - wlo < nb) reload(); }
  // peek nb (<= 32) bits ending at pos, without consuming; requires ensure(nb) beforehand
  __device__ __forceinline__ u32 peek(int nb) const {
    return (u32)((w >> (pos - nb - wlo)) & ((1ull << nb) - 1));
  }
  __device__ __forceinline__ void skip(int nb) { pos -= nb; }
  __device__ __forceinline__ u32 read(int nb) {   // nb <= 32, may be 0
    if (nb == 0) return 0;
    ensure(nb);
    u32 v = peek(nb);
    pos -= nb;
    return v;
  }
};

// BitR for long streams that are read from end to start in one go (the Huffman literal streams): the same window at the same
// positions, but assembled from an 8-byte grid that is loaded one word AHEAD of the window. Which bytes a backward stream needs next
// does not depend on what is decoded — only the bit position inside them does — so the load latency (a DRAM round trip per four
// symbols with BitR, the whole cost of the Huffman stage) stays out of the decode chain.
struct BitRS {
  const u8* base; const u8* lim;
  i32 pos, wlo; u64 w;
  i32 gtop;                      // byte index of grid word g0; g1 = the 8 bytes below it, g2 the 8 below those (in flight)
  u64 g0, g1, g2;
  // 8 stream bytes at byte index idx; bytes below the stream's start read as zero (BitR::reload does the same with one shifted load)
  __device__ __forceinline__ u64 grid(i32 idx) const {
    if (idx >= 0) return ld64_safe(base + idx, lim);
    if (idx <= -8) return 0;
    return ld64_safe(base, lim) << (u32)(-idx * 8);
  }
  __device__ __forceinline__ void reload() {
    const i32 byteIdx = ((pos + 7) >> 3) - 8;
    while (byteIdx <= gtop - 8) { g0 = g1; g1 = g2; gtop -= 8; g2 = grid(gtop - 16); }
    const u32 d = (u32)(gtop - byteIdx);               // 0..7 bytes of g1 under the window
    w = d ? ((g0 << (8 * d)) | (g1 >> (64 - 8 * d))) : g0;
    wlo = byteIdx * 8;
  }
  __device__ __forceinline__ int init(const u8* b, u32 n, const u8* bufLim) {
    base = b; lim = bufLim;
    if (n == 0) return 1;
    u32 last = b[n - 1];
    if (last == 0) return 1;
    pos = (i32)(n - 1) * 8 + (i32)hb32(last);
    gtop = ((pos + 7) >> 3) - 8;
    g0 = grid(gtop); g1 = grid(gtop - 8); g2 = grid(gtop - 16);
    w = g0; wlo = gtop * 8;
    return 0;
  }
  // the same reader, positioned on `p` unread bits of a stream whose bytes [b, ...) are readable up to bufLim
  __device__ __forceinline__ void init_at(const u8* b, const u8* bufLim, i32 p) {
    base = b; lim = bufLim; pos = p;
    gtop = ((pos + 7) >> 3) - 8;
    g0 = grid(gtop); g1 = grid(gtop - 8); g2 = grid(gtop - 16);
    w = g0; wlo = gtop * 8;
  }
  __device__ __forceinline__ void ensure(int nb) { if (pos - wlo < nb) reload(); }
  __device__ __forceinline__ u32 peek(int nb) const { return (u32)((w >> (pos - nb - wlo)) & ((1ull << nb) - 1)); }
  __device__ __forceinline__ void skip(int nb) { pos -= nb; }
};


// ---- XXH64 (seed 0), zstd content checksum
constexpr u64 XP1 = 11400714785074694791ULL, XP2 = 14029467366897019727ULL, XP3 = 1609587929392839161ULL,
              XP4 = 9650029242287828579ULL, XP5 = 2870177450012600261ULL;
__device__ __forceinline__ u64 rotl64(u64 x, int r) { return (x << r) | (x >> (64 - r)); }
__device__ __forceinline__ u64 xxround(u64 acc, u64 x) { return rotl64(acc + x * XP2, 31) * XP1; }
__device__ __forceinline__ u64 xxmerge(u64 h, u64 v) { return (h ^ xxround(0, v)) * XP1 + XP4; }


// XXH64(seed 0) of n bytes at p, computed by a group of 4 consecutive lanes (j = lane&3): the four stripe
// accumulators are independent chains, one per lane. Result valid on j == 0. Shared with the encoder.
__device__ inline u64 zra_xxh64_quad(const u8* p, u32 n, int j) {
  u64 v = j == 0 ? XP1 + XP2 : j == 1 ? XP2 : j == 2 ? 0 : (0 - XP1);
  const u32 stripes = n >> 5;
  const u8* q = p + 8 * j;
  // (eight stripes' loads in flight per lane: the accumulator chain is ~60 cycles per stripe, a lone dependent load was ~700 — frames of
  //  256 KiB and more spent longer in their checksum than in their execute stage, round 6)
  u32 s = 0;
  for (; s + 8 <= stripes; s += 8) {
    u64 x[8];
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = ld64(q + 32 * (size_t)(s + k));
#pragma unroll
    for (int k = 0; k < 8; k++) v = xxround(v, x[k]);
  }
  for (; s < stripes; s++) v = xxround(v, ld64(q + 32 * (size_t)s));
  const int base = (threadIdx.x & 63) & ~3;
  u64 v1 = __shfl(v, base + 0, 64), v2 = __shfl(v, base + 1, 64), v3 = __shfl(v, base + 2, 64), v4 = __shfl(v, base + 3, 64);
  u64 h;
  if (n >= 32) {
    h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
    h = xxmerge(h, v1); h = xxmerge(h, v2); h = xxmerge(h, v3); h = xxmerge(h, v4);
  } else h = XP5;
  h += (u64)n;
  if (j == 0) {
    const u8* t = p + ((size_t)stripes << 5); u32 rem = n & 31;
    while (rem >= 8) { h = rotl64(h ^ xxround(0, ld64(t)), 27) * XP1 + XP4; t += 8; rem -= 8; }
    if (rem >= 4) { h = rotl64(h ^ ((u64)ld32(t) * XP1), 23) * XP2 + XP3; t += 4; rem -= 4; }
    while (rem) { h = rotl64(h ^ ((u64)*t * XP5), 11) * XP1; t++; rem--; }
    h ^= h >> 33; h *= XP2; h ^= h >> 29; h *= XP3; h ^= h >> 32;
  }
  return h;
}


}  // namespace zra_dev
