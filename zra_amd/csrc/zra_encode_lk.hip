// zra_amd — ENCODE stage 1, "link" formulation of dfast (zstd 1.4.9 levels 3-4) for frames of at most 64 KiB, bit-exact.
//
// Replaces the match-finding work of the reference's per-frame ZSTD_compress2 (zra.cpp:219,331) where zra_mf_dfast_kernel did it through
// hash tables in HBM. That kernel is bound by random fabric requests (97 k table / candidate reads and 78 k partial-line table writes per
// 64 KiB frame). Here no table exists:
//   * zra_lk_prepass_kernel (parse-independent, one 1024-thread workgroup per frame, everything in LDS): for EVERY position p the three
//     previous positions of its long-hash bucket and of its short-hash bucket (q1 > q2 > q3) and one bit per predecessor "same 8 / 4
//     bytes as p", streamed out coalesced (16 B per position);
//   * zra_lk_parse_kernel (one wave per frame): dfast's insert positions never decrease, so a table cell holds the most recent INSERTED
//     position of its bucket = the first chain predecessor that was inserted. The parse keeps two bitmaps "inserted into the long /
//     short table" in LDS (16 KiB per frame), reads the entries of a 64-position window coalesced, and never writes a table or
//     computes a hash.
// Exactness argument, window rules (pending predecessors, re-walks, deep chains) and a fuzz driver: tools/model/dfast_link_model.c, the
// CPU restatement of THIS file (sequence-identical to oracle/zo_encode.c on the bench corpus and on random inputs).
#include "zra_dev.h"
#include "zra_kernels.h"
#include "zra_encode_wave.h"

using namespace zra_dev;
using namespace zra_wave;

// Phase timing for bring-up (build with -DZRA_MF_PROFILE; never in the shipped library): s_memtime sums of every wave's parse
#ifdef ZRA_MF_PROFILE
__device__ unsigned long long zra_lk_prof[32];
#define LPROF_DECL u64 pt_[24]; for (int k_ = 0; k_ < 24; k_++) pt_[k_] = 0; u64 pl_ = __builtin_amdgcn_s_memtime();
#define LPROF(k) { __builtin_amdgcn_s_waitcnt(0); const u64 n_ = __builtin_amdgcn_s_memtime(); pt_[k] += n_ - pl_; pl_ = n_; }
#define LCNT(k) { pt_[k]++; }
#define LPROF_END { if (lane == 0) for (int k_ = 0; k_ < 24; k_++) atomicAdd(&zra_lk_prof[k_], pt_[k_]); }
extern "C" __attribute__((visibility("default"))) void ZraHipDebugReadLkProfile(unsigned long long* out32, int reset) {
  (void)hipMemcpyFromSymbol(out32, HIP_SYMBOL(zra_lk_prof), sizeof(unsigned long long) * 32);
  if (reset) { unsigned long long z[32] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(zra_lk_prof), z, sizeof(z)); }
}
#else
#define LPROF_DECL
#define LPROF(k)
#define LCNT(k)
#define LPROF_END
#endif

namespace {

__device__ __forceinline__ u32 lk_bucket_long(u64 v, u32 hlog) { return (u32)((v * 0xCF1BBCDCB7A56463ULL) >> (64 - hlog)); }
__device__ __forceinline__ u32 lk_bucket_short(u64 v, u32 clog, u32 mls) {
  if (mls <= 4) return ((u32)v * 2654435761u) >> (32 - clog);
  const u64 prime = mls == 5 ? 889523592379ULL : mls == 6 ? 227718039650203ULL : mls == 7 ? 58295818150454627ULL : 0xCF1BBCDCB7A56463ULL;
  const u32 sh = mls >= 8 ? 0u : 64 - 8 * mls;
  return (u32)(((v << sh) * prime) >> (64 - clog));
}

// ================================================================================================ pre-pass
struct PpLds {
  u8* big;          // 128 KiB: bucket heads (u16) / chain links (u16 per position) / a copy of the frame
  u8* lastOf;       // [1024][16] per bucket group of the chunk and 64-position block: the block's last position of the group (lane)
  u32* bmask;       // [2][512]   per bucket group, 16 bits: which blocks of the chunk hold a position of the group (two chunks alternate)
  u32* succ;        // [2048]     one bit per position: a later position of the frame shares its bucket
};

__device__ __forceinline__ void pp_wave_sync() {      // (a wave's LDS accesses execute in issue order: volatile accesses + a scheduling barrier)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// links of one bucket chain: lk[p] = the previous position with p's bucket (0: none), for p in [1, last].
// A chunk of 1024 consecutive positions goes through the head table at once, in four steps:
//   1. every position reads the head of its bucket (the same value for all positions of a bucket);
//   2. every position writes its chunk index into the head: one of a bucket's positions wins, its index names the GROUP;
//   3. inside a 64-position block (one wave) the positions of a group are chained in lane order (ballots; two positions of a bucket in
//      one block are rare outside runs and short periods); the group's first position of the block sets the block's bit in the group's
//      mask, its last one leaves its lane in lastOf[group][block];
//   4. a block's first position of a group links to the last position of the nearest earlier block of the group (mask + lastOf), or to
//      the head read in step 1; the group's last position of the chunk becomes the new head.
// On log-like data most positions of a chunk share their bucket with a position one line further (different blocks): steps 3-4 resolve
// all of them side by side; nothing is serial.
// lk == nullptr: only the two bit sets are wanted (bit i = the thread's position of chunk i): pred = the bucket has an earlier position,
// succ = it has a later one (the final head of the bucket is another position).
template <bool LONG>
__device__ void pp_links(const PpLds& S, const u8* src, u32 last, u32 log, u32 mls, u16* lk, u64& predBits, u64& succBits) {
  const u32 t = threadIdx.x, blk = t >> 6, lane = t & 63;
  volatile u16* H = (volatile u16*)S.big;
  volatile u8* lastOf = S.lastOf;
  const u32 nPass = log > 16 ? 1u << (log - 16) : 1u;       // 64 Ki heads fit; a 17-bit table takes two passes over the positions
  const u32 headBytes = 2u << (log > 16 ? 16 : log);
  // bit p of `succ`: a later position links to p (p is not the last position of its bucket)
  for (u32 i = t; i < 2048; i += ZRA_LK_PP_THREADS) S.succ[i] = 0;
  auto src_of = [&](u32 ci) -> u64 { const u32 p = 1 + ci * ZRA_LK_PP_THREADS + t; return p <= last ? ld64(src + p) : 0ull; };
#ifdef ZRA_MF_PROFILE
  u64 pq_[6] = {0, 0, 0, 0, 0, 0}; u64 pql_ = __builtin_amdgcn_s_memtime();
#define PPT(k) { const u64 n_ = __builtin_amdgcn_s_memtime(); pq_[k] += n_ - pql_; pql_ = n_; }
#else
#define PPT(k)
#endif
  for (u32 pass = 0; pass < nPass; pass++) {
    for (u32 i = t; i < headBytes / 16; i += ZRA_LK_PP_THREADS) ((uint4*)S.big)[i] = make_uint4(0, 0, 0, 0);
    S.bmask[t] = 0;                                          // (two chunks' masks, 16 bits per group: 1024 words)
    // the source bytes of the chunk after the next are requested before a chunk is worked on: a chunk is four barriers of LDS work,
    // and nothing else would hide the load
    u64 vA = src_of(0), vB = src_of(1);
    __syncthreads();
    u32 par = 0, ci = 0;
    for (u32 c0 = 1; c0 <= last; c0 += ZRA_LK_PP_THREADS, par ^= 1, ci++) {
      const u32 p = c0 + t;
      bool act = p <= last;
      const u64 v = vA; vA = vB; vB = src_of(ci + 2);
      u32 b = LONG ? lk_bucket_long(v, log) : lk_bucket_short(v, log, mls);
      if (nPass > 1) { act = act && (b >> 16) == pass; b &= 0xFFFFu; }
      u32* const BM = S.bmask + par * 512;
      u32 h0 = 0;
      if (act) h0 = H[b];
      if (t < 512) S.bmask[(par ^ 1) * 512 + t] = 0;   // the other chunk's masks: free since the barrier that ended it
      PPT(0)
      __syncthreads();
      if (act) H[b] = (u16)t;
      __syncthreads();
      PPT(1)
      const u32 w = act ? (u32)H[b] : 0xFFFFu;
      volatile u8* const slot = lastOf + (w & 1023u) * 16 + blk;
      if (act) *slot = (u8)lane;
      pp_wave_sync();
      const bool lost = act && *slot != (u8)lane;
      pp_wave_sync();
      if (lost) *slot = (u8)lane;
      pp_wave_sync();
      const bool shared = act && (lost || *slot != (u8)lane);
      bool first = act, lastIn = act; u32 predIn = 0;
      u64 rem = __ballot(shared);
      PPT(2)
      while (rem) {
#ifdef ZRA_MF_PROFILE
        pq_[5]++;
#endif
        const u32 l = (u32)__builtin_ctzll(rem);
        const u32 wl = bcast(w, l);
        const u64 same = __ballot(act && w == wl);
        if (act && w == wl) {
          const u64 before = same & ((1ull << lane) - 1ull), after = same >> lane >> 1;
          if (before) { first = false; predIn = 63u - (u32)__builtin_clzll(before); }
          lastIn = after == 0;
        }
        rem &= ~same;
      }
      pp_wave_sync();
      if (lastIn) *slot = (u8)lane;
      const u32 wsh = (w & 1) * 16;
      if (first) atomicOr(&BM[(w & 1023u) >> 1], (1u << blk) << wsh);
      PPT(3)
      __syncthreads();
      if (act) {
        const u32 m = (BM[w >> 1] >> wsh) & 0xFFFFu;
        u32 link;
        if (first) {
          const u32 lower = m & ((1u << blk) - 1u);
          if (lower) { const u32 pb = 31u - (u32)__builtin_clz(lower); link = c0 + pb * 64 + lastOf[w * 16 + pb]; }
          else link = h0;
        } else link = c0 + blk * 64 + predIn;
        if (lastIn && (m >> blk >> 1) == 0) H[b] = (u16)p;
        if (lk) lk[p] = (u16)link;
        if (link) { predBits |= 1ull << ci; atomicOr(&S.succ[link >> 5], 1u << (link & 31)); }
      }
      __syncthreads();
      PPT(4)
    }
  }
#ifdef ZRA_MF_PROFILE
  if (t == 0 && blockIdx.x == 0) { for (int k_ = 0; k_ < 6; k_++) atomicAdd(&zra_lk_prof[26 + k_], pq_[k_]); }
#endif
  {
    u32 cj = 0;
    for (u32 c0 = 1; c0 <= last; c0 += ZRA_LK_PP_THREADS, cj++) {
      const u32 p = c0 + t;
      if (p <= last && ((S.succ[p >> 5] >> (p & 31)) & 1)) succBits |= 1ull << cj;
    }
  }
  __syncthreads();
}

// the three nearest predecessors of every position: q1 = lk[p], q2 = lk[q1], q3 = lk[q2] with the links in LDS
__device__ void pp_prefix(const PpLds& S, u32 last, u32 nEnt, const u16* lk, u64* ent, u32 which) {
  const u32 t = threadIdx.x;
  const u32 words = ((last + 1) * 2 + 15) / 16;
  for (u32 i = t; i < words; i += ZRA_LK_PP_THREADS) ((uint4*)S.big)[i] = ((const uint4*)lk)[i];
  __syncthreads();
  const u16* L = (const u16*)S.big;
  for (u32 p = t; p < nEnt; p += ZRA_LK_PP_THREADS) {
    u64 e = 0;
    if (p >= 1 && p <= last) {
      const u32 q1 = L[p], q2 = q1 ? L[q1] : 0u, q3 = q2 ? L[q2] : 0u;
      e = (u64)q1 | ((u64)q2 << 16) | ((u64)q3 << 32);
    }
    ent[2 * (size_t)p + which] = e;
  }
  __syncthreads();
}

// 8 bytes at byte offset `off` of the LDS copy (any alignment): three aligned words, two funnel shifts
__device__ __forceinline__ u64 lds_ld64(const u32* w, u32 off) {
  const u32 a = off >> 2, sh = off & 3;
  const u32 w0 = w[a], w1 = w[a + 1], w2 = w[a + 2];
  const u32 lo = __builtin_amdgcn_alignbyte(w1, w0, sh), hi = __builtin_amdgcn_alignbyte(w2, w1, sh);
  return (u64)lo | ((u64)hi << 32);
}
__device__ __forceinline__ u32 lds_ld32(const u32* w, u32 off) {
  const u32 a = off >> 2, sh = off & 3;
  return __builtin_amdgcn_alignbyte(w[a + 1], w[a], sh);
}

// equal-content bits: predecessor k of p carries p's 8 bytes (long chain) / 4 bytes (short chain) — what the parse's candidate tests
// (MEM_read64 / MEM_read32 compares of ZSTD_compressBlock_doubleFast) would find, so it never reads a candidate to test it
__device__ void pp_flags(const PpLds& S, const u8* src, u32 n, u32 last, u64* ent) {
  const u32 t = threadIdx.x;
  u32* W = (u32*)S.big;
  const u32 words = (n + 3) / 4;
  for (u32 i = t; i < words + 3; i += ZRA_LK_PP_THREADS) {
    u32 v = 0;
    if (4 * i + 4 <= n) v = ld32(src + 4 * i);
    else for (u32 k = 0; k < 4; k++) if (4 * i + k < n) v |= (u32)src[4 * i + k] << (8 * k);
    W[i] = v;
  }
  __syncthreads();
  for (u32 p = 1 + t; p <= last; p += ZRA_LK_PP_THREADS) {
    u64 eL = ent[2 * (size_t)p], eS = ent[2 * (size_t)p + 1];
    const u64 v = lds_ld64(W, p);
#pragma unroll
    for (u32 k = 0; k < 3; k++) {
      const u32 qL = (u32)(eL >> (16 * k)) & 0xFFFFu, qS = (u32)(eS >> (16 * k)) & 0xFFFFu;
      if (qL && lds_ld64(W, qL) == v) eL |= 1ull << (48 + k);
      if (qS && lds_ld32(W, qS) == (u32)v) eS |= 1ull << (48 + k);
    }
    ent[2 * (size_t)p] = eL; ent[2 * (size_t)p + 1] = eS;
  }
  __syncthreads();
}

// patience of the waits between the two persistent kernels (s_memtime ticks of the 100 MHz constant clock: 3 s)
constexpr u64 LK_PATIENCE = 300000000ull;

}  // namespace

extern "C" __global__ void __launch_bounds__(ZRA_LK_PP_THREADS)
zra_lk_prepass_kernel(ZraEncArgs a, ZraLkArgs k) {
  extern __shared__ u32 ppLds[];
  PpLds S;
  S.big = (u8*)ppLds;
  S.lastOf = S.big + 131072;
  S.bmask = (u32*)(S.lastOf + 16384);
  S.succ = S.bmask + 1024;
  __shared__ u32 sFrame;
  u16* lkL = k.lkTmp ? k.lkTmp + (size_t)blockIdx.x * 2 * 65536 : nullptr;
  u16* lkS = lkL + 65536;
#define PPDBG(i, v) { if (k.dbg && threadIdx.x == 0 && blockIdx.x == 0) { k.dbg[i] = (v); __threadfence_system(); } }
  PPDBG(0, 1u)
  if (k.started && threadIdx.x == 0) atomicAdd(k.started, 1u);
  PPDBG(0, 2u)   // (the host launches the consumers once every workgroup of this kernel is resident)
  for (;;) {
    PPDBG(8, 1u)
    if (threadIdx.x == 0) sFrame = atomicAdd(k.ppQueue, 1u);
    PPDBG(8, 2u)
    __syncthreads();
    PPDBG(8, 3u)
    // (wave-uniform for the compiler too: with the frame index in a vector register the loop's exit is "divergent", and the
    //  structurised loop then lets thread 0 run the publish step apart from its wave — into the next iteration's barriers on its own)
    const u32 fi = rfl(sFrame);
    __syncthreads();
    PPDBG(1, fi + 1)
    if (fi >= k.count) return;
    const u32 f = k.first + fi;
    const u64 fstart = (u64)(a.firstFrame + f) * a.frameSize;
    const u64 remaining = a.inSize - fstart;
    const u32 n = (u32)(remaining < a.frameSize ? remaining : a.frameSize);
    const ZraEncParams& P = (n == a.frameSize) ? a.full : a.tail;
    const u32 slot = f % k.ringSlots;
    if (k.flagsOut && k.subDone && f >= k.ringSlots) {
      // flags ring of the table kernel: the slot is free once the sub-batch of the frame that used it before is through the match finder
      if (threadIdx.x == 0) {
        const u32 j = (f - k.ringSlots) / a.mfSubFrames;
        const u32 need = min(a.mfSubFrames, a.nFrames - j * a.mfSubFrames) - ((k.oddTail && (j + 1) * a.mfSubFrames >= a.nFrames) ? 1u : 0u);
        const u64 t0 = __builtin_amdgcn_s_memtime();
        while (__hip_atomic_load(&k.subDone[j], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < need) {
          __builtin_amdgcn_s_sleep(64);
          if (__builtin_amdgcn_s_memtime() - t0 > LK_PATIENCE) { atomicExch(k.fail, 1u); break; }
        }
      }
      __syncthreads();
    } else if (k.consumed && f >= k.ringSlots) {
      // the slot is free once the frame that used it before has been parsed
      if (threadIdx.x == 0) {
        const u64 t0 = __builtin_amdgcn_s_memtime();
        while (__hip_atomic_load(&k.consumed[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != f - k.ringSlots + 1) {
          __builtin_amdgcn_s_sleep(32);
          if (__builtin_amdgcn_s_memtime() - t0 > LK_PATIENCE) { atomicExch(k.fail, 1u); break; }
        }
      }
      __syncthreads();
    }
    if (k.flagsOut) {
      // flags for the table kernel (zra_mf_dfast_kernel): one byte per position, bit 0 / 1: the long bucket has an earlier / a later
      // position of the frame, bit 2 / 3: the short bucket. A table read without an earlier position finds the cleared cell, a table
      // write without a later position is never read: the kernel skips both (0x0F = no knowledge: behind the last hashed position)
      u8* fl = k.flagsOut + (size_t)slot * k.flagStride;
      if (P.strategy == 2 && n >= 9 && n <= ZRA_LK_MAX_FRAME) {
        const u8* src = a.in + fstart;
        const u32 last = n - 8;
        u64 pL = 0, sL = 0, pS = 0, sS = 0;
        PPDBG(2, f + 1)
        pp_links<true>(S, src, last, P.hashLog, P.minMatch, nullptr, pL, sL);
        PPDBG(3, f + 1)
        pp_links<false>(S, src, last, P.chainLog, P.minMatch, nullptr, pS, sS);
        PPDBG(4, f + 1)
        u32 ci = 0;
        for (u32 c0 = 1; c0 <= last; c0 += ZRA_LK_PP_THREADS, ci++) {
          const u32 p = c0 + threadIdx.x;
          if (p <= last) fl[p] = (u8)(((pL >> ci) & 1) | (((sL >> ci) & 1) << 1) | (((pS >> ci) & 1) << 2) | (((sS >> ci) & 1) << 3));
        }
        if (threadIdx.x == 0) fl[0] = 0x0F;
        for (u32 p = last + 1 + threadIdx.x; p < n; p += ZRA_LK_PP_THREADS) fl[p] = 0x0F;
      } else {
        for (u32 p = threadIdx.x; p < n; p += ZRA_LK_PP_THREADS) fl[p] = 0x0F;
      }
    } else if (P.strategy == 2 && n >= 9 && n <= ZRA_LK_MAX_FRAME) {
      u64* ent = k.ent + (size_t)slot * 2 * k.entPositions;
      const u8* src = a.in + fstart;
      const u32 last = n - 8;
      const u32 nEnt = (u32)min((u64)((n + 63) & ~63u), k.entPositions);
      u64 d0 = 0, d1 = 0;
      pp_links<true>(S, src, last, P.hashLog, P.minMatch, lkL, d0, d1);
      pp_links<false>(S, src, last, P.chainLog, P.minMatch, lkS, d0, d1);
      __threadfence_block();
      pp_prefix(S, last, nEnt, lkL, ent, 0);
      pp_prefix(S, last, nEnt, lkS, ent, 1);
      pp_flags(S, src, n, last, ent);
    }
    PPDBG(5, f + 1)
    if (k.ready) {
      __threadfence();
      __syncthreads();
      PPDBG(6, f + 1)
      if (threadIdx.x == 0) { *(volatile u32*)&k.ready[slot] = k.readyBase + f + 1; __threadfence(); }
      PPDBG(7, f + 1)
    }
  }
}

// ================================================================================================ parse
namespace {

// the first inserted predecessor beyond `from` (a position of the chain that is known not to be inserted), through memory: exact, one
// entry (three predecessors) per round trip. `which`: 0 long chain / bitmap, 1 short
__device__ u32 lk_walk_mem(const uint4* ent, u32 from, const u32* bm, u32 which) {
  u32 q = from;
  for (;;) {
    const uint4 e = ent[q];
    const u32 lo = which ? e.z : e.x, hi = which ? e.w : e.y;
    const u32 t1 = lo & 0xFFFFu, t2 = lo >> 16, t3 = hi & 0xFFFFu;
    if (!t1) return 0;
    if ((bm[t1 >> 5] >> (t1 & 31)) & 1) return t1;
    if (!t2) return 0;
    if ((bm[t2 >> 5] >> (t2 & 31)) & 1) return t2;
    if (!t3) return 0;
    if ((bm[t3 >> 5] >> (t3 & 31)) & 1) return t3;
    q = t3;
  }
}

// one chain of one lane: the first predecessor that is inserted (bitmap, positions below ipNow) or pending (at or beyond ipNow: the lane
// assumes it will be inserted and remembers which lane it depends on). Returns cand | hit << 16 | dep << 17 | deep << 24 (dep 64: none)
__device__ __forceinline__ u32 lk_walk3(u32 lo, u32 hi, const u32* bm, u32 ipNow, u32 base, u32 s) {
  const u32 q1 = lo & 0xFFFFu, q2 = lo >> 16, q3 = hi & 0xFFFFu;
  const u32 b1 = (bm[q1 >> 5] >> (q1 & 31)) & 1, b2 = (bm[q2 >> 5] >> (q2 & 31)) & 1, b3 = (bm[q3 >> 5] >> (q3 & 31)) & 1;
  bool p1 = q1 >= ipNow, p2 = q2 >= ipNow, p3 = q3 >= ipNow;
  u32 d1 = q1 - base, d2 = q2 - base, d3 = q3 - base;
  if (s != 1) {
    // between the lanes of a strided window nothing is inserted before the window ends
    const u32 r1 = d1 / s, r2 = d2 / s, r3 = d3 / s;
    const bool g1 = r1 * s == d1, g2 = r2 * s == d2, g3 = r3 * s == d3;
    const bool s1 = q1 && (p1 ? g1 : b1 != 0), s2 = q2 && (p2 ? g2 : b2 != 0), s3 = q3 && (p3 ? g3 : b3 != 0);
    d1 = r1; d2 = r2; d3 = r3;
    const u32 cand = s1 ? q1 : s2 ? q2 : s3 ? q3 : 0u;
    const u32 hit = s1 ? (hi >> 16) & 1 : s2 ? (hi >> 17) & 1 : s3 ? (hi >> 18) & 1 : 0u;
    const u32 dep = s1 ? (p1 ? d1 : 64u) : s2 ? (p2 ? d2 : 64u) : s3 ? (p3 ? d3 : 64u) : 64u;
    // (a pending predecessor off the lane grid is skipped, not "not inserted for good": the chain goes on behind it)
    const u32 deep = (!s1 && !s2 && !s3 && q3) ? 1u : 0u;
    return cand | (hit << 16) | (dep << 17) | (deep << 24);
  }
  const bool s1 = q1 && (p1 || b1), s2 = q2 && (p2 || b2), s3 = q3 && (p3 || b3);
  const u32 cand = s1 ? q1 : s2 ? q2 : s3 ? q3 : 0u;
  const u32 hit = s1 ? (hi >> 16) & 1 : s2 ? (hi >> 17) & 1 : s3 ? (hi >> 18) & 1 : 0u;
  const u32 dep = s1 ? (p1 ? d1 : 64u) : s2 ? (p2 ? d2 : 64u) : s3 ? (p3 ? d3 : 64u) : 64u;
  const u32 deep = (!s1 && !s2 && !s3 && q3) ? 1u : 0u;
  return cand | (hit << 16) | (dep << 17) | (deep << 24);
}

// One frame (one block, at most 64 KiB), one wave. src: the frame; ent: its entries; bmL / bmS: 2048 words each, zeroed.
__device__ u32 lk_parse(const u8* src, u32 be, const uint4* ent, u32* bmL, u32* bmS, u32* rep, u64* seqs, u32* nOut, int lane) {
  be = rfl(be);
  u32 o1 = rfl(rep[0]), o2 = rfl(rep[1]), saved = 0;
  u32 anchor = 0, nseq = 0;
  const u32 ilimit = be >= 8 ? be - 8 : 0;
  u32 ip = 1;                                          // ZSTD_compressBlock_doubleFast: ip += (ip == prefixLowest)
  { const u32 maxRep = 1; if (o2 > maxRep) { saved = o2; o2 = 0; } if (o1 > maxRep) { saved = o1; o1 = 0; } }
  u32 sqLo = 0, sqHi = 0;
  LPROF_DECL
  auto emit = [&](u32 ll, u32 ml, u32 offVal) {
    const u64 q = (u64)ll | ((u64)ml << 20) | ((u64)offVal << 40);
    sqLo = wlane(sqLo, (u32)q, nseq & 63); sqHi = wlane(sqHi, (u32)(q >> 32), nseq & 63);
    nseq++;
    if ((nseq & 63) == 0) seqs[nseq - 64 + (u32)lane] = (u64)sqLo | ((u64)sqHi << 32);
  };
  while (ip < ilimit) {
    // ---------------------------------------------------------------- window build
    const u32 run = ip - anchor;
    u32 s = 1, base, lo, hi;
    if (run < 256) { base = ip & ~63u; lo = ip - base; hi = min(min(64u, anchor + 256 - base), ilimit - base); }
    else { s = (run >> 8) + 1; base = ip; lo = 0; hi = min(64u, min((256 * s - run + s - 1) / s, (ilimit - ip + s - 1) / s)); }
    const u64 AM = (hi >= 64 ? ~0ull : (bit64(hi) - 1)) & (~0ull << lo);
    const bool active = lane_in(AM);
    const u32 p = base + (u32)lane * s;
    const uint4 e = active ? ent[p] : make_uint4(0, 0, 0, 0);
    const u64 v8 = active ? ld64(src + p) : 0;
    u32 repFor = o1;
    bool rv = active && o1 > 0 && p + 1 >= o1;
    u32 repVal = rv ? ld32(src + p + 1 - o1) : 0;
    // pending insertions of the window: for stride 1 the two bitmap words of the window's 64 positions (bits may already be there:
    // insertions behind the end of an earlier match), for a strided window one bit per lane
    u64 mkL = 0, mkS = 0;
    if (s == 1) {
      mkL = (u64)rfl(bmL[base >> 5]) | ((u64)rfl(bmL[(base >> 5) + 1]) << 32);
      mkS = (u64)rfl(bmS[base >> 5]) | ((u64)rfl(bmS[(base >> 5) + 1]) << 32);
    }
    auto flush = [&]() {
      if (s == 1) {
        if (lane < 4) { u32* w = (lane < 2 ? bmL : bmS) + (base >> 5) + (lane & 1); const u64 m = lane < 2 ? mkL : mkS; *w = (lane & 1) ? (u32)(m >> 32) : (u32)m; }
      } else {
        if (lane_in(mkL)) atomicOr(&bmL[p >> 5], 1u << (p & 31));
        if (lane_in(mkS)) atomicOr(&bmS[p >> 5], 1u << (p & 31));
      }
    };
    // insertion of position q (>= base) into the long and / or short table
    auto ins = [&](u32 q, bool doL, bool doS) {
      const u32 r = q - base;
      if (s == 1 && r < 64) { if (doL) mkL |= bit64(r); if (doS) mkS |= bit64(r); }
      else if (lane == 0) { if (doL) atomicOr(&bmL[q >> 5], 1u << (q & 31)); if (doS) atomicOr(&bmS[q >> 5], 1u << (q & 31)); }
    };
    LPROF(0) LCNT(12)
    u32 wL = 0, wS = 0;                                 // cand | hit << 16 | dep << 17 | deep << 24
    if (active) { wL = lk_walk3(e.x, e.y, bmL, ip, base, s); wS = lk_walk3(e.z, e.w, bmS, ip, base, s); }
    u64 LH = __ballot(active && ((wL >> 16) & 1)), SH = __ballot(active && ((wS >> 16) & 1));
    u64 DL = __ballot(active && (wL >> 24)), DS = __ballot(active && (wS >> 24));
    const u32 v8s = (u32)(v8 >> 8);
    u64 RH = __ballot(rv && repVal == v8s);
    u32 repOld = 0, repOldFor = 0xFFFFFFFFu; u64 ROV = 0;
    u64 RV = __ballot(rv);
    LPROF(1)
    // ---------------------------------------------------------------- resolve the window
    u32 cur = lo;
    bool done = false;
    for (;;) {
      if (repFor != o1) {                               // only after the repcode loop swapped the offsets
        rv = active && o1 > 0 && p + 1 >= o1;
        repVal = rv ? ld32(src + p + 1 - o1) : 0; repFor = o1;
        RH = __ballot(rv && repVal == v8s); RV = __ballot(rv);
      }
      const u64 live = AM & (~0ull << cur);
      const u64 hm = (RH | LH | SH | DL | DS) & live;
      if (!hm) { mkL |= live; mkS |= live; ip = base + hi * s; flush(); LPROF(2) break; }
      const u32 f = (u32)__builtin_ctzll(hm);
      const u32 top = base + f * s;
      const bool isRep = (RH >> f) & 1;
      if (!isRep && (((DL | DS) >> f) & 1)) {
        // a lane whose three predecessors are all not inserted has been reached: its chain is walked through memory
        const u64 vis = live & (bit64(f) - 1);
        mkL |= vis; mkS |= vis;
        flush();
        u32 nl = wL, ns = wS;
        if ((u32)lane == f) {
          if ((DL >> f) & 1) { const u32 c = lk_walk_mem(ent, e.y & 0xFFFFu, bmL, 0); nl = c | ((c && ld64(src + c) == v8) ? 1u << 16 : 0u) | (64u << 17); }
          if ((DS >> f) & 1) { const u32 c = lk_walk_mem(ent, e.w & 0xFFFFu, bmS, 1); ns = c | ((c && ld32(src + c) == (u32)v8) ? 1u << 16 : 0u) | (64u << 17); }
        }
        wL = nl; wS = ns;
        DL &= ~bit64(f); DS &= ~bit64(f);
        LH = (LH & ~bit64(f)) | (__ballot((u32)lane == f && ((wL >> 16) & 1)));
        SH = (SH & ~bit64(f)) | (__ballot((u32)lane == f && ((wS >> 16) & 1)));
        LPROF(3) LCNT(13)
        continue;
      }
      { const u64 vis = live & ((bit64(f) << 1) - 1); mkL |= vis; mkS |= vis; }
      ip = top;
      u32 m, known, offVal = 1;
      if (isRep) { ip = top + 1; m = ip - o1; known = 4; }
      else if ((LH >> f) & 1) { m = bcast(wL, f) & 0xFFFFu; known = 8; }
      else {
        // short hit: long-table probe at top+1. In the window this IS lane f+1's long test.
        bool hit3; u32 m3;
        if (s == 1 && f + 1 < hi && !((DL >> (f + 1)) & 1)) { hit3 = (LH >> (f + 1)) & 1; m3 = bcast(wL, f + 1) & 0xFFFFu; mkL |= bit64(f + 1); }
        else {
          flush();
          u32 c3 = 0, h3 = 0;
          if (lane == 0) {
            const uint4 e3 = ent[top + 1];
            const u32 w3 = lk_walk3(e3.x, e3.y, bmL, top + 1, 0, 1);      // everything below top+1 is decided by the bitmap
            c3 = w3 & 0xFFFFu; h3 = (w3 >> 16) & 1;
            if (w3 >> 24) { c3 = lk_walk_mem(ent, e3.y & 0xFFFFu, bmL, 0); h3 = (c3 && ld64(src + c3) == ld64(src + top + 1)) ? 1u : 0u; }
          }
          hit3 = rfl(h3) != 0; m3 = rfl(c3);
          ins(top + 1, true, false);
          LCNT(14)
        }
        if (hit3) { m = m3; ip = top + 1; known = 8; }
        else { m = bcast(wS, f) & 0xFFFFu; known = 4; }
      }
      LPROF(4)
      const u32 off = ip - m;
      // ---- issue together: forward compare (16 x 8 B), backward compare (64 x 1 B), rep gather for the next o1
      const u32 fa = ip + known + 8 * (u32)lane, fb = m + known + 8 * (u32)lane;
      const bool fv = (u32)lane < 16 && fa + 8 <= be;
      const u64 xa = fv ? ld64(src + fa) : 0, xb = fv ? ld64(src + fb) : 0;
      const u32 lim = isRep ? 0u : min(ip - anchor, m);
      u32 ya = 0, yb = 1;
      if (lim) { const bool bv = (u32)lane < lim; ya = bv ? src[ip - 1 - lane] : 0u; yb = bv ? src[m - 1 - lane] : 1u; }
      u32 rnext = 0; bool rvn = false;
      if (!isRep) { rvn = active && p + 1 >= off; rnext = rvn ? ld32(src + p + 1 - off) : 0; }
      u32 ml;
      {
        const u64 d = xa ^ xb;
        const u32 eq = d ? ((u32)__builtin_ctzll(d) >> 3) : 8;
        const u64 stop = __ballot(!fv || d != 0);
        const u32 l = stop ? (u32)__builtin_ctzll(stop) : 64u;
        const bool clean = stop && ((__ballot(fv) >> l) & 1);
        if (clean) ml = known + 8 * l + bcast(eq, l);
        else ml = known + wave_count_eq(src, ip + known, m + known, be, lane);
      }
      LPROF(5)
      if (!isRep) {
        u32 back = 0;
        if (lim) {
          const u64 bad = ~__ballot(ya == yb);
          back = bad ? (u32)__builtin_ctzll(bad) : wave_count_back(src, ip, m, anchor, lane);
        }
        ip -= back; ml += back;
        o2 = o1; o1 = off; offVal = off + 3;
        repOld = repVal; ROV = RV; repOldFor = repFor;
        repVal = rnext; repFor = off; RV = __ballot(rvn); RH = __ballot(rvn && rnext == v8s);
      }
      emit(ip - anchor, ml, offVal);
      ip += ml; anchor = ip;
      LPROF(6) LCNT(15)
      if (ip > ilimit) { flush(); done = true; break; }
      // ---- complementary insertions (top+2 into both tables, ip-2 long, ip-1 short) and the immediate repcode test
      ins(top + 2, true, true); ins(ip - 2, true, false); ins(ip - 1, false, true);
      const u32 relE = ip - base;
      u32 here = 0, there = 1;
      if (o2) {
        const bool inI = s == 1 && relE < hi;
        if (inI && repOldFor == o2 && ((ROV >> (relE - 1)) & 1)) { here = bcast((u32)v8, relE); there = bcast(repOld, relE - 1); }
        else {
          const u32 x = lane < 2 ? ld32(src + (lane == 0 ? ip : ip - o2)) : 0u;
          here = bcast(x, 0); there = bcast(x, 1);
          LCNT(16)
        }
      }
      if (here == there) {
        // immediate repcode sequences (rare)
        for (;;) {
          const u32 rl = wave_count_eq(src, ip + 4, ip + 4 - o2, be, lane) + 4;
          const u32 t = o2; o2 = o1; o1 = t;
          ins(ip, true, true);
          emit(0, rl, 1);
          ip += rl; anchor = ip;
          if (!(ip <= ilimit && o2 > 0)) break;
          if (rfl(ld32(src + ip)) != rfl(ld32(src + ip - o2))) break;
        }
      }
      LPROF(7)
      if (s != 1 || ip >= base + hi || ip >= ilimit) { flush(); LPROF(2) break; }
      cur = ip - base;
      // lanes behind the match whose pending predecessor was skipped (or went into the other table only): walked again
      {
        const u32 dL = (wL >> 17) & 127u, dS = (wS >> 17) & 127u;
        const bool bad = active && (u32)lane >= cur && ((dL < cur && !((mkL >> dL) & 1)) || (dS < cur && !((mkS >> dS) & 1)));
        const u64 BM = __ballot(bad);
        if (BM) {
          flush();
          if (bad) { wL = lk_walk3(e.x, e.y, bmL, ip, base, 1); wS = lk_walk3(e.z, e.w, bmS, ip, base, 1); }
          LH = (LH & ~BM) | __ballot(bad && ((wL >> 16) & 1)); SH = (SH & ~BM) | __ballot(bad && ((wS >> 16) & 1));
          DL = (DL & ~BM) | __ballot(bad && (wL >> 24)); DS = (DS & ~BM) | __ballot(bad && (wS >> 24));
          LCNT(17)
        }
      }
      LPROF(8)
    }
    if (done) break;
  }
  if (nseq & 63) { if ((u32)lane < (nseq & 63)) seqs[(nseq & ~63u) + (u32)lane] = (u64)sqLo | ((u64)sqHi << 32); }
  LPROF(9) LPROF_END
  rep[0] = o1 ? o1 : saved; rep[1] = o2 ? o2 : saved;
  *nOut = nseq;
  return be - anchor;
}

}  // namespace

// One workgroup of ZRA_LK_PARSE_WAVES independent waves per CU (its LDS, 16 KiB of bitmaps per wave, keeps a second one off the CU and
// leaves room for an entropy-stage workgroup); every wave pulls frames from the queue.
extern "C" __global__ void __launch_bounds__(ZRA_LK_PARSE_WAVES * 64)
zra_lk_parse_kernel(ZraEncArgs a, ZraLkArgs k) {
  extern __shared__ u32 lkLds[];
  const int lane = threadIdx.x & 63;
  const u32 wave = threadIdx.x >> 6;
  u32* bmL = lkLds + wave * 4096;                      // (blockDim.x / 64 waves, 16 KiB each)
  u32* bmS = bmL + 2048;
  for (;;) {
    u32 t = 0;
    if (lane == 0) t = atomicAdd(a.mfQueue, 1u);
    const u32 fi = rfl(t);
    if (fi >= k.count) return;
    const u32 f = k.first + fi;
    const u64 fstart = (u64)(a.firstFrame + f) * a.frameSize;
    const u64 remaining = a.inSize - fstart;
    const u32 n = (u32)(remaining < a.frameSize ? remaining : a.frameSize);
    const ZraEncParams& P = (n == a.frameSize) ? a.full : a.tail;
    const u32 slot = f % k.ringSlots;
    ZraEncFrameState* st = &a.state[f];
    ZraEncBlockOut* bo = &a.blockOut[f];
    const bool mine = P.strategy == 2;                 // a short last frame with other cparams is parsed by zra_mf_kernel (host launches it)
    if (mine) {
      if (k.ready) {
        u32 ok = 1;
        if (lane == 0) {
          const u64 t0 = __builtin_amdgcn_s_memtime();
          while (__hip_atomic_load(&k.ready[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != k.readyBase + f + 1) {
            __builtin_amdgcn_s_sleep(32);
            if (__builtin_amdgcn_s_memtime() - t0 > LK_PATIENCE) { atomicExch(k.fail, 1u); ok = 0; break; }
          }
        }
        (void)rfl(ok);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      if (lane == 0) { st->rep[0] = 1; st->rep[1] = 4; st->rep[2] = 8; st->nextToUpdate = 1; st->insEnd = 1; st->idxShift = 0; }
      if (n < 7) { if (lane == 0) { bo->nbSeq = 0; bo->lastLL = n; bo->skip = 1; } }
      else {
        for (u32 i = lane; i < 1024; i += 64) ((uint4*)bmL)[i] = make_uint4(0, 0, 0, 0);
        u32 rep[3] = {1, 4, 8};
        u32 nseq = 0;
        const uint4* ent = (const uint4*)(k.ent + (size_t)slot * 2 * k.entPositions);
#ifdef ZRA_MF_PROFILE
        const u64 kt0_ = __builtin_amdgcn_s_memtime();
#endif
        const u32 lastLL = lk_parse(a.in + fstart, n, ent, bmL, bmS, rep, a.seqs + (size_t)f * a.seqStride, &nseq, lane);
#ifdef ZRA_MF_PROFILE
        if (lane == 0) { atomicAdd(&zra_lk_prof[24], __builtin_amdgcn_s_memtime() - kt0_); atomicAdd(&zra_lk_prof[25], 1ull); }
#endif
        if (lane == 0) {
          bo->nbSeq = nseq; bo->lastLL = lastLL; bo->skip = 0;
          bo->rep[0] = rep[0]; bo->rep[1] = rep[1]; bo->rep[2] = rep[2];
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      if (lane == 0) {
        if (k.consumed) __hip_atomic_store(&k.consumed[slot], f + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        atomicAdd(&a.mfDone[f / a.mfSubFrames], 1u);
      }
    } else if (k.consumed && lane == 0) {
      // not parsed here, but the slot's turn passes all the same (the pre-pass skipped the frame too)
      if (k.ready) { const u64 t0 = __builtin_amdgcn_s_memtime(); while (__hip_atomic_load(&k.ready[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != k.readyBase + f + 1) { __builtin_amdgcn_s_sleep(32); if (__builtin_amdgcn_s_memtime() - t0 > LK_PATIENCE) { atomicExch(k.fail, 1u); break; } } }
      __hip_atomic_store(&k.consumed[slot], f + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
