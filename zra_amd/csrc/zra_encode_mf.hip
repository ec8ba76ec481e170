// zra_amd — ENCODE stage 1 for gfx950: the match finders (sequence producers) of zstd 1.4.9, bit-exact.
//
// Replaces the match-finding 78-97 % of the reference's per-frame ZSTD_compress2 work (zra.cpp:219,331). Frames are independent; a
// frame's hash tables (384 KiB at level 3 @ 64 KiB, 3.3 MiB at level 9 @ 256 KiB) live in an HBM table slot, not in the 160 KiB LDS.
// Five kernels, one per way a strategy can be spread over a wave — each holds its own finder only, because a kernel pays the register
// budget of everything it can call (tests/test_kernel_budgets.py):
//   zra_mf_dfast_kernel  dfast (levels 3-4): one wave per frame, window-resolve parse (64 positions looked up at once, resolved in
//                        registers), persistent waves pulling frames from a queue
//   zra_mf_dfast_ls_kernel  the same parse over a copy of the frame in LDS: calls of a few hundred frames at most (latency mode)
//   zra_mf_hc_kernel     greedy / lazy / lazy2 (levels 5-10): one wave per frame, a window of 64 positions inserted and searched at
//                        once (one lane per chain, four candidates per round trip), the parse consumes the answers
//   zra_mf_fast_kernel   fast (levels 1-2, negative levels): lane = frame, up to 8 frames per wave
//   zra_mf_kernel        btlazy2, frames larger than the level's window (sliding-window rules), single odd tails: lane = frame
//   zra_mf_opt_kernel    the same plus btopt / btultra / btultra2 (zra_encode_opt.h)
// Rules restated from SURVEY.md Appendix A.4.3 (validated there against libzstd 1.4.9) and checked against oracle/zo_encode.c.
#include "zra_dev.h"
#include "zra_kernels.h"
#include "zra_encode_wave.h"

using namespace zra_dev;
using namespace zra_wave;

// Phase timing for bring-up (build with -DZRA_MF_PROFILE; never in the shipped library): per-phase s_memtime sums of lane 0
// of every wave, accumulated into zra_mf_prof[] (read with hipMemcpyFromSymbol through ZraHipDebugReadMfProfile).
#ifdef ZRA_MF_PROFILE
__device__ unsigned long long zra_mf_prof[24];
#define PROF_DECL u64 pt_[20]; for (int k_ = 0; k_ < 20; k_++) pt_[k_] = 0; u64 pl_ = __builtin_amdgcn_s_memtime();
#define PROF(k) { __builtin_amdgcn_s_waitcnt(0); const u64 n_ = __builtin_amdgcn_s_memtime(); pt_[k] += n_ - pl_; pl_ = n_; }
#define PROF_CNT(k) { pt_[k]++; }
#define PROF_END { if (lane == 0) for (int k_ = 0; k_ < 20; k_++) atomicAdd(&zra_mf_prof[k_], pt_[k_]); }
#elif defined(ZRA_MF_MARK)
// (-DZRA_MF_MARK with -S: the phase boundaries as comments in the assembly, for tools/isa_phases.py — never built into the library)
#define PROF_DECL
#define PROF(k) asm volatile("; ZMARK " #k);
#define PROF_CNT(k)
#define PROF_END
#else
#define PROF_DECL
#define PROF(k)
#define PROF_CNT(k)
#define PROF_END
#endif

namespace {

__device__ __forceinline__ u32 hash4(const u8* p, u32 bits) { return (ld32(p) * 2654435761u) >> (32 - bits); }
__device__ __forceinline__ u32 hash5(const u8* p, u32 bits) { return (u32)(((ld64(p) << 24) * 889523592379ULL) >> (64 - bits)); }
__device__ __forceinline__ u32 hash6(const u8* p, u32 bits) { return (u32)(((ld64(p) << 16) * 227718039650203ULL) >> (64 - bits)); }
__device__ __forceinline__ u32 hash7(const u8* p, u32 bits) { return (u32)(((ld64(p) << 8) * 58295818150454627ULL) >> (64 - bits)); }
__device__ __forceinline__ u32 hash8(const u8* p, u32 bits) { return (u32)((ld64(p) * 0xCF1BBCDCB7A56463ULL) >> (64 - bits)); }
__device__ __forceinline__ u32 hashN(const u8* p, u32 bits, u32 mls) {
  switch (mls) { case 5: return hash5(p, bits); case 6: return hash6(p, bits); case 7: return hash7(p, bits); case 8: return hash8(p, bits); default: return hash4(p, bits); }
}
// common-prefix length of src[a..] and src[b..] (b < a), a limited to `end`
__device__ __forceinline__ u32 count_eq(const u8* src, u32 a, u32 b, u32 end) {
  u32 l = 0;
  while (a + l + 8 <= end) {
    u64 d = ld64(src + a + l) ^ ld64(src + b + l);
    if (d) return l + ((u32)__builtin_ctzll(d) >> 3);
    l += 8;
  }
  while (a + l < end && src[a + l] == src[b + l]) l++;
  return l;
}

struct Emit {
  u64* seqs; u32 n;
  __device__ __forceinline__ void put(u32 ll, u32 ml, u32 offVal) { seqs[n++] = (u64)ll | ((u64)ml << 20) | ((u64)offVal << 40); }
};

// ZSTD_getLowestMatchIndex / ZSTD_getLowestPrefixIndex without a dictionary: the lowest index a match may have when the position with
// index `curr` is searched — the frame's first index, or curr - 2^windowLog once the frame is longer than the window (serial finders
// only: the wave-cooperative kernels are given frames that fit their window)
__device__ __forceinline__ u32 lowest_at(u32 curr, u32 windowLog, u32 idxShift = 0) {
  const u32 maxDist = 1u << windowLog, lowValid = 1 + idxShift;
  return (curr - lowValid > maxDist) ? curr - maxDist : lowValid;
}
__device__ __forceinline__ u32 mf_prologue(u32 bs, u32& o1, u32& o2, u32& saved, u32 windowLog = 31, u32 prefixStartPos = 0) {
  u32 ip = bs + (bs == prefixStartPos);
  const u32 maxRep = (ip + 1) - lowest_at(ip + 1, windowLog);
  saved = 0;
  if (o2 > maxRep) { saved = o2; o2 = 0; }
  if (o1 > maxRep) { saved = o1; o1 = 0; }
  return ip;
}

// ---- A.4.3 "fast" (levels 1-2)
__device__ u32 mf_fast(const ZraEncParams& P, u32* T, const u8* src, u32 bs, u32 be, u32* rep, Emit& E) {
  const u32 hlog = P.hashLog, mls = P.minMatch;
  const u32 step0 = P.targetLength + (P.targetLength == 0) + 1;
  u32 o1 = rep[0], o2 = rep[1], saved;
  u32 anchor = bs;
  const u32 ilimit = be >= 8 ? be - 8 : 0;            // a 7-byte first block: iend-8 lies before the start, nothing is searched
  const u32 psi = lowest_at(be + 1, P.windowLog);    // prefixStartIndex, from the block's end
  u32 ip0 = mf_prologue(bs, o1, o2, saved, P.windowLog, psi - 1), ip1 = ip0 + 1;
  while (ip1 < ilimit) {
    const u32 ip2 = ip0 + 2, top = ip0;
    const u32 h0 = hashN(src + ip0, hlog, mls), h1 = hashN(src + ip1, hlog, mls);
    const u32 m0 = T[h0], m1 = T[h1];
    u32 match, ml, offVal;
    T[h0] = ip0 + 1; T[h1] = ip1 + 1;
    if (o1 > 0 && ld32(src + ip2 - o1) == ld32(src + ip2)) {
      const u32 back = src[ip2 - 1] == src[ip2 - o1 - 1];
      ip0 = ip2 - back; match = ip2 - o1 - back; ml = 4 + back; offVal = 1;
    } else {
      if (m0 > psi && ld32(src + m0 - 1) == ld32(src + ip0)) match = m0 - 1;
      else if (m1 > psi && ld32(src + m1 - 1) == ld32(src + ip1)) { ip0 = ip1; match = m1 - 1; }
      else { const u32 st = ((ip0 - anchor) >> 7) + step0; ip0 += st; ip1 += st; continue; }
      o2 = o1; o1 = ip0 - match; offVal = o1 + 3; ml = 4;
      while (ip0 > anchor && match > psi - 1 && src[ip0 - 1] == src[match - 1]) { ip0--; match--; ml++; }
    }
    ml += count_eq(src, ip0 + ml, match + ml, be);
    E.put(ip0 - anchor, ml, offVal);
    ip0 += ml; anchor = ip0;
    if (ip0 <= ilimit) {
      T[hashN(src + top + 2, hlog, mls)] = top + 3;
      T[hashN(src + ip0 - 2, hlog, mls)] = ip0 - 1;
      if (o2 > 0) {
        while (ip0 <= ilimit && ld32(src + ip0) == ld32(src + ip0 - o2)) {
          const u32 rl = count_eq(src, ip0 + 4, ip0 + 4 - o2, be) + 4;
          const u32 t = o2; o2 = o1; o1 = t;
          T[hashN(src + ip0, hlog, mls)] = ip0 + 1;
          E.put(0, rl, 1);
          ip0 += rl; anchor = ip0;
        }
      }
    }
    ip1 = ip0 + 1;
  }
  rep[0] = o1 ? o1 : saved; rep[1] = o2 ? o2 : saved;
  return be - anchor;
}


// ---- A.4.3 "dfast", serial (oracle/zo_encode.c: mf_dfast). The window-resolve kernel below is the fast path; this one serves frames
// larger than the level's window, which that kernel is never given.
__device__ u32 mf_dfast_serial(const ZraEncParams& P, u32* HL, u32* HS, const u8* src, u32 bs, u32 be, u32* rep, Emit& E) {
  const u32 hlog = P.hashLog, clog = P.chainLog, mls = P.minMatch;
  u32 o1 = rep[0], o2 = rep[1], saved;
  u32 anchor = bs;
  const u32 ilimit = be >= 8 ? be - 8 : 0;
  const u32 psi = lowest_at(be + 1, P.windowLog);    // prefixLowestIndex, from the block's end
  u32 ip = mf_prologue(bs, o1, o2, saved, P.windowLog, psi - 1);
  while (ip < ilimit) {
    const u32 top = ip; u32 ml, offVal;
    const u32 hL = hash8(src + ip, hlog), hS = hashN(src + ip, clog, mls);
    const u32 curr = ip + 1, mL = HL[hL], mS = HS[hS];
    HL[hL] = curr; HS[hS] = curr;
    if (o1 > 0 && ld32(src + ip + 1 - o1) == ld32(src + ip + 1)) {
      ml = count_eq(src, ip + 5, ip + 5 - o1, be) + 4; ip++; offVal = 1;
    } else {
      u32 m;
      if (mL > psi && ld64(src + mL - 1) == ld64(src + ip)) {
        m = mL - 1; ml = count_eq(src, ip + 8, m + 8, be) + 8;
      } else if (mS > psi && ld32(src + mS - 1) == ld32(src + ip)) {
        const u32 h3 = hash8(src + ip + 1, hlog), m3 = HL[h3];
        HL[h3] = curr + 1;
        if (m3 > psi && ld64(src + m3 - 1) == ld64(src + ip + 1)) { m = m3 - 1; ip++; ml = count_eq(src, ip + 8, m + 8, be) + 8; }
        else { m = mS - 1; ml = count_eq(src, ip + 4, m + 4, be) + 4; }
      } else { ip += ((ip - anchor) >> 8) + 1; continue; }
      const u32 off = ip - m;
      while (ip > anchor && m > psi - 1 && src[ip - 1] == src[m - 1]) { ip--; m--; ml++; }
      o2 = o1; o1 = off; offVal = off + 3;
    }
    E.put(ip - anchor, ml, offVal);
    ip += ml; anchor = ip;
    if (ip <= ilimit) {
      const u32 q = top + 2;
      HL[hash8(src + q, hlog)] = q + 1;
      HL[hash8(src + ip - 2, hlog)] = ip - 1;
      HS[hashN(src + q, clog, mls)] = q + 1;
      HS[hashN(src + ip - 1, clog, mls)] = ip;
      while (ip <= ilimit && o2 > 0 && ld32(src + ip) == ld32(src + ip - o2)) {
        const u32 rl = count_eq(src, ip + 4, ip + 4 - o2, be) + 4;
        const u32 t = o2; o2 = o1; o1 = t;
        HS[hashN(src + ip, clog, mls)] = ip + 1;
        HL[hash8(src + ip, hlog)] = ip + 1;
        E.put(0, rl, 1);
        ip += rl; anchor = ip;
      }
    }
  }
  rep[0] = o1 ? o1 : saved; rep[1] = o2 ? o2 : saved;
  return be - anchor;
}

// ---- A.4.3 hash chain (greedy depth 0 / lazy 1 / lazy2 2) and, with `bt`, the binary tree with delayed updates of btlazy2
// (ZSTD_updateDUBT / ZSTD_insertDUBT1 / ZSTD_DUBT_findBestMatch of zstd_lazy.c; oracle/zo_encode.c: bt_search). Serial: one lane.
struct HC {
  u32* hashT; u32* chainT; u32 hlog, mls, cmask, chainSize, searchLog, nextToUpdate;
  u32 windowLog;
  bool bt;
  __device__ u32 search(const u8* src, u32 ip, u32 be, u32& offCode) {
    if (bt) return bt_search(src, ip, be, offCode);
    const u32 target = ip + 1;
    for (u32 idx = nextToUpdate; idx < target; idx++) {
      const u32 h = hashN(src + idx - 1, hlog, mls);
      chainT[idx & cmask] = hashT[h];
      hashT[h] = idx;
    }
    nextToUpdate = target;
    u32 mi = hashT[hashN(src + ip, hlog, mls)];
    const u32 curr = target, minChain = curr > chainSize ? curr - chainSize : 0;
    const u32 lowLimit = lowest_at(curr, windowLog);
    int attempts = 1 << searchLog;
    u32 ml = 3;
    offCode = 999999999u;
    for (; mi >= lowLimit && attempts > 0; attempts--) {
      const u32 m = mi - 1;
      u32 cur = 0;
      if (src[m + ml] == src[ip + ml]) cur = count_eq(src, ip, m, be);
      if (cur > ml) { ml = cur; offCode = curr - mi + 2; if (ip + cur == be) break; }
      if (mi <= minChain) break;
      mi = chainT[mi & cmask];
    }
    return ml;
  }
  // the tree lives in chainT as pairs {smaller, larger} at 2 * (index & btMask), btLog = chainLog - 1; an inserted but unsorted
  // position holds {previous head of its bucket, 1}
  __device__ void bt_insert1(const u8* src, u32 curr, u32 iend, u32 nbCompares, u32 btLow) {
    const u32 btMask = (chainSize >> 1) - 1;
    u32 commonSmaller = 0, commonLarger = 0;
    const u32 ipos = curr - 1;
    u32* smallerPtr = chainT + 2 * (curr & btMask);
    u32* largerPtr = smallerPtr + 1;
    u32 matchIndex = *smallerPtr;
    u32 dummy32;
    const u32 windowLow = lowest_at(curr, windowLog);
    while (nbCompares-- && matchIndex > windowLow) {
      u32* const nextPtr = chainT + 2 * (matchIndex & btMask);
      u32 ml = min(commonSmaller, commonLarger);
      const u32 m = matchIndex - 1;
      ml += count_eq(src, ipos + ml, m + ml, iend);
      if (ipos + ml == iend) break;
      if (src[m + ml] < src[ipos + ml]) {
        *smallerPtr = matchIndex; commonSmaller = ml;
        if (matchIndex <= btLow) { smallerPtr = &dummy32; break; }
        smallerPtr = nextPtr + 1; matchIndex = nextPtr[1];
      } else {
        *largerPtr = matchIndex; commonLarger = ml;
        if (matchIndex <= btLow) { largerPtr = &dummy32; break; }
        largerPtr = nextPtr; matchIndex = nextPtr[0];
      }
    }
    *smallerPtr = 0; *largerPtr = 0;
  }
  __device__ u32 bt_search(const u8* src, u32 ip, u32 be, u32& offCode) {
    const u32 btMask = (chainSize >> 1) - 1;
    const u32 curr = ip + 1;
    offCode = 999999999u;
    if (curr < nextToUpdate) return 0;                 // skipped area
    for (u32 idx = nextToUpdate; idx < curr; idx++) {  // ZSTD_updateDUBT
      const u32 h = hashN(src + idx - 1, hlog, mls);
      u32* p = chainT + 2 * (idx & btMask);
      p[0] = hashT[h]; p[1] = 1u;
      hashT[h] = idx;
    }
    nextToUpdate = curr;
    const u32 h = hashN(src + ip, hlog, mls);
    u32 matchIndex = hashT[h];
    const u32 btLow = btMask >= curr ? 0 : curr - btMask;
    const u32 windowLow = lowest_at(curr, windowLog);
    const u32 unsortLimit = max(btLow, windowLow);
    u32* nextCandidate = chainT + 2 * (matchIndex & btMask);
    u32* unsortedMark = nextCandidate + 1;
    u32 nbCompares = 1u << searchLog, nbCandidates = nbCompares, previousCandidate = 0;
    while (matchIndex > unsortLimit && *unsortedMark == 1u && nbCandidates > 1) {
      *unsortedMark = previousCandidate;               // the mark becomes a reversed chain
      previousCandidate = matchIndex;
      matchIndex = *nextCandidate;
      nextCandidate = chainT + 2 * (matchIndex & btMask);
      unsortedMark = nextCandidate + 1;
      nbCandidates--;
    }
    if (matchIndex > unsortLimit && *unsortedMark == 1u) { *nextCandidate = 0; *unsortedMark = 0; }
    matchIndex = previousCandidate;                    // batch sort the stacked candidates
    while (matchIndex) {
      const u32 nextIdx = chainT[2 * (matchIndex & btMask) + 1];
      bt_insert1(src, matchIndex, be, nbCandidates, unsortLimit);
      matchIndex = nextIdx;
      nbCandidates++;
    }
    u32 commonSmaller = 0, commonLarger = 0, bestLength = 0;
    u32* smallerPtr = chainT + 2 * (curr & btMask);
    u32* largerPtr = smallerPtr + 1;
    u32 matchEndIdx = curr + 8 + 1, dummy32;
    matchIndex = hashT[h];
    hashT[h] = curr;
    while (nbCompares-- && matchIndex > windowLow) {
      u32* const nextPtr = chainT + 2 * (matchIndex & btMask);
      u32 ml = min(commonSmaller, commonLarger);
      const u32 m = matchIndex - 1;
      ml += count_eq(src, ip + ml, m + ml, be);
      if (ml > bestLength) {
        if (ml > matchEndIdx - matchIndex) matchEndIdx = matchIndex + ml;
        if (4 * (int)(ml - bestLength) > (int)(hb32(curr - matchIndex + 1) - hb32(offCode + 1))) { bestLength = ml; offCode = 2 + curr - matchIndex; }
        if (ip + ml == be) break;
      }
      if (src[m + ml] < src[ip + ml]) {
        *smallerPtr = matchIndex; commonSmaller = ml;
        if (matchIndex <= btLow) { smallerPtr = &dummy32; break; }
        smallerPtr = nextPtr + 1; matchIndex = nextPtr[1];
      } else {
        *largerPtr = matchIndex; commonLarger = ml;
        if (matchIndex <= btLow) { largerPtr = &dummy32; break; }
        largerPtr = nextPtr; matchIndex = nextPtr[0];
      }
    }
    *smallerPtr = 0; *largerPtr = 0;
    nextToUpdate = matchEndIdx - 8;                    // skip repetitive patterns
    return bestLength;
  }
};

#include "zra_encode_opt.h"

__device__ u32 mf_lazy(HC& H, const u8* src, u32 bs, u32 be, u32* rep, Emit& E, int depth) {
  u32 o1 = rep[0], o2 = rep[1], saved;
  u32 anchor = bs;
  const u32 ilimit = be >= 8 ? be - 8 : 0;            // a 7-byte first block: iend-8 lies before the start, nothing is searched
  u32 ip = mf_prologue(bs, o1, o2, saved, H.windowLog, 0);
  while (ip < ilimit) {
    u32 ml = 0, start = ip + 1, off = 0; bool stored = false;
    if (o1 > 0 && ld32(src + ip + 1 - o1) == ld32(src + ip + 1)) {
      ml = count_eq(src, ip + 5, ip + 5 - o1, be) + 4;
      if (depth == 0) stored = true;
    }
    if (!stored) {
      u32 oc2; u32 m2 = H.search(src, ip, be, oc2);
      if (m2 > ml) { ml = m2; start = ip; off = oc2; }
      if (ml < 4) { ip += ((ip - anchor) >> 8) + 1; continue; }
      if (depth >= 1) {
        while (ip < ilimit) {
          ip++;
          if (off && o1 > 0 && ld32(src + ip) == ld32(src + ip - o1)) {
            const u32 mr = count_eq(src, ip + 4, ip + 4 - o1, be) + 4;
            const int g2 = (int)(mr * 3), g1 = (int)(ml * 3 - hb32(off + 1) + 1);
            if (mr >= 4 && g2 > g1) { ml = mr; off = 0; start = ip; }
          }
          {
            m2 = H.search(src, ip, be, oc2);
            const int g2 = (int)(m2 * 4 - hb32(oc2 + 1)), g1 = (int)(ml * 4 - hb32(off + 1) + 4);
            if (m2 >= 4 && g2 > g1) { ml = m2; off = oc2; start = ip; continue; }
          }
          if (depth == 2 && ip < ilimit) {
            ip++;
            if (off && o1 > 0 && ld32(src + ip) == ld32(src + ip - o1)) {
              const u32 mr = count_eq(src, ip + 4, ip + 4 - o1, be) + 4;
              const int g2 = (int)(mr * 4), g1 = (int)(ml * 4 - hb32(off + 1) + 1);
              if (mr >= 4 && g2 > g1) { ml = mr; off = 0; start = ip; }
            }
            {
              m2 = H.search(src, ip, be, oc2);
              const int g2 = (int)(m2 * 4 - hb32(oc2 + 1)), g1 = (int)(ml * 4 - hb32(off + 1) + 7);
              if (m2 >= 4 && g2 > g1) { ml = m2; off = oc2; start = ip; continue; }
            }
          }
          break;
        }
      }
      if (off) {
        const u32 ro = off - 2;
        while (start > anchor && start > ro && src[start - 1] == src[start - ro - 1]) { start--; ml++; }
        o2 = o1; o1 = ro;
      }
    }
    E.put(start - anchor, ml, off ? off + 1 : 1);
    anchor = ip = start + ml;
    while (ip <= ilimit && o2 > 0 && ld32(src + ip) == ld32(src + ip - o2)) {
      const u32 rl = count_eq(src, ip + 4, ip + 4 - o2, be) + 4;
      const u32 t = o2; o2 = o1; o1 = t;
      E.put(0, rl, 1);
      ip += rl; anchor = ip;
    }
  }
  rep[0] = o1 ? o1 : saved; rep[1] = o2 ? o2 : saved;
  return be - anchor;
}


// ================================================================================================
// Wave-cooperative dfast (A.4.3), bit-exact with the serial formulation above.
//
// The serial parse visits positions ip, ip+s, ip+2s, ... (s = 1 + literal-run/256) and at each one (a) looks two hash
// buckets up, (b) inserts the position, (c) tests three candidates; only a hit changes control flow. So up to 64 consecutive
// visit positions are evaluated at once, one per lane: all bucket reads, candidate reads and compares of a batch share ONE
// memory round trip each instead of one per position. Exactness: a lane's speculative lookup equals the serial one unless an
// EARLIER lane of the batch inserts into one of its buckets; an LDS scatter (atomicMax of epoch|reversed-lane) finds the first
// lane with such an earlier bucket-mate (conservatively, buckets folded to 10 bits) and the batch is cut there. Lanes up to
// the first hit commit their inserts (no two share a bucket, so store order is irrelevant); the hit lane's match is then
// handled wave-uniformly with wave-wide forward/backward length counts.
// ================================================================================================
// Window-resolve dfast. A window of up to 64 consecutive parse positions is looked up ONCE (one table-gather round trip + one
// tag-filtered candidate round trip), then the parse is resolved INSIDE the window: after a match the positions behind it are still
// in registers, so the next sequences of the window need no further trip to the tables. This is exact because no two lanes of a
// window share a bucket (exactly verified: the LDS scatter only flags suspects, flagged lanes are compared against all other lanes
// and the window is cut at the first duplicate), so inserts made while resolving the window can never change what another lane of
// the window would have read. Rep-offset tests depend on the parse state and are re-evaluated per sequence. Layout choices:
//   * every in-window insertion (visited positions, the ip+1 long probe, the complementary insertions) is "lane q-wip stores its
//     own (bucket, value)": one exec-masked store per table and sequence, masks built on the scalar unit;
//   * the ip+1 long probe of a short hit is lane f+1's own long-table test (already evaluated at window build);
//   * the immediate-repcode test after a match reuses the rep gather of the previous offset (o2 == old o1) — no load;
//   * out-of-window work of a sequence (insertions behind the match end, repcode test) is one 6-lane load + one vector hash;
//   * sequences are collected in a register (lane = index & 63) and stored 64 at a time;
//   * duplicate-bucket detection uses byte-wide LDS slots (2048 per table, no epochs, no atomics);
//   * tags use all bits above the index (ib = bits needed for position+1), so every dfast frame size is tagged.

// EB (round 6): the lowest EB bits of a cell's tag field hold the EPOCH of the frame that wrote the cell (0: none), so that the persistent
// kernel's waves clear their table slot once per 2^EB frames instead of once per frame (384 KiB of full-line writes per 64 KiB frame at
// level 3). A cell of another epoch can never equal a lane's tag: it reads as empty. (The epoch needs bits of its own: folded into the hash
// bits — XOR — a stale cell could pass the tag test with OTHER bytes behind it, and the bytes that are then checked are the current frame's.)
template <u32 MLS, u32 EB = 0>
struct DfHash {
  u32 shL, shS, shT, ib, tagMask;
  u32 epoch;
  __device__ __forceinline__ void init(u32 hlog, u32 clog, u32 ibits, u32 ep = 0) {
    ib = ibits; tagMask = ~((1u << ib) - 1u); shL = 64 - hlog; shS = (MLS == 4 ? 32 : 64) - clog; shT = shL - (32 - ib);
    epoch = ep;
  }
  // bucket indices and tags (tag = the hash bits just below the bucket bits, moved above the index bits)
  __device__ __forceinline__ void both(u64 v, u32& bL, u32& bS, u32& tL, u32& tS) const {
    const u64 pl = v * 0xCF1BBCDCB7A56463ULL;
    const u32 p4 = (u32)v * 2654435761u;
    bL = (u32)(pl >> shL);
    if (EB == 0) { tL = (u32)(pl >> shT) << ib; tS = p4 & tagMask; }
    else {
      // (one scalar operand per instruction: the epoch goes in below the shift)
      constexpr u32 keep = ~((1u << EB) - 1u);
      tL = (((u32)(pl >> shT) & keep) | epoch) << ib;
      tS = (((p4 >> ib) & keep) | epoch) << ib;
    }
    if (MLS == 5) bS = (u32)(((v << 24) * 889523592379ULL) >> shS);
    else if (MLS == 6) bS = (u32)(((v << 16) * 227718039650203ULL) >> shS);
    else if (MLS == 7) bS = (u32)(((v << 8) * 58295818150454627ULL) >> shS);
    else bS = p4 >> shS;
  }
};

struct __attribute__((packed, aligned(1))) quad_u { u32 x, y, z, w; };
__device__ __forceinline__ uint4 ld128(const u8* p) { const quad_u q = *(const quad_u*)p; return make_uint4(q.x, q.y, q.z, q.w); }

// LDS working set of one frame's parse: duplicate-bucket scratch + the bucket filter (1 bit per 2^sh buckets)
struct LeanLds {
  u8* dup; u32 dupSlots; u32* bmL; u32* bmS; u32 shL, shS;
  // round 5: a span of the frame's source bytes (and of the bucket flags that go with them) kept in LDS ahead of the parse: [span:
  // spanBytes + 16][flags: 16 per window of 64 positions, spanBytes / 64 + 1 windows]; spanBytes == 0: none
  u8* span; u8* spanFlg; u32 spanBytes;
  __device__ __forceinline__ void markL(u32 b) const { const u32 g = b >> shL; atomicOr(&bmL[g >> 5], 1u << (g & 31)); }
  __device__ __forceinline__ void markS(u32 b) const { const u32 g = b >> shS; atomicOr(&bmS[g >> 5], 1u << (g & 31)); }
};

// table accesses of the lean dfast parse: with -DZRA_MF_NT they carry the non-temporal hint (bring-up A/B: do the random table lines
// leave the L2 to the frame's source bytes?)
#ifdef ZRA_MF_NT
#define TLD(p) __builtin_nontemporal_load(p)
#define TST(p, v) __builtin_nontemporal_store((u32)(v), p)
#elif defined(ZRA_MF_SC1ST)
// bring-up A/B (round 6): table stores with the sc1 policy — the line leaves the XCD's L2 behind the write (MI355X_MICROARCH.md, "stores of
// each flavour") instead of staying as a single-use line that pushes the frame's source lines out
#ifdef ZRA_MF_SC1LD
#define TLD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)     /* ... and table loads past the L1 */
#else
#define TLD(p) (*(p))
#endif
#define TST(p, v) __hip_atomic_store((p), (u32)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#else
#define TLD(p) (*(p))
#define TST(p, v) (*(p) = (v))
#endif
// the flag nibble of one position as mf_dfast_lean reads it: bit 1 / 3 = later position in the long / short bucket (bits 0 / 2 always set)
__device__ __forceinline__ u32 df_flags_at(const u8* flg, u32 p) {
  const u8* const w = flg + (size_t)(p >> 6) * 16 + ((p & 63u) >> 3);
  const u32 b = p & 7u;
  return 0x05u | (((w[0] >> b) & 1u) << 1) | (((w[8] >> b) & 1u) << 3);
}
#define ZRA_DF_EPOCH_BITS 4u      /* epoch bits in the flag kernel's cells: a wave's table slot is cleared every 16th frame */
template <u32 MLS, bool FLAGS>
__device__ u32 mf_dfast_lean(const ZraEncParams& P, u32* HL, u32* HS, const u8* src, u32 bs, u32 be, u32* rep, u64* seqs, u32* nOut,
                             const LeanLds& W, int lane, u32 ib, const u8* flgIn, u32 epoch = 0) {
  const u8* const flg = FLAGS ? flgIn : nullptr;       // (the kernel without flags carries none of their code or registers)
  // flg (df_later_flags' masks, read through df_flags_at; nullptr: none): per position bit 1 / bit 3 = its long / short bucket has a later
  // position of the frame. An insertion without one is never looked up: the table write is skipped — half of the table's write requests.
  // (Bits 0 / 2, "has an earlier position", are always set: a lookup's table read is decided by the LDS filter alone.)
  DfHash<MLS, FLAGS ? ZRA_DF_EPOCH_BITS : 0u> H; H.init(P.hashLog, P.chainLog, ib, FLAGS ? rfl(epoch) : 0u);
  const u32 idxMask = ~H.tagMask;
  // block-level scalars arrive in VGPRs (vector loads of the frame state): pin them to SGPRs once so that the whole
  // parse state stays on the scalar unit instead of being dragged onto the VALU
  bs = rfl(bs); be = rfl(be);
  u32 o1 = rfl(rep[0]), o2 = rfl(rep[1]), saved;
  u32 anchor = bs, nseq = 0;
  const u32 ilimit = be >= 8 ? be - 8 : 0;            // a 7-byte first block: iend-8 lies before the start, nothing is searched
  u32 ip = mf_prologue(bs, o1, o2, saved);
  u32 sqLo = 0, sqHi = 0;                              // pending sequences, lane = index & 63
  // duplicate detection of a window (below): one 32-bit slot per bucket class and table, keyed (round << 6) | (63 - lane) and written with
  // an LDS max: the slot's winner is the LOWEST lane of the round, and an older round never wins. Cleared once per block.
  u32* const dL = (u32*)W.dup; u32* const dS = dL + W.dupSlots;
  const u32 dmask = W.dupSlots - 1;
  for (u32 i = (u32)lane; i < 2 * W.dupSlots; i += 64) dL[i] = 0;
  u32 dupRound = 0;
  u32* const bmL = W.bmL; u32* const bmS = W.bmS;
  PROF_DECL
  auto emit = [&](u32 ll, u32 ml, u32 offVal) {
    const u64 q = (u64)ll | ((u64)ml << 20) | ((u64)offVal << 40);
    sqLo = wlane(sqLo, (u32)q, nseq & 63); sqHi = wlane(sqHi, (u32)(q >> 32), nseq & 63);
    nseq++;
    if ((nseq & 63) == 0) seqs[nseq - 64 + (u32)lane] = (u64)sqLo | ((u64)sqHi << 32);
  };
  // scalar insertion of one position (rare paths only)
  auto insert_slow = [&](u32 pos, bool doL, bool doS) {
    if (lane == 0) {
      const u32 fx = flg ? df_flags_at(flg, pos) : 0x0Fu;
      u32 bl, bs_, tl, ts; H.both(ld64(src + pos), bl, bs_, tl, ts);
      if (doL && (fx & 2)) { TST(HL + bl, (pos + 1) | tl); W.markL(bl); }
      if (doS && (fx & 8)) { TST(HS + bs_, (pos + 1) | ts); W.markS(bs_); }
    }
  };
  // SOURCE SPAN (round 5). A window's first dependent memory round trip was the load of its own 64 + 8 source bytes (and of its flag
  // bytes): the address is the end of the previous window's last match. The parse moves forward ~40 bytes per window, so the wave keeps
  // the next spanBytes of the frame in LDS — one 16-byte load per lane, every ~(spanBytes - 72) bytes of progress, instead of one trip per
  // window (1,626 per 64 KiB frame of the bench corpus) — and a window reads its bytes with one ds_read. Windows in stride mode (s > 1:
  // behind 256 literals without a match) read from memory as before.
  const bool useSpan = FLAGS && W.spanBytes != 0;
  u32 sBase = 0, sEnd = 0;                              // the span holds positions [sBase, sEnd)
  auto span_fill = [&](u32 at) {
    // (start on a 16-byte boundary of the ADDRESS where the frame allows it: the loads are then whole 16-byte pieces of one line)
    const u32 mis = (u32)((uintptr_t)(src + at) & 15u);
    const u32 base = rfl(at - min(at, mis));
    const u32 len = min(W.spanBytes, be - base);
    const u32 o = 16u * (u32)lane;
    if (o + 16 <= len) lds_write128(W.span + o, ld128(src + base + o));
    else if (o < len) { for (u32 k = o; k < len; k++) lds_write8(W.span + k, src[base + k]); }      // (the frame's last, partial piece: the input may end there)
    if (flg) {
      // the flag bytes of the windows [base / 64, ...) the span touches, 16 per window, by the first lanes
      const u32 l2 = (u32)lane, w0 = base >> 6;
      if (l2 <= (W.spanBytes >> 6) && (w0 + l2) * 64u < be) lds_write128(W.spanFlg + 16u * l2, *(const uint4*)(flg + 16u * (size_t)(w0 + l2)));
    }
    __builtin_amdgcn_wave_barrier();
    sBase = base; sEnd = base + len;
  };
  while (ip < ilimit) {
    // ---------------------------------------------------------------- window build
    const u32 wip = ip, run = ip - anchor;
    u32 s = 1, nAct;
    if (run < 256) nAct = min(min(64u, 256u - run), ilimit - ip);
    else { s = (run >> 8) + 1; nAct = min(64u, min((256 * s - run + s - 1) / s, (ilimit - ip + s - 1) / s)); }
    bool active = (u32)lane < nAct;
    const u32 p = wip + (u32)lane * s;
    u64 v8; u32 bflags;
    if (useSpan && s == 1) {
      if (wip < sBase || wip + nAct + 7 > sEnd) span_fill(wip);
      const u32 r = p - sBase;
      v8 = active ? *(const lds_u64u_t*)(W.span + r) : 0;
      bflags = 0x0Fu;
      if (flg && active) {
        const u8* const w = W.spanFlg + ((p >> 6) - (sBase >> 6)) * 16u + ((p & 63u) >> 3);
        const u32 b = p & 7u;
        bflags = 0x05u | (((lds_read8(w) >> b) & 1u) << 1) | (((lds_read8(w + 8) >> b) & 1u) << 3);
      }
    } else {
      v8 = active ? ld64(src + p) : 0;
      bflags = (flg && active) ? df_flags_at(flg, p) : 0x0Fu;   // (NOT `fb`: that name is the match's forward-compare address further down)
    }
    // rep gather for the current o1 (independent of the tables: in flight together with them)
    u32 repFor = o1;
    bool rv = active && o1 > 0 && p + 1 >= o1;
    u32 repVal = rv ? ld32(src + p + 1 - o1) : 0;
    PROF(0) PROF_CNT(12)
    u32 bL, bS, tL, tS; H.both(v8, bL, bS, tL, tS);
    if (nAct > 1) {
      // Which is the first lane of the window that shares a bucket (of either table) with an EARLIER lane? The window ends before it.
      // Every lane enters its bucket's slot (low bucket bits) with an LDS max of (round << 6) | (63 - lane): the slot's winner is the
      // lowest lane of the round. A lane that finds another lane there compares BUCKETS with it (one cross-lane read): equal — it has
      // an earlier bucket-mate, it is marked and done; different — only the slot is shared, it tries again in the next round among
      // the lanes still unsettled. Winners are done. Exact: the first lane j with an earlier bucket-mate i meets i in the round i wins
      // (i cannot be marked itself — it would be an earlier end of the window — and j cannot win before i), and only real mates are
      // marked. ~2.5 rounds per window. The serial check of every lane that shared a SLOT, which stood here before, was 28 % of a lone
      // frame's time and most of the kernel's scalar instructions (round 4, profiles/r04_experiments.md §5).
      // (A wave's LDS accesses execute in issue order: volatile reads and a scheduling barrier are all the ordering it takes.)
      bool pL = active, pS = active;
      u64 mates = 0;
      for (;;) {
        dupRound++;
        const u32 key = (dupRound << 6) | (63u - (u32)lane);
        if (pL) atomicMax(&dL[bL & dmask], key);
        if (pS) atomicMax(&dS[bS & dmask], key);
        __builtin_amdgcn_wave_barrier();
        const u32 wl = pL ? 63u - (lds_read32(&dL[bL & dmask]) & 63u) : (u32)lane;
        const u32 ws = pS ? 63u - (lds_read32(&dS[bS & dmask]) & 63u) : (u32)lane;
        __builtin_amdgcn_wave_barrier();
        const u32 obL = (u32)__shfl((int)bL, (int)wl, 64), obS = (u32)__shfl((int)bS, (int)ws, 64);   // (by every lane: the lane read from must be executing)
        const bool lostL = wl != (u32)lane, lostS = ws != (u32)lane;
        const bool sameL = lostL && obL == bL, sameS = lostS && obS == bS;
        mates |= __ballot(sameL || sameS);
        pL = lostL && !sameL; pS = lostS && !sameS;
        PROF_CNT(17)
        if (!__ballot(pL || pS)) break;
      }
      if (mates) { nAct = (u32)__builtin_ctzll(mates); active = (u32)lane < nAct; }
    }
    PROF(1)
    // LDS filter: a clear bit means no position of this frame was ever inserted into the bucket (group) -> no table read
    const u32 gL = bL >> W.shL, gS = bS >> W.shS;
    const u32 wL = gL >> 5, wS = gS >> 5, qL = 1u << (gL & 31), qS = 1u << (gS & 31);
    u32 mL = 0, mS = 0;
    if (active) {
      const bool needL = (bflags & 1) && (bmL[wL] & qL) != 0, needS = (bflags & 4) && (bmS[wS] & qS) != 0;
      const u32 rL = needL ? TLD(HL + bL) : 0u, rS = needS ? TLD(HS + bS) : 0u;
      mL = ((rL & H.tagMask) == tL) ? (rL & idxMask) : 0u;
      mS = ((rS & H.tagMask) == tS) ? (rS & idxMask) : 0u;
    }
    PROF(2)
    // PROBABLE hits (round 5): a cell whose tag — 16 further bits of the position's hash — equals the lane's own is a candidate whose
    // bytes match with all but 2^-16 of the chance, so they are not fetched here. The candidate a sequence takes is checked by the
    // loads that measure its length anyway (they start at the match's first byte instead of behind the 8 / 4 known ones); one that
    // fails loses its bit and the window is resolved again from where it stood. A lane without a tag match has no match: exact.
    // Rounds 1-4 verified every candidate here: a third dependent memory round trip per window, 9-11 % of a frame's time.
    u64 LH = __ballot(active && mL > 1);
    u64 SH = __ballot(active && mS > 1);
    const u32 valL = (p + 1) | tL, valS = (p + 1) | tS;
    const u32 v8s = (u32)(v8 >> 8);
    const u64 AM = nAct >= 64 ? ~0ull : (bit64(nAct) - 1);
    u64 RH = __ballot(rv && repVal == v8s);
    u32 repOld = 0, repOldFor = 0xFFFFFFFFu; u64 ROV = 0;   // the gather for the previous o1 (== o2 after a non-rep match)
    u64 RV = __ballot(rv);
    PROF(3)
    // ---------------------------------------------------------------- resolve the window
    u32 cur = 0;
    for (;;) {
      if (repFor != o1) {                             // only after the repcode loop swapped the offsets
        rv = active && o1 > 0 && p + 1 >= o1;
        repVal = rv ? ld32(src + p + 1 - o1) : 0; repFor = o1;
        RH = __ballot(rv && repVal == v8s); RV = __ballot(rv);
      }
      const u64 live = AM & (~0ull << cur);
      const u64 hm = (RH | LH | SH) & live;
      if (!hm) {
        if (lane_in(live)) { if (bflags & 2) { TST(HL + bL, valL); atomicOr(&bmL[wL], qL); } if (bflags & 8) { TST(HS + bS, valS); atomicOr(&bmS[wS], qS); } }
        ip = wip + nAct * s; PROF(4)
        break;
      }
      const u32 f = (u32)__builtin_ctzll(hm);
      u64 mkL = live & ((bit64(f) << 1) - 1), mkS = mkL;          // visited positions cur..f
      const u32 top = wip + f * s;
      const bool isRep = (RH >> f) & 1, isLong = (LH >> f) & 1;
      ip = top;
      u32 m, known, offVal = 1;
      u32 vneed = 0, mS2 = 0; bool dual = false;         // vneed: bytes that must be equal from the candidate's first byte for a probable hit to be one
      if (isRep) { ip = top + 1; m = ip - o1; known = 4; }
      else if (isLong) { m = bcast(mL, f) - 1; known = 0; vneed = 8; }
      else if (s == 1 && f + 1 < nAct) {
        // short hit: long-table probe at ip+1 (A.4.3 case 3). In the window this IS lane f+1's long test. Both candidates are checked by
        // one trip: the probed long match on lanes 0-15, the short one on lanes 16-31 (the probe happens only behind a REAL short hit)
        mkL |= bit64(f + 1);
        if ((LH >> (f + 1)) & 1) { m = bcast(mL, f + 1) - 1; ip = top + 1; known = 0; vneed = 8; dual = true; mS2 = bcast(mS, f) - 1; }
        else { m = bcast(mS, f) - 1; known = 0; vneed = 4; }
      } else {
        bool hit3; u32 m3;
        {
          PROF_CNT(15)
          // (this path stores before it knows the match: the short hit is checked first, with a load of its own)
          if (rfl(ld32(src + bcast(mS, f) - 1)) != bcast((u32)v8, f)) { SH &= ~bit64(f); continue; }
          // the probed bucket is not covered by the window's no-duplicate guarantee: commit the visited positions first
          if (lane_in(mkL) && (bflags & 2)) { TST(HL + bL, valL); atomicOr(&bmL[wL], qL); }
          if (lane_in(mkS) && (bflags & 8)) { TST(HS + bS, valS); atomicOr(&bmS[wS], qS); }
          mkL = 0; mkS = 0;
          u32 m3v = 0; bool h3v = false;                // rare: done on lane 0's vector path (keeps the parse state scalar)
          if (lane == 0) {
            const u64 v9 = ld64(src + top + 1);
            const u32 f3 = flg ? df_flags_at(flg, top + 1) : 0x0Fu;
            u32 b3, bx, t3, tx; H.both(v9, b3, bx, t3, tx);
            const u32 r3 = (f3 & 1) ? TLD(HL + b3) : 0u;
            m3v = ((r3 & H.tagMask) == t3) ? (r3 & idxMask) : 0u;
            if (f3 & 2) { TST(HL + b3, (top + 2) | t3); W.markL(b3); }
            h3v = m3v > 1 && ld64(src + m3v - 1) == v9;
          }
          hit3 = __ballot(h3v) & 1; m3 = bcast(m3v, 0);
        }
        if (hit3) { m = m3 - 1; ip = top + 1; known = 8; }
        else { m = bcast(mS, f) - 1; known = 4; }
      }
      PROF(5)
#ifdef ZRA_MF_PAD_SALU
      // bring-up experiment (round 4, profiles/r04_experiments.md §5): N idle scalar instructions per sequence — does the kernel's time follow
      // its instruction count (issue-bound) or not (latency-bound)?
      { u32 sd_ = nseq;
#pragma unroll
        for (int k_ = 0; k_ < ZRA_MF_PAD_SALU; k_++) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sd_) :: "scc"); }
#endif
#ifdef ZRA_MF_PAD_VALU
      { u32 vd_ = (u32)lane;
#pragma unroll
        for (int k_ = 0; k_ < ZRA_MF_PAD_VALU; k_++) asm volatile("v_add_u32 %0, %0, 1" : "+v"(vd_)); }
#endif
      u32 off = ip - m;
      // ---- issue together: forward compare (16 lanes x 8 B; with a probed long match a second one for the short match on lanes 16-31),
      // backward compare (64 x 1 B), rep gather for the next o1
      const u32 l16 = (u32)lane & 15u;
      const bool sec = dual && (u32)lane >= 16 && (u32)lane < 32;
      const u32 fa = sec ? top + 8 * l16 : ip + known + 8 * (u32)lane, fb = sec ? mS2 + 8 * l16 : m + known + 8 * (u32)lane;
      // 16 lanes x 8 B: matches are ~10 bytes on average, and every 64-byte line of the match source is a likely DRAM request
      // (the kernel is bound by the DRAM request rate); longer matches take the wave_count_eq loop below
      const bool fv = ((u32)lane < 16 || sec) && fa + 8 <= be;
      const u64 xa = fv ? ld64(src + fa) : 0, xb = fv ? ld64(src + fb) : 0;
      const u32 lim = isRep ? 0u : min(ip - anchor, m);
      u32 ya = 0, yb = 1;
      if (lim) { const bool bv = (u32)lane < lim; ya = bv ? src[ip - 1 - lane] : 0u; yb = bv ? src[m - 1 - lane] : 1u; }
      u32 rnext = 0; bool rvn = false;
      if (!isRep) { rvn = active && p + 1 >= off; rnext = rvn ? ld32(src + p + 1 - off) : 0; }
      u32 ml;
      bool redo = false;                                 // the short match stands behind a probed long candidate that was none: lengths again
      {
        const u64 d = xa ^ xb;
        const u32 eq = d ? ((u32)__builtin_ctzll(d) >> 3) : 8;
        const u64 stopAll = __ballot(!fv || d != 0), fvAll = __ballot(fv);
        const u64 stop = stopAll & 0xFFFFull;            // the taken candidate: lanes 0-15 (no stop among them: more than 128 equal bytes)
        const u32 l = stop ? (u32)__builtin_ctzll(stop) : 16u;
        const bool clean = stop && ((fvAll >> l) & 1);              // first stopping lane compared a full 8-byte word
        if (clean) ml = known + 8 * l + bcast(eq, l);
        else ml = known + wave_count_eq(src, ip + known, m + known, be, lane);   // block end inside the compare, or > 128 equal bytes
        if (dual) {
          const u64 stopS = (stopAll >> 16) & 0xFFFFull;
          const u32 lS = stopS ? (u32)__builtin_ctzll(stopS) : 16u;
          const bool cleanS = stopS && ((fvAll >> (16 + lS)) & 1);
          const u32 mlS = cleanS ? 8 * lS + bcast(eq, 16 + lS) : wave_count_eq(src, top, mS2, be, lane);
          if (mlS < 4) { SH &= ~bit64(f); continue; }     // the short hit was none: the reference does not probe ip+1, nothing has happened yet
          if (ml < 8) { m = mS2; ip = top; ml = mlS; off = ip - m; redo = true; }   // the probe misses (its insertion stays): the short match
        } else if (ml < vneed) {                          // a tag without its bytes (2^-16): not a hit
          if (isLong) LH &= ~bit64(f); else SH &= ~bit64(f);
          continue;
        }
      }
      PROF(6)
      if (!isRep) {
        u32 back = 0;
        if (redo) {
          back = wave_count_back(src, ip, m, anchor, lane);
          rvn = active && p + 1 >= off; rnext = rvn ? ld32(src + p + 1 - off) : 0;
        } else if (lim) {
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
      PROF(7) PROF_CNT(13)
      if (ip > ilimit) {
        if (lane_in(mkL) && (bflags & 2)) { TST(HL + bL, valL); atomicOr(&bmL[wL], qL); }
        if (lane_in(mkS) && (bflags & 8)) { TST(HS + bS, valS); atomicOr(&bmS[wS], qS); }
        break;
      }
      // ---- complementary insertions (top+2 into both tables, then ip-2 long / ip-1 short) and the immediate repcode test
      const u32 relE = ip - wip;                       // >= 4
      const bool in2 = s == 1 && f + 2 < nAct;          // top+2 in the window
      const bool inE = s == 1 && relE - 1 < nAct;       // ip-1 (and ip-2) in the window
      const bool inI = s == 1 && relE < nAct;           // ip itself in the window
      if (in2) { mkL |= bit64(f + 2); mkS |= bit64(f + 2); }
      if (s == 1 && relE - 2 < nAct) mkL |= bit64(relE - 2);
      if (inE) mkS |= bit64(relE - 1);
      if (lane_in(mkL) && (bflags & 2)) { TST(HL + bL, valL); atomicOr(&bmL[wL], qL); }
      if (lane_in(mkS) && (bflags & 8)) { TST(HS + bS, valS); atomicOr(&bmS[wS], qS); }
      u32 here = 0, there = 1;
      const bool thereIn = inI && repOldFor == o2 && ((ROV >> (relE - 1)) & 1);   // src[ip - o2] == old rep gather of lane ip-1-wip
      if (in2 && inE && (o2 == 0 || thereIn)) {
        if (o2) { here = bcast((u32)v8, relE); there = bcast(repOld, relE - 1); }
      } else {
        // lanes 0..3: insertions (pos, table) = (top+2,L) (top+2,S) (ip-2,L) (ip-1,S); lane 4: src[ip]; lane 5: src[ip-o2]
        PROF_CNT(14)
        const u32 ipos = lane < 2 ? top + 2 : lane == 2 ? ip - 2 : lane == 3 ? ip - 1 : lane == 4 ? ip : ip - o2;
        const u64 x = lane < 6 ? ld64(src + ipos) : 0;
        const u32 fx = (flg && lane < 4) ? df_flags_at(flg, ipos) : 0x0Fu;
        const bool wr = (lane & 1) ? (fx & 8) != 0 : (fx & 2) != 0;           // the position's bucket has a later position: the cell will be read
        u32 xbL, xbS, xtL, xtS; H.both(x, xbL, xbS, xtL, xtS);
        u32* const tp = (lane & 1) ? HS + xbS : HL + xbL;
        const u32 tv = (ipos + 1) | ((lane & 1) ? xtS : xtL);
        if (lane < 4 && wr) { if (lane & 1) W.markS(xbS); else W.markL(xbL); }  // (a superset of the stores below: harmless)
        if (!in2 && lane < 2 && wr) TST(tp, tv);                              // top+2 first ...
        asm volatile("" ::: "memory");                                     // two instructions: lanes 0/2 (1/3) may hit the same bucket
        const u64 later = (s == 1 && relE - 2 < nAct ? 0ull : 4ull) | (inE ? 0ull : 8ull);
        if (lane_in(later) && wr) TST(tp, tv);                                // ... then ip-2 / ip-1 (same-bucket order per table)
        here = bcast((u32)x, 4); there = o2 ? bcast((u32)x, 5) : here + 1;
        if (o2 == 0) { here = 0; there = 1; }
      }
      PROF(8)
      if (here == there) {
        // immediate repcode sequences (rare): scalar walk
        for (;;) {
          PROF_CNT(16)
          const u32 rl = wave_count_eq(src, ip + 4, ip + 4 - o2, be, lane) + 4;
          const u32 t = o2; o2 = o1; o1 = t;
          insert_slow(ip, true, true);
          emit(0, rl, 1);
          ip += rl; anchor = ip;
          if (!(ip <= ilimit && o2 > 0)) break;
          if (rfl(ld32(src + ip)) != rfl(ld32(src + ip - o2))) break;
        }
      }
      PROF(9)
      if (s != 1 || ip >= wip + nAct || ip >= ilimit) break;    // left the window: build the next one at ip
      cur = ip - wip;
    }
  }
  if (nseq & 63) { if ((u32)lane < (nseq & 63)) seqs[(nseq & ~63u) + (u32)lane] = (u64)sqLo | ((u64)sqHi << 32); }
  PROF(10) PROF_END
  rep[0] = o1 ? o1 : saved; rep[1] = o2 ? o2 : saved;
  *nOut = nseq;
  return be - anchor;
}



// ================================================================================================
// Wave-cooperative hash chain (greedy / lazy / lazy2, levels 5-10), bit-exact with mf_lazy + HC::search above.
//
// What makes the serial formulation slow on this machine is one dependent memory round trip after the other: every insertion, every
// chain step, every candidate compare. Two facts about the reference remove most of them:
//   * ZSTD_insertAndFindFirstIndex inserts EVERY position below the search position, in order, whatever the parse decided; so the
//     tables at the moment position p is searched are a function of p alone, and the head the search starts from is exactly what
//     p's own chain slot receives when p is inserted (chainT[(p+1) & mask] = hashT[h(p)] at that time).
//   * ZSTD_HcFindBestMatch only reads: the answer for p does not depend on whether later positions are already in the tables.
// So a window of 64 consecutive positions is inserted at once (one table gather + two scatters; lanes that share a bucket are linked
// in lane order), all 64 are searched at once (each lane walks its own chain: the round trips of 64 searches overlap), and the parse
// then consumes the answers from registers. Repcode tests and match extensions depend on the parse and stay serial, but use the
// wave-wide counters. Only what the reference actually searched moves nextToUpdate; if the next block's "limited update after a
// very long match" skips the (at most 66) positions inserted ahead of it at the end of a block, hcw_undo takes them out again.
#ifdef ZRA_MF_PROFILE
#define HPROF(k) { __builtin_amdgcn_s_waitcnt(0); const u64 n_ = __builtin_amdgcn_s_memtime(); H.pt[k] += n_ - H.pl; H.pl = n_; }
#define HCNT(k, v) { H.pt[k] += (v); }
#else
#define HPROF(k)
#define HCNT(k, v)
#endif
struct HCW {
#ifdef ZRA_MF_PROFILE
  u64 pt[12]; u64 pl;
#endif
  // chainT (round 4): FOUR links per chain slot {x, y, z, w} — what the slot's index linked to when it was inserted (x), what THAT index
  // linked to (y), and so on: the next four candidates of the chain. A search step is one dependent round trip; with four links in hand
  // it tests four candidates per trip. A later link is exact whenever it is used: a slot keeps its links from its index's insertion
  // until index + chainSize overwrites it, and the search stops behind a candidate at or below minChain (the only candidates whose slot
  // can be gone) before it would follow it. One 16-byte load costs the address unit what an 8-byte one does.
  u32* hashT; uint4* chainT; u32 hlog, mls, cmask, chainSize, searchLog;
  u32 insEnd;            // first index (position + 1) not inserted yet
  u32 ntuRef;            // the reference's nextToUpdate
  u32 w;                 // window: answers for positions [w, w + 64) are in rml / roff of lane p - w
  u32 rml, roff;
#ifdef ZRA_MF_PROFILE
  u32 rsteps;            // chain steps this lane walked for the window
#endif
  bool haveWin;
  // chain tables smaller than the frame (chainLog < windowLog): inserting index i overwrites the link of index i - chainSize. The
  // reference never follows that link once i is inserted (i - chainSize is below its minChain by then), but a window is inserted
  // AHEAD of the positions searched in it, so the links it overwrote (at most 66, ring of 128 by index) are kept in LDS
  uint4* oldLink;
  u8* dup;               // 1024 byte slots: which lanes of an insert step may share a bucket (hcw_insert; accessed as volatile LDS)
};

__device__ __forceinline__ void hcw_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// takes the indices [from, H.insEnd) out of the tables again, newest first: the head of an index's bucket goes back to what the index
// linked to, its chain slot gets back what it held before (oldLink ring). One lane; at most 66 indices, at most once per block.
__device__ void hcw_undo(HCW& H, const u8* src, u32 from, int lane) {
  if (lane == 0) {
    for (u32 idx = H.insEnd; idx-- > from;) {
      const u32 h = hashN(src + idx - 1, H.hlog, H.mls);
      H.hashT[h] = H.chainT[idx & H.cmask].x;
      H.chainT[idx & H.cmask] = H.oldLink[idx & 127u];
    }
  }
  H.insEnd = from;
  hcw_sync();
}

// inserts indices [H.insEnd, endIdx) in order, 64 per step
__device__ void hcw_insert(HCW& H, const u8* src, u32 endIdx, int lane) {
  while (H.insEnd < endIdx) {
    HCNT(6, 1)
    const u32 idx = H.insEnd + (u32)lane;
    const bool act = idx < endIdx;
    u32 h = 0, link = 0;
    if (act) { h = hashN(src + idx - 1, H.hlog, H.mls); link = H.hashT[h]; }
    // lanes with the same bucket: each links to the nearest earlier one, the last one becomes the head. Most lanes share their bucket
    // with nobody (43 distinct buckets per step on average): two rounds through 1024 byte slots in LDS (slot = low hash bits, value =
    // lane) find every lane of a shared bucket — the lanes that lose a slot in round one, and in round two the winner they lost to —
    // plus a few that only share a slot; the exact grouping below runs over those only. (A wave's LDS accesses execute in issue
    // order: volatile accesses and a scheduling barrier are all the ordering it takes.)
    bool head = act;
    u32 from = 64;                                       // lane whose index this lane links to (64: the bucket's head in the table)
    const u32 slot = h & 1023u;
    if (act) lds_write8(H.dup + slot, (u32)lane);
    __builtin_amdgcn_wave_barrier();
    const bool lost = act && lds_read8(H.dup + slot) != (u32)lane;
    __builtin_amdgcn_wave_barrier();
    if (lost) lds_write8(H.dup + slot, (u32)lane);
    __builtin_amdgcn_wave_barrier();
    const bool shared = act && (lost || lds_read8(H.dup + slot) != (u32)lane);
    u64 rem = __ballot(shared);
    while (rem) {
      const u32 l = (u32)__builtin_ctzll(rem);
      const u32 hv = bcast(h, l);
      const u64 same = __ballot(act && h == hv);
      if (act && h == hv) {
        const u64 before = same & ((1ull << lane) - 1ull), after = same >> lane >> 1;
        if (before) { from = 63u - (u32)__builtin_clzll(before); link = H.insEnd + from; }
        head = after == 0;
      }
      rem &= ~same;
      HCNT(7, 1)
    }
    // the links behind the first: the first three of the index linked to — from the table for the bucket's old head (loads before this
    // step's stores), from the linked lane for a bucket-mate of this step (three rounds: a lane's second link is its mate's first, ...)
    uint4 E = make_uint4(0, 0, 0, 0);
    if (from >= 64 && act && link) E = H.chainT[link & H.cmask];
    const int srcLane = (int)(from < 64 ? from : (u32)lane);
    u32 l2 = (u32)__shfl((int)link, srcLane, 64); if (from >= 64) l2 = E.x;
    u32 l3 = (u32)__shfl((int)l2, srcLane, 64); if (from >= 64) l3 = E.y;
    u32 l4 = (u32)__shfl((int)l3, srcLane, 64); if (from >= 64) l4 = E.z;
    if (act) {
      H.oldLink[idx & 127u] = idx > H.chainSize ? H.chainT[idx & H.cmask] : make_uint4(0, 0, 0, 0);
      H.chainT[idx & H.cmask] = make_uint4(link, l2, l3, l4);
      if (head) H.hashT[h] = idx;
    }
    H.insEnd = min(endIdx, H.insEnd + 64u);
    hcw_sync();
  }
}

// answers of ZSTD_HcFindBestMatch for the positions [w, w + 64) that can be searched in this block (p <= ilimit)
__device__ void hcw_search_window(HCW& H, const u8* src, u32 w, u32 ilimit, u32 be, int lane) {
  const u32 lastPos = min(w + 63u, ilimit);
  HPROF(2)
  hcw_insert(H, src, lastPos + 2, lane);                 // indices <= lastPos + 1: every position of the window has its chain slot
  HPROF(0) HCNT(3, 1)
  const u32 p = w + (u32)lane;
  u32 ml = 3, offCode = 999999999u;
#ifdef ZRA_MF_PROFILE
  u32 steps_ = 0;
#endif
  if (p <= lastPos) {
    const u32 curr = p + 1, minChain = curr > H.chainSize ? curr - H.chainSize : 0;
    int attempts = 1 << H.searchLog;
    uint4 e = H.chainT[curr & H.cmask];
    // the lane's own 16 bytes stay in registers; per round trip up to FOUR candidates: their 16 bytes each (one load per candidate)
    // and the chain slot of the fourth (which names the next four) are requested together. (The reference's "byte at ml first" is only
    // a shortcut: a candidate that differs there cannot be longer than ml.) A candidate is touched only if the reference would reach it:
    // the one before it lies above minChain (so the link that named it is intact), an attempt is left, and it exists.
    // (Measured and dropped: two positions per lane in one loop, window of 128 — 101 VGPRs, 4 waves per SIMD, 1.5 instead of 2.8 GiB/s.)
    const bool wide = p + 16 <= be;
    const uint4 ownQ = wide ? ld128(src + p) : make_uint4(0, 0, 0, 0);
    const u64 own0 = (u64)ownQ.x | ((u64)ownQ.y << 32), own1 = (u64)ownQ.z | ((u64)ownQ.w << 32);
    auto measure = [&](u32 m, const uint4& c) -> u32 {
      if (wide) {
        const u64 d0 = ((u64)c.x | ((u64)c.y << 32)) ^ own0, d1 = ((u64)c.z | ((u64)c.w << 32)) ^ own1;
        if (d0) return (u32)__builtin_ctzll(d0) >> 3;
        if (d1) return 8 + ((u32)__builtin_ctzll(d1) >> 3);
        return 16 + count_eq(src, p + 16, m + 16, be);
      }
      return src[m + ml] == src[p + ml] ? count_eq(src, p, m, be) : 0u;
    };
    while (e.x >= 1 && attempts > 0) {
      const bool t2 = e.x > minChain && attempts > 1 && e.y >= 1;
      const bool t3 = t2 && e.y > minChain && attempts > 2 && e.z >= 1;
      const bool t4 = t3 && e.z > minChain && attempts > 3 && e.w >= 1;
      const uint4 z = make_uint4(0, 0, 0, 0);
      uint4 qa = z, qb = z, qc = z, qd = z, en = z;
      if (wide) {
        qa = ld128(src + e.x - 1);
        if (t2) qb = ld128(src + e.y - 1);
        if (t3) qc = ld128(src + e.z - 1);
        if (t4) qd = ld128(src + e.w - 1);
      }
      if (t4) {
        const u32 over = e.w + H.chainSize;                               // the index that shares e.w's chain slot
        // (two loads of two kinds, not one load through a selected pointer: that would be a flat load in the search loop)
        if (over > curr && over < H.insEnd) { const u64 lo = lds_read64((const u64*)&H.oldLink[over & 127u]), hi = lds_read64((const u64*)&H.oldLink[over & 127u] + 1);
                                              en = make_uint4((u32)lo, (u32)(lo >> 32), (u32)hi, (u32)(hi >> 32)); }
        else en = H.chainT[e.w & H.cmask];
      }
#ifdef ZRA_MF_PROFILE
      steps_++;
#endif
      u32 cur = measure(e.x - 1, qa);
      if (cur > ml) { ml = cur; offCode = curr - e.x + 2; if (p + cur == be) break; }
      if (e.x <= minChain) break;
      attempts--;
      if (!t2) break;                                    // (no further candidate, or no attempt left for it)
      cur = measure(e.y - 1, qb);
      if (cur > ml) { ml = cur; offCode = curr - e.y + 2; if (p + cur == be) break; }
      if (e.y <= minChain) break;
      attempts--;
      if (!t3) break;
      cur = measure(e.z - 1, qc);
      if (cur > ml) { ml = cur; offCode = curr - e.z + 2; if (p + cur == be) break; }
      if (e.z <= minChain) break;
      attempts--;
      if (!t4) break;
      cur = measure(e.w - 1, qd);
      if (cur > ml) { ml = cur; offCode = curr - e.w + 2; if (p + cur == be) break; }
      if (e.w <= minChain) break;
      attempts--;
      e = en;
    }
  }
  HPROF(1)
#ifdef ZRA_MF_PROFILE
  H.rsteps = steps_;
  for (u32 l_ = 0; l_ < 64; l_++) H.pt[8] += bcast(steps_, l_);
#endif
  H.w = w; H.rml = ml; H.roff = offCode; H.haveWin = true;
}

__device__ u32 mf_lazy_wave(HCW& H, const u8* src, u32 bs, u32 be, u32* rep, u64* seqs, u32* nOut, int depth, int lane) {
  u32 o1 = rep[0], o2 = rep[1], saved;
  u32 anchor = bs, n = 0;
  const u32 ilimit = be >= 8 ? be - 8 : 0;
  u32 ip = mf_prologue(bs, o1, o2, saved);
  H.haveWin = false;
#ifdef ZRA_MF_PROFILE
  for (int k_ = 0; k_ < 12; k_++) H.pt[k_] = 0;
  H.pl = __builtin_amdgcn_s_memtime();
#endif
  // the reference's search at position q (q <= ilimit): answer from the window, which slides when q leaves it
  auto search = [&](u32 q, u32& oc) -> u32 {
    if (!H.haveWin || q < H.w || q >= H.w + 64u) hcw_search_window(H, src, q, ilimit, be, lane);
    if (q + 1 > H.ntuRef) H.ntuRef = q + 1;
#ifdef ZRA_MF_PROFILE
    H.pt[9] += bcast(H.rsteps, q - H.w); H.pt[10] += 1;
#endif
    oc = bcast(H.roff, q - H.w);
    return bcast(H.rml, q - H.w);
  };
  auto put = [&](u32 ll, u32 ml, u32 offVal) { if (lane == 0) seqs[n] = (u64)ll | ((u64)ml << 20) | ((u64)offVal << 40); n++; };
  // repcode tests "4 bytes at t equal 4 bytes at t - o": the parse asks them at consecutive positions (ip+1, then ip+1 again from the
  // lazy loop, ip+2, after a match ip with the second offset, ...), one dependent round trip each in the serial form. Here one round
  // trip (rounds 2-5) fetched 8 bytes at cb and at cb - o1 and cb - o2 together and answered the tests at cb .. cb+4 for both offsets.
  // Round 6: the whole wave answers them — lane l compares the 4 bytes at cb + l with those at cb + l - o1 and cb + l - o2, two ballots
  // hold the answers for cb .. cb + 63 and both offsets: one round trip per 64 positions (and per change of the offsets) instead of
  // one per 5. The same tests, bit for bit: 4 bytes at t against 4 bytes at t - o for 0 < o <= t; t + 4 <= be holds for every t asked.
  u32 cb = 0xFFFFFFF0u, cbO1 = 0, cbO2 = 0; u64 m1 = 0, m2r = 0;
  auto rep_test = [&](u32 t, bool first) -> bool {          // first: against o1, else against o2 (the offset is > 0 and <= t)
    if (t < cb || t > cb + 63 || cbO1 != o1 || cbO2 != o2) {
      cb = t; cbO1 = o1; cbO2 = o2;
      const u32 q = t + (u32)lane;
      const bool in = q + 4 <= be;
      const u32 vs = in ? ld32(src + q) : 0u;
      const bool a1 = in && o1 > 0 && o1 <= q, a2 = in && o2 > 0 && o2 <= q;
      const u32 v1 = a1 ? ld32(src + q - o1) : 0u, v2 = a2 ? ld32(src + q - o2) : 0u;
      m1 = __ballot(a1 && v1 == vs); m2r = __ballot(a2 && v2 == vs);
    }
    return (((first ? m1 : m2r) >> (t - cb)) & 1ull) != 0;
  };
  while (ip < ilimit) {
    u32 ml = 0, start = ip + 1, off = 0; bool stored = false;
    if (o1 > 0 && rep_test(ip + 1, true)) {
      ml = wave_count_eq(src, ip + 5, ip + 5 - o1, be, lane) + 4;
      if (depth == 0) stored = true;
    }
    if (!stored) {
      u32 oc2; u32 m2 = search(ip, oc2);
      if (m2 > ml) { ml = m2; start = ip; off = oc2; }
      if (ml < 4) { ip += ((ip - anchor) >> 8) + 1; continue; }
      if (depth >= 1) {
        while (ip < ilimit) {
          ip++;
          if (off && o1 > 0 && rep_test(ip, true)) {
            const u32 mr = wave_count_eq(src, ip + 4, ip + 4 - o1, be, lane) + 4;
            const int g2 = (int)(mr * 3), g1 = (int)(ml * 3 - hb32(off + 1) + 1);
            if (mr >= 4 && g2 > g1) { ml = mr; off = 0; start = ip; }
          }
          {
            m2 = search(ip, oc2);
            const int g2 = (int)(m2 * 4 - hb32(oc2 + 1)), g1 = (int)(ml * 4 - hb32(off + 1) + 4);
            if (m2 >= 4 && g2 > g1) { ml = m2; off = oc2; start = ip; continue; }
          }
          if (depth == 2 && ip < ilimit) {
            ip++;
            if (off && o1 > 0 && rep_test(ip, true)) {
              const u32 mr = wave_count_eq(src, ip + 4, ip + 4 - o1, be, lane) + 4;
              const int g2 = (int)(mr * 4), g1 = (int)(ml * 4 - hb32(off + 1) + 1);
              if (mr >= 4 && g2 > g1) { ml = mr; off = 0; start = ip; }
            }
            {
              m2 = search(ip, oc2);
              const int g2 = (int)(m2 * 4 - hb32(oc2 + 1)), g1 = (int)(ml * 4 - hb32(off + 1) + 7);
              if (m2 >= 4 && g2 > g1) { ml = m2; off = oc2; start = ip; continue; }
            }
          }
          break;
        }
      }
      if (off) {
        const u32 ro = off - 2;
        // while (start > anchor && start > ro && src[start-1] == src[start-ro-1]) { start--; ml++; }
        const u32 back = (start > ro) ? wave_count_back(src, start, start - ro, anchor, lane) : 0u;
        start -= back; ml += back;
        o2 = o1; o1 = ro;
      }
    }
    put(start - anchor, ml, off ? off + 1 : 1);
    anchor = ip = start + ml;
    while (ip <= ilimit && o2 > 0 && rep_test(ip, false)) {
      const u32 rl = wave_count_eq(src, ip + 4, ip + 4 - o2, be, lane) + 4;
      const u32 t = o2; o2 = o1; o1 = t;
      put(0, rl, 1);
      ip += rl; anchor = ip;
    }
  }
  rep[0] = o1 ? o1 : saved; rep[1] = o2 ? o2 : saved;
  *nOut = n;
#ifdef ZRA_MF_PROFILE
  HPROF(2) HCNT(4, n) HCNT(5, 1)
  if (lane == 0) for (int k_ = 0; k_ < 12; k_++) atomicAdd(&zra_mf_prof[k_], H.pt[k_]);
#endif
  return be - anchor;
}

}  // namespace

// ---- per-frame setup shared by the match-finder kernels: which block of which frame, cleared tables on a fresh frame
struct MfFrame {
  const ZraEncParams* P; const u8* src; ZraEncFrameState* st; ZraEncBlockOut* bo; u32* hashT; u32* chainT; u64* seqs;
  u32 fsize, bs, be;
};
// returns false when this workgroup has nothing to parse (block beyond the frame, or a block too small to compress)
// hcwOnly: the caller is the wave-cooperative hash-chain finder — its chain slots need no clearing (below)
__device__ __forceinline__ bool mf_frame_setup(const ZraEncArgs& a, u32 block, int lane, MfFrame& F, u32 f, u32 tableSlot, bool hcwOnly = false, bool keepTables = false) {
  const u64 fstart = (u64)(a.firstFrame + f) * a.frameSize;
  const u64 remaining = a.inSize - fstart;
  F.fsize = (u32)(remaining < a.frameSize ? remaining : a.frameSize);
  F.P = (F.fsize == a.frameSize) ? &a.full : &a.tail;                  // the short last frame has its own cparams (A.4.1)
  const u32 blockSize = F.P->blockSize;
  F.bs = block * blockSize;
  if (F.bs >= F.fsize && !(F.fsize == 0 && block == 0)) return false;
  F.be = min(F.fsize, F.bs + blockSize);
  F.src = a.in + fstart;
  F.st = &a.state[f];
  F.hashT = a.tables + (size_t)tableSlot * a.tableStride;
  F.chainT = F.hashT + ((size_t)1 << F.P->hashLog);
  F.bo = &a.blockOut[f];
  F.seqs = a.seqs + (size_t)f * a.seqStride;
  if (block == 0) {
    // fresh frame: zeroed tables, repcodes {1,4,8}, nextToUpdate 1 (A.4.8); 16-byte coalesced clears by the whole wave
    // The wave-cooperative hash-chain finder (four links per chain slot: 16 bytes per position of the window, 4 MiB per 256 KiB frame
    // at level 9) clears the hash table only: a chain slot is written when its index is inserted and read only for inserted indices —
    // the slot of the position being searched, of a candidate above the chain's lower bound (the other candidates end the search before
    // their links are followed), of the bucket's head at insertion. What an untouched slot holds is copied around at most (the ring of
    // overwritten slots), never looked at. Everybody else clears hash + chain (+ what the optimal parsers keep behind them).
    size_t words = (size_t)1 << F.P->hashLog;
    if (!hcwOnly) words += (size_t)1 << F.P->chainLog;
    if (F.P->strategy >= 7)           // optimal parsers: the 3-byte hash table and the statistics of the price model behind the tree
      words += (F.P->minMatch == 3 ? (size_t)1 << min(17u, F.P->windowLog) : 0) + 512;
    // (keepTables, round 6: the dfast kernel's cells carry the epoch of the frame that wrote them — what an earlier frame of this wave left
    //  in the slot reads as empty, nothing to clear)
    if (keepTables) words = 0;
    uint4* t4 = (uint4*)F.hashT;
    for (size_t i = lane; i < words / 4; i += 64) t4[i] = make_uint4(0, 0, 0, 0);
    for (size_t i = (words / 4) * 4 + lane; i < words; i += 64) F.hashT[i] = 0;
    if (lane == 0) { F.st->rep[0] = 1; F.st->rep[1] = 4; F.st->rep[2] = 8; F.st->nextToUpdate = 1; F.st->insEnd = 1; F.st->idxShift = 0; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
  if (F.be - F.bs < 7) {                               // too small to compress (A.4.2) -> raw block
    if (lane == 0) { F.bo->nbSeq = 0; F.bo->lastLL = F.be - F.bs; F.bo->skip = 1; }
    return false;
  }
  return true;
}

// One block of one frame parsed by ONE lane: fast, the serial hash chain (odd tails), btlazy2 and the optimal parsers.
// MODE 1: the caller has checked strategy == 1; MODE 2: no frame with an optimal parser (strategy >= 7) comes here — the finders left
// out stay out of the caller's kernel and out of its register budget (a kernel pays for everything it can call). MODE 0: all of them.
template <int MODE>
__device__ __forceinline__ void mf_serial_block_t(const MfFrame& F, u32 ntu0) {
  const ZraEncParams& P = *F.P;
  const u8* src = F.src; ZraEncFrameState* st = F.st; ZraEncBlockOut* bo = F.bo;
  u32* hashT = F.hashT; u32* chainT = F.chainT;
  const u32 bs = F.bs, be = F.be;
  u32 rep[3] = {st->rep[0], st->rep[1], st->rep[2]};
  u32 lastLL;
  bo->skip = 0;
  Emit E; E.seqs = F.seqs; E.n = 0;
  // limited update after a very long match (A.4.3 hash chain prologue; harmless for the other finders)
  u32 ntu = ntu0;
  { const u32 cur = bs + 1; if (cur > ntu + 384) { const u32 d = cur - ntu - 384; ntu = cur - (d < 192 ? d : 192); } }
  if constexpr (MODE == 1) lastLL = mf_fast(P, hashT, src, bs, be, rep, E);
  else if (MODE == 0 && P.strategy >= 7) {
    // btopt / btultra / btultra2 (zra_encode_opt.h); the limited update above is redone with the window shift of btultra2
    OptCtx O;
    O.hashT = hashT; O.bt = chainT; O.hashLog = P.hashLog; O.chainLog = P.chainLog; O.searchLog = P.searchLog;
    O.minMatchParam = P.minMatch; O.targetLength = P.targetLength; O.lvl = P.strategy == 7 ? 0 : 2; O.windowLog = P.windowLog;
    O.hashLog3 = P.minMatch == 3 ? min(17u, P.windowLog) : 0;
    O.hash3 = chainT + ((size_t)1 << P.chainLog);
    O.o = (ZraOptState*)(O.hash3 + (O.hashLog3 ? (size_t)1 << O.hashLog3 : 0));
    O.idxShift = st->idxShift;
    ntu = ntu0;
    { const u32 cur = bs + 1 + O.idxShift; if (cur > ntu + 384) { const u32 d = cur - ntu - 384; ntu = cur - (d < 192 ? d : 192); } }
    O.nextToUpdate = ntu;
    if (P.strategy == 9 && O.o->litLengthSum == 0 && bs == 0 && ntu == 1 && be - bs > 1024) {
      // btultra2: a first pass over the first block only to collect statistics, then the window is moved past it
      u32 tmpRep[3] = {rep[0], rep[1], rep[2]};
      (void)O.parse(src, bs, be, tmpRep, E, true);
      O.idxShift += be - bs;
      O.nextToUpdate = 1 + O.idxShift;
      O.upscale_stats();
    }
    lastLL = O.parse(src, bs, be, rep, E, false);
    ntu = O.nextToUpdate;
    st->idxShift = O.idxShift;
  } else if (P.strategy == 1) lastLL = mf_fast(P, hashT, src, bs, be, rep, E);
  else if (P.strategy == 2) lastLL = mf_dfast_serial(P, hashT, chainT, src, bs, be, rep, E);
  else {
    HC H; H.hashT = hashT; H.chainT = chainT; H.hlog = P.hashLog; H.mls = P.minMatch < 4 ? 4 : P.minMatch > 6 ? 6 : P.minMatch;
    H.chainSize = 1u << P.chainLog; H.cmask = H.chainSize - 1; H.searchLog = P.searchLog; H.nextToUpdate = ntu;
    H.windowLog = P.windowLog;
    H.bt = P.strategy == 6;
    lastLL = mf_lazy(H, src, bs, be, rep, E, P.strategy == 6 ? 2 : (int)P.strategy - 3);
    ntu = H.nextToUpdate;
  }
  st->nextToUpdate = ntu; st->insEnd = ntu;
  bo->nbSeq = E.n; bo->lastLL = lastLL;
  bo->rep[0] = rep[0]; bo->rep[1] = rep[1]; bo->rep[2] = rep[2];   // confirmed by stage 2 only if the block is emitted compressed
}

// ---- bucket flags of a frame, computed by the frame's own wave ahead of its parse (round 5). Two bits per position for
// mf_dfast_lean<.., FLAGS>: the position's long / short bucket has a LATER position of the frame — an insertion without one is never looked
// up, so its table write is skipped (59 % / 43 % of the long / short table writes on the bench corpus, and the random 4-byte writes are what
// the memory system is slowest at). "Has an earlier position" is not computed: the parse's LDS filter of inserted buckets decides the reads.
// Layout: per window of 64 positions two u64 masks {long, short} at flg + 16 * window.
// Backward sweep over the frame, 64 positions per window, the last window first: a "seen" bit per bucket in LDS tells whether a later WINDOW
// holds the bucket; inside a window every lane that shares a byte counter (bucket & 255) with another lane counts as having a later mate —
// a superset (an earlier mate or another bucket in the slot only cost an unnecessary write, never a missing one; tools/model/dfast_flags_stats.c:
// 4-8 % more writes than the exact rule). The bitmaps of both tables are 12 KiB at hashLog 16 / chainLog 15 and the wave owns 6.5 KiB, so
// the sweep runs `1 << npLog` times, each over the buckets whose top bits equal the pass; a pass ORs into the masks of the passes before it.
// lds: ldsWords words — [0, 128) byte counters (256 per table), then the seen bits.
// Source bytes: 512 positions per load (lane l holds the 8 bytes at block + 8 l; a window's own 8 bytes per lane come out of them with four
// cross-lane reads), two blocks in flight ahead of the one being hashed (a block's eight windows take about as long as a loaded round trip): one memory round trip per 512 positions, off the critical path.
struct DfBlock { u64 a; u64 m; };                       // the block's chunk of this lane; lanes 0..15: the 16 masks of the block's 8 windows (passes > 0)
__device__ __forceinline__ DfBlock df_block_load(const u8* src, u32 fsize, const u8* flg, u32 blk, bool haveMasks, int lane) {
  DfBlock B; B.a = 0; B.m = 0;
  const u32 c = blk * 512 + 8 * (u32)lane;
  if (c + 8 <= fsize) B.a = ld64(src + c);
  else if (c < fsize) B.a = ld64_safe(src + c, src + fsize);
  if (haveMasks && lane < 16) B.m = *(const u64*)(flg + (size_t)blk * 128 + 8 * (u32)lane);
  return B;
}
__device__ __forceinline__ void df_later_flags(const u8* src, u32 fsize, u32 hlog, u32 clog, u32 mls, u32* lds, u32 ldsWords, u8* flg, int lane) {
  if (fsize < 8) return;                                 // (no position is ever hashed)
  const u32 lastPos = fsize - 8;
  const u32 bmWords = ldsWords - 128;
  u32 npLog = 0;
  while ((((1u << hlog) + (1u << clog)) >> npLog) > bmWords * 32) npLog++;
  const u32 hlogP = hlog - npLog, clogP = clog - npLog;
  u32* const cnt = lds; u32* const bmL = lds + 128; u32* const bmS = bmL + ((1u << hlogP) >> 5);
  const u64 primeS = mls == 5 ? 889523592379ULL : mls == 6 ? 227718039650203ULL : 58295818150454627ULL;
  const u32 shV = 64 - 8 * min(mls, 7u);
  const u32 nBlk = lastPos / 512 + 1;
  const u32 srcLane = (u32)lane >> 3, sh = ((u32)lane & 7u) * 8;
  for (u32 pass = 0; pass < (1u << npLog); pass++) {
    for (u32 i = (u32)lane; i < 128 + ((1u << hlogP) >> 5) + ((1u << clogP) >> 5); i += 64) lds[i] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    const bool hm = pass != 0;
    // blocks nBlk-1 .. 0; B0 is the one being hashed, B1 the next one below it (in flight meanwhile)
    DfBlock B0 = df_block_load(src, fsize, flg, nBlk - 1, hm, lane), B1{0, 0};
    // the 8 bytes behind the block (lane 0 of the block above). Behind the TOP block: the frame's own bytes there — the last hashed
    // position's 8 bytes reach up to 7 bytes past its block when fsize % 512 is 1..7 (found by the round-5 soak, seed 90047: a tail frame
    // of 40,967 bytes; with zeros here its last positions marked the wrong buckets and an earlier bucket-mate's write was skipped)
    u64 above = nBlk * 512 < fsize ? ld64_safe(src + nBlk * 512, src + fsize) : 0;
    for (u32 blk = nBlk; blk-- > 0;) {
      if (blk >= 1) B1 = df_block_load(src, fsize, flg, blk - 1, hm, lane);
      const u32 aLo = (u32)B0.a, aHi = (u32)(B0.a >> 32);
      u64 outM = 0;                                      // lanes 2k / 2k+1: the long / short mask of window k
      // two windows per step, the upper one first: their cross-lane reads go out together, then their LDS sequences back to back (LDS
      // executes a wave's operations in issue order: the lower window's "seen" reads come behind the upper window's marks), one wait
      // for all of it — the sweep is a chain of LDS round trips, and at 22 waves per CU each one costs hundreds of cycles
#pragma unroll 1
      for (u32 kk = 0; kk < 4; kk++) {
        if (blk * 512 + (6 - 2 * kk) * 64 > lastPos) continue;      // (wave-uniform: both windows of the pair lie behind the last position)
        u32 iLw[2], iSw[2]; bool pLw[2], pSw[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const u32 k = 7 - 2 * kk - h;
          const u32 p = blk * 512 + k * 64 + (u32)lane;
          const bool act = p <= lastPos;
          // this lane's 8 bytes: chunks 8k + lane/8 and the one behind it, shifted by lane % 8 bytes
          const u32 ci = 8 * k + srcLane;
          const u32 c0l = (u32)__builtin_amdgcn_ds_bpermute((int)(4 * ci), (int)aLo), c0h = (u32)__builtin_amdgcn_ds_bpermute((int)(4 * ci), (int)aHi);
          u32 c1l = (u32)__builtin_amdgcn_ds_bpermute((int)(4 * ((ci + 1) & 63)), (int)aLo), c1h = (u32)__builtin_amdgcn_ds_bpermute((int)(4 * ((ci + 1) & 63)), (int)aHi);
          if (ci == 63) { c1l = (u32)above; c1h = (u32)(above >> 32); }
          const u64 c0 = (u64)c0l | ((u64)c0h << 32), c1 = (u64)c1l | ((u64)c1h << 32);
          const u64 v = sh ? (c0 >> sh) | (c1 << (64 - sh)) : c0;
          const u32 bL = (u32)((v * 0xCF1BBCDCB7A56463ULL) >> (64 - hlog));
          const u32 bS = mls <= 4 ? ((u32)v * 2654435761u) >> (32 - clog) : (u32)(((v << shV) * primeS) >> (64 - clog));
          pLw[h] = act && (bL >> hlogP) == pass; pSw[h] = act && (bS >> clogP) == pass;
          iLw[h] = bL & ((1u << hlogP) - 1); iSw[h] = bS & ((1u << clogP) - 1);
        }
        u32 seenL[2], seenS[2], cL[2], cS[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const u32 iL = iLw[h], iS = iSw[h], sL = iL & 255u, sS = 256u + (iS & 255u);
          const bool pL = pLw[h], pS = pSw[h];
          seenL[h] = pL ? lds_read32(&bmL[iL >> 5]) : 0u; seenS[h] = pS ? lds_read32(&bmS[iS >> 5]) : 0u;
          if (pL) atomicAdd(&cnt[sL >> 2], 1u << (8 * (sL & 3)));
          if (pS) atomicAdd(&cnt[sS >> 2], 1u << (8 * (sS & 3)));
          __builtin_amdgcn_wave_barrier();
          cL[h] = pL ? lds_read32(&cnt[sL >> 2]) : 0u;
          cS[h] = pS ? lds_read32(&cnt[sS >> 2]) : 0u;
          __builtin_amdgcn_wave_barrier();
          if (pL) { atomicSub(&cnt[sL >> 2], 1u << (8 * (sL & 3))); atomicOr(&bmL[iL >> 5], 1u << (iL & 31)); }
          if (pS) { atomicSub(&cnt[sS >> 2], 1u << (8 * (sS & 3))); atomicOr(&bmS[iS >> 5], 1u << (iS & 31)); }
          __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const u32 k = 7 - 2 * kk - h;
          if (blk * 512 + k * 64 > lastPos) continue;    // (the pair's upper window alone lies behind the last position)
          const u32 iL = iLw[h], iS = iSw[h], sL = iL & 255u, sS = iS & 255u;
          const u64 mL = __ballot(((seenL[h] >> (iL & 31)) & 1u) || ((cL[h] >> (8 * (sL & 3))) & 255u) > 1);
          const u64 mS = __ballot(((seenS[h] >> (iS & 31)) & 1u) || ((cS[h] >> (8 * (sS & 3))) & 255u) > 1);
          if ((u32)lane == 2 * k) outM = mL;
          if ((u32)lane == 2 * k + 1) outM = mS;
        }
      }
      if (lane < 16) *(u64*)(flg + (size_t)blk * 128 + 8 * (u32)lane) = outM | B0.m;
      above = (u64)bcast(aLo, 0) | ((u64)bcast(aHi, 0) << 32);
      B0 = B1;
    }
    // the next pass reads these masks back (and the parse after the last one): stores done and visible to this CU's loads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
}
// Match finder for strategy 2 (dfast, levels 3-4): one wave per frame, lean window-resolve parse. Launched when the batch's
// full-size frames use dfast; a short last frame with another strategy is left to zra_mf_kernel (second launch, `only`).
// LSRC (round 4, calls of a few hundred frames at most): the frame's source bytes are copied into LDS first and the parse reads them
// there — every source access of the parse (hash input, repcode streams, candidate compares, extensions) is an LDS access; only the
// table gathers go to memory. A lone frame is one dependent chain of round trips: this shortens most of them from memory latency to
// LDS latency. One or two frames per CU by LDS, so it is a latency mode, not the throughput kernel.
template <bool FLAGS, bool LSRC = false>
__device__ __forceinline__ void mf_dfast_body(const ZraEncArgs& a, u32 block, u32 only, u32 onlySlot, const ZraFlagArgs* g = nullptr) {
  const int lane = threadIdx.x;
  // dynamic LDS: [LSRC: source bytes, (a.mfFilter >> 16) * 64][dup slots 2 x dupSlots x 4 B][filter L][filter S]; geometry chosen by the host
  // (a.mfFilter: shL | shS<<4 | log2(dupSlots)<<8 | source granules << 16)
  extern __shared__ u32 dynLds[];
  LeanLds W;
  W.shL = a.mfFilter & 15; W.shS = (a.mfFilter >> 4) & 15; W.dupSlots = 1u << ((a.mfFilter >> 8) & 15);
  u32* const fltLds = LSRC ? dynLds + (a.mfFilter >> 16) * 16 : dynLds;
  W.dup = (u8*)fltLds;
  W.bmL = fltLds + 2 * W.dupSlots;
  W.span = nullptr; W.spanFlg = nullptr; W.spanBytes = 0;
  if (FLAGS && g && g->spanBytes) { W.spanBytes = g->spanBytes; W.span = (u8*)(fltLds + g->ldsWords); W.spanFlg = W.span + g->spanBytes + 16; }
  const bool persistent = a.mfQueue != nullptr;
  // launch telemetry (persistent launches): where this wave sits and its shader cycles against the constant 100 MHz clock
  const bool tele = persistent && a.mfTele != nullptr;
  // (start values parked in the wave's own record of the telemetry buffer: nothing of this stays in registers while frames are parsed)
  u64* const tl = tele ? a.mfTele + ZRA_TELE_HEAD + 4 * (size_t)min(blockIdx.x, ZRA_TELE_WAVES - 1u) : nullptr;
  if (tele && lane == 0) {
    const u32 hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);          // HW_REG_HW_ID: cu [11:8], sh [12], se [15:13]
    const u32 xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;    // HW_REG_XCC_ID [3:0]
    const u64 tc0 = __builtin_readcyclecounter(), tr0 = wall_clock64();
    const u32 key = (xcc << 8) | (((hw >> 13) & 7u) << 5) | (((hw >> 12) & 1u) << 4) | ((hw >> 8) & 15u);
    atomicAdd((unsigned long long*)&a.mfTele[32 + key], 1ull << (16u * ((hw >> 4) & 3u)));   // four 16-bit counts: the CU's SIMDs (HW_ID simd_id [5:4])
    atomicAdd((unsigned long long*)&a.mfTele[8 + xcc], 1ull);
    atomicMax((unsigned long long*)&a.mfTele[4], ~tr0);
    atomicMax((unsigned long long*)&a.mfTele[6], tr0);
    tl[0] = tc0; tl[1] = tr0; tl[2] = xcc; tl[3] = 0;
  }
  if (persistent && a.mfStarted && lane == 0) atomicAdd(a.mfStarted, 1u);
  // round 6: epoch-tagged cells (DfHash). The wave's table slot is cleared before the first frame it takes and then once per 2^epochBits
  // full-size frames; a short last frame (other cparams: another cell layout) clears before and forces a clear behind it.
  const u32 epochBits = (FLAGS && g && persistent) ? min(g->epochBits, ZRA_DF_EPOCH_BITS) : 0u;   // (host knob: fewer than the cell holds, 0 = clear per frame)
  u32 epochCtr = 0;                                    // full-size frames parsed since the slot's last clear
  for (;;) {
    u32 f = only == 0xFFFFFFFFu ? blockIdx.x : only;   // `only`: a single-workgroup launch for that frame on table slot `onlySlot`
    if (persistent) {
      u32 t = 0;
      if (lane == 0) t = atomicAdd(a.mfQueue, 1u);
      f = rfl(t);
      if (f >= a.nFrames) {
        if (tele && lane == 0) {
          const u64 tc1 = __builtin_readcyclecounter(), tr1 = wall_clock64();
          const u64 tc0 = tl[0], tr0 = tl[1];
          const u32 xcc = (u32)tl[2] & 7u;
          atomicAdd((unsigned long long*)&a.mfTele[0], tc1 - tc0);
          atomicAdd((unsigned long long*)&a.mfTele[1], tr1 - tr0);
          atomicAdd((unsigned long long*)&a.mfTele[2], 1ull);
          atomicMax((unsigned long long*)&a.mfTele[3], tr1 - tr0);
          atomicMax((unsigned long long*)&a.mfTele[5], tr1);
          atomicMax((unsigned long long*)&a.mfTele[7], ~tr1);
          atomicAdd((unsigned long long*)&a.mfTele[16 + xcc], (unsigned long long)tl[3]);
          atomicAdd((unsigned long long*)&a.mfTele[24 + xcc], tr1 - tr0);
        }
        return;
      }
      if (tele && lane == 0) tl[3]++;
    }
#ifdef ZRA_MF_PROFILE
    const u64 kt0_ = __builtin_amdgcn_s_memtime();
#endif
    bool mine = true;
    {
      // frames whose cparams select another strategy (only the short last frame of the input can differ from the rest) are parsed
      // by zra_mf_kernel, launched for that frame by the host
      const u64 fstart = (u64)(a.firstFrame + f) * a.frameSize;
      const ZraEncParams& Pf = (a.inSize - fstart < a.frameSize) ? a.tail : a.full;
      if (Pf.strategy != 2) mine = false;
    }
    MfFrame F;
    bool keepTables = false; u32 epoch = 0;
    if (epochBits && mine) {
      const u64 fstart = (u64)(a.firstFrame + f) * a.frameSize;
      const bool fullFrame = a.inSize - fstart >= a.frameSize;
      if (!fullFrame) epochCtr = 0;                   // (cleared now; epoch 0)
      epoch = epochCtr & ((1u << epochBits) - 1u);
      keepTables = epoch != 0;
      epochCtr = fullFrame ? epochCtr + 1 : 0;
    }
    if (mine && mf_frame_setup(a, block, lane, F, f, persistent ? blockIdx.x : only == 0xFFFFFFFFu ? f : onlySlot, false, keepTables)) {
#ifdef ZRA_MF_PROFILE
      __builtin_amdgcn_s_waitcnt(0);
      if (lane == 0) atomicAdd(&zra_mf_prof[20], __builtin_amdgcn_s_memtime() - kt0_);
#endif
      const u32 wordsL = ((1u << F.P->hashLog) >> W.shL) / 32, wordsS = ((1u << F.P->chainLog) >> W.shS) / 32;
      W.bmS = W.bmL + wordsL;
      u32 rep[3] = {F.st->rep[0], F.st->rep[1], F.st->rep[2]};
      u32 lastLL, nseq = 0;
      const u32 ib = 32 - __builtin_clz(F.fsize - 1);  // bits for position+1 < fsize (fsize >= 7 here)
      const u8* flg = nullptr;
      if (FLAGS && g && g->flags && block == 0 && persistent) {
        // round 5: the wave computes its frame's flags itself, ahead of the parse, over the LDS the parse uses afterwards (the filter is
        // cleared below, the duplicate slots by the parse)
        u8* const fw = g->flags + (size_t)blockIdx.x * g->flagStride;
        df_later_flags(F.src, F.fsize, F.P->hashLog, F.P->chainLog, F.P->minMatch, fltLds, g->ldsWords, fw, lane);
        flg = fw;
      }
      {
        // block 0 starts with empty tables (all bits clear); later blocks of a frame inherit tables filled by earlier launches
        const u32 fill = block == 0 ? 0u : 0xFFFFFFFFu;
        for (u32 i = lane; i < wordsL + wordsS; i += 64) W.bmL[i] = fill;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      }
      const u8* srcP = F.src;
      if (LSRC) {
        // the frame into LDS, 16 bytes per lane and step (the last, partial, 16 bytes one by one: the input ends there), 64 zero bytes behind it
        u8* const ls = (u8*)dynLds;
        const u32 n16 = F.fsize >> 4;
        for (u32 i = (u32)lane; i < n16; i += 64) *(uint4*)(ls + 16 * i) = ld128(F.src + 16 * i);
        for (u32 i = (n16 << 4) + (u32)lane; i < F.fsize; i += 64) ls[i] = F.src[i];
        ls[F.fsize + (u32)lane] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        srcP = ls;
      }
      switch (F.P->minMatch) {
        case 5: lastLL = mf_dfast_lean<5, FLAGS>(*F.P, F.hashT, F.chainT, srcP, F.bs, F.be, rep, F.seqs, &nseq, W, lane, ib, flg, epoch); break;
        case 6: lastLL = mf_dfast_lean<6, FLAGS>(*F.P, F.hashT, F.chainT, srcP, F.bs, F.be, rep, F.seqs, &nseq, W, lane, ib, flg, epoch); break;
        case 7: lastLL = mf_dfast_lean<7, FLAGS>(*F.P, F.hashT, F.chainT, srcP, F.bs, F.be, rep, F.seqs, &nseq, W, lane, ib, flg, epoch); break;
        default: lastLL = mf_dfast_lean<4, FLAGS>(*F.P, F.hashT, F.chainT, srcP, F.bs, F.be, rep, F.seqs, &nseq, W, lane, ib, flg, epoch); break;
      }
      if (lane == 0) {
        F.bo->nbSeq = nseq; F.bo->lastLL = lastLL; F.bo->skip = 0;
        F.bo->rep[0] = rep[0]; F.bo->rep[1] = rep[1]; F.bo->rep[2] = rep[2];
      }
#ifdef ZRA_MF_PROFILE
      __builtin_amdgcn_s_waitcnt(0);
      if (lane == 0) { atomicAdd(&zra_mf_prof[21], __builtin_amdgcn_s_memtime() - kt0_); atomicAdd(&zra_mf_prof[22], 1ull); }
#endif
    }
    if (!persistent) return;
    if (mine) {
      // publish: every store of this frame (sequences, block record) is visible device-wide before its stamp
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      if (lane == 0) { __hip_atomic_store(&a.blockOut[f].ready, a.readyStamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); if (a.mfDone) atomicAdd(&a.mfDone[f / a.entSubFrames], 1u); }
    }
  }
}

extern "C" __global__ void __launch_bounds__(64)
zra_mf_dfast_kernel(ZraEncArgs a, u32 block, u32 only, u32 onlySlot) { mf_dfast_body<false>(a, block, only, onlySlot); }
// the same parse behind the wave's own bucket-flag sweep (round 5, df_later_flags): skips the table writes nobody can read
extern "C" __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(6)))   // (22 resident waves per CU: six on two of the SIMDs)
zra_mf_dfast_fl_kernel(ZraEncArgs a, ZraFlagArgs g, u32 block, u32 only, u32 onlySlot) { mf_dfast_body<true>(a, block, only, onlySlot, &g); }
// the same parse over a copy of the frame in LDS (round 4): the latency mode for calls of a few hundred frames at most
extern "C" __global__ void __launch_bounds__(64)
zra_mf_dfast_ls_kernel(ZraEncArgs a, u32 block, u32 only, u32 onlySlot) { mf_dfast_body<false, true>(a, block, only, onlySlot); }

// Match finder for everything the lean kernels do not take (btlazy2, the optimal parsers, frames larger than the level's window, single
// odd tails of any strategy). The parse of a frame is one dependent pointer chase (hash head
// -> chain -> candidate bytes), so a frame gets ONE lane and throughput is frames in flight / frame latency: a wave carries `perWave`
// frames in its first lanes (SIMT across frames: divergent, but the lanes' memory round trips overlap), and the whole wave clears
// their tables. `only` = 0xFFFFFFFF: every frame of the batch; otherwise just that frame (the short last frame whose strategy differs
// from the batch's), launched as one workgroup on table slot `onlySlot`.
template <bool OPT>
__device__ __forceinline__ void mf_generic(const ZraEncArgs& a, u32 block, u32 only, u32 onlySlot, u32 perWave) {
  const int lane = threadIdx.x;
  const bool all = only == 0xFFFFFFFFu;
  if (!all) perWave = 1;
  // cooperative per-frame setup (table clears), one frame after the other; every lane keeps the descriptor of "its" frame
  MfFrame F; bool mine = false;
  for (u32 k = 0; k < perWave; k++) {
    const u32 f = all ? blockIdx.x * perWave + k : only;
    if (f >= a.nFrames) break;
    MfFrame G;
    const bool go = mf_frame_setup(a, block, lane, G, f, all ? f : onlySlot);
    if ((u32)lane == k || perWave == 1) { F = G; mine = go && (G.P->strategy != 2 || a.serialAll); }   // dfast frames belong to zra_mf_dfast_kernel
  }
  // (hash-chain frames — greedy / lazy / lazy2 inside the window — never come here: zra_mf_hc_kernel, also for a lone short last frame.
  //  Reached with such a frame anyway, the serial chain finder below would read the two-link chain slots of HCW as single links.)
  if (!mine || (lane != 0 && perWave == 1)) return;
  const u32 ntu0 = F.st->nextToUpdate;
  mf_serial_block_t<OPT ? 0 : 2>(F, ntu0);
}
// the generic finder without / with the optimal parsers (215 VGPRs with them: batches and tails of levels 13-22 only)
extern "C" __global__ void __launch_bounds__(64)
zra_mf_kernel(ZraEncArgs a, u32 block, u32 only, u32 onlySlot, u32 perWave) {
  mf_generic<false>(a, block, only, onlySlot, perWave);
}
extern "C" __global__ void __launch_bounds__(64)
zra_mf_opt_kernel(ZraEncArgs a, u32 block, u32 only, u32 onlySlot, u32 perWave) {
  mf_generic<true>(a, block, only, onlySlot, perWave);
}

// Match finder for batches whose full-size frames use "fast" (levels 1-2 and all negative levels): lane = frame, `perWave` frames in
// the first lanes of a wave (SIMT across frames: divergent, but the lanes' memory round trips overlap), the whole wave clears their
// tables. Nothing but mf_fast lives here: in one kernel with the tree and optimal-parser code the register budget was 215 VGPRs and
// 412 B of scratch, and level 1 ran at 9.2 instead of 14 GiB/s. A short last frame with other cparams: second launch by the host.
extern "C" __global__ void __launch_bounds__(64)
zra_mf_fast_kernel(ZraEncArgs a, u32 block, u32 perWave) {
  const int lane = threadIdx.x;
  MfFrame F; bool mine = false;
  for (u32 k = 0; k < perWave; k++) {
    const u32 f = blockIdx.x * perWave + k;
    if (f >= a.nFrames) break;
    {
      const u64 fstart = (u64)(a.firstFrame + f) * a.frameSize;
      const ZraEncParams& Pf = (a.inSize - fstart < a.frameSize) ? a.tail : a.full;
      if (Pf.strategy != 1) continue;
    }
    MfFrame G;
    const bool go = mf_frame_setup(a, block, lane, G, f, f);
    if ((u32)lane == k) { F = G; mine = go; }
  }
  if (!mine) return;
  mf_serial_block_t<1>(F, F.st->nextToUpdate);
}

// Match finder for batches whose full-size frames use a hash-chain strategy (greedy / lazy / lazy2): one wave per frame, the
// wave-cooperative finder and nothing else (37 VGPRs; with the serial finders in the same kernel it was 119 and half the waves).
// A short last frame whose cparams select another strategy is left to a second single-frame launch by the host
// (zra_mf_dfast_kernel or zra_mf_kernel with `only`).
extern "C" __global__ void __launch_bounds__(64)
zra_mf_hc_kernel(ZraEncArgs a, u32 block, u32 only, u32 onlySlot) {
  const int lane = threadIdx.x;
  const bool all = only == 0xFFFFFFFFu;                // else: frame `only` alone, with table slot onlySlot (a short last frame)
  const u32 f = all ? blockIdx.x : only;
  if (f >= a.nFrames) return;
  {
    const u64 fstart = (u64)(a.firstFrame + f) * a.frameSize;
    const ZraEncParams& Pf = (a.inSize - fstart < a.frameSize) ? a.tail : a.full;
    if (Pf.strategy < 3 || Pf.strategy > 5) return;
  }
  MfFrame F;
  if (!mf_frame_setup(a, block, lane, F, f, all ? f : onlySlot, true)) return;
  const ZraEncParams& P = *F.P;
  ZraEncFrameState* st = F.st; ZraEncBlockOut* bo = F.bo;
  const u32 bs = F.bs, be = F.be;
  u32 rep[3] = {st->rep[0], st->rep[1], st->rep[2]};
  // limited update after a very long match (A.4.3 hash chain prologue)
  const u32 ntu0 = st->nextToUpdate;
  u32 ntu = ntu0;
  {
    const u32 cur = bs + 1;
    if (cur > ntu + 384) { const u32 d = cur - ntu - 384; ntu = cur - (d < 192 ? d : 192); }
  }
  __shared__ uint4 hcOld[128];
  __shared__ u8 hcDup[1024];
  HCW H; H.oldLink = hcOld; H.dup = hcDup; H.hashT = F.hashT; H.chainT = (uint4*)F.chainT; H.hlog = P.hashLog; H.mls = P.minMatch < 4 ? 4 : P.minMatch > 6 ? 6 : P.minMatch;
  H.chainSize = 1u << P.chainLog; H.cmask = H.chainSize - 1; H.searchLog = P.searchLog;
  H.insEnd = st->insEnd; H.ntuRef = ntu;
  for (u32 i = (u32)lane; i < 128; i += 64) hcOld[i] = *(const uint4*)st->ring[i];
  hcw_sync();
  if (ntu > ntu0) {
    // indices [ntu0, ntu) are never inserted by the reference: those inserted ahead of the parse come out again, the rest is skipped
    if (H.insEnd > ntu0) hcw_undo(H, F.src, ntu0, lane);
    H.insEnd = ntu;
  }
  u32 nSeq = 0;
  const u32 lastLL = mf_lazy_wave(H, F.src, bs, be, rep, F.seqs, &nSeq, (int)P.strategy - 3, lane);
  hcw_sync();
  for (u32 i = (u32)lane; i < 128; i += 64) *(uint4*)st->ring[i] = hcOld[i];
  if (lane == 0) {
    st->nextToUpdate = H.ntuRef; st->insEnd = H.insEnd;
    bo->skip = 0; bo->nbSeq = nSeq; bo->lastLL = lastLL;
    bo->rep[0] = rep[0]; bo->rep[1] = rep[1]; bo->rep[2] = rep[2];   // confirmed by stage 2 only if the block is emitted compressed
  }
}

#ifdef ZRA_MF_PROFILE
// bring-up only: copies (and optionally clears) the phase counters
extern "C" __attribute__((visibility("default"))) void ZraHipDebugReadMfProfile(unsigned long long* out24, int reset) {
  (void)hipMemcpyFromSymbol(out24, HIP_SYMBOL(zra_mf_prof), sizeof(unsigned long long) * 24);
  if (reset) { unsigned long long z[24] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(zra_mf_prof), z, sizeof(z)); }
}
#endif
