// zra_amd — ENCODE stage 1 for gfx950: the match finder (sequence producer) of zstd 1.4.9, bit-exact.
//
// Replaces the match-finding ~78-97 % of the reference's per-frame ZSTD_compress2 work (zra.cpp:219,331).
// One independent frame per workgroup; the workgroup is a single 64-lane wave. The greedy/lazy parse is a
// strict dependency chain (every table insertion depends on where the previous match ended), so one lane walks
// it while the whole wave clears the hash tables with coalesced 16-byte stores; throughput comes from thousands
// of frames in flight (up to 32 waves per CU), not from SIMD inside a frame. Hash tables live in HBM/L2 scratch
// (64 KiB frames at level 3 need 384 KiB: more than the 160 KiB LDS).
// Rules restated from SURVEY.md Appendix A.4.3 (validated there against libzstd 1.4.9).
#include "zra_dev.h"
#include "zra_kernels.h"

using namespace zra_dev;

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

__device__ __forceinline__ u32 mf_prologue(u32 bs, u32& o1, u32& o2, u32& saved) {
  u32 ip = bs + (bs == 0), maxRep = ip;
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
  const u32 ilimit = be - 8;
  u32 ip0 = mf_prologue(bs, o1, o2, saved), ip1 = ip0 + 1;
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
      if (m0 > 1 && ld32(src + m0 - 1) == ld32(src + ip0)) match = m0 - 1;
      else if (m1 > 1 && ld32(src + m1 - 1) == ld32(src + ip1)) { ip0 = ip1; match = m1 - 1; }
      else { const u32 st = ((ip0 - anchor) >> 7) + step0; ip0 += st; ip1 += st; continue; }
      o2 = o1; o1 = ip0 - match; offVal = o1 + 3; ml = 4;
      while (ip0 > anchor && match > 0 && src[ip0 - 1] == src[match - 1]) { ip0--; match--; ml++; }
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

// ---- A.4.3 "dfast" (level 3-4): long table keyed by 8 bytes, short table keyed by minMatch bytes
__device__ u32 mf_dfast(const ZraEncParams& P, u32* HL, u32* HS, const u8* src, u32 bs, u32 be, u32* rep, Emit& E) {
  const u32 hlog = P.hashLog, clog = P.chainLog, mls = P.minMatch;
  u32 o1 = rep[0], o2 = rep[1], saved;
  u32 anchor = bs;
  const u32 ilimit = be - 8;
  u32 ip = mf_prologue(bs, o1, o2, saved);
  while (ip < ilimit) {
    const u32 top = ip;
    const u64 v8 = ld64(src + ip);
    const u32 hL = (u32)((v8 * 0xCF1BBCDCB7A56463ULL) >> (64 - hlog));
    const u32 hS = mls == 5 ? (u32)(((v8 << 24) * 889523592379ULL) >> (64 - clog)) : hashN(src + ip, clog, mls);
    const u32 curr = ip + 1;
    const u32 mL = HL[hL], mS = HS[hS];
    HL[hL] = curr; HS[hS] = curr;
    u32 ml, offVal;
    if (o1 > 0 && ld32(src + ip + 1 - o1) == ld32(src + ip + 1)) {
      ml = count_eq(src, ip + 5, ip + 5 - o1, be) + 4; ip++; offVal = 1;
    } else {
      u32 m;
      if (mL > 1 && ld64(src + mL - 1) == v8) {
        m = mL - 1; ml = count_eq(src, ip + 8, m + 8, be) + 8;
      } else if (mS > 1 && ld32(src + mS - 1) == (u32)v8) {
        const u32 h3 = hash8(src + ip + 1, hlog), m3 = HL[h3];
        HL[h3] = curr + 1;
        if (m3 > 1 && ld64(src + m3 - 1) == ld64(src + ip + 1)) { m = m3 - 1; ip++; ml = count_eq(src, ip + 8, m + 8, be) + 8; }
        else { m = mS - 1; ml = count_eq(src, ip + 4, m + 4, be) + 4; }
      } else { ip += ((ip - anchor) >> 8) + 1; continue; }
      const u32 off = ip - m;
      while (ip > anchor && m > 0 && src[ip - 1] == src[m - 1]) { ip--; m--; ml++; }
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

// ---- A.4.3 hash chain (greedy depth 0 / lazy 1 / lazy2 2)
struct HC {
  u32* hashT; u32* chainT; u32 hlog, mls, cmask, chainSize, searchLog, nextToUpdate;
  __device__ u32 search(const u8* src, u32 ip, u32 be, u32& offCode) {
    const u32 target = ip + 1;
    for (u32 idx = nextToUpdate; idx < target; idx++) {
      const u32 h = hashN(src + idx - 1, hlog, mls);
      chainT[idx & cmask] = hashT[h];
      hashT[h] = idx;
    }
    nextToUpdate = target;
    u32 mi = hashT[hashN(src + ip, hlog, mls)];
    const u32 curr = target, minChain = curr > chainSize ? curr - chainSize : 0;
    int attempts = 1 << searchLog;
    u32 ml = 3;
    offCode = 999999999u;
    for (; mi >= 1 && attempts > 0; attempts--) {
      const u32 m = mi - 1;
      u32 cur = 0;
      if (src[m + ml] == src[ip + ml]) cur = count_eq(src, ip, m, be);
      if (cur > ml) { ml = cur; offCode = curr - mi + 2; if (ip + cur == be) break; }
      if (mi <= minChain) break;
      mi = chainT[mi & cmask];
    }
    return ml;
  }
};

__device__ u32 mf_lazy(HC& H, const u8* src, u32 bs, u32 be, u32* rep, Emit& E, int depth) {
  u32 o1 = rep[0], o2 = rep[1], saved;
  u32 anchor = bs;
  const u32 ilimit = be - 8;
  u32 ip = mf_prologue(bs, o1, o2, saved);
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
__device__ __forceinline__ u32 rfl(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ u64 rfl64(u64 v) { return (u64)rfl((u32)v) | ((u64)rfl((u32)(v >> 32)) << 32); }
__device__ __forceinline__ u32 bcast(u32 v, u32 l) { return (u32)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ u64 bcast64(u64 v, u32 l) { return (u64)bcast((u32)v, l) | ((u64)bcast((u32)(v >> 32), l) << 32); }

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

template <typename T, bool TAG>
__device__ u32 mf_dfast_wave(const ZraEncParams& P, T* HL, T* HS, const u8* src, u32 bs, u32 be, u32* rep, u64* seqs, u32* nOut,
                             u32* dupL, u32* dupS, int lane, u32 tune) {
  const u32 hlog = P.hashLog, clog = P.chainLog, mls = P.minMatch;
  u32 o1 = rep[0], o2 = rep[1], saved;
  u32 anchor = bs, nseq = 0;
  const u32 ilimit = be - 8;
  u32 ip = mf_prologue(bs, o1, o2, saved);
  u32 W = 8, epoch = 1;
  auto hashS64 = [&](u64 v) -> u32 {
    switch (mls) {
      case 5: return (u32)(((v << 24) * 889523592379ULL) >> (64 - clog));
      case 6: return (u32)(((v << 16) * 227718039650203ULL) >> (64 - clog));
      case 7: return (u32)(((v << 8) * 58295818150454627ULL) >> (64 - clog));
      default: return ((u32)v * 2654435761u) >> (32 - clog);
    }
  };
  auto hashL64 = [&](u64 v) -> u32 { return (u32)((v * 0xCF1BBCDCB7A56463ULL) >> (64 - hlog)); };
  // TAG mode (32-bit entries, positions < 65536): the upper half of an entry carries 16 more hash bits of the bytes the
  // candidate test compares (8 for the long table, 4 for the short one); a tag mismatch proves the candidate test would
  // fail, so the random read of the candidate's bytes is skipped. Results are unchanged.
  auto tagL64 = [&](u64 v) -> u32 { return TAG ? ((u32)((v * 0xCF1BBCDCB7A56463ULL) >> (48 - hlog)) & 0xFFFFu) << 16 : 0u; };
  auto tagS64 = [&](u64 v) -> u32 { return TAG ? (((u32)v * 2654435761u) >> 16) << 16 : 0u; };
  while (ip < ilimit) {
    const u32 run = ip - anchor, s = (run >> 8) + 1;
    u32 nAct = min(W, min((256 * s - run + s - 1) / s, (ilimit - ip + s - 1) / s));
    bool active = (u32)lane < nAct;
    const u32 p = ip + (u32)lane * s;
    const u64 v8 = active ? ld64(src + p) : 0;
    const u32 hL = hashL64(v8), hS = hashS64(v8);
    const u32 tL = tagL64(v8), tS = tagS64(v8);
    u32 mL = 0, mS = 0;
    bool tagLok = true, tagSok = true;
    if (active) {
      const u32 rL = HL[hL], rS = HS[hS];
      if (TAG) { mL = rL & 0xFFFFu; mS = rS & 0xFFFFu; tagLok = (rL & 0xFFFF0000u) == tL; tagSok = (rS & 0xFFFF0000u) == tS; }
      else { mL = rL; mS = rS; }
    }
    if (nAct > 1) {
      const u32 tag = (epoch << 6) | (63u - (u32)lane);
      if (active) { atomicMax(&dupL[hL & 511], tag); atomicMax(&dupS[hS & 511], tag); }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      const bool earlier = active && (((dupL[hL & 511] & 63u) != 63u - (u32)lane) || ((dupS[hS & 511] & 63u) != 63u - (u32)lane));
      const u64 cut = __ballot(earlier);
      if (cut) { nAct = (u32)__builtin_ctzll(cut); active = (u32)lane < nAct; }
      epoch++;
    }
    const bool repHit = active && o1 > 0 && ld32(src + p + 1 - o1) == (u32)(v8 >> 8);
    const bool longHit = active && mL > 1 && tagLok && ld64(src + mL - 1) == v8;
    const bool shortHit = active && mS > 1 && tagSok && ld32(src + mS - 1) == (u32)v8;
    const u64 hm = __ballot(repHit || longHit || shortHit);
    const u32 f = hm ? (u32)__builtin_ctzll(hm) : nAct - 1;
    if (active && (u32)lane <= f) { HL[hL] = (T)((p + 1) | tL); HS[hS] = (T)((p + 1) | tS); }
    if (!hm) { ip += nAct * s; W = tune == 3 ? 64u : tune == 4 ? 16u : min(64u, W * 2); continue; }
    W = tune == 1 ? min(64u, max(2u, f + 2)) : tune == 2 ? min(64u, max(8u, 4 * (f + 1))) : tune == 3 ? 64u : tune == 4 ? 16u : tune == 5 ? min(64u, max(4u, f + 4)) : min(64u, max(4u, 2 * (f + 1)));
    // ---- the hit lane's values, wave-uniform from here on
    const u32 top = ip + f * s;
    const u64 v8f = bcast64(v8, f);
    const u32 mLf = bcast(mL, f), mSf = bcast(mS, f);
    const bool isRep = (__ballot(repHit) >> f) & 1, isLong = (__ballot(longHit) >> f) & 1;
    const u32 curr = top + 1;
    ip = top;
    u32 ml, offVal;
    if (isRep) {
      ml = wave_count_eq(src, ip + 5, ip + 5 - o1, be, lane) + 4; ip++; offVal = 1;
    } else {
      u32 m;
      if (isLong) { m = mLf - 1; ml = wave_count_eq(src, ip + 8, m + 8, be, lane) + 8; }
      else {
        const u64 v9 = rfl64(ld64(src + ip + 1));
        const u32 h3 = hashL64(v9);
        const u32 r3 = rfl(HL[h3]);
        const u32 m3 = TAG ? (r3 & 0xFFFFu) : r3;
        const bool tag3ok = !TAG || (r3 & 0xFFFF0000u) == tagL64(v9);
        if (lane == 0) HL[h3] = (T)((curr + 1) | tagL64(v9));
        if (m3 > 1 && tag3ok && rfl64(ld64(src + m3 - 1)) == v9) { m = m3 - 1; ip++; ml = wave_count_eq(src, ip + 8, m + 8, be, lane) + 8; }
        else { m = mSf - 1; ml = wave_count_eq(src, ip + 4, m + 4, be, lane) + 4; }
      }
      const u32 off = ip - m;
      const u32 back = wave_count_back(src, ip, m, anchor, lane);
      ip -= back; ml += back;
      o2 = o1; o1 = off; offVal = off + 3;
    }
    (void)v8f;
    if (lane == 0) seqs[nseq] = (u64)(ip - anchor) | ((u64)ml << 20) | ((u64)offVal << 40);
    nseq++;
    ip += ml; anchor = ip;
    if (ip <= ilimit) {
      // complementary insertions (order per table preserved: q first, then ip-2 / ip-1)
      const u32 q = top + 2;
      { const u64 vq = ld64(src + q);
        if (lane == 0) HL[hashL64(vq)] = (T)((q + 1) | tagL64(vq));
        if (lane == 1) HS[hashS64(vq)] = (T)((q + 1) | tagS64(vq)); }
      { const u64 va = ld64(src + ip - 2), vb = va >> 8 | ((u64)src[ip + 6] << 56);
        if (lane == 0) HL[hashL64(va)] = (T)((ip - 1) | tagL64(va));
        if (lane == 1) HS[hashS64(vb)] = (T)(ip | tagS64(vb)); }
      while (ip <= ilimit && o2 > 0 && rfl(ld32(src + ip)) == rfl(ld32(src + ip - o2))) {
        const u32 rl = wave_count_eq(src, ip + 4, ip + 4 - o2, be, lane) + 4;
        const u32 t = o2; o2 = o1; o1 = t;
        const u64 vi = ld64(src + ip);
        if (lane == 0) HS[hashS64(vi)] = (T)((ip + 1) | tagS64(vi));
        if (lane == 1) HL[hashL64(vi)] = (T)((ip + 1) | tagL64(vi));
        if (lane == 0) seqs[nseq] = (u64)0 | ((u64)rl << 20) | ((u64)1 << 40);
        nseq++;
        ip += rl; anchor = ip;
      }
    }
  }
  rep[0] = o1 ? o1 : saved; rep[1] = o2 ? o2 : saved;
  *nOut = nseq;
  return be - anchor;
}


// ================================================================================================
// Window-resolve dfast: the batch formulation above, taken one step further. A window of up to 64 consecutive parse
// positions is looked up ONCE (one table-gather round trip + one tag-filtered candidate round trip), then the parse is
// resolved INSIDE the window: after a match the positions behind it are still in registers, so the next sequences of the
// window need no further trip to the tables. This is exact because no two lanes of a window share a bucket (exactly
// verified: the LDS scatter only flags suspects, flagged lanes are compared against all earlier lanes), so inserts made while
// resolving the window (visited positions, the ip+1 long probe, the complementary and repcode insertions — all positions of
// the window) can never change what another lane of the window would have read. Rep-offset tests depend on the parse state
// and are re-evaluated per sequence (a cached, sequential read).
template <typename T, bool TAG>
__device__ u32 mf_dfast_window(const ZraEncParams& P, T* HL, T* HS, const u8* src, u32 bs, u32 be, u32* rep, u64* seqs, u32* nOut,
                               u32* dupL, u32* dupS, int lane, u32 wcap) {
  const u32 hlog = P.hashLog, clog = P.chainLog, mls = P.minMatch;
  u32 o1 = rep[0], o2 = rep[1], saved;
  u32 anchor = bs, nseq = 0;
  const u32 ilimit = be - 8;
  u32 ip = mf_prologue(bs, o1, o2, saved);
  u32 epoch = 1;
  auto hashS64 = [&](u64 v) -> u32 {
    switch (mls) {
      case 5: return (u32)(((v << 24) * 889523592379ULL) >> (64 - clog));
      case 6: return (u32)(((v << 16) * 227718039650203ULL) >> (64 - clog));
      case 7: return (u32)(((v << 8) * 58295818150454627ULL) >> (64 - clog));
      default: return ((u32)v * 2654435761u) >> (32 - clog);
    }
  };
  auto hashL64 = [&](u64 v) -> u32 { return (u32)((v * 0xCF1BBCDCB7A56463ULL) >> (64 - hlog)); };
  auto tagL64 = [&](u64 v) -> u32 { return TAG ? ((u32)((v * 0xCF1BBCDCB7A56463ULL) >> (48 - hlog)) & 0xFFFFu) << 16 : 0u; };
  auto tagS64 = [&](u64 v) -> u32 { return TAG ? (((u32)v * 2654435761u) >> 16) << 16 : 0u; };
  while (ip < ilimit) {
    // ---------------------------------------------------------------- window build
    const u32 wip = ip;
    const u32 run = ip - anchor, s = (run >> 8) + 1;
    u32 nAct = min(wcap, min((256 * s - run + s - 1) / s, (ilimit - ip + s - 1) / s));
    bool active = (u32)lane < nAct;
    const u32 p = wip + (u32)lane * s;
    const u64 v8 = active ? ld64(src + p) : 0;
    const u32 hL = hashL64(v8), hS = hashS64(v8);
    const u32 tL = tagL64(v8), tS = tagS64(v8);
    if (nAct > 1) {
      const u32 tag = (epoch << 6) | (63u - (u32)lane);
      if (active) { atomicMax(&dupL[hL & 511], tag); atomicMax(&dupS[hS & 511], tag); }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      const bool suspect = active && (((dupL[hL & 511] & 63u) != 63u - (u32)lane) || ((dupS[hS & 511] & 63u) != 63u - (u32)lane));
      u64 sm = __ballot(suspect);
      while (sm) {                                   // exact check of the suspects, in ascending lane order
        const u32 i = (u32)__builtin_ctzll(sm); sm &= sm - 1;
        const u32 hLi = bcast(hL, i), hSi = bcast(hS, i);
        if (__ballot((u32)lane < i && (hL == hLi || hS == hSi))) { nAct = i; active = (u32)lane < nAct; break; }
      }
      epoch++;
    }
    u32 mL = 0, mS = 0;
    bool tagLok = true, tagSok = true;
    if (active) {
      const u32 rL = HL[hL], rS = HS[hS];
      if (TAG) { mL = rL & 0xFFFFu; mS = rS & 0xFFFFu; tagLok = (rL & 0xFFFF0000u) == tL; tagSok = (rS & 0xFFFF0000u) == tS; }
      else { mL = rL; mS = rS; }
    }
    // candidate tests that do not depend on the parse state
    const bool longHit = active && mL > 1 && tagLok && ld64(src + mL - 1) == v8;
    const bool shortHit = active && mS > 1 && tagSok && ld32(src + mS - 1) == (u32)v8;
    // insert position `pos` into the long / short table; bucket + tag come from the window's registers when pos is in it
    auto insert = [&](u32 pos, bool doL, bool doS) {
      u32 hl, hs, tl, ts;
      if (s == 1 && pos >= wip && pos - wip < nAct) {
        const u32 l = pos - wip;
        hl = bcast(hL, l); hs = bcast(hS, l); tl = bcast(tL, l); ts = bcast(tS, l);
      } else {
        const u64 v = rfl64(ld64(src + pos));
        hl = hashL64(v); hs = hashS64(v); tl = tagL64(v); ts = tagS64(v);
      }
      if (lane == 0 && doL) HL[hl] = (T)((pos + 1) | tl);
      if (lane == 1 && doS) HS[hs] = (T)((pos + 1) | ts);
    };
    // ---------------------------------------------------------------- resolve the window
    // Per sequence the dependent round trips are kept to two: (1) forward count + backward extension + the NEXT sequence's
    // rep-offset gather are issued together, (2) the immediate-repcode test. The rep gather is speculative on o1 (redone if
    // the repcode loop swaps the offsets).
    u32 cur = 0;
    u32 repFor = 0xFFFFFFFFu, repVal = 0;            // repVal = ld32(src + p + 1 - repFor), gathered ahead of time
    for (;;) {
      const bool live = active && (u32)lane >= cur;
      if (repFor != o1) { repVal = (active && o1 > 0 && p + 1 >= o1) ? ld32(src + p + 1 - o1) : 0; repFor = o1; }
      const bool repHit = live && o1 > 0 && p + 1 >= o1 && repVal == (u32)(v8 >> 8);
      const u64 hm = __ballot(live && (repHit || longHit || shortHit));
      const u32 f = hm ? (u32)__builtin_ctzll(hm) : nAct - 1;
      if (live && (u32)lane <= f) { HL[hL] = (T)((p + 1) | tL); HS[hS] = (T)((p + 1) | tS); }   // visited positions
      if (!hm) { ip = wip + nAct * s; break; }
      const u32 top = wip + f * s;
      const u32 mLf = bcast(mL, f), mSf = bcast(mS, f);
      const bool isRep = (__ballot(repHit) >> f) & 1, isLong = (__ballot(longHit) >> f) & 1;
      const u32 curr = top + 1;
      ip = top;
      // ---- which match: start position `ip`, source `m`, bytes already known equal `known`
      u32 m, known, offVal;
      if (isRep) { ip = top + 1; m = ip - o1; known = 4; offVal = 1; }
      else if (isLong) { m = mLf - 1; known = 8; offVal = 0; }
      else {
        // short hit: probe the long table at ip+1 (A.4.3 case 3) — lane f+1 already holds that lookup when it is in the window
        u64 v9; u32 h3, m3, t3; bool tag3ok;
        if (s == 1 && f + 1 < nAct) {
          v9 = bcast64(v8, f + 1); h3 = bcast(hL, f + 1); m3 = bcast(mL, f + 1); t3 = bcast(tL, f + 1);
          tag3ok = (__ballot(tagLok) >> (f + 1)) & 1;
        } else {
          v9 = rfl64(ld64(src + ip + 1)); h3 = hashL64(v9); t3 = tagL64(v9);
          const u32 r3 = rfl((u32)HL[h3]);
          m3 = TAG ? (r3 & 0xFFFFu) : r3;
          tag3ok = !TAG || (r3 & 0xFFFF0000u) == t3;
        }
        if (lane == 0) HL[h3] = (T)((curr + 1) | t3);
        if (m3 > 1 && tag3ok && rfl64(ld64(src + m3 - 1)) == v9) { m = m3 - 1; ip = top + 1; known = 8; }
        else { m = mSf - 1; known = 4; }
        offVal = 0;
      }
      const u32 off = ip - m;
      const u32 o1n = isRep ? o1 : off;             // o1 after this sequence (unless the repcode loop swaps)
      // ---- issue together: forward compare (64 x 8 B), backward compare (64 x 1 B), next rep gather
      const u32 fa = ip + known + 8 * (u32)lane, fb = m + known + 8 * (u32)lane;
      const bool fv = fa + 8 <= be;
      const u64 xa = fv ? ld64(src + fa) : 0, xb = fv ? ld64(src + fb) : 0;
      const u32 lim = isRep ? 0u : min(ip - anchor, m);
      const bool bv = (u32)lane < lim;
      const u32 ya = bv ? src[ip - 1 - lane] : 0u, yb = bv ? src[m - 1 - lane] : 1u;
      const u32 rnext = (active && o1n > 0 && p + 1 >= o1n) ? ld32(src + p + 1 - o1n) : 0;
      u32 ml;
      {
        const u64 d = xa ^ xb;
        const u32 eq = d ? ((u32)__builtin_ctzll(d) >> 3) : 8;
        const u64 stop = __ballot(!fv || d != 0);
        const u32 l = stop ? (u32)__builtin_ctzll(stop) : 64u;
        const bool clean = stop && ((__ballot(fv) >> l) & 1);       // first stopping lane compared a full 8-byte word
        if (clean) ml = known + 8 * l + bcast(eq, l);
        else ml = known + wave_count_eq(src, ip + known, m + known, be, lane);   // block end inside the window, or > 512 equal bytes
      }
      u32 back = 0;
      if (!isRep) {
        const u64 bad = ~__ballot(bv && ya == yb);
        back = bad ? (u32)__builtin_ctzll(bad) : wave_count_back(src, ip, m, anchor, lane);
        o2 = o1; o1 = off; offVal = off + 3;
      }
      ip -= back; ml += back;
      repVal = rnext; repFor = o1n;
      if (lane == 0) seqs[nseq] = (u64)(ip - anchor) | ((u64)ml << 20) | ((u64)offVal << 40);
      nseq++;
      ip += ml; anchor = ip;
      if (ip <= ilimit) {
        insert(top + 2, true, true);                  // complementary insertions (order per table: q first)
        insert(ip - 2, true, false);
        insert(ip - 1, false, true);
        for (;;) {
          if (!(ip <= ilimit && o2 > 0)) break;
          const u32 here = (s == 1 && ip >= wip && ip - wip < nAct) ? (u32)bcast64(v8, ip - wip) : rfl(ld32(src + ip));
          if (here != rfl(ld32(src + ip - o2))) break;
          const u32 rl = wave_count_eq(src, ip + 4, ip + 4 - o2, be, lane) + 4;
          const u32 t = o2; o2 = o1; o1 = t;
          insert(ip, true, true);
          if (lane == 0) seqs[nseq] = (u64)0 | ((u64)rl << 20) | ((u64)1 << 40);
          nseq++;
          ip += rl; anchor = ip;
        }
      }
      if (s != 1 || ip >= wip + nAct || ip >= ilimit) break;    // left the window: build the next one at ip
      cur = ip - wip;
    }
  }
  rep[0] = o1 ? o1 : saved; rep[1] = o2 ? o2 : saved;
  *nOut = nseq;
  return be - anchor;
}

}  // namespace

// One wave per frame; `block` = index of the <=128 KiB block being parsed in this round (A.4.2 driver).
extern "C" __global__ void __launch_bounds__(64)
zra_mf_kernel(ZraEncArgs a, u32 block) {
  const u32 f = blockIdx.x;
  const int lane = threadIdx.x;
  const u64 fstart = (u64)(a.firstFrame + f) * a.frameSize;
  const u64 remaining = a.inSize - fstart;
  const u32 fsize = (u32)(remaining < a.frameSize ? remaining : a.frameSize);
  const ZraEncParams& P = (fsize == a.frameSize) ? a.full : a.tail;   // the short last frame has its own cparams (A.4.1)
  const u32 blockSize = P.blockSize;
  const u32 bs = block * blockSize;
  if (bs >= fsize && !(fsize == 0 && block == 0)) return;
  const u32 be = min(fsize, bs + blockSize);
  const u8* src = a.in + fstart;
  ZraEncFrameState* st = &a.state[f];
  u32* hashT = a.tables + (size_t)f * a.tableStride;
  u32* chainT = hashT + ((size_t)1 << P.hashLog);

  // frames that are one block of <= 64 KiB keep 16-bit table entries (index = position+1 <= 65529 fits): half the table
  // footprint in HBM/L2 and half the clear traffic; everything else uses 32-bit entries
  const bool narrow = false;   // 16-bit entries superseded by tagged 32-bit entries (see mf_dfast_wave TAG mode)
  const bool tagged = P.strategy == 2 && fsize <= 65536;
  if (block == 0) {
    // fresh frame: zeroed tables, repcodes {1,4,8}, nextToUpdate 1 (A.4.8); 16-byte coalesced clears by the whole wave
    const size_t entries = ((size_t)1 << P.hashLog) + ((size_t)1 << P.chainLog);
    const size_t words = narrow ? entries / 2 : entries;
    uint4* t4 = (uint4*)hashT;
    for (size_t i = lane; i < words / 4; i += 64) t4[i] = make_uint4(0, 0, 0, 0);
    for (size_t i = (words / 4) * 4 + lane; i < words; i += 64) hashT[i] = 0;
    if (lane == 0) { st->rep[0] = 1; st->rep[1] = 4; st->rep[2] = 8; st->nextToUpdate = 1; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
  const u32 L = be - bs;
  ZraEncBlockOut* bo = &a.blockOut[f];
  if (L < 7) {                                         // too small to compress (A.4.2) -> raw block
    if (lane == 0) { bo->nbSeq = 0; bo->lastLL = L; bo->skip = 1; }
    return;
  }
  u64* seqs = a.seqs + (size_t)f * a.seqStride;
  u32 rep[3] = {st->rep[0], st->rep[1], st->rep[2]};
  u32 lastLL, nseq = 0;
  if (P.strategy == 2) {
    __shared__ u32 dupL[512], dupS[512];
    for (int i = lane; i < 512; i += 64) { dupL[i] = 0; dupS[i] = 0; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    const u32 wcap = (a.mfTune >= 8 && a.mfTune <= 64 && a.mfTune != 9) ? a.mfTune : 64u;   // window width (bring-up knob ZRA_MF_TUNE=8..64)
    if (a.mfTune == 9) {                             // previous formulation (one batch per sequence), kept for A/B measurements
      if (tagged) lastLL = mf_dfast_wave<u32, true>(P, hashT, chainT, src, bs, be, rep, seqs, &nseq, dupL, dupS, lane, 0);
      else lastLL = mf_dfast_wave<u32, false>(P, hashT, chainT, src, bs, be, rep, seqs, &nseq, dupL, dupS, lane, 0);
    } else if (tagged) lastLL = mf_dfast_window<u32, true>(P, hashT, chainT, src, bs, be, rep, seqs, &nseq, dupL, dupS, lane, wcap);
    else lastLL = mf_dfast_window<u32, false>(P, hashT, chainT, src, bs, be, rep, seqs, &nseq, dupL, dupS, lane, wcap);
    if (lane == 0) {
      bo->nbSeq = nseq; bo->lastLL = lastLL; bo->skip = 0;
      bo->rep[0] = rep[0]; bo->rep[1] = rep[1]; bo->rep[2] = rep[2];
    }
    return;
  }
  if (lane != 0) return;
  bo->skip = 0;
  Emit E; E.seqs = seqs; E.n = 0;
  // limited update after a very long match (A.4.3 hash chain prologue; harmless for the other finders)
  u32 ntu = st->nextToUpdate;
  {
    const u32 cur = bs + 1;
    if (cur > ntu + 384) { const u32 d = cur - ntu - 384; ntu = cur - (d < 192 ? d : 192); }
  }
  if (P.strategy == 1) lastLL = mf_fast(P, hashT, src, bs, be, rep, E);
  else {
    HC H; H.hashT = hashT; H.chainT = chainT; H.hlog = P.hashLog; H.mls = P.minMatch < 4 ? 4 : P.minMatch > 6 ? 6 : P.minMatch;
    H.chainSize = 1u << P.chainLog; H.cmask = H.chainSize - 1; H.searchLog = P.searchLog; H.nextToUpdate = ntu;
    lastLL = mf_lazy(H, src, bs, be, rep, E, (int)P.strategy - 3);
    ntu = H.nextToUpdate;
  }
  st->nextToUpdate = ntu;
  bo->nbSeq = E.n; bo->lastLL = lastLL;
  bo->rep[0] = rep[0]; bo->rep[1] = rep[1]; bo->rep[2] = rep[2];   // confirmed by stage 2 only if the block is emitted compressed
}
