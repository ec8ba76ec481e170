/* ORACLE — TEST INFRASTRUCTURE ONLY (see zo_common.h).
 *
 * zstd 1.4.9 frame ENCODER restated from SURVEY.md Appendix A.4 (parameters, frame/block driver,
 * fast / dfast / greedy / lazy / lazy2 match finders, literal + sequence entropy stages).
 * Stands in for ZSTD_compress2 at the reference call sites zra.cpp:219 and zra.cpp:331 with the
 * parameters the reference sets at zra.cpp:210-213 / 305-308 (level, no content size, checksum
 * flag, no dict id). Scalar and sequential on purpose: it is the checker for the HIP kernels.
 */
#include "zo_internal.h"
#include <stdlib.h>

/* ------------------------------------------------------------------ A.4.1 parameters */
size_t zo_compress_bound(size_t n) { return n + (n >> 8) + (n < (128u << 10) ? (((128u << 10) - n) >> 11) : 0); }

static const u16 CP16[23][7] = {{14,14,15,2,4,0,2},{14,14,15,1,5,0,1},{14,14,15,1,4,0,1},{14,14,15,2,4,0,2},{14,14,14,4,4,2,3},{14,14,14,3,4,4,4},
  {14,14,14,4,4,8,5},{14,14,14,6,4,8,5},{14,14,14,8,4,8,5},{14,15,14,5,4,8,6},{14,15,14,9,4,8,6},{14,15,14,3,4,12,7},{14,15,14,4,3,24,7},
  {14,15,14,5,3,32,8},{14,15,15,6,3,64,8},{14,15,15,7,3,256,8},{14,15,15,5,3,48,9},{14,15,15,6,3,128,9},{14,15,15,7,3,256,9},{14,15,15,8,3,256,9},{14,15,15,8,3,512,9},{14,15,15,9,3,512,9},{14,15,15,10,3,999,9}};
static const u16 CP128[23][7] = {{17,15,16,2,5,0,2},{17,12,13,1,6,0,1},{17,13,15,1,5,0,1},{17,15,16,2,5,0,2},{17,17,17,2,4,0,2},{17,16,17,3,4,2,3},
  {17,17,17,3,4,4,4},{17,17,17,3,4,8,5},{17,17,17,4,4,8,5},{17,17,17,5,4,8,5},{17,17,17,6,4,8,5},{17,17,17,5,4,8,6},{17,18,17,7,4,12,6},
  {17,18,17,3,4,12,7},{17,18,17,4,3,32,7},{17,18,17,6,3,256,7},{17,18,17,6,3,128,8},{17,18,17,8,3,256,8},{17,18,17,10,3,512,8},{17,18,17,5,3,256,9},{17,18,17,7,3,512,9},{17,18,17,9,3,512,9},{17,18,17,11,3,999,9}};
static const u16 CP256[23][7] = {{18,16,16,1,4,0,2},{18,13,14,1,6,0,1},{18,14,14,1,5,0,2},{18,16,16,1,4,0,2},{18,16,17,2,5,2,3},{18,18,18,3,5,2,3},
  {18,18,19,3,5,4,4},{18,18,19,4,4,4,4},{18,18,19,4,4,8,5},{18,18,19,5,4,8,5},{18,18,19,6,4,8,5},{18,18,19,5,4,12,6},{18,19,19,7,4,12,6},
  {18,18,19,4,4,16,7},{18,18,19,4,3,32,7},{18,18,19,6,3,128,7},{18,19,19,6,3,128,8},{18,19,19,8,3,256,8},{18,19,19,6,3,128,9},{18,19,19,8,3,256,9},{18,19,19,10,3,512,9},{18,19,19,12,3,512,9},{18,19,19,13,3,999,9}};

/* row 0 of the level tables of ZSTD_getCParams_internal: "base for negative levels" (strategy fast); the level itself becomes the
 * acceleration: targetLength = -level */
static const u16 CPNEG16[7] = {14,12,13,1,5,1,1}, CPNEG128[7] = {17,12,12,1,5,1,1}, CPNEG256[7] = {18,12,13,1,5,1,1};

/* the "default" table (srcSize > 256 KB), levels 0..22 */
static const u16 CPDEF[23][7] = {{21,16,17,1,5,0,2},{19,13,14,1,7,0,1},{20,15,16,1,6,0,1},{21,16,17,1,5,0,2},{21,18,18,1,5,0,2},{21,18,19,2,5,2,3},
  {21,19,19,3,5,4,3},{21,19,19,3,5,8,4},{21,19,19,3,5,16,5},{21,19,20,4,5,16,5},{22,20,21,4,5,16,5},{22,21,22,4,5,16,5},{22,21,22,5,5,16,5},
  {22,21,22,5,5,32,6},{22,22,23,5,5,32,6},{22,23,23,6,5,32,6},
  {22,22,22,5,5,48,7},{23,23,22,5,4,64,7},{23,23,22,6,3,64,8},{23,24,22,7,3,256,9},{25,25,23,7,3,256,9},{26,26,24,7,3,512,9},{27,27,25,9,3,999,9}};
static const u16 CPNEGDEF[7] = {19,12,13,1,6,1,1};

int zo_get_cparams(int level, size_t S, zo_cparams* cp) {
  if (level == 0) level = 3;
  if (level > 22) level = 22;                                           /* ZSTD_maxCLevel */
  const u16* r = level < 0 ? (S <= (16u << 10) ? CPNEG16 : S <= (128u << 10) ? CPNEG128 : S <= (256u << 10) ? CPNEG256 : CPNEGDEF)
                          : (S <= (16u << 10) ? CP16[level] : S <= (128u << 10) ? CP128[level] : S <= (256u << 10) ? CP256[level] : CPDEF[level]);
  /* a frame larger than the level's window: matches are limited to the last 2^windowLog bytes (lowest_at below) */
  cp->windowLog = r[0]; cp->chainLog = r[1]; cp->hashLog = r[2]; cp->searchLog = r[3];
  cp->minMatch = r[4]; cp->targetLength = level < 0 ? (u32)(-level) : r[5]; cp->strategy = r[6];
  u32 srcLog = S < 64 ? 6 : hb32((u32)S - 1) + 1;
  if (cp->windowLog > srcLog) cp->windowLog = srcLog;
  if (cp->hashLog > cp->windowLog + 1) cp->hashLog = cp->windowLog + 1;
  u32 cycleLog = cp->chainLog - (cp->strategy >= 6);
  if (cycleLog > cp->windowLog) cp->chainLog -= cycleLog - cp->windowLog;
  if (cp->windowLog < 10) cp->windowLog = 10;
  return 0;
}

/* ------------------------------------------------------------------ encoder state */
typedef struct {
  u32 rep[3];
  zo_huf_ctable huf; int hufRepeat;            /* 0 none, 1 check, 2 valid */
  zo_fse_ctable ll, of, ml; int llRepeat, ofRepeat, mlRepeat;
} estate;

typedef struct {
  zo_cparams cp;
  u32* hashTable;   /* fast: the table; dfast: long table; lazy: hash table */
  u32* chainTable;  /* dfast: short table; lazy: chain table */
  u32 nextToUpdate;
  estate prev, next;
  zo_seq* seqs; size_t nbSeq;
  u8* lits; size_t litSize;
  /* optimal parsers (btopt / btultra / btultra2) */
  u32* hashTable3; u32 hashLog3;
  u32 idxShift;                 /* index = position + 1 + idxShift (btultra2 moves the window base after its first pass) */
  struct zo_opt_state* opt;
} cctx;

/* ------------------------------------------------------------------ A.4.3 match finders */
static inline u32 hash4(const u8* p, u32 bits) { return (rd32(p) * 2654435761u) >> (32 - bits); }
static inline u32 hash5(const u8* p, u32 bits) { return (u32)(((rd64(p) << 24) * 889523592379ULL) >> (64 - bits)); }
static inline u32 hash6(const u8* p, u32 bits) { return (u32)(((rd64(p) << 16) * 227718039650203ULL) >> (64 - bits)); }
static inline u32 hash7(const u8* p, u32 bits) { return (u32)(((rd64(p) << 8) * 58295818150454627ULL) >> (64 - bits)); }
static inline u32 hash8(const u8* p, u32 bits) { return (u32)((rd64(p) * 0xCF1BBCDCB7A56463ULL) >> (64 - bits)); }
static inline u32 hashN(const u8* p, u32 bits, u32 mls) {
  switch (mls) { case 5: return hash5(p, bits); case 6: return hash6(p, bits); case 7: return hash7(p, bits); case 8: return hash8(p, bits); default: return hash4(p, bits); }
}
static inline size_t count_eq(const u8* src, size_t a, size_t b, size_t end) {
  size_t l = 0;
  while (a + l < end && src[a + l] == src[b + l]) l++;
  return l;
}
static void emit(cctx* c, const u8* src, size_t anchor, size_t ll, size_t ml, u32 offsetValue) {
  memcpy(c->lits + c->litSize, src + anchor, ll); c->litSize += ll;
  c->seqs[c->nbSeq].litLength = (u32)ll; c->seqs[c->nbSeq].matchLength = (u32)ml; c->seqs[c->nbSeq].offsetValue = offsetValue;
  c->nbSeq++;
}

/* common prologue; returns start ip */
/* ZSTD_getLowestMatchIndex / ZSTD_getLowestPrefixIndex without a dictionary: the lowest index a match may have when the position
 * with index `curr` is searched — the first index of the frame, or curr - 2^windowLog once the frame is longer than the window */
static inline u32 lowest_at(const cctx* c, u32 curr) {
  u32 maxDist = 1u << c->cp.windowLog, lowValid = 1 + c->idxShift;
  return (curr - lowValid > maxDist) ? curr - maxDist : lowValid;
}
/* block prologue: skip the very first byte of the prefix, drop repeat offsets that reach below the window */
static size_t mf_prologue(const cctx* c, size_t bs, size_t prefixStartPos, u32* o1, u32* o2, u32* saved) {
  size_t ip = bs + (bs == prefixStartPos);
  u32 curr = (u32)ip + 1;
  u32 maxRep = curr - lowest_at(c, curr);
  *saved = 0;
  if (*o2 > maxRep) { *saved = *o2; *o2 = 0; }
  if (*o1 > maxRep) { *saved = *o1; *o1 = 0; }
  return ip;
}

static size_t mf_fast(cctx* c, const u8* src, size_t bs, size_t be, u32 rep[3]) {
  u32* T = c->hashTable; u32 hlog = c->cp.hashLog, mls = c->cp.minMatch;
  u32 tl = c->cp.targetLength; size_t step0 = tl + (tl == 0) + 1;
  u32 o1 = rep[0], o2 = rep[1], saved;
  size_t anchor = bs, ilimit = be >= 8 ? be - 8 : 0;   /* a 7-byte first block: iend-8 lies before the start, nothing is searched */
  u32 psi = lowest_at(c, (u32)be + 1);                 /* prefixStartIndex, from the block's end */
  size_t ip0 = mf_prologue(c, bs, psi - 1, &o1, &o2, &saved), ip1 = ip0 + 1;
  while (ip1 < ilimit) {
    size_t ip2 = ip0 + 2, top = ip0;
    u32 h0 = hashN(src + ip0, hlog, mls), h1 = hashN(src + ip1, hlog, mls);
    u32 m0 = T[h0], m1 = T[h1];
    size_t match, ml; u32 offVal;
    T[h0] = (u32)ip0 + 1; T[h1] = (u32)ip1 + 1;
    if (o1 > 0 && rd32(src + ip2 - o1) == rd32(src + ip2)) {
      size_t back = src[ip2 - 1] == src[ip2 - o1 - 1];
      ip0 = ip2 - back; match = ip2 - o1 - back; ml = 4 + back; offVal = 1;
    } else {
      if (m0 > psi && rd32(src + m0 - 1) == rd32(src + ip0)) match = m0 - 1;
      else if (m1 > psi && rd32(src + m1 - 1) == rd32(src + ip1)) { ip0 = ip1; match = m1 - 1; }
      else { size_t st = ((ip0 - anchor) >> 7) + step0; ip0 += st; ip1 += st; continue; }
      o2 = o1; o1 = (u32)(ip0 - match); offVal = o1 + 3; ml = 4;
      while (ip0 > anchor && match > psi - 1 && src[ip0 - 1] == src[match - 1]) { ip0--; match--; ml++; }
    }
    ml += count_eq(src, ip0 + ml, match + ml, be);
    emit(c, src, anchor, ip0 - anchor, ml, offVal);
    ip0 += ml; anchor = ip0;
    if (ip0 <= ilimit) {
      T[hashN(src + top + 2, hlog, mls)] = (u32)top + 3;
      T[hashN(src + ip0 - 2, hlog, mls)] = (u32)ip0 - 1;
      if (o2 > 0) {
        while (ip0 <= ilimit && rd32(src + ip0) == rd32(src + ip0 - o2)) {
          size_t rl = count_eq(src, ip0 + 4, ip0 + 4 - o2, be) + 4;
          u32 t = o2; o2 = o1; o1 = t;
          T[hashN(src + ip0, hlog, mls)] = (u32)ip0 + 1;
          emit(c, src, anchor, 0, rl, 1);
          ip0 += rl; anchor = ip0;
        }
      }
    }
    ip1 = ip0 + 1;
  }
  rep[0] = o1 ? o1 : saved; rep[1] = o2 ? o2 : saved;
  return be - anchor;
}

static size_t mf_dfast(cctx* c, const u8* src, size_t bs, size_t be, u32 rep[3]) {
  u32* HL = c->hashTable; u32* HS = c->chainTable;
  u32 hlog = c->cp.hashLog, clog = c->cp.chainLog, mls = c->cp.minMatch;
  u32 o1 = rep[0], o2 = rep[1], saved;
  size_t anchor = bs, ilimit = be >= 8 ? be - 8 : 0;   /* a 7-byte first block: iend-8 lies before the start, nothing is searched */
  u32 psi = lowest_at(c, (u32)be + 1);                 /* prefixLowestIndex, from the block's end */
  size_t ip = mf_prologue(c, bs, psi - 1, &o1, &o2, &saved);
  while (ip < ilimit) {
    size_t top = ip, ml; u32 offVal;
    u32 hL = hash8(src + ip, hlog), hS = hashN(src + ip, clog, mls);
    u32 curr = (u32)ip + 1, mL = HL[hL], mS = HS[hS];
    HL[hL] = HS[hS] = curr;
    if (o1 > 0 && rd32(src + ip + 1 - o1) == rd32(src + ip + 1)) {
      ml = count_eq(src, ip + 5, ip + 5 - o1, be) + 4; ip++; offVal = 1;
    } else {
      size_t m;
      if (mL > psi && rd64(src + mL - 1) == rd64(src + ip)) {
        m = mL - 1; ml = count_eq(src, ip + 8, m + 8, be) + 8;
      } else if (mS > psi && rd32(src + mS - 1) == rd32(src + ip)) {
        u32 h3 = hash8(src + ip + 1, hlog), m3 = HL[h3];
        HL[h3] = curr + 1;
        if (m3 > psi && rd64(src + m3 - 1) == rd64(src + ip + 1)) { m = m3 - 1; ip++; ml = count_eq(src, ip + 8, m + 8, be) + 8; }
        else { m = mS - 1; ml = count_eq(src, ip + 4, m + 4, be) + 4; }
      } else { ip += ((ip - anchor) >> 8) + 1; continue; }
      u32 off = (u32)(ip - m);
      while (ip > anchor && m > psi - 1 && src[ip - 1] == src[m - 1]) { ip--; m--; ml++; }
      o2 = o1; o1 = off; offVal = off + 3;
    }
    emit(c, src, anchor, ip - anchor, ml, offVal);
    ip += ml; anchor = ip;
    if (ip <= ilimit) {
      size_t q = top + 2;
      HL[hash8(src + q, hlog)] = (u32)q + 1;
      HL[hash8(src + ip - 2, hlog)] = (u32)ip - 1;
      HS[hashN(src + q, clog, mls)] = (u32)q + 1;
      HS[hashN(src + ip - 1, clog, mls)] = (u32)ip;
      while (ip <= ilimit && o2 > 0 && rd32(src + ip) == rd32(src + ip - o2)) {
        size_t rl = count_eq(src, ip + 4, ip + 4 - o2, be) + 4;
        u32 t = o2; o2 = o1; o1 = t;
        HS[hashN(src + ip, clog, mls)] = (u32)ip + 1;
        HL[hash8(src + ip, hlog)] = (u32)ip + 1;
        emit(c, src, anchor, 0, rl, 1);
        ip += rl; anchor = ip;
      }
    }
  }
  rep[0] = o1 ? o1 : saved; rep[1] = o2 ? o2 : saved;
  return be - anchor;
}

/* hash-chain search (ZSTD_HcFindBestMatch); returns best length, *offCode = offset+2 */
static size_t hc_search(cctx* c, const u8* src, size_t ip, size_t be, u32* offCode) {
  u32 hlog = c->cp.hashLog, mls = c->cp.minMatch < 4 ? 4 : c->cp.minMatch > 6 ? 6 : c->cp.minMatch;
  u32 chainSize = 1u << c->cp.chainLog, cmask = chainSize - 1;
  u32 target = (u32)ip + 1, idx = c->nextToUpdate;
  while (idx < target) {
    u32 h = hashN(src + idx - 1, hlog, mls);
    c->chainTable[idx & cmask] = c->hashTable[h];
    c->hashTable[h] = idx;
    idx++;
  }
  c->nextToUpdate = target;
  u32 mi = c->hashTable[hashN(src + ip, hlog, mls)];
  u32 curr = target, minChain = curr > chainSize ? curr - chainSize : 0;
  u32 lowLimit = lowest_at(c, curr);
  int attempts = 1 << c->cp.searchLog;
  size_t ml = 3;
  *offCode = 999999999u;
  for (; mi >= lowLimit && attempts > 0; attempts--) {
    size_t m = mi - 1, cur = 0;
    if (src[m + ml] == src[ip + ml]) cur = count_eq(src, ip, m, be);
    if (cur > ml) { ml = cur; *offCode = curr - mi + 2; if (ip + cur == be) break; }
    if (mi <= minChain) break;
    mi = c->chainTable[mi & cmask];
  }
  return ml;
}


/* ---- A.4.3 binary tree with delayed updates (btlazy2): ZSTD_updateDUBT / ZSTD_insertDUBT1 / ZSTD_DUBT_findBestMatch /
 * ZSTD_BtFindBestMatch of zstd_lazy.c (1.4.9), single-segment prefix mode. Index = position + 1 as everywhere here; the tree lives
 * in chainTable as pairs {smaller, larger} indexed by (index & btMask), btLog = chainLog - 1; an inserted but not yet sorted
 * position holds {previous head of its bucket, UNSORTED_MARK}. */
#define ZO_DUBT_UNSORTED 1u
static void bt_insert1(cctx* c, const u8* src, u32 curr, size_t iend, u32 nbCompares, u32 btLow) {
  u32* bt = c->chainTable; u32 btMask = (1u << (c->cp.chainLog - 1)) - 1;
  size_t commonSmaller = 0, commonLarger = 0;
  const u8* ip = src + curr - 1;
  u32* smallerPtr = bt + 2 * (curr & btMask);
  u32* largerPtr = smallerPtr + 1;
  u32 matchIndex = *smallerPtr;
  u32 dummy32;
  u32 windowLow = lowest_at(c, curr);
  while (nbCompares-- && matchIndex > windowLow) {
    u32* nextPtr = bt + 2 * (matchIndex & btMask);
    size_t ml = commonSmaller < commonLarger ? commonSmaller : commonLarger;
    const u8* match = src + matchIndex - 1;
    ml += count_eq(src, (size_t)(ip - src) + ml, (size_t)(match - src) + ml, iend);
    if ((size_t)(ip - src) + ml == iend) break;        /* equal: no way to know if inf or sup */
    if (match[ml] < ip[ml]) {
      *smallerPtr = matchIndex; commonSmaller = ml;
      if (matchIndex <= btLow) { smallerPtr = &dummy32; break; }
      smallerPtr = nextPtr + 1; matchIndex = nextPtr[1];
    } else {
      *largerPtr = matchIndex; commonLarger = ml;
      if (matchIndex <= btLow) { largerPtr = &dummy32; break; }
      largerPtr = nextPtr; matchIndex = nextPtr[0];
    }
  }
  *smallerPtr = *largerPtr = 0;
}
static size_t bt_search(cctx* c, const u8* src, size_t ip, size_t be, u32* offCode) {
  u32 hlog = c->cp.hashLog, mls = c->cp.minMatch < 4 ? 4 : c->cp.minMatch > 6 ? 6 : c->cp.minMatch;
  u32* bt = c->chainTable; u32 btMask = (1u << (c->cp.chainLog - 1)) - 1;
  u32 curr = (u32)ip + 1;
  *offCode = 999999999u;
  if (curr < c->nextToUpdate) return 0;                /* skipped area */
  for (u32 idx = c->nextToUpdate; idx < curr; idx++) { /* ZSTD_updateDUBT */
    u32 h = hashN(src + idx - 1, hlog, mls);
    u32* p = bt + 2 * (idx & btMask);
    p[0] = c->hashTable[h]; p[1] = ZO_DUBT_UNSORTED;
    c->hashTable[h] = idx;
  }
  c->nextToUpdate = curr;
  u32 h = hashN(src + ip, hlog, mls);
  u32 matchIndex = c->hashTable[h];
  u32 windowLow = lowest_at(c, curr);
  u32 btLow = btMask >= curr ? 0 : curr - btMask;
  u32 unsortLimit = btLow > windowLow ? btLow : windowLow;
  u32* nextCandidate = bt + 2 * (matchIndex & btMask);
  u32* unsortedMark = nextCandidate + 1;
  u32 nbCompares = 1u << c->cp.searchLog, nbCandidates = nbCompares, previousCandidate = 0;
  while (matchIndex > unsortLimit && *unsortedMark == ZO_DUBT_UNSORTED && nbCandidates > 1) {
    *unsortedMark = previousCandidate;                 /* the mark becomes a reversed chain */
    previousCandidate = matchIndex;
    matchIndex = *nextCandidate;
    nextCandidate = bt + 2 * (matchIndex & btMask);
    unsortedMark = nextCandidate + 1;
    nbCandidates--;
  }
  if (matchIndex > unsortLimit && *unsortedMark == ZO_DUBT_UNSORTED) *nextCandidate = *unsortedMark = 0;
  matchIndex = previousCandidate;                      /* batch sort the stacked candidates */
  while (matchIndex) {
    u32* nextIdxPtr = bt + 2 * (matchIndex & btMask) + 1;
    u32 nextIdx = *nextIdxPtr;
    bt_insert1(c, src, matchIndex, be, nbCandidates, unsortLimit);
    matchIndex = nextIdx;
    nbCandidates++;
  }
  size_t commonSmaller = 0, commonLarger = 0, bestLength = 0;
  u32* smallerPtr = bt + 2 * (curr & btMask);
  u32* largerPtr = smallerPtr + 1;
  u32 matchEndIdx = curr + 8 + 1, dummy32;
  matchIndex = c->hashTable[h];
  c->hashTable[h] = curr;
  while (nbCompares-- && matchIndex > windowLow) {
    u32* nextPtr = bt + 2 * (matchIndex & btMask);
    size_t ml = commonSmaller < commonLarger ? commonSmaller : commonLarger;
    size_t m = matchIndex - 1;
    ml += count_eq(src, ip + ml, m + ml, be);
    if (ml > bestLength) {
      if (ml > matchEndIdx - matchIndex) matchEndIdx = matchIndex + (u32)ml;
      if (4 * (int)(ml - bestLength) > (int)(hb32(curr - matchIndex + 1) - hb32(*offCode + 1))) { bestLength = ml; *offCode = 2 + curr - matchIndex; }
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
  *smallerPtr = *largerPtr = 0;
  c->nextToUpdate = matchEndIdx - 8;                   /* skip repetitive patterns */
  return bestLength;
}
static size_t lazy_search(cctx* c, const u8* src, size_t ip, size_t be, u32* offCode);
static size_t mf_lazy(cctx* c, const u8* src, size_t bs, size_t be, u32 rep[3], int depth) {
  u32 o1 = rep[0], o2 = rep[1], saved;
  size_t anchor = bs, ilimit = be >= 8 ? be - 8 : 0;   /* a 7-byte first block: iend-8 lies before the start, nothing is searched */
  size_t ip = mf_prologue(c, bs, 0, &o1, &o2, &saved);
  while (ip < ilimit) {
    size_t ml = 0, start = ip + 1; u32 off = 0; int stored = 0;
    if (o1 > 0 && rd32(src + ip + 1 - o1) == rd32(src + ip + 1)) {
      ml = count_eq(src, ip + 5, ip + 5 - o1, be) + 4;
      if (depth == 0) stored = 1;
    }
    if (!stored) {
      u32 oc2; size_t m2 = lazy_search(c, src, ip, be, &oc2);
      if (m2 > ml) { ml = m2; start = ip; off = oc2; }
      if (ml < 4) { ip += ((ip - anchor) >> 8) + 1; continue; }
      if (depth >= 1) {
        while (ip < ilimit) {
          ip++;
          if (off && o1 > 0 && rd32(src + ip) == rd32(src + ip - o1)) {
            size_t mr = count_eq(src, ip + 4, ip + 4 - o1, be) + 4;
            int g2 = (int)(mr * 3), g1 = (int)(ml * 3 - hb32(off + 1) + 1);
            if (mr >= 4 && g2 > g1) { ml = mr; off = 0; start = ip; }
          }
          {
            m2 = lazy_search(c, src, ip, be, &oc2);
            int g2 = (int)(m2 * 4 - hb32(oc2 + 1)), g1 = (int)(ml * 4 - hb32(off + 1) + 4);
            if (m2 >= 4 && g2 > g1) { ml = m2; off = oc2; start = ip; continue; }
          }
          if (depth == 2 && ip < ilimit) {
            ip++;
            if (off && o1 > 0 && rd32(src + ip) == rd32(src + ip - o1)) {
              size_t mr = count_eq(src, ip + 4, ip + 4 - o1, be) + 4;
              int g2 = (int)(mr * 4), g1 = (int)(ml * 4 - hb32(off + 1) + 1);
              if (mr >= 4 && g2 > g1) { ml = mr; off = 0; start = ip; }
            }
            {
              m2 = lazy_search(c, src, ip, be, &oc2);
              int g2 = (int)(m2 * 4 - hb32(oc2 + 1)), g1 = (int)(ml * 4 - hb32(off + 1) + 7);
              if (m2 >= 4 && g2 > g1) { ml = m2; off = oc2; start = ip; continue; }
            }
          }
          break;
        }
      }
      if (off) {
        u32 ro = off - 2;
        while (start > anchor && start - ro > 0 && src[start - 1] == src[start - ro - 1]) { start--; ml++; }
        o2 = o1; o1 = ro;
      }
    }
    emit(c, src, anchor, start - anchor, ml, off ? off + 1 : 1);
    anchor = ip = start + ml;
    while (ip <= ilimit && o2 > 0 && rd32(src + ip) == rd32(src + ip - o2)) {
      size_t rl = count_eq(src, ip + 4, ip + 4 - o2, be) + 4;
      u32 t = o2; o2 = o1; o1 = t;
      emit(c, src, anchor, 0, rl, 1);
      ip += rl; anchor = ip;
    }
  }
  rep[0] = o1 ? o1 : saved; rep[1] = o2 ? o2 : saved;
  return be - anchor;
}
static size_t lazy_search(cctx* c, const u8* src, size_t ip, size_t be, u32* offCode) {
  return c->cp.strategy == 6 ? bt_search(c, src, ip, be, offCode) : hc_search(c, src, ip, be, offCode);
}

/* ------------------------------------------------------------------ A.4.5 literals */
static size_t raw_literals(u8* dst, const u8* lit, size_t n, int rle) {
  size_t fl = 1 + (n > 31) + (n > 4095);
  u32 t = rle ? 1 : 0;
  if (fl == 1) dst[0] = (u8)(t + (n << 3));
  else if (fl == 2) wr16(dst, (u16)(t + (1 << 2) + (n << 4)));
  else wr32(dst, (u32)(t + (3 << 2) + (n << 4)));
  if (rle) { dst[fl] = lit[0]; return fl + 1; }
  memcpy(dst + fl, lit, n);
  return fl + n;
}

static size_t huf_encode_1x(u8* dst, size_t cap, const u8* src, size_t n, const zo_huf_ctable* ct) {
  zo_bitw bw;
  if (cap < 8) return 0;
  zo_bitw_init(&bw, dst, cap);
  for (size_t i = n; i > 0; i--) zo_bitw_add(&bw, ct->val[src[i - 1]], ct->nbBits[src[i - 1]]);
  return zo_bitw_close(&bw);
}
static size_t huf_encode_4x(u8* dst, size_t cap, const u8* src, size_t n, const zo_huf_ctable* ct) {
  size_t seg = (n + 3) / 4; u8* op = dst + 6;
  if (cap < 17 || n < 12) return 0;
  for (int i = 0; i < 4; i++) {
    size_t len = i < 3 ? seg : n - 3 * seg;
    size_t c = huf_encode_1x(op, (size_t)(dst + cap - op), src + i * seg, len, ct);
    if (!c) return 0;
    if (i < 3) wr16(dst + 2 * i, (u16)c);
    op += c;
  }
  return (size_t)(op - dst);
}
#define HUF_ERR ((size_t)-1)
static size_t huf_with_table(u8* ostart, u8* op, size_t cap, const u8* src, size_t n, int streams, const zo_huf_ctable* ct) {
  size_t c = streams == 1 ? huf_encode_1x(op, (size_t)(ostart + cap - op), src, n, ct) : huf_encode_4x(op, (size_t)(ostart + cap - op), src, n, ct);
  if (!c) return 0;
  op += c;
  if ((size_t)(op - ostart) >= n - 1) return 0;
  return (size_t)(op - ostart);
}
/* HUF_compress_internal; `old` is in/out (next block's table), *repeat in/out */
static size_t huf_compress(u8* dst, size_t cap, const u8* src, size_t n, int streams, zo_huf_ctable* old, int* repeat, int preferRepeat) {
  u32 count[256] = {0}; unsigned maxSym = 0; u32 largest = 0;
  if (!n || !cap) return 0;
  if (preferRepeat && *repeat == 2) return huf_with_table(dst, dst, cap, src, n, streams, old);
  for (size_t i = 0; i < n; i++) count[src[i]]++;
  for (unsigned s = 0; s < 256; s++) { if (count[s]) maxSym = s; if (count[s] > largest) largest = count[s]; }
  if (largest == n) { dst[0] = src[0]; return 1; }
  if (largest <= (n >> 7) + 4) return 0;
  if (*repeat == 1) {
    int bad = 0;
    for (unsigned s = 0; s <= maxSym; s++) bad |= (count[s] != 0) & (old->nbBits[s] == 0);
    if (bad) *repeat = 0;
  }
  if (preferRepeat && *repeat != 0) return huf_with_table(dst, dst, cap, src, n, streams, old);
  zo_huf_ctable nt;
  unsigned log = zo_fse_optimal_tablelog(11, n, maxSym, 1);
  zo_huf_build(&nt, count, maxSym, log);
  size_t h = zo_huf_write_ctable(dst, cap, &nt);
  if (!h) return HUF_ERR;
  if (*repeat != 0) {
    size_t oldSize = 0, newSize = 0;
    for (unsigned s = 0; s <= maxSym; s++) { oldSize += (size_t)old->nbBits[s] * count[s]; newSize += (size_t)nt.nbBits[s] * count[s]; }
    oldSize >>= 3; newSize >>= 3;
    if (oldSize <= h + newSize || h + 12 >= n) return huf_with_table(dst, dst, cap, src, n, streams, old);
  }
  if (h + 12 >= n) return 0;
  *repeat = 0;
  *old = nt;
  return huf_with_table(dst, dst + h, cap, src, n, streams, &nt);
}

static size_t min_gain(size_t n, unsigned strategy) { return (n >> (strategy >= 8 ? strategy - 1 : 6)) + 2; }

static size_t compress_literals(const estate* prev, estate* next, unsigned strategy, int disabled, u8* dst, size_t cap, const u8* lit, size_t n) {
  size_t minGain = min_gain(n, strategy);
  size_t lh = 3 + (n >= 1024) + (n >= 16384);
  int single = n < 256;
  next->huf = prev->huf; next->hufRepeat = prev->hufRepeat;
  if (disabled) return raw_literals(dst, lit, n, 0);   /* ZSTD_compressLiterals: disableLiteralCompression (fast strategy with targetLength > 0) */
  if (n <= (prev->hufRepeat == 2 ? 6u : 63u)) return raw_literals(dst, lit, n, 0);
  if (cap < lh + 1) return ZO_ERR(ZO_E_DSTSIZE_TOOSMALL);
  int repeat = prev->hufRepeat, preferRepeat = strategy < 4 ? n <= 1024 : 0;
  if (repeat == 2 && lh == 3) single = 1;
  size_t c = huf_compress(dst + lh, cap - lh, lit, n, single ? 1 : 4, &next->huf, &repeat, preferRepeat);
  unsigned hType = repeat != 0 ? 3 : 2;
  if (c == 0 || c == HUF_ERR || c >= n - minGain) { next->huf = prev->huf; next->hufRepeat = prev->hufRepeat; return raw_literals(dst, lit, n, 0); }
  if (c == 1) { next->huf = prev->huf; next->hufRepeat = prev->hufRepeat; return raw_literals(dst, lit, n, 1); }
  if (hType == 2) next->hufRepeat = 1;
  if (lh == 3) wr24(dst, (u32)(hType + ((!single) << 2) + ((u32)n << 4) + ((u32)c << 14)));
  else if (lh == 4) wr32(dst, (u32)(hType + (2 << 2) + ((u32)n << 4) + ((u32)c << 18)));
  else { wr32(dst, (u32)(hType + (3 << 2) + ((u32)n << 4) + ((u32)c << 22))); dst[4] = (u8)(c >> 10); }
  return lh + c;
}

/* ------------------------------------------------------------------ A.4.4 / A.4.7 sequences */
static unsigned ll_code(u32 v) { unsigned c = 35; while (zo_ll_base[c] > v) c--; return c; }
static unsigned ml_code(u32 v) { unsigned c = 52; while (zo_ml_base[c] > v) c--; return c; }

static u32 inv_prob_log256(unsigned i) {
  /* kInverseProbabilityLog256: floor(-log2(i/256)*256), [0] = 0. Integer-only evaluation. */
  static u32 tab[256]; static int init = 0;
  if (!init) {
    tab[0] = 0;
    for (unsigned k = 1; k < 256; k++) {
      /* find largest v with 2^(-v/256) >= k/256  <=>  k^256 <= 2^(2048 - v); use long double log2 then fix up exactly */
      extern double log2(double);
      double x = -log2((double)k / 256.0) * 256.0;
      u32 v = (u32)x;
      tab[k] = v;
    }
    init = 1;
  }
  return tab[i];
}

static size_t cross_entropy_cost(const s16* norm, unsigned accLog, const u32* count, unsigned max) {
  unsigned shift = 8 - accLog; size_t cost = 0;
  for (unsigned s = 0; s <= max; s++) {
    unsigned na = norm[s] != -1 ? (unsigned)norm[s] : 1;
    cost += (size_t)count[s] * inv_prob_log256(na << shift);
  }
  return cost >> 8;
}
static size_t entropy_cost(const u32* count, unsigned max, size_t total) {
  unsigned cost = 0;
  for (unsigned s = 0; s <= max; s++) {
    unsigned norm = (unsigned)((256 * count[s]) / total);
    if (count[s] != 0 && norm == 0) norm = 1;
    cost += count[s] * inv_prob_log256(norm);
  }
  return cost >> 8;
}
#define COST_ERR ((size_t)-1)
static size_t fse_bit_cost(const zo_fse_ctable* ct, const u32* count, unsigned max) {
  size_t cost = 0;
  if (ct->rle) {
    /* FSE_buildCTable_rle: tableLog 0, one symbol with deltaNbBits 0 */
    if (ct->maxSym < max) return COST_ERR;
    for (unsigned s = 0; s <= max; s++) {
      if (!count[s]) continue;
      return COST_ERR; /* badCost = 1<<8, bitCost of an rle table's symbol = 256 - ... >= badCost in practice; rle tables always carry repeat=none */
    }
    return 0;
  }
  if (ct->maxSym < max) return COST_ERR;
  unsigned tl = ct->tableLog;
  for (unsigned s = 0; s <= max; s++) {
    u32 badCost = (tl + 1) << 8;
    u32 minNb = ct->deltaNbBits[s] >> 16, thr = (minNb + 1) << 16;
    u32 d = thr - (ct->deltaNbBits[s] + (1u << tl));
    u32 bitCost = (minNb + 1) * 256 - ((d << 8) >> tl);
    if (!count[s]) continue;
    if (bitCost >= badCost) return COST_ERR;
    cost += (size_t)count[s] * bitCost;
  }
  return cost >> 8;
}
static size_t ncount_cost(const u32* count, unsigned max, size_t nbSeq, unsigned FSELog) {
  u8 w[512]; s16 norm[64];
  unsigned tl = zo_fse_optimal_tablelog(FSELog, nbSeq, max, 2);
  if (zo_fse_normalize(norm, tl, count, nbSeq, max, nbSeq >= 2048) <= 0) return COST_ERR;
  size_t r = zo_fse_write_ncount(w, sizeof(w), norm, max, tl);
  return r ? r : COST_ERR;
}

/* modes: 0 basic, 1 rle, 2 compressed, 3 repeat */
static unsigned select_encoding(int* repeatMode, const u32* count, unsigned max, size_t mostFrequent, size_t nbSeq, unsigned FSELog,
                                const zo_fse_ctable* prevCT, const s16* defNorm, unsigned defLog, int defaultAllowed, unsigned strategy) {
  if (mostFrequent == nbSeq) {
    *repeatMode = 0;
    if (defaultAllowed && nbSeq <= 2) return 0;
    return 1;
  }
  if (strategy < 4) {
    if (defaultAllowed) {
      size_t mult = 10 - strategy;
      size_t dynMin = (((size_t)1 << defLog) * mult) >> 3;
      if (*repeatMode == 2 && nbSeq < 1000) return 3;
      if (nbSeq < dynMin || mostFrequent < (nbSeq >> (defLog - 1))) { *repeatMode = 0; return 0; }
    }
  } else {
    size_t basic = defaultAllowed ? cross_entropy_cost(defNorm, defLog, count, max) : COST_ERR;
    size_t repeat = *repeatMode != 0 ? fse_bit_cost(prevCT, count, max) : COST_ERR;
    size_t nc = ncount_cost(count, max, nbSeq, FSELog);
    size_t compressed = (nc << 3) + entropy_cost(count, max, nbSeq);
    if (basic <= repeat && basic <= compressed) { *repeatMode = 0; return 0; }
    if (repeat <= compressed) return 3;
  }
  *repeatMode = 1;
  return 2;
}

/* builds next table + writes its description; returns bytes written or error */
static size_t build_ctable(u8* dst, size_t cap, zo_fse_ctable* next, unsigned FSELog, unsigned type, u32* count, unsigned max,
                           const u8* codes, size_t nbSeq, const s16* defNorm, unsigned defLog, unsigned defMax, const zo_fse_ctable* prev) {
  switch (type) {
    case 1: zo_fse_build_ctable_rle(next, max); if (!cap) return ZO_ERR(ZO_E_DSTSIZE_TOOSMALL); dst[0] = codes[0]; return 1;
    case 3: *next = *prev; return 0;
    case 0: zo_fse_build_ctable(next, defNorm, defMax, defLog); return 0;
    default: {
      s16 norm[64]; size_t n1 = nbSeq;
      unsigned tl = zo_fse_optimal_tablelog(FSELog, nbSeq, max, 2);
      if (count[codes[nbSeq - 1]] > 1) { count[codes[nbSeq - 1]]--; n1--; }
      if (zo_fse_normalize(norm, tl, count, n1, max, n1 >= 2048) <= 0) return ZO_ERR(ZO_E_GENERIC);
      size_t h = zo_fse_write_ncount(dst, cap, norm, max, tl);
      if (!h) return ZO_ERR(ZO_E_GENERIC);
      if (zo_fse_build_ctable(next, norm, max, tl)) return ZO_ERR(ZO_E_GENERIC);
      return h;
    }
  }
}

/* ZSTD_compressSequences_internal: literals + sequences of one block; 0 = not compressible */
static size_t entropy_compress(cctx* c, u8* dst, size_t cap) {
  unsigned strategy = c->cp.strategy;
  const estate* prev = &c->prev; estate* next = &c->next;
  u8* op = dst; u8* oend = dst + cap;
  size_t nbSeq = c->nbSeq;
  {
    size_t r = compress_literals(prev, next, strategy, c->cp.strategy == 1 && c->cp.targetLength > 0, op, cap, c->lits, c->litSize);
    if (ZO_ISERR(r)) return r;
    op += r;
  }
  if (oend - op < 4) return ZO_ERR(ZO_E_DSTSIZE_TOOSMALL);
  if (nbSeq < 128) *op++ = (u8)nbSeq;
  else if (nbSeq < 0x7F00) { op[0] = (u8)((nbSeq >> 8) + 0x80); op[1] = (u8)nbSeq; op += 2; }
  else { op[0] = 0xFF; wr16(op + 1, (u16)(nbSeq - 0x7F00)); op += 3; }
  if (nbSeq == 0) {
    next->ll = prev->ll; next->of = prev->of; next->ml = prev->ml;
    next->llRepeat = prev->llRepeat; next->ofRepeat = prev->ofRepeat; next->mlRepeat = prev->mlRepeat;
    return (size_t)(op - dst);
  }
  u8* seqHead = op++;
  u8* llc = (u8*)malloc(nbSeq * 3); u8* ofc = llc + nbSeq; u8* mlc = ofc + nbSeq;
  u8* lastNCount = NULL;
  unsigned LLt, OFt, MLt;
  for (size_t i = 0; i < nbSeq; i++) {
    llc[i] = (u8)ll_code(c->seqs[i].litLength);
    ofc[i] = (u8)hb32(c->seqs[i].offsetValue);
    mlc[i] = (u8)ml_code(c->seqs[i].matchLength);
  }
#define HIST(codes, maxv)                                                     \
  memset(count, 0, sizeof(count)); max = 0; most = 0;                         \
  for (size_t i = 0; i < nbSeq; i++) count[codes[i]]++;                       \
  for (unsigned s = 0; s <= (maxv); s++) { if (count[s]) max = s; if (count[s] > most) most = count[s]; }
  u32 count[64]; unsigned max; u32 most; size_t r;
  HIST(llc, 35)
  next->llRepeat = prev->llRepeat;
  LLt = select_encoding(&next->llRepeat, count, max, most, nbSeq, 9, &prev->ll, zo_ll_defnorm, 6, 1, strategy);
  r = build_ctable(op, (size_t)(oend - op), &next->ll, 9, LLt, count, max, llc, nbSeq, zo_ll_defnorm, 6, 35, &prev->ll);
  if (ZO_ISERR(r)) { free(llc); return r; }
  if (LLt == 2) lastNCount = op;
  op += r;
  HIST(ofc, 31)
  next->ofRepeat = prev->ofRepeat;
  OFt = select_encoding(&next->ofRepeat, count, max, most, nbSeq, 8, &prev->of, zo_of_defnorm, 5, max <= 28, strategy);
  r = build_ctable(op, (size_t)(oend - op), &next->of, 8, OFt, count, max, ofc, nbSeq, zo_of_defnorm, 5, 28, &prev->of);
  if (ZO_ISERR(r)) { free(llc); return r; }
  if (OFt == 2) lastNCount = op;
  op += r;
  HIST(mlc, 52)
  next->mlRepeat = prev->mlRepeat;
  MLt = select_encoding(&next->mlRepeat, count, max, most, nbSeq, 9, &prev->ml, zo_ml_defnorm, 6, 1, strategy);
  r = build_ctable(op, (size_t)(oend - op), &next->ml, 9, MLt, count, max, mlc, nbSeq, zo_ml_defnorm, 6, 52, &prev->ml);
  if (ZO_ISERR(r)) { free(llc); return r; }
  if (MLt == 2) lastNCount = op;
  op += r;
#undef HIST
  /* mode numbering on the wire: 0 predefined, 1 rle, 2 compressed, 3 repeat */
  *seqHead = (u8)((LLt << 6) + (OFt << 4) + (MLt << 2));
  {
    zo_bitw bw; zo_bitw_init(&bw, op, (size_t)(oend - op));
    size_t n = nbSeq - 1;
    u32 sML = zo_fse_init_state(&next->ml, mlc[n]), sOF = zo_fse_init_state(&next->of, ofc[n]), sLL = zo_fse_init_state(&next->ll, llc[n]);
    u32 bits, nb;
    zo_bitw_add(&bw, c->seqs[n].litLength, zo_ll_bits[llc[n]]);
    zo_bitw_add(&bw, c->seqs[n].matchLength - 3, zo_ml_bits[mlc[n]]);
    zo_bitw_add(&bw, c->seqs[n].offsetValue, ofc[n]);
    while (n-- > 0) {
      nb = zo_fse_encode(&next->of, &sOF, ofc[n], &bits); zo_bitw_add(&bw, bits, nb);
      nb = zo_fse_encode(&next->ml, &sML, mlc[n], &bits); zo_bitw_add(&bw, bits, nb);
      nb = zo_fse_encode(&next->ll, &sLL, llc[n], &bits); zo_bitw_add(&bw, bits, nb);
      zo_bitw_add(&bw, c->seqs[n].litLength, zo_ll_bits[llc[n]]);
      zo_bitw_add(&bw, c->seqs[n].matchLength - 3, zo_ml_bits[mlc[n]]);
      zo_bitw_add(&bw, c->seqs[n].offsetValue, ofc[n]);
    }
    zo_bitw_add(&bw, sML, next->ml.tableLog);
    zo_bitw_add(&bw, sOF, next->of.tableLog);
    zo_bitw_add(&bw, sLL, next->ll.tableLog);
    size_t bs = zo_bitw_close(&bw);
    free(llc);
    if (!bs) return ZO_ERR(ZO_E_DSTSIZE_TOOSMALL);
    op += bs;
    if (lastNCount && (op - lastNCount) < 4) return 0;
  }
  return (size_t)(op - dst);
}

/* ------------------------------------------------------------------ A.4.2 frame / block driver */

/* ---- A.4.3 optimal parsers: zstd_opt.c of 1.4.9 (ZSTD_insertBt1, ZSTD_updateTree_internal, ZSTD_insertBtAndGetAllMatches, the price
 * model and ZSTD_compressBlock_opt_generic), single-segment prefix mode, no dictionary. Index = position + 1 + c->idxShift. */
#define ZO_OPT_NUM (1u << 12)
#define ZO_BITCOST_ACC 8
#define ZO_BITCOST_MUL (1u << ZO_BITCOST_ACC)
#define ZO_MAX_PRICE (1 << 30)
typedef struct { int price; u32 off, mlen, litlen, rep[3]; } zo_optimal;
typedef struct { u32 off, len; } zo_match;
struct zo_opt_state {
  u32 litFreq[256], litLengthFreq[36], matchLengthFreq[53], offCodeFreq[32];
  u32 litSum, litLengthSum, matchLengthSum, offCodeSum;
  u32 litSumBasePrice, litLengthSumBasePrice, matchLengthSumBasePrice, offCodeSumBasePrice;
  int predef;
  zo_optimal table[ZO_OPT_NUM + 1];
  zo_match matches[ZO_OPT_NUM + 1];
};
static inline u32 opt_bit_weight(u32 stat) { return hb32(stat + 1) * ZO_BITCOST_MUL; }
static inline u32 opt_frac_weight(u32 raw) { u32 stat = raw + 1, hb = hb32(stat); return hb * ZO_BITCOST_MUL + ((stat << ZO_BITCOST_ACC) >> hb); }
#define OPT_WEIGHT(stat, lvl) ((lvl) ? opt_frac_weight(stat) : opt_bit_weight(stat))
static void opt_set_base_prices(struct zo_opt_state* o, int lvl) {
  o->litSumBasePrice = OPT_WEIGHT(o->litSum, lvl);
  o->litLengthSumBasePrice = OPT_WEIGHT(o->litLengthSum, lvl);
  o->matchLengthSumBasePrice = OPT_WEIGHT(o->matchLengthSum, lvl);
  o->offCodeSumBasePrice = OPT_WEIGHT(o->offCodeSum, lvl);
}
static u32 opt_downscale(u32* t, u32 last, int malus) {
  u32 sum = 0;
  for (u32 s = 0; s <= last; s++) { t[s] = 1 + (t[s] >> (4 + malus)); sum += t[s]; }
  return sum;
}
static u32 opt_upscale(u32* t, u32 last, int bonus) {      /* ZSTD_upscaleStat (2-pass strategy) */
  u32 sum = 0;
  for (u32 s = 0; s <= last; s++) { t[s] <<= 4 + bonus; t[s]--; sum += t[s]; }
  return sum;
}
static void opt_upscale_stats(struct zo_opt_state* o) {
  o->litSum = opt_upscale(o->litFreq, 255, 0);
  o->litLengthSum = opt_upscale(o->litLengthFreq, 35, 0);
  o->matchLengthSum = opt_upscale(o->matchLengthFreq, 52, 0);
  o->offCodeSum = opt_upscale(o->offCodeFreq, 31, 0);
}
static void opt_rescale_freqs(struct zo_opt_state* o, const u8* src, size_t n, int lvl) {
  o->predef = 0;
  if (o->litLengthSum == 0) {                          /* first block */
    if (n <= 1024) o->predef = 1;
    memset(o->litFreq, 0, sizeof(o->litFreq));
    for (size_t i = 0; i < n; i++) o->litFreq[src[i]]++;
    o->litSum = opt_downscale(o->litFreq, 255, 1);
    for (u32 k = 0; k <= 35; k++) o->litLengthFreq[k] = 1;
    o->litLengthSum = 36;
    for (u32 k = 0; k <= 52; k++) o->matchLengthFreq[k] = 1;
    o->matchLengthSum = 53;
    for (u32 k = 0; k <= 31; k++) o->offCodeFreq[k] = 1;
    o->offCodeSum = 32;
  } else {
    o->litSum = opt_downscale(o->litFreq, 255, 1);
    o->litLengthSum = opt_downscale(o->litLengthFreq, 35, 0);
    o->matchLengthSum = opt_downscale(o->matchLengthFreq, 52, 0);
    o->offCodeSum = opt_downscale(o->offCodeFreq, 31, 0);
  }
  opt_set_base_prices(o, lvl);
}
static u32 opt_raw_literals_cost(const u8* lit, u32 n, const struct zo_opt_state* o, int lvl) {
  if (n == 0) return 0;
  if (o->predef) return (n * 6) * ZO_BITCOST_MUL;
  u32 price = n * o->litSumBasePrice;
  for (u32 u = 0; u < n; u++) price -= OPT_WEIGHT(o->litFreq[lit[u]], lvl);
  return price;
}
static u32 opt_ll_price(u32 ll, const struct zo_opt_state* o, int lvl) {
  if (o->predef) return OPT_WEIGHT(ll, lvl);
  unsigned code = ll_code(ll);
  return (zo_ll_bits[code] * ZO_BITCOST_MUL) + o->litLengthSumBasePrice - OPT_WEIGHT(o->litLengthFreq[code], lvl);
}
static u32 opt_match_price(u32 offset, u32 ml, const struct zo_opt_state* o, int lvl) {
  u32 offCode = hb32(offset + 1), mlBase = ml - 3;
  if (o->predef) return OPT_WEIGHT(mlBase, lvl) + ((16 + offCode) * ZO_BITCOST_MUL);
  u32 price = (offCode * ZO_BITCOST_MUL) + (o->offCodeSumBasePrice - OPT_WEIGHT(o->offCodeFreq[offCode], lvl));
  if (lvl < 2 && offCode >= 20) price += (offCode - 19) * 2 * ZO_BITCOST_MUL;
  unsigned mlCode = ml_code(ml);
  price += (zo_ml_bits[mlCode] * ZO_BITCOST_MUL) + (o->matchLengthSumBasePrice - OPT_WEIGHT(o->matchLengthFreq[mlCode], lvl));
  price += ZO_BITCOST_MUL / 5;
  return price;
}
static void opt_update_stats(struct zo_opt_state* o, u32 ll, const u8* lit, u32 offCode, u32 ml) {
  for (u32 u = 0; u < ll; u++) o->litFreq[lit[u]] += 2;
  o->litSum += ll * 2;
  o->litLengthFreq[ll_code(ll)]++; o->litLengthSum++;
  o->offCodeFreq[hb32(offCode + 1)]++; o->offCodeSum++;
  o->matchLengthFreq[ml_code(ml)]++; o->matchLengthSum++;
}
static void opt_update_rep(u32 out[3], const u32 rep[3], u32 offset, u32 ll0) {
  if (offset >= 3) { out[2] = rep[1]; out[1] = rep[0]; out[0] = offset - 2; }
  else {
    u32 repCode = offset + ll0;
    if (repCode > 0) {
      u32 cur = repCode == 3 ? rep[0] - 1 : rep[repCode];
      u32 r2 = repCode >= 2 ? rep[1] : rep[2], r1 = rep[0];
      out[2] = r2; out[1] = r1; out[0] = cur;
    } else { out[0] = rep[0]; out[1] = rep[1]; out[2] = rep[2]; }
  }
}
static inline u32 opt_hash3(const u8* p, u32 h) { return ((rd32(p) << 8) * 506832829u) >> (32 - h); }
/* ZSTD_insertBt1: inserts index curr into the tree, returns how many positions may be skipped */
static u32 opt_insert_bt1(cctx* c, const u8* src, u32 curr, size_t iend, u32 mls) {
  u32* bt = c->chainTable; u32 btMask = (1u << (c->cp.chainLog - 1)) - 1;
  const u8* ip = src + (curr - 1 - c->idxShift);
  size_t ipos = (size_t)(ip - src);
  u32 h = hashN(ip, c->cp.hashLog, mls);
  u32 matchIndex = c->hashTable[h];
  size_t commonSmaller = 0, commonLarger = 0;
  u32 btLow = btMask >= curr ? 0 : curr - btMask;
  u32* smallerPtr = bt + 2 * (curr & btMask); u32* largerPtr = smallerPtr + 1;
  u32 dummy32, windowLow = 1 + c->idxShift, matchEndIdx = curr + 8 + 1;
  size_t bestLength = 8;
  u32 nbCompares = 1u << c->cp.searchLog;
  c->hashTable[h] = curr;
  while (nbCompares-- && matchIndex >= windowLow) {
    u32* nextPtr = bt + 2 * (matchIndex & btMask);
    size_t ml = commonSmaller < commonLarger ? commonSmaller : commonLarger;
    size_t m = matchIndex - 1 - c->idxShift;
    ml += count_eq(src, ipos + ml, m + ml, iend);
    if (ml > bestLength) { bestLength = ml; if (ml > matchEndIdx - matchIndex) matchEndIdx = matchIndex + (u32)ml; }
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
  *smallerPtr = *largerPtr = 0;
  u32 positions = 0;
  if (bestLength > 384) positions = (u32)(bestLength - 384) < 192 ? (u32)(bestLength - 384) : 192;
  u32 fwd = matchEndIdx - (curr + 8);
  return positions > fwd ? positions : fwd;
}
/* ZSTD_BtGetAllMatches: matches at position ip in increasing length */
static u32 opt_get_all_matches(cctx* c, zo_match* matches, u32* nextToUpdate3, const u8* src, size_t ip, size_t iend, const u32 rep[3], u32 ll0, u32 lengthToBeat) {
  u32 mls = c->cp.minMatch, curr = (u32)ip + 1 + c->idxShift;
  if (curr < c->nextToUpdate) return 0;                /* skipped area */
  u32 mlsH = mls < 4 ? (mls == 3 ? 3 : 4) : mls > 6 ? 6 : mls;   /* the template: 3, 4, 5, 6 (7 -> 6) */
  for (u32 idx = c->nextToUpdate; idx < curr;) idx += opt_insert_bt1(c, src, idx, iend, mlsH);
  c->nextToUpdate = curr;
  u32 sufficient_len = c->cp.targetLength < ZO_OPT_NUM - 1 ? c->cp.targetLength : ZO_OPT_NUM - 1;
  u32 minMatch = mlsH == 3 ? 3 : 4;
  u32 h = hashN(src + ip, c->cp.hashLog, mlsH);
  u32 matchIndex = c->hashTable[h];
  u32* bt = c->chainTable; u32 btMask = (1u << (c->cp.chainLog - 1)) - 1;
  size_t commonSmaller = 0, commonLarger = 0;
  u32 dictLimit = 1 + c->idxShift;
  u32 btLow = btMask >= curr ? 0 : curr - btMask;
  u32 windowLow = lowest_at(c, curr), matchLow = windowLow ? windowLow : 1;
  u32* smallerPtr = bt + 2 * (curr & btMask); u32* largerPtr = smallerPtr + 1;
  u32 matchEndIdx = curr + 8 + 1, dummy32, mnum = 0;
  u32 nbCompares = 1u << c->cp.searchLog;
  size_t bestLength = lengthToBeat - 1;
  {   /* repcodes */
    u32 lastR = 3 + ll0;
    for (u32 repCode = ll0; repCode < lastR; repCode++) {
      u32 repOffset = repCode == 3 ? rep[0] - 1 : rep[repCode];
      u32 repIndex = curr - repOffset;
      u32 repLen = 0;
      if (repOffset - 1 < curr - dictLimit) {
        int same = minMatch == 3 ? ((rd32(src + ip) << 8) == (rd32(src + ip - repOffset) << 8)) : (rd32(src + ip) == rd32(src + ip - repOffset));
        if (repIndex >= windowLow && same) repLen = (u32)count_eq(src, ip + minMatch, ip + minMatch - repOffset, iend) + minMatch;
      }
      if (repLen > bestLength) {
        bestLength = repLen;
        matches[mnum].off = repCode - ll0; matches[mnum].len = repLen; mnum++;
        if (repLen > sufficient_len || ip + repLen == iend) return mnum;
      }
    }
  }
  if (mlsH == 3 && bestLength < 3) {                   /* HC3 */
    u32 idx = *nextToUpdate3, target = curr;
    u32 hash3 = opt_hash3(src + ip, c->hashLog3);
    while (idx < target) { c->hashTable3[opt_hash3(src + (idx - 1 - c->idxShift), c->hashLog3)] = idx; idx++; }
    *nextToUpdate3 = target;
    u32 matchIndex3 = c->hashTable3[hash3];
    if (matchIndex3 >= matchLow && curr - matchIndex3 < (1u << 18)) {
      size_t mlen = count_eq(src, ip, matchIndex3 - 1 - c->idxShift, iend);
      if (mlen >= 3) {
        bestLength = mlen;
        matches[0].off = (curr - matchIndex3) + 2; matches[0].len = (u32)mlen; mnum = 1;
        if (mlen > sufficient_len || ip + mlen == iend) { c->nextToUpdate = curr + 1; return 1; }
      }
    }
  }
  c->hashTable[h] = curr;
  while (nbCompares-- && matchIndex >= matchLow) {
    u32* nextPtr = bt + 2 * (matchIndex & btMask);
    size_t ml = commonSmaller < commonLarger ? commonSmaller : commonLarger;
    size_t m = matchIndex - 1 - c->idxShift;
    ml += count_eq(src, ip + ml, m + ml, iend);
    if (ml > bestLength) {
      if (ml > matchEndIdx - matchIndex) matchEndIdx = matchIndex + (u32)ml;
      bestLength = ml;
      matches[mnum].off = (curr - matchIndex) + 2; matches[mnum].len = (u32)ml; mnum++;
      if (ml > ZO_OPT_NUM || ip + ml == iend) break;
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
  *smallerPtr = *largerPtr = 0;
  c->nextToUpdate = matchEndIdx - 8;
  return mnum;
}
/* ZSTD_compressBlock_opt_generic; emits through emit() unless dry (btultra2's statistics pass) */
static size_t mf_opt(cctx* c, const u8* src, size_t bs, size_t be, u32 rep[3], int lvl, int dry) {
  struct zo_opt_state* o = c->opt;
  zo_optimal* opt = o->table; zo_match* matches = o->matches;
  size_t anchor = bs, ilimit = be >= 8 ? be - 8 : 0, ip = bs;
  u32 sufficient_len = c->cp.targetLength < ZO_OPT_NUM - 1 ? c->cp.targetLength : ZO_OPT_NUM - 1;
  u32 minMatch = c->cp.minMatch == 3 ? 3 : 4;
  u32 nextToUpdate3 = c->nextToUpdate;
  zo_optimal lastSequence; memset(&lastSequence, 0, sizeof(lastSequence));
  opt_rescale_freqs(o, src + bs, be - bs, lvl);
  ip += (ip + 1 + c->idxShift == 1 + c->idxShift) && bs == 0 ? 1 : 0;      /* ip += (ip == prefixStart) */
  while (ip < ilimit) {
    u32 cur, last_pos = 0;
    {
      u32 litlen = (u32)(ip - anchor), ll0 = !litlen;
      u32 nb = opt_get_all_matches(c, matches, &nextToUpdate3, src, ip, be, rep, ll0, minMatch);
      if (!nb) { ip++; continue; }
      for (int i = 0; i < 3; i++) opt[0].rep[i] = rep[i];
      opt[0].mlen = 0; opt[0].litlen = litlen;
      opt[0].price = (int)opt_ll_price(litlen, o, lvl);
      {
        u32 maxML = matches[nb - 1].len, maxOffset = matches[nb - 1].off;
        if (maxML > sufficient_len) {
          lastSequence.litlen = litlen; lastSequence.mlen = maxML; lastSequence.off = maxOffset;
          cur = 0; last_pos = lastSequence.litlen + lastSequence.mlen;
          goto shortest_path;
        }
      }
      {
        u32 literalsPrice = (u32)opt[0].price + opt_ll_price(0, o, lvl);
        u32 pos;
        for (pos = 1; pos < minMatch; pos++) opt[pos].price = ZO_MAX_PRICE;
        for (u32 k = 0; k < nb; k++) {
          u32 offset = matches[k].off, end = matches[k].len;
          for (; pos <= end; pos++) {
            u32 matchPrice = opt_match_price(offset, pos, o, lvl);
            opt[pos].mlen = pos; opt[pos].off = offset; opt[pos].litlen = litlen; opt[pos].price = (int)(literalsPrice + matchPrice);
          }
        }
        last_pos = pos - 1;
      }
    }
    for (cur = 1; cur <= last_pos; cur++) {
      size_t inr = ip + cur;
      {
        u32 litlen = opt[cur - 1].mlen == 0 ? opt[cur - 1].litlen + 1 : 1;
        int price = opt[cur - 1].price + (int)opt_raw_literals_cost(src + ip + cur - 1, 1, o, lvl) + (int)opt_ll_price(litlen, o, lvl) - (int)opt_ll_price(litlen - 1, o, lvl);
        if (price <= opt[cur].price) { opt[cur].mlen = 0; opt[cur].off = 0; opt[cur].litlen = litlen; opt[cur].price = price; }
      }
      if (opt[cur].mlen != 0) {
        u32 prev = cur - opt[cur].mlen;
        u32 nr[3]; opt_update_rep(nr, opt[prev].rep, opt[cur].off, opt[cur].litlen == 0);
        memcpy(opt[cur].rep, nr, sizeof(nr));
      } else memcpy(opt[cur].rep, opt[cur - 1].rep, sizeof(opt[cur].rep));
      if (inr > ilimit) continue;
      if (cur == last_pos) break;
      if (lvl == 0 && opt[cur + 1].price <= opt[cur].price + (int)(ZO_BITCOST_MUL / 2)) continue;
      {
        u32 ll0 = opt[cur].mlen != 0;
        u32 litlen = opt[cur].mlen == 0 ? opt[cur].litlen : 0;
        u32 previousPrice = (u32)opt[cur].price;
        u32 basePrice = previousPrice + opt_ll_price(0, o, lvl);
        u32 nb = opt_get_all_matches(c, matches, &nextToUpdate3, src, inr, be, opt[cur].rep, ll0, minMatch);
        if (!nb) continue;
        {
          u32 maxML = matches[nb - 1].len;
          if (maxML > sufficient_len || cur + maxML >= ZO_OPT_NUM) {
            lastSequence.mlen = maxML; lastSequence.off = matches[nb - 1].off; lastSequence.litlen = litlen;
            cur -= opt[cur].mlen == 0 ? opt[cur].litlen : 0;
            last_pos = cur + lastSequence.litlen + lastSequence.mlen;
            if (cur > ZO_OPT_NUM) cur = 0;
            goto shortest_path;
          }
        }
        for (u32 k = 0; k < nb; k++) {
          u32 offset = matches[k].off, lastML = matches[k].len, startML = k > 0 ? matches[k - 1].len + 1 : minMatch;
          for (u32 mlen = lastML; mlen >= startML; mlen--) {
            u32 pos = cur + mlen;
            int price = (int)(basePrice + opt_match_price(offset, mlen, o, lvl));
            if (pos > last_pos || price < opt[pos].price) {
              while (last_pos < pos) { opt[last_pos + 1].price = ZO_MAX_PRICE; last_pos++; }
              opt[pos].mlen = mlen; opt[pos].off = offset; opt[pos].litlen = litlen; opt[pos].price = price;
            } else if (lvl == 0) break;
          }
        }
      }
    }
    lastSequence = opt[last_pos];
    cur = last_pos > lastSequence.litlen + lastSequence.mlen ? last_pos - (lastSequence.litlen + lastSequence.mlen) : 0;
shortest_path:
    if (lastSequence.mlen != 0) { u32 nr[3]; opt_update_rep(nr, opt[cur].rep, lastSequence.off, lastSequence.litlen == 0); memcpy(rep, nr, sizeof(nr)); }
    else memcpy(rep, opt[cur].rep, 3 * sizeof(u32));
    {
      u32 storeEnd = cur + 1, storeStart = storeEnd, seqPos = cur;
      opt[storeEnd] = lastSequence;
      while (seqPos > 0) {
        u32 backDist = opt[seqPos].litlen + opt[seqPos].mlen;
        storeStart--;
        opt[storeStart] = opt[seqPos];
        seqPos = seqPos > backDist ? seqPos - backDist : 0;
      }
      for (u32 sp = storeStart; sp <= storeEnd; sp++) {
        u32 llen = opt[sp].litlen, mlen = opt[sp].mlen, offCode = opt[sp].off, advance = llen + mlen;
        if (mlen == 0) { ip = anchor + llen; continue; }
        opt_update_stats(o, llen, src + anchor, offCode, mlen);
        if (!dry) emit(c, src, anchor, llen, mlen, offCode + 1);
        anchor += advance; ip = anchor;
      }
      opt_set_base_prices(o, lvl);
    }
  }
  return be - anchor;
}

static size_t run_match_finder(cctx* c, const u8* src, size_t bs, size_t be, u32 rep[3]) {
  /* limited update after a very long match (ZSTD_buildSeqStore) */
  u32 cur = (u32)bs + 1 + c->idxShift;
  if (cur > c->nextToUpdate + 384) {
    u32 d = cur - c->nextToUpdate - 384;
    c->nextToUpdate = cur - (d < 192 ? d : 192);
  }
  switch (c->cp.strategy) {
    case 1: return mf_fast(c, src, bs, be, rep);
    case 2: return mf_dfast(c, src, bs, be, rep);
    case 3: return mf_lazy(c, src, bs, be, rep, 0);
    case 4: return mf_lazy(c, src, bs, be, rep, 1);
    case 5: case 6: return mf_lazy(c, src, bs, be, rep, 2);
    case 7: return mf_opt(c, src, bs, be, rep, 0, 0);
    case 8: return mf_opt(c, src, bs, be, rep, 2, 0);
    default:
      /* btultra2: a first pass over the first block only to collect statistics, then the window is moved past it */
      if (c->opt->litLengthSum == 0 && bs == 0 && c->nextToUpdate == 1 && be - bs > 1024) {
        u32 tmpRep[3] = {rep[0], rep[1], rep[2]};
        size_t L = be - bs;
        (void)mf_opt(c, src, bs, be, tmpRep, 2, 1);
        c->idxShift += (u32)L;
        c->nextToUpdate = 1 + c->idxShift;
        opt_upscale_stats(c->opt);                     /* "re-inforce weight of collected statistics" */
      }
      return mf_opt(c, src, bs, be, rep, 2, 0);
  }
}

static int cctx_init(cctx* c, int level, size_t n) {
  memset(c, 0, sizeof(*c));
  if (zo_get_cparams(level, n, &c->cp)) return -1;
  if (c->cp.strategy >= 7) {
    c->opt = (struct zo_opt_state*)calloc(1, sizeof(struct zo_opt_state));
    c->hashLog3 = c->cp.minMatch == 3 ? (c->cp.windowLog < 17 ? c->cp.windowLog : 17) : 0;
    c->hashTable3 = (u32*)calloc((size_t)1 << c->hashLog3, 4);
    if (!c->opt || !c->hashTable3) return -1;
  }
  size_t hsz = (size_t)1 << c->cp.hashLog, csz = (size_t)1 << c->cp.chainLog;
  size_t blockMax = n < (128u << 10) ? n : (128u << 10);
  c->hashTable = (u32*)calloc(hsz, 4);
  c->chainTable = (u32*)calloc(csz, 4);
  c->seqs = (zo_seq*)malloc((blockMax / 3 + 2) * sizeof(zo_seq));
  c->lits = (u8*)malloc(blockMax + 16);
  c->nextToUpdate = 1;
  c->prev.rep[0] = 1; c->prev.rep[1] = 4; c->prev.rep[2] = 8;
  return (c->hashTable && c->chainTable && c->seqs && c->lits) ? 0 : -1;
}
static void cctx_free(cctx* c) { free(c->hashTable); free(c->chainTable); free(c->seqs); free(c->lits); free(c->hashTable3); free(c->opt); }

size_t zo_compress_frame(void* dstv, size_t cap, const void* srcv, size_t n, int level, int checksum) {
  u8* dst = (u8*)dstv; const u8* src = (const u8*)srcv;
  cctx c;
  if (cctx_init(&c, level, n)) { cctx_free(&c); return ZO_ERR(ZO_E_PARAM_UNSUPPORTED); }
  if (cap < 6) { cctx_free(&c); return ZO_ERR(ZO_E_DSTSIZE_TOOSMALL); }
  u8* op = dst; u8* oend = dst + cap;
  wr32(op, 0xFD2FB528u); op[4] = checksum ? 4 : 0; op[5] = (u8)((c.cp.windowLog - 10) << 3); op += 6;
  size_t blockSize = (size_t)1 << c.cp.windowLog;
  if (blockSize > (128u << 10)) blockSize = 128u << 10;
  size_t pos = 0; int first = 1;
  /* ZSTD_compressEnd on empty input writes one empty raw last block; ZRA never calls it with n == 0 (zra.cpp:216) */
  do {
    size_t L = n - pos < blockSize ? n - pos : blockSize;
    int last = pos + L == n;
    size_t cSize = 0;
    if ((size_t)(oend - op) < 3 + 3) { cctx_free(&c); return ZO_ERR(ZO_E_DSTSIZE_TOOSMALL); }
    if (L >= 7) {
      c.nbSeq = 0; c.litSize = 0;
      memcpy(c.next.rep, c.prev.rep, sizeof(c.prev.rep));
      size_t lastLL = run_match_finder(&c, src, pos, pos + L, c.next.rep);
      memcpy(c.lits + c.litSize, src + pos + L - lastLL, lastLL); c.litSize += lastLL;
      cSize = entropy_compress(&c, op + 3, (size_t)(oend - op) - 3);
      if (ZO_ISERR(cSize)) {
        if (ZO_ERRCODE(cSize) == ZO_E_DSTSIZE_TOOSMALL && L <= (size_t)(oend - op) - 3) cSize = 0;
        else { cctx_free(&c); return cSize; }
      }
      if (cSize && cSize >= L - min_gain(L, c.cp.strategy)) cSize = 0;
      if (!first && cSize < 25) {
        size_t k = 1; while (k < L && src[pos + k] == src[pos]) k++;
        if (k == L) { cSize = 1; op[3] = src[pos]; }
      }
      if (cSize > 1) c.prev = c.next;
    }
    if (cSize == 0) {
      if ((size_t)(oend - op) < 3 + L) { cctx_free(&c); return ZO_ERR(ZO_E_DSTSIZE_TOOSMALL); }
      wr24(op, (u32)last + (0u << 1) + ((u32)L << 3));
      memcpy(op + 3, src + pos, L); op += 3 + L;
    } else if (cSize == 1) { wr24(op, (u32)last + (1u << 1) + ((u32)L << 3)); op += 4; }
    else { wr24(op, (u32)last + (2u << 1) + ((u32)cSize << 3)); op += 3 + cSize; }
    pos += L; first = 0;
  } while (pos < n);
  if (checksum) {
    if (oend - op < 4) { cctx_free(&c); return ZO_ERR(ZO_E_DSTSIZE_TOOSMALL); }
    wr32(op, (u32)zo_xxh64(src, n, 0)); op += 4;
  }
  cctx_free(&c);
  return (size_t)(op - dst);
}

size_t zo_generate_sequences(zo_seq* out, size_t cap, const void* srcv, size_t n, int level) {
  /* match finder only, block by block, with the rep/entropy confirmation rule of the real driver */
  const u8* src = (const u8*)srcv;
  cctx c; size_t total = 0;
  if (cctx_init(&c, level, n)) { cctx_free(&c); return ZO_ERR(ZO_E_PARAM_UNSUPPORTED); }
  size_t blockSize = (size_t)1 << c.cp.windowLog; if (blockSize > (128u << 10)) blockSize = 128u << 10;
  u8* scratch = (u8*)malloc(zo_compress_bound(blockSize) + 64);
  size_t pos = 0;
  while (pos < n) {
    size_t L = n - pos < blockSize ? n - pos : blockSize;
    if (L >= 7) {
      c.nbSeq = 0; c.litSize = 0;
      memcpy(c.next.rep, c.prev.rep, sizeof(c.prev.rep));
      size_t lastLL = run_match_finder(&c, src, pos, pos + L, c.next.rep);
      for (size_t i = 0; i < c.nbSeq && total < cap; i++) out[total++] = c.seqs[i];
      if (total < cap) { out[total].litLength = (u32)lastLL; out[total].matchLength = 0; out[total].offsetValue = 0; total++; }
      memcpy(c.lits + c.litSize, src + pos + L - lastLL, lastLL); c.litSize += lastLL;
      size_t cSize = entropy_compress(&c, scratch, zo_compress_bound(blockSize) + 64);
      if (ZO_ISERR(cSize) || cSize >= L - min_gain(L, c.cp.strategy)) cSize = 0;
      if (cSize > 1) c.prev = c.next;
    } else if (total < cap) { out[total].litLength = (u32)L; out[total].matchLength = 0; out[total].offsetValue = 0; total++; }
    pos += L;
  }
  free(scratch); cctx_free(&c);
  return total;
}
