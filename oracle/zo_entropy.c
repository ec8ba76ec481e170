/* ORACLE — TEST INFRASTRUCTURE ONLY (see zo_common.h).
 *
 * Entropy primitives of zstd 1.4.9 restated from SURVEY.md Appendix A (A.2, A.3, A.4.5-A.4.7):
 * CRC-32, XXH64, FSE (normalise / NCount write+read / encode+decode tables), Huffman
 * (tree build with depth limit, weight serialisation, decode table).
 * The dependency itself (facebook/zstd, submodule of the reference, .gitmodules:1-3) is absent from
 * /root/reference; the call sites this serves are zra.cpp:219,249,280,289,293,331,397,406,410,435.
 */
#include "zo_internal.h"
#include <string.h>

/* ------------------------------------------------------------------ CRC-32 (zra.cpp:128-133) */
u32 zo_crc32(u32 crc, const void* data, size_t n) {
  static u32 table[256];
  static int init = 0;
  if (!init) {
    for (u32 i = 0; i < 256; i++) {
      u32 c = i;
      for (int k = 0; k < 8; k++) c = (c & 1) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
      table[i] = c;
    }
    init = 1;
  }
  const u8* p = (const u8*)data;
  crc = ~crc;
  for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
  return ~crc;
}

/* ------------------------------------------------------------------ XXH64 (A.1) */
static const u64 P1 = 11400714785074694791ULL, P2 = 14029467366897019727ULL, P3 = 1609587929392839161ULL,
                 P4 = 9650029242287828579ULL, P5 = 2870177450012600261ULL;
static u64 rotl64(u64 x, int r) { return (x << r) | (x >> (64 - r)); }
static u64 xxround(u64 acc, u64 x) { return rotl64(acc + x * P2, 31) * P1; }
static u64 xxmerge(u64 h, u64 v) { return (h ^ xxround(0, v)) * P1 + P4; }
u64 zo_xxh64(const void* data, size_t n, u64 seed) {
  const u8* p = (const u8*)data;
  const u8* end = p + n;
  u64 h;
  if (n >= 32) {
    u64 v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
    while (end - p >= 32) {
      v1 = xxround(v1, rd64(p));
      v2 = xxround(v2, rd64(p + 8));
      v3 = xxround(v3, rd64(p + 16));
      v4 = xxround(v4, rd64(p + 24));
      p += 32;
    }
    h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
    h = xxmerge(h, v1); h = xxmerge(h, v2); h = xxmerge(h, v3); h = xxmerge(h, v4);
  } else {
    h = seed + P5;
  }
  h += (u64)n;
  while (end - p >= 8) { h = rotl64(h ^ xxround(0, rd64(p)), 27) * P1 + P4; p += 8; }
  if (end - p >= 4) { h = rotl64(h ^ ((u64)rd32(p) * P1), 23) * P2 + P3; p += 4; }
  while (p < end) { h = rotl64(h ^ ((u64)*p * P5), 11) * P1; p++; }
  h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
  return h;
}

/* ------------------------------------------------------------------ FSE (A.4.6) */
unsigned zo_fse_optimal_tablelog(unsigned maxLog, size_t n, unsigned maxSym, unsigned minus) {
  u32 maxBitsSrc = hb32((u32)(n - 1)) - minus; /* unsigned wrap intended */
  u32 t = maxLog;
  u32 minBitsSrc = hb32((u32)n) + 1, minBitsSym = hb32(maxSym) + 2;
  u32 minBits = minBitsSrc < minBitsSym ? minBitsSrc : minBitsSym;
  if (maxBitsSrc < t) t = maxBitsSrc;
  if (minBits > t) t = minBits;
  if (t < 5) t = 5;
  if (t > 12) t = 12;
  return t;
}

static int fse_normalize_m2(s16* norm, u32 t, const u32* cnt, size_t total, u32 maxSym, s16 low) {
  const s16 NYA = -2;
  u32 s, distributed = 0, toDist;
  u32 lowThr = (u32)(total >> t);
  u32 lowOne = (u32)((total * 3) >> (t + 1));
  for (s = 0; s <= maxSym; s++) {
    if (cnt[s] == 0) { norm[s] = 0; continue; }
    if (cnt[s] <= lowThr) { norm[s] = low; distributed++; total -= cnt[s]; continue; }
    if (cnt[s] <= lowOne) { norm[s] = 1; distributed++; total -= cnt[s]; continue; }
    norm[s] = NYA;
  }
  toDist = (1u << t) - distributed;
  if (toDist == 0) return 0;
  if ((total / toDist) > lowOne) {
    lowOne = (u32)((total * 3) / (toDist * 2));
    for (s = 0; s <= maxSym; s++)
      if (norm[s] == NYA && cnt[s] <= lowOne) { norm[s] = 1; distributed++; total -= cnt[s]; }
    toDist = (1u << t) - distributed;
  }
  if (distributed == maxSym + 1) {
    u32 maxV = 0, maxC = 0;
    for (s = 0; s <= maxSym; s++) if (cnt[s] > maxC) { maxV = s; maxC = cnt[s]; }
    norm[maxV] += (s16)toDist;
    return 0;
  }
  if (total == 0) {
    for (s = 0; toDist > 0; s = (s + 1) % (maxSym + 1))
      if (norm[s] > 0) { toDist--; norm[s]++; }
    return 0;
  }
  {
    u64 vLog = 62 - t;
    u64 mid = (1ULL << (vLog - 1)) - 1;
    u64 rStep = ((((u64)1 << vLog) * toDist) + mid) / (u32)total;
    u64 acc = mid;
    for (s = 0; s <= maxSym; s++) {
      if (norm[s] == NYA) {
        u64 end = acc + (u64)cnt[s] * rStep;
        u32 w = (u32)(end >> vLog) - (u32)(acc >> vLog);
        if (w < 1) return -1;
        norm[s] = (s16)w;
        acc = end;
      }
    }
  }
  return 0;
}

/* returns tableLog, 0 for the rle special case, <0 on error */
int zo_fse_normalize(s16* norm, unsigned t, const u32* cnt, size_t total, unsigned maxSym, int useLowProb) {
  static const u32 rtb[8] = {0, 473195, 504333, 520860, 550000, 700000, 750000, 830000};
  s16 low = useLowProb ? -1 : 1;
  u64 scale = 62 - t;
  u64 step = ((u64)1 << 62) / (u32)total;
  u64 vStep = 1ULL << (scale - 20);
  int still = 1 << t;
  unsigned s, largest = 0;
  s16 largestP = 0;
  u32 lowThr = (u32)(total >> t);
  if (t < 5 || t > 12) return -1;
  for (s = 0; s <= maxSym; s++) {
    if (cnt[s] == total) return 0;
    if (cnt[s] == 0) { norm[s] = 0; continue; }
    if (cnt[s] <= lowThr) { norm[s] = low; still--; }
    else {
      s16 p = (s16)(((u64)cnt[s] * step) >> scale);
      if (p < 8) {
        u64 rest = vStep * rtb[p];
        p += ((u64)cnt[s] * step) - ((u64)p << scale) > rest;
      }
      if (p > largestP) { largestP = p; largest = s; }
      norm[s] = p;
      still -= p;
    }
  }
  if (-still >= (norm[largest] >> 1)) {
    if (fse_normalize_m2(norm, t, cnt, total, maxSym, low)) return -1;
  } else norm[largest] += (s16)still;
  return (int)t;
}

/* forward LSB-first bit writer for NCount; returns byte count or 0 on error */
size_t zo_fse_write_ncount(u8* out, size_t cap, const s16* norm, unsigned maxSym, unsigned t) {
  u8 tmp[512];
  u64 acc = 0; int nacc = 0; size_t pos = 0;
  int tableSize = 1 << t, remaining = tableSize + 1, thr = tableSize, nb = (int)t + 1;
  unsigned sym = 0, alpha = maxSym + 1;
  int prev0 = 0;
#define PUT(v, n) do { acc |= ((u64)(v)) << nacc; nacc += (n); while (nacc >= 8) { tmp[pos++] = (u8)acc; acc >>= 8; nacc -= 8; } } while (0)
  PUT(t - 5, 4);
  while (sym < alpha && remaining > 1) {
    if (prev0) {
      unsigned start = sym;
      while (sym < alpha && !norm[sym]) sym++;
      if (sym == alpha) break;
      while (sym >= start + 24) { start += 24; PUT(0xFFFFu, 16); }
      while (sym >= start + 3) { start += 3; PUT(3, 2); }
      PUT(sym - start, 2);
    }
    {
      int c = norm[sym++];
      int max = (2 * thr - 1) - remaining;
      remaining -= c < 0 ? -c : c;
      c++;
      if (c >= thr) c += max;
      PUT((u32)c, nb - (c < max));
      prev0 = (c == 1);
      if (remaining < 1) return 0;
      while (remaining < thr) { nb--; thr >>= 1; }
    }
    if (pos > sizeof(tmp) - 16) return 0;
  }
#undef PUT
  if (remaining != 1) return 0;
  if (nacc > 0) tmp[pos++] = (u8)acc;
  if (pos > cap) return 0;
  memcpy(out, tmp, pos);
  return pos;
}

/* A.3 "FSE table description" reader. returns bytes consumed, 0 on corruption.
 * Restates FSE_readNCount of libzstd 1.4.9 (lib/common/entropy_common.c, FSE_readNCount_body) INCLUDING what it does near the end of its
 * input, which a reader written from the format description does not: it reads 32 bits at a byte pointer that never passes end - 4, and
 * when a read position would, the pointer is clamped there and the bit offset taken modulo 32 — a description that runs past the end of
 * the input then reads bits it has read before instead of failing, and only the position behind the LAST symbol is checked (> 32).
 * Inputs shorter than 8 bytes are read from a zero-padded copy. Round 5: the plain reader (zeros past the end, every overrun an error)
 * refused a damaged frame that libzstd decodes (soak seed 91417: checksum_wrong there, corruption_detected here). */
static u32 zo_rd32le(const u8* p) { return (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16) | ((u32)p[3] << 24); }
size_t zo_fse_read_ncount(s16* norm, unsigned* maxSymPtr, unsigned* tPtr, const u8* src, size_t n, unsigned maxAL) {
  u8 pad[8];
  if (n < 1) return 0;
  if (n < 8) {
    memset(pad, 0, 8); memcpy(pad, src, n);
    { const size_t h = zo_fse_read_ncount(norm, maxSymPtr, tPtr, pad, 8, maxAL);
      return (h == 0 || h > n) ? 0 : h; }
  }
  {
    const u8* const iend = src + n;
    const u8* ip = src;
    const unsigned maxSV1 = *maxSymPtr + 1;
    u32 bitStream = zo_rd32le(ip);
    int nbBits = (int)(bitStream & 0xF) + 5, remaining, threshold, bitCount = 4, previous0 = 0;
    memset(norm, 0, (size_t)maxSV1 * sizeof(norm[0]));          /* symbols the description does not reach have probability 0 */
    unsigned charnum = 0;
    if (nbBits > 15) return 0;                                   /* FSE_TABLELOG_ABSOLUTE_MAX: tableLog_tooLarge */
    bitStream >>= 4;
    *tPtr = (unsigned)nbBits;
    remaining = (1 << nbBits) + 1; threshold = 1 << nbBits; nbBits++;
    for (;;) {
      if (previous0) {
        /* pairs of 11 = three more zero-probability symbols each */
        int repeats = __builtin_ctz(~bitStream | 0x80000000u) >> 1;
        while (repeats >= 12) {
          charnum += 3 * 12;
          if (ip <= iend - 7) ip += 3;
          else { bitCount -= (int)(8 * (iend - 7 - ip)); bitCount &= 31; ip = iend - 4; }
          bitStream = zo_rd32le(ip) >> bitCount;
          repeats = __builtin_ctz(~bitStream | 0x80000000u) >> 1;
        }
        charnum += 3 * (unsigned)repeats;
        bitStream >>= 2 * repeats; bitCount += 2 * repeats;
        charnum += bitStream & 3; bitCount += 2;
        if (charnum >= maxSV1) break;
        if (ip <= iend - 7 || ip + (bitCount >> 3) <= iend - 4) { ip += bitCount >> 3; bitCount &= 7; }
        else { bitCount -= (int)(8 * (iend - 4 - ip)); bitCount &= 31; ip = iend - 4; }
        bitStream = zo_rd32le(ip) >> bitCount;
      }
      {
        const int max = (2 * threshold - 1) - remaining;
        int count;
        if ((bitStream & (u32)(threshold - 1)) < (u32)max) { count = (int)(bitStream & (u32)(threshold - 1)); bitCount += nbBits - 1; }
        else { count = (int)(bitStream & (u32)(2 * threshold - 1)); if (count >= threshold) count -= max; bitCount += nbBits; }
        count--;
        if (count >= 0) remaining -= count; else remaining += count;
        norm[charnum++] = (s16)count;
        previous0 = !count;
        if (remaining < threshold) {
          if (remaining <= 1) break;
          nbBits = (31 - __builtin_clz((u32)remaining)) + 1;
          threshold = 1 << (nbBits - 1);
        }
        if (charnum >= maxSV1) break;
        if (ip <= iend - 7 || ip + (bitCount >> 3) <= iend - 4) { ip += bitCount >> 3; bitCount &= 7; }
        else { bitCount -= (int)(8 * (iend - 4 - ip)); bitCount &= 31; ip = iend - 4; }
        bitStream = zo_rd32le(ip) >> bitCount;
      }
    }
    if (remaining != 1) return 0;
    if (charnum > maxSV1) return 0;                              /* maxSymbolValue_tooSmall */
    if (bitCount > 32) return 0;
    if (*tPtr > maxAL) return 0;                                 /* the callers' "tableLog > maxLog" (checked behind the read there) */
    *maxSymPtr = charnum - 1;
    ip += (bitCount + 7) >> 3;
    return (size_t)(ip - src);
  }
}

/* symbol spreading shared by encode and decode tables (A.3) */
static int fse_spread(u8* cell, const s16* norm, unsigned maxSym, unsigned t) {
  u32 size = 1u << t, mask = size - 1, high = size - 1, step = (size >> 1) + (size >> 3) + 3, pos = 0;
  for (unsigned s = 0; s <= maxSym; s++) if (norm[s] == -1) cell[high--] = (u8)s;
  for (unsigned s = 0; s <= maxSym; s++) {
    for (int i = 0; i < norm[s]; i++) {
      cell[pos] = (u8)s;
      pos = (pos + step) & mask;
      while (pos > high) pos = (pos + step) & mask;
    }
  }
  return pos == 0 ? 0 : -1;
}

int zo_fse_build_ctable(zo_fse_ctable* ct, const s16* norm, unsigned maxSym, unsigned t) {
  u8 cell[1 << 12];
  u32 cumul[ZO_MAXSYM + 2];
  u32 size = 1u << t;
  ct->tableLog = t; ct->maxSym = maxSym; ct->rle = 0;
  cumul[0] = 0;
  for (unsigned u = 1; u <= maxSym + 1; u++) cumul[u] = cumul[u - 1] + (norm[u - 1] == -1 ? 1 : (u32)norm[u - 1]);
  if (fse_spread(cell, norm, maxSym, t)) return -1;
  for (u32 u = 0; u < size; u++) ct->stateTable[cumul[cell[u]]++] = (u16)(size + u);
  {
    u32 total = 0;
    for (unsigned s = 0; s <= maxSym; s++) {
      int p = norm[s];
      if (p == 0) { ct->deltaNbBits[s] = ((t + 1) << 16) - (1u << t); ct->deltaFindState[s] = 0; }
      else if (p == 1 || p == -1) { ct->deltaNbBits[s] = (t << 16) - (1u << t); ct->deltaFindState[s] = (int)total - 1; total++; }
      else {
        u32 maxBitsOut = t - hb32((u32)p - 1);
        ct->deltaNbBits[s] = (maxBitsOut << 16) - ((u32)p << maxBitsOut);
        ct->deltaFindState[s] = (int)total - p;
        total += (u32)p;
      }
    }
  }
  return 0;
}

void zo_fse_build_ctable_rle(zo_fse_ctable* ct, unsigned sym) {
  memset(ct, 0, sizeof(*ct));
  ct->rle = 1; ct->tableLog = 0; ct->maxSym = sym;
}

u32 zo_fse_init_state(const zo_fse_ctable* ct, unsigned sym) {
  if (ct->rle) return 0;
  u32 d = ct->deltaNbBits[sym];
  u32 nb = (d + (1u << 15)) >> 16;
  u32 v = (nb << 16) - d;
  return ct->stateTable[(v >> nb) + ct->deltaFindState[sym]];
}

/* returns nbBits; *bits = the low nbBits of the old state; updates *state */
u32 zo_fse_encode(const zo_fse_ctable* ct, u32* state, unsigned sym, u32* bits) {
  if (ct->rle) { *bits = 0; return 0; }
  u32 nb = (*state + ct->deltaNbBits[sym]) >> 16;
  *bits = *state & ((1u << nb) - 1);
  *state = ct->stateTable[(*state >> nb) + ct->deltaFindState[sym]];
  return nb;
}

int zo_fse_build_dtable(zo_fse_dtable* dt, const s16* norm, unsigned maxSym, unsigned t) {
  u8 cell[1 << 12];
  u16 next[ZO_MAXSYM + 1];
  u32 size = 1u << t;
  if (t > 12) return -1;
  dt->tableLog = t;
  for (unsigned s = 0; s <= maxSym; s++) next[s] = norm[s] == -1 ? 1 : (u16)norm[s];
  if (fse_spread(cell, norm, maxSym, t)) return -1;
  for (u32 u = 0; u < size; u++) {
    unsigned s = cell[u];
    u32 x = next[s]++;
    u32 nb = t - hb32(x);
    dt->sym[u] = (u8)s; dt->nbBits[u] = (u8)nb; dt->base[u] = (u16)((x << nb) - size);
  }
  return 0;
}

/* ------------------------------------------------------------------ Huffman (A.4.5) */
typedef struct { u32 count; u16 parent; u8 byte; u8 nbBits; } hnode;

static u32 huf_set_max_height(hnode* h, u32 lastNonNull, u32 maxNbBits) {
  u32 largestBits = h[lastNonNull].nbBits;
  if (largestBits <= maxNbBits) return largestBits;
  int totalCost = 0;
  u32 baseCost = 1u << (largestBits - maxNbBits);
  int n = (int)lastNonNull;
  while (h[n].nbBits > maxNbBits) {
    totalCost += (int)(baseCost - (1u << (largestBits - h[n].nbBits)));
    h[n].nbBits = (u8)maxNbBits;
    n--;
  }
  while (h[n].nbBits == maxNbBits) n--;
  totalCost >>= (largestBits - maxNbBits);
  {
    const u32 none = 0xF0F0F0F0u;
    u32 rankLast[14];
    for (int i = 0; i < 14; i++) rankLast[i] = none;
    {
      u32 cur = maxNbBits;
      for (int pos = n; pos >= 0; pos--) {
        if (h[pos].nbBits >= cur) continue;
        cur = h[pos].nbBits;
        rankLast[maxNbBits - cur] = (u32)pos;
      }
    }
    while (totalCost > 0) {
      u32 d = hb32((u32)totalCost) + 1;
      for (; d > 1; d--) {
        u32 hp = rankLast[d], lp = rankLast[d - 1];
        if (hp == none) continue;
        if (lp == none) break;
        if (h[hp].count <= 2 * h[lp].count) break;
      }
      while (d <= 12 && rankLast[d] == none) d++;
      totalCost -= 1 << (d - 1);
      if (rankLast[d - 1] == none) rankLast[d - 1] = rankLast[d];
      h[rankLast[d]].nbBits++;
      if (rankLast[d] == 0) rankLast[d] = none;
      else {
        rankLast[d]--;
        if (h[rankLast[d]].nbBits != maxNbBits - d) rankLast[d] = none;
      }
    }
    while (totalCost < 0) {
      if (rankLast[1] == none) {
        while (h[n].nbBits == maxNbBits) n--;
        h[n + 1].nbBits--;
        rankLast[1] = (u32)(n + 1);
        totalCost++;
        continue;
      }
      h[rankLast[1] + 1].nbBits--;
      rankLast[1]++;
      totalCost++;
    }
  }
  return maxNbBits;
}

/* builds code lengths + values for symbols 0..maxSym; returns maxNbBits actually used */
unsigned zo_huf_build(zo_huf_ctable* ct, const u32* count, unsigned maxSym, unsigned maxNbBits) {
  hnode node0[512 + 2];
  hnode* h = node0 + 1;
  memset(node0, 0, sizeof(node0));
  /* sort: count descending, ties by ascending symbol */
  {
    int n = 0;
    for (unsigned s = 0; s <= maxSym; s++) {
      int pos = n++;
      while (pos > 0 && count[s] > h[pos - 1].count) { h[pos] = h[pos - 1]; pos--; }
      h[pos].count = count[s]; h[pos].byte = (u8)s;
    }
  }
  int nonNull = (int)maxSym;
  while (h[nonNull].count == 0) nonNull--;
  const int START = 256;
  int lowS = nonNull, nodeNb = START, nodeRoot = nodeNb + lowS - 1, lowN = nodeNb;
  h[nodeNb].count = h[lowS].count + h[lowS - 1].count;
  h[lowS].parent = h[lowS - 1].parent = (u16)nodeNb;
  nodeNb++; lowS -= 2;
  for (int n = nodeNb; n <= nodeRoot; n++) h[n].count = 1u << 30;
  node0[0].count = 1u << 31;
  while (nodeNb <= nodeRoot) {
    int n1 = (h[lowS].count < h[lowN].count) ? lowS-- : lowN++;
    int n2 = (h[lowS].count < h[lowN].count) ? lowS-- : lowN++;
    h[nodeNb].count = h[n1].count + h[n2].count;
    h[n1].parent = h[n2].parent = (u16)nodeNb;
    nodeNb++;
  }
  h[nodeRoot].nbBits = 0;
  for (int n = nodeRoot - 1; n >= START; n--) h[n].nbBits = h[h[n].parent].nbBits + 1;
  for (int n = 0; n <= nonNull; n++) h[n].nbBits = h[h[n].parent].nbBits + 1;
  maxNbBits = huf_set_max_height(h, (u32)nonNull, maxNbBits);
  {
    u16 nbPerRank[14] = {0}, valPerRank[14] = {0};
    for (int n = 0; n <= nonNull; n++) nbPerRank[h[n].nbBits]++;
    {
      u16 min = 0;
      for (int n = (int)maxNbBits; n > 0; n--) { valPerRank[n] = min; min += nbPerRank[n]; min >>= 1; }
    }
    memset(ct, 0, sizeof(*ct));
    for (unsigned n = 0; n <= maxSym; n++) ct->nbBits[h[n].byte] = h[n].nbBits;
    for (unsigned n = 0; n <= maxSym; n++) ct->val[n] = valPerRank[ct->nbBits[n]]++;
  }
  ct->maxSym = maxSym;
  ct->tableLog = maxNbBits;
  return maxNbBits;
}

/* FSE-compress the weight string (HUF_compressWeights); 0 = not compressible, 1 = rle */
static size_t huf_compress_weights(u8* dst, size_t cap, const u8* w, size_t n) {
  u32 count[13] = {0};
  s16 norm[13];
  unsigned maxSym = 0;
  u32 maxCount = 0;
  if (n <= 1) return 0;
  for (size_t i = 0; i < n; i++) count[w[i]]++;
  for (unsigned s = 0; s <= 12; s++) { if (count[s]) maxSym = s; if (count[s] > maxCount) maxCount = count[s]; }
  if (maxCount == n) return 1;
  if (maxCount == 1) return 0;
  unsigned t = zo_fse_optimal_tablelog(6, n, maxSym, 2);
  if (zo_fse_normalize(norm, t, count, n, maxSym, 0) <= 0) return 0;
  size_t h = zo_fse_write_ncount(dst, cap, norm, maxSym, t);
  if (!h) return 0;
  zo_fse_ctable ct;
  if (zo_fse_build_ctable(&ct, norm, maxSym, t)) return 0;
  if (n <= 2) return 0;
  zo_bitw bw; zo_bitw_init(&bw, dst + h, cap - h);
  const u8* ip = w + n;
  u32 s1, s2, bits, nb;
  size_t rem = n;
  if (rem & 1) {
    s1 = zo_fse_init_state(&ct, *--ip);
    s2 = zo_fse_init_state(&ct, *--ip);
    nb = zo_fse_encode(&ct, &s1, *--ip, &bits); zo_bitw_add(&bw, bits, nb);
  } else {
    s2 = zo_fse_init_state(&ct, *--ip);
    s1 = zo_fse_init_state(&ct, *--ip);
  }
  while (ip > w) {
    nb = zo_fse_encode(&ct, &s2, *--ip, &bits); zo_bitw_add(&bw, bits, nb);
    if (ip > w) { nb = zo_fse_encode(&ct, &s1, *--ip, &bits); zo_bitw_add(&bw, bits, nb); }
  }
  zo_bitw_add(&bw, s2, t);
  zo_bitw_add(&bw, s1, t);
  size_t c = zo_bitw_close(&bw);
  if (!c) return 0;
  return h + c;
}

/* HUF_writeCTable; returns size, 0 on failure */
size_t zo_huf_write_ctable(u8* dst, size_t cap, const zo_huf_ctable* ct) {
  u8 w[256];
  unsigned maxSym = ct->maxSym, log = ct->tableLog;
  for (unsigned n = 0; n < maxSym; n++) w[n] = ct->nbBits[n] ? (u8)(log + 1 - ct->nbBits[n]) : 0;
  if (cap < 1) return 0;
  {
    size_t h = huf_compress_weights(dst + 1, cap - 1, w, maxSym);
    if (h > 1 && h < maxSym / 2) { dst[0] = (u8)h; return h + 1; }
  }
  if (maxSym > 128) return 0;
  if (((maxSym + 1) / 2) + 1 > cap) return 0;
  dst[0] = (u8)(128 + (maxSym - 1));
  w[maxSym] = 0;
  for (unsigned n = 0; n < maxSym; n += 2) dst[(n / 2) + 1] = (u8)((w[n] << 4) + w[n + 1]);
  return ((maxSym + 1) / 2) + 1;
}

/* A.2 tree description reader. Fills weights[0..nSym-1] incl. the implied last; returns bytes consumed or 0 */
size_t zo_huf_read_weights(u8* weights, unsigned* nSymPtr, unsigned* maxBitsPtr, const u8* src, size_t n) {
  if (n < 1) return 0;
  unsigned hbyte = src[0], nw = 0;
  size_t used;
  if (hbyte >= 128) {
    nw = hbyte - 127;
    used = 1 + (nw + 1) / 2;
    if (used > n) return 0;
    for (unsigned i = 0; i < nw; i += 2) {
      weights[i] = src[1 + i / 2] >> 4;
      weights[i + 1] = src[1 + i / 2] & 15;
    }
  } else {
    used = 1 + hbyte;
    if (used > n || hbyte < 1) return 0;
    s16 norm[256]; unsigned maxSym = 255, t;
    memset(norm, 0, sizeof(norm));
    size_t h = zo_fse_read_ncount(norm, &maxSym, &t, src + 1, hbyte, 6);
    if (!h) return 0;
    zo_fse_dtable dt;
    if (zo_fse_build_dtable(&dt, norm, maxSym, t)) return 0;
    zo_bitr br;
    if (zo_bitr_init(&br, src + 1 + h, hbyte - h)) return 0;
    u32 s1 = zo_bitr_read(&br, t), s2 = zo_bitr_read(&br, t);
    for (;;) {
      if (nw >= 254) return 0;
      weights[nw++] = dt.sym[s1];
      s1 = dt.base[s1] + zo_bitr_read(&br, dt.nbBits[s1]);
      if (br.pos < 0) { weights[nw++] = dt.sym[s2]; break; }
      if (nw >= 254) return 0;
      weights[nw++] = dt.sym[s2];
      s2 = dt.base[s2] + zo_bitr_read(&br, dt.nbBits[s2]);
      if (br.pos < 0) { weights[nw++] = dt.sym[s1]; break; }
    }
  }
  u32 total = 0;
  for (unsigned i = 0; i < nw; i++) { if (weights[i] > 11) return 0; total += (1u << weights[i]) >> 1; }
  if (total == 0) return 0;
  unsigned maxBits = hb32(total) + 1;
  if (maxBits > 12) return 0;                              /* HUF_readStats: tableLog > HUF_TABLELOG_MAX (the format allows 11, libzstd 12) */
  u32 rest = (1u << maxBits) - total;
  if (rest == 0 || (rest & (rest - 1))) return 0;
  weights[nw] = (u8)(hb32(rest) + 1);
  { unsigned r1 = 0; for (unsigned i = 0; i <= nw; i++) r1 += weights[i] == 1; if (r1 < 2) return 0; }   /* "at least 2 elts of rank 1" */
  *nSymPtr = nw + 1;
  *maxBitsPtr = maxBits;
  return used;
}

int zo_huf_build_dtable(zo_huf_dtable* dt, const u8* weights, unsigned nSym, unsigned maxBits) {
  u32 pos = 0;
  dt->maxBits = maxBits;
  for (unsigned w = 1; w <= maxBits; w++) {
    for (unsigned s = 0; s < nSym; s++) {
      if (weights[s] != w) continue;
      u32 len = 1u << (w - 1);
      if (pos + len > (1u << maxBits)) return -1;
      for (u32 i = 0; i < len; i++) { dt->sym[pos + i] = (u8)s; dt->nbBits[pos + i] = (u8)(maxBits + 1 - w); }
      pos += len;
    }
  }
  return pos == (1u << maxBits) ? 0 : -1;
}
