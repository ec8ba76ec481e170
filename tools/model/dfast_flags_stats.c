/* BRING-UP MODEL (round 5) — what the bucket flags of the in-wave pre-pass leave of the dfast table traffic.
 *
 * Not product code and not the oracle. Runs zstd 1.4.9's double-fast parse (oracle/zo_encode.c: mf_dfast, reference call site zra.cpp:219)
 * with the real tables and counts, per frame, the table reads and writes that remain under each skipping rule:
 *   reads : none | the kernel's LDS filter (1 bit per 2 long / 4 short buckets, set at insertion) | filter + "bucket has an earlier position"
 *           | an exact inserted bit per bucket
 *   writes: all | "bucket has a later position" exact | the same as the in-wave backward sweep computes it (a superset: inside a 64-position
 *           window every lane that shares a count slot (bucket & (SL-1)) with another participating lane counts as having a later mate)
 * Build + run: gcc -O2 -I../../oracle -o /tmp/dfast_flags_stats dfast_flags_stats.c ../../oracle/zo_entropy.c ../../oracle/zo_decode.c -lm -ldl
 *              /tmp/dfast_flags_stats /tmp/corpus64m.bin 65536 1024 3
 */
#include "../../oracle/zo_encode.c"
#include <stdio.h>
static u64 rdAllL, rdAllS, rdAllEL, rdAllES;
static u64 nFrames, lookL, lookS, insL_, insS_, rdFiltL, rdFiltS, rdFEarlL, rdFEarlS, rdExactL, rdExactS, wrLaterL, wrLaterS, wrSweepL, wrSweepS, nSeq;
int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* f = fopen(argv[1], "rb"); if (!f) return 2;
  size_t fs = argc > 2 ? (size_t)atol(argv[2]) : 65536, nf = argc > 3 ? (size_t)atol(argv[3]) : 64; int level = argc > 4 ? atoi(argv[4]) : 3;
  u32 SL = argc > 5 ? (u32)atoi(argv[5]) : 512, NP = argc > 6 ? (u32)atoi(argv[6]) : 2;
  u8* src = (u8*)malloc(fs + 16);
  u8* earlL = (u8*)malloc(fs), *earlS = (u8*)malloc(fs), *lateL = (u8*)malloc(fs), *lateS = (u8*)malloc(fs), *swL = (u8*)malloc(fs), *swS = (u8*)malloc(fs);
  while (nFrames < nf && fread(src, 1, fs, f) == fs) {
    cctx c; if (cctx_init(&c, level, fs)) return 3;
    if (c.cp.strategy != 2) { printf("not dfast\n"); return 4; }
    u32 hlog = c.cp.hashLog, clog = c.cp.chainLog, mls = c.cp.minMatch;
    size_t last = fs - 8;
    u8* seenL = (u8*)calloc((size_t)1 << hlog, 1), *seenS = (u8*)calloc((size_t)1 << clog, 1);
    for (size_t p = 0; p <= last; p++) { u32 bl = hash8(src + p, hlog), bs = hashN(src + p, clog, mls); earlL[p] = seenL[bl]; seenL[bl] = 1; earlS[p] = seenS[bs]; seenS[bs] = 1; }
    memset(seenL, 0, (size_t)1 << hlog); memset(seenS, 0, (size_t)1 << clog);
    for (size_t p = last + 1; p-- > 0;) { u32 bl = hash8(src + p, hlog), bs = hashN(src + p, clog, mls); lateL[p] = seenL[bl]; seenL[bl] = 1; lateS[p] = seenS[bs]; seenS[bs] = 1; }
    /* the sweep as the kernel does it: NP passes over bucket ranges, windows of 64 from the end, seen bits from earlier windows + slot mates */
    memset(swL, 0, fs); memset(swS, 0, fs);
    for (u32 h = 0; h < NP; h++) {
      memset(seenL, 0, (size_t)1 << hlog); memset(seenS, 0, (size_t)1 << clog);
      for (size_t w = (last / 64) + 1; w-- > 0;) {
        u32 cntL[4096] = {0}, cntS[4096] = {0};
        for (u32 l = 0; l < 64; l++) { size_t p = w * 64 + l; if (p > last) continue; u32 bl = hash8(src + p, hlog), bs = hashN(src + p, clog, mls);
          if (bl * NP >> hlog == h) cntL[bl & (SL - 1)]++; if (bs * NP >> clog == h) cntS[bs & (SL - 1)]++; }
        for (u32 l = 0; l < 64; l++) { size_t p = w * 64 + l; if (p > last) continue; u32 bl = hash8(src + p, hlog), bs = hashN(src + p, clog, mls);
          if (bl * NP >> hlog == h) swL[p] = seenL[bl] || cntL[bl & (SL - 1)] > 1; if (bs * NP >> clog == h) swS[p] = seenS[bs] || cntS[bs & (SL - 1)] > 1; }
        for (u32 l = 0; l < 64; l++) { size_t p = w * 64 + l; if (p > last) continue; seenL[hash8(src + p, hlog)] = 1; seenS[hashN(src + p, clog, mls)] = 1; }
      }
    }
    for (size_t p = 0; p <= last; p++) if ((lateL[p] && !swL[p]) || (lateS[p] && !swS[p])) { printf("sweep misses a later mate at %zu\n", p); return 5; }
    memset(seenL, 0, (size_t)1 << hlog); memset(seenS, 0, (size_t)1 << clog);       /* now: inserted bits per bucket */
    u8* fL = (u8*)calloc(((size_t)1 << hlog) / 2, 1), *fS = (u8*)calloc(((size_t)1 << clog) / 4, 1);
    u8* gL = (u8*)calloc(((size_t)1 << hlog) / 2, 1), *gS = (u8*)calloc(((size_t)1 << clog) / 4, 1);
    u32* HL = c.hashTable; u32* HS = c.chainTable;
    u32 o1 = 1, o2 = 4, saved;
    size_t bs0 = 0, be = fs, anchor = 0, ilimit = be - 8;
    u32 psi = lowest_at(&c, (u32)be + 1);
    size_t ip = mf_prologue(&c, bs0, psi - 1, &o1, &o2, &saved);
#define LOOK_L(p, b) { lookL++; rdFiltL += fL[(b) >> 1]; rdFEarlL += fL[(b) >> 1] && earlL[p]; rdExactL += seenL[b]; rdAllL += gL[(b) >> 1]; rdAllEL += gL[(b) >> 1] && earlL[p]; }
#define LOOK_S(p, b) { lookS++; rdFiltS += fS[(b) >> 2]; rdFEarlS += fS[(b) >> 2] && earlS[p]; rdExactS += seenS[b]; rdAllS += gS[(b) >> 2]; rdAllES += gS[(b) >> 2] && earlS[p]; }
#define INS_L(p) { u32 b_ = hash8(src + (p), hlog); HL[b_] = (u32)(p) + 1; insL_++; wrLaterL += lateL[p]; wrSweepL += swL[p]; if (lateL[p]) { fL[b_ >> 1] = 1; } seenL[b_] = 1; gL[b_ >> 1] = 1; }
#define INS_S(p) { u32 b_ = hashN(src + (p), clog, mls); HS[b_] = (u32)(p) + 1; insS_++; wrLaterS += lateS[p]; wrSweepS += swS[p]; if (lateS[p]) { fS[b_ >> 2] = 1; } seenS[b_] = 1; gS[b_ >> 2] = 1; }
    while (ip < ilimit) {
      size_t top = ip, ml;
      u32 hL = hash8(src + ip, hlog), hS = hashN(src + ip, clog, mls);
      u32 mL = HL[hL], mS = HS[hS];
      LOOK_L(ip, hL); LOOK_S(ip, hS);
      INS_L(ip); INS_S(ip);
      if (o1 > 0 && rd32(src + ip + 1 - o1) == rd32(src + ip + 1)) { ml = count_eq(src, ip + 5, ip + 5 - o1, be) + 4; ip++; }
      else {
        size_t m;
        if (mL > psi && rd64(src + mL - 1) == rd64(src + ip)) { m = mL - 1; ml = count_eq(src, ip + 8, m + 8, be) + 8; }
        else if (mS > psi && rd32(src + mS - 1) == rd32(src + ip)) {
          u32 h3 = hash8(src + ip + 1, hlog), m3 = HL[h3];
          LOOK_L(ip + 1, h3);
          INS_L(ip + 1);
          if (m3 > psi && rd64(src + m3 - 1) == rd64(src + ip + 1)) { m = m3 - 1; ip++; ml = count_eq(src, ip + 8, m + 8, be) + 8; }
          else { m = mS - 1; ml = count_eq(src, ip + 4, m + 4, be) + 4; }
        } else { ip += ((ip - anchor) >> 8) + 1; continue; }
        u32 off = (u32)(ip - m);
        while (ip > anchor && m > psi - 1 && src[ip - 1] == src[m - 1]) { ip--; m--; ml++; }
        o2 = o1; o1 = off;
      }
      nSeq++;
      ip += ml; anchor = ip;
      if (ip <= ilimit) {
        size_t q = top + 2;
        INS_L(q); INS_L(ip - 2); INS_S(q); INS_S(ip - 1);
        while (ip <= ilimit && o2 > 0 && rd32(src + ip) == rd32(src + ip - o2)) {
          size_t rl = count_eq(src, ip + 4, ip + 4 - o2, be) + 4;
          u32 t = o2; o2 = o1; o1 = t;
          INS_S(ip); INS_L(ip);
          nSeq++;
          ip += rl; anchor = ip;
        }
      }
    }
    free(seenL); free(seenS); free(fL); free(fS); free(gL); free(gS);
    cctx_free(&c);
    nFrames++;
  }
  double n = (double)nFrames;
  printf("frames %llu level %d fs %zu slots %u passes %u | per frame: sequences %.0f\n", (unsigned long long)nFrames, level, fs, SL, NP, nSeq / n);
  printf("reads  long : lookups %.0f | LDS filter (of buckets written under the later rule) %.0f | filter + earlier %.0f | exact inserted bit %.0f\n", lookL / n, rdFiltL / n, rdFEarlL / n, rdExactL / n);
  printf("reads  short: lookups %.0f | LDS filter %.0f | filter + earlier %.0f | exact inserted bit %.0f\n", lookS / n, rdFiltS / n, rdFEarlS / n, rdExactS / n);
  printf("reads with the filter marked by EVERY insertion (today): long %.0f short %.0f | + earlier: %.0f %.0f\n", rdAllL / n, rdAllS / n, rdAllEL / n, rdAllES / n);
  printf("writes long : insertions %.0f | bucket has a later position %.0f | as the in-wave sweep sees it %.0f\n", insL_ / n, wrLaterL / n, wrSweepL / n);
  printf("writes short: insertions %.0f | bucket has a later position %.0f | as the in-wave sweep sees it %.0f\n", insS_ / n, wrLaterS / n, wrSweepS / n);
  return 0;
}
