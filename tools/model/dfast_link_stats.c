/* BRING-UP MODEL — statistics for the "parse-independent links + inserted bitmaps" formulation of dfast (round 4).
 *
 * Not product code and not the oracle. It runs zstd 1.4.9's double-fast parse (oracle/zo_encode.c: mf_dfast, reference call site
 * zra.cpp:219) once with the real hash tables and, beside it, answers every table lookup a second way:
 *   prevL[p] / prevS[p] = the previous position with the same long / short bucket (built for EVERY position, parse-independent),
 *   insL / insS         = one bit per position: has the parse inserted it into the long / short table,
 *   lookup(p)           = walk prev[] from p to the first position whose inserted bit is set.
 * It checks that both answers agree on every lookup (the exactness claim: insert positions are non-decreasing, so the table holds
 * the most recent inserted position of the bucket) and prints what the walk costs: steps per lookup, histogram, lookups per frame.
 * Build + run: gcc -O2 -I../../oracle -o /tmp/dfast_link_stats dfast_link_stats.c ../../oracle/zo_entropy.c ../../oracle/zo_decode.c -lm -ldl
 *              /tmp/dfast_link_stats /tmp/corpus64m.bin 65536 1024 3
 */
#include "../../oracle/zo_encode.c"
#include <stdio.h>

static u64 H_L[66], H_S[66], nLook, nVisited, nSeq, nFrames, bad, insLcnt, insScnt, tagHitL, tagHitS, realHitL, realHitS;
static u64 depL_hist[4], firstK[8], lookPredL, lookPredS, insSuccL, insSuccS, lookL, lookS;
static u8 *sucL, *sucS;
/* what a write-back cache of table cells in LDS would leave of the table traffic (per table, direct mapped by the bucket's low bits) */
#define NC 4
static const u32 CS[NC] = {256, 512, 1024, 2048};
static u32 cTagL[NC][2048], cTagS[NC][2048]; static u8 cDirL[NC][2048], cDirS[NC][2048];
static u64 cRdL[NC], cRdS[NC], cWrL[NC], cWrS[NC];
static void c_look(u32 tag[][2048], u8 dir[][2048], u64* rd, u64* wr, u32 b) { for (int k = 0; k < NC; k++) { u32 s_ = b & (CS[k] - 1); if (tag[k][s_] != b + 1) { rd[k]++; if (dir[k][s_]) wr[k]++; tag[k][s_] = b + 1; dir[k][s_] = 0; } } }
static void c_ins(u32 tag[][2048], u8 dir[][2048], u64* wr, u32 b) { for (int k = 0; k < NC; k++) { u32 s_ = b & (CS[k] - 1); if (tag[k][s_] != b + 1) { if (dir[k][s_]) wr[k]++; tag[k][s_] = b + 1; } dir[k][s_] = 1; } }

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* f = fopen(argv[1], "rb"); if (!f) return 2;
  size_t fs = argc > 2 ? (size_t)atol(argv[2]) : 65536, nf = argc > 3 ? (size_t)atol(argv[3]) : 64; int level = argc > 4 ? atoi(argv[4]) : 3;
  size_t skip = argc > 5 ? (size_t)atol(argv[5]) : 0;
  u8* src = (u8*)malloc(fs + 16);
  u32* prevL = (u32*)malloc(4 * fs), *prevS = (u32*)malloc(4 * fs);
  u8* insL = (u8*)malloc(fs), *insS = (u8*)malloc(fs);
  fseek(f, (long)(skip * fs), SEEK_SET);
  while (nFrames < nf && fread(src, 1, fs, f) == fs) {
    cctx c; if (cctx_init(&c, level, fs)) return 3;
    if (c.cp.strategy != 2) { printf("not dfast\n"); return 4; }
    u32 hlog = c.cp.hashLog, clog = c.cp.chainLog, mls = c.cp.minMatch;
    /* ---- parse-independent pre-pass */
    u32* headL = (u32*)calloc((size_t)1 << hlog, 4); u32* headS = (u32*)calloc((size_t)1 << clog, 4);
    for (size_t p = 1; p + 8 <= fs; p++) {
      u32 bl = hash8(src + p, hlog), bs = hashN(src + p, clog, mls);
      prevL[p] = headL[bl]; headL[bl] = (u32)p; prevS[p] = headS[bs]; headS[bs] = (u32)p;
    }
    if (!sucL) { sucL = (u8*)malloc(fs); sucS = (u8*)malloc(fs); }
    memset(sucL, 0, fs); memset(sucS, 0, fs);
    for (size_t p = 1; p + 8 <= fs; p++) { if (prevL[p]) sucL[prevL[p]] = 1; if (prevS[p]) sucS[prevS[p]] = 1; }
    free(headL); free(headS);
    memset(insL, 0, fs); memset(insS, 0, fs);
    for (int k = 0; k < NC; k++) for (u32 i_ = 0; i_ < 2048; i_++) { if (cDirL[k][i_]) cWrL[k]++; if (cDirS[k][i_]) cWrS[k]++; }
    memset(cTagL, 0, sizeof(cTagL)); memset(cTagS, 0, sizeof(cTagS)); memset(cDirL, 0, sizeof(cDirL)); memset(cDirS, 0, sizeof(cDirS));
    /* ---- the serial parse with the real tables, every lookup answered twice */
    u32* HL = c.hashTable; u32* HS = c.chainTable;
    u32 o1 = 1, o2 = 4, saved;
    size_t bs0 = 0, be = fs, anchor = 0, ilimit = be - 8;
    u32 psi = lowest_at(&c, (u32)be + 1);
    size_t ip = mf_prologue(&c, bs0, psi - 1, &o1, &o2, &saved);
#define WALK(prev, ins, p, out, hist) { u32 q_ = prev[p], st_ = 1; while (q_ && !ins[q_]) { q_ = prev[q_]; st_++; } out = q_; hist[st_ > 64 ? 65 : st_]++; if (q_ == 0) hist[0]++; }
#define INS_L(p) { HL[hash8(src + (p), hlog)] = (u32)(p) + 1; insL[p] = 1; insLcnt++; insSuccL += sucL[p]; c_ins(cTagL, cDirL, cWrL, hash8(src + (p), hlog)); }
#define INS_S(p) { HS[hashN(src + (p), clog, mls)] = (u32)(p) + 1; insS[p] = 1; insScnt++; insSuccS += sucS[p]; c_ins(cTagS, cDirS, cWrS, hashN(src + (p), clog, mls)); }
    while (ip < ilimit) {
      size_t top = ip, ml; u32 offVal;
      u32 hL = hash8(src + ip, hlog), hS = hashN(src + ip, clog, mls);
      u32 mL = HL[hL], mS = HS[hS];
      c_look(cTagL, cDirL, cRdL, cWrL, hL); c_look(cTagS, cDirS, cRdS, cWrS, hS);
      u32 wL, wS; WALK(prevL, insL, ip, wL, H_L); WALK(prevS, insS, ip, wS, H_S);
      nLook += 2; nVisited++; lookL++; lookS++; lookPredL += prevL[ip] != 0; lookPredS += prevS[ip] != 0;
      if ((mL ? mL - 1 : 0) != wL || (mS ? mS - 1 : 0) != wS) { bad++; if (bad < 5) printf("MISMATCH frame %llu ip %zu: table L %u S %u, walk L %u S %u\n", (unsigned long long)nFrames, ip, mL, mS, wL, wS); }
      INS_L(ip); INS_S(ip);
      if (mL > 1 && rd64(src + mL - 1) == rd64(src + ip)) realHitL++;
      if (mS > 1 && rd32(src + mS - 1) == rd32(src + ip)) realHitS++;
      if (o1 > 0 && rd32(src + ip + 1 - o1) == rd32(src + ip + 1)) {
        ml = count_eq(src, ip + 5, ip + 5 - o1, be) + 4; ip++; offVal = 1;
      } else {
        size_t m;
        if (mL > psi && rd64(src + mL - 1) == rd64(src + ip)) {
          m = mL - 1; ml = count_eq(src, ip + 8, m + 8, be) + 8;
        } else if (mS > psi && rd32(src + mS - 1) == rd32(src + ip)) {
          u32 h3 = hash8(src + ip + 1, hlog), m3 = HL[h3];
          c_look(cTagL, cDirL, cRdL, cWrL, h3);
          u32 w3; WALK(prevL, insL, ip + 1, w3, H_L); nLook++; lookL++; lookPredL += prevL[ip + 1] != 0;
          if ((m3 ? m3 - 1 : 0) != w3) { bad++; if (bad < 5) printf("MISMATCH3 frame %llu ip %zu: table %u walk %u\n", (unsigned long long)nFrames, ip, m3, w3); }
          INS_L(ip + 1);
          if (m3 > psi && rd64(src + m3 - 1) == rd64(src + ip + 1)) { m = m3 - 1; ip++; ml = count_eq(src, ip + 8, m + 8, be) + 8; }
          else { m = mS - 1; ml = count_eq(src, ip + 4, m + 4, be) + 4; }
        } else { ip += ((ip - anchor) >> 8) + 1; continue; }
        u32 off = (u32)(ip - m);
        while (ip > anchor && m > psi - 1 && src[ip - 1] == src[m - 1]) { ip--; m--; ml++; }
        o2 = o1; o1 = off; offVal = off + 3;
      }
      (void)offVal; nSeq++;
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
    cctx_free(&c);
    nFrames++;
  }
  printf("frames %llu level %d fs %zu | lookup mismatches %llu | per frame: visited %.0f lookups %.0f seqs %.0f insL %.0f insS %.0f realHitL %.0f realHitS %.0f\n",
         (unsigned long long)nFrames, level, fs, (unsigned long long)bad, (double)nVisited / nFrames, (double)nLook / nFrames, (double)nSeq / nFrames,
         (double)insLcnt / nFrames, (double)insScnt / nFrames, (double)realHitL / nFrames, (double)realHitS / nFrames);
  printf("per frame: long lookups %.0f of which the bucket has an earlier position %.0f | short %.0f / %.0f | long insertions %.0f of which the bucket has a later position %.0f | short %.0f / %.0f\n",
         (double)lookL / nFrames, (double)lookPredL / nFrames, (double)lookS / nFrames, (double)lookPredS / nFrames, (double)insLcnt / nFrames, (double)insSuccL / nFrames, (double)insScnt / nFrames, (double)insSuccS / nFrames);
  for (int k = 0; k < NC; k++) printf("LDS cell cache, %4u entries per table: per frame table reads %.0f + %.0f (of %.0f + %.0f lookups), table writes %.0f + %.0f (of %.0f + %.0f insertions)\n", CS[k],
         (double)cRdL[k] / nFrames, (double)cRdS[k] / nFrames, (double)lookL / nFrames, (double)lookS / nFrames, (double)cWrL[k] / nFrames, (double)cWrS[k] / nFrames, (double)insLcnt / nFrames, (double)insScnt / nFrames);
  for (int t = 0; t < 2; t++) {
    u64* H = t ? H_S : H_L; u64 tot = 0, steps = 0; for (int i = 1; i < 66; i++) { tot += H[i]; steps += (u64)i * H[i]; }
    printf("%s walk: lookups/frame %.0f  mean steps %.2f  empty-ended %.1f %%  hist(1..8, 9-16, 17-64, >64):", t ? "short" : "long ", (double)tot / nFrames, (double)steps / tot, 100.0 * H[0] / tot);
    u64 a = 0; for (int i = 1; i <= 8; i++) printf(" %.1f", 100.0 * H[i] / tot);
    for (int i = 9; i <= 16; i++) a += H[i]; printf(" | %.2f", 100.0 * a / tot);
    a = 0; for (int i = 17; i <= 64; i++) a += H[i]; printf(" %.2f", 100.0 * a / tot);
    printf(" %.3f\n", 100.0 * H[65] / tot);
  }
  return bad != 0;
}
