/* BRING-UP MODEL — CPU emulation (lane arrays) of the round-4 "link" dfast parse of zra_amd/csrc/zra_encode_lk.hip.
 *
 * Not product code and not the oracle: it restates the KERNEL's algorithm (64 lanes as arrays, ballots as loops) so that its
 * exactness argument can be fuzzed on the CPU against oracle/zo_encode.c (the restatement of zstd 1.4.9's
 * ZSTD_compressBlock_doubleFast, reference call site zra.cpp:219) before a GPU minute is spent. Build + run: tools/model/run_link.sh
 *
 * The formulation (frames of at most 64 KiB, one block):
 *   pre-pass (parse-independent, zra_lk_prepass_kernel): for EVERY position p its three predecessors in the long-hash bucket chain and
 *     in the short-hash bucket chain (q1 > q2 > q3, 0 = none) plus one bit per predecessor: "its 8 (long) / 4 (short) bytes equal p's";
 *   parse (zra_lk_parse_kernel): the hash tables are replaced by two bitmaps "position was inserted into the long / short table".
 *     dfast's insert positions never decrease, so a table cell holds the most recent INSERTED position of its bucket = the first
 *     chain predecessor whose bit is set. No table is written, no hash is computed in the parse.
 *   windows sit on a fixed 64-position grid (stride-1 parse) or slide with the stride once 256 literals went by without a match;
 *   a predecessor at or beyond the parse position is "pending": the lane assumes it will be inserted (true in a literal run) and
 *     remembers the lane it depends on; after every match the lanes behind the match whose dependency was skipped are walked again;
 *   a chain whose first three predecessors are all not inserted is "deep": the lane is a suspect that is walked through memory only
 *     when the parse reaches it.
 */
#include "../../oracle/zo_encode.c"
#include <stdio.h>
#include <time.h>

typedef struct { u16 q[3]; u8 fl; } ent;

static struct { u64 frames, windows, strideWin, seqs, repSeq, immRep, rewalkEvents, rewalkLanes, deepLanes, deepResolved, deepSteps, slowProbe,
                tripB, longFwd, backAny, visited, winWithDeep, deepAnyEq, deepHit, deepEmpty,
                v2Win, v2Rep, v2Sat, v2Back, v2Deep, v2Probe, v2ImmOut, v2Seqs; } ST;

static void prepass(const u8* src, u32 n, u32 hlog, u32 clog, u32 mls, ent* EL, ent* ES) {
  u16* headL = (u16*)calloc((size_t)1 << hlog, 2); u16* headS = (u16*)calloc((size_t)1 << clog, 2);
  u16* lkL = (u16*)calloc(n + 1, 2); u16* lkS = (u16*)calloc(n + 1, 2);
  memset(EL, 0, sizeof(ent) * n); memset(ES, 0, sizeof(ent) * n);
  for (u32 p = 1; p + 8 <= n; p++) {
    u32 bl = hash8(src + p, hlog), bs = hashN(src + p, clog, mls);
    lkL[p] = headL[bl]; headL[bl] = (u16)p; lkS[p] = headS[bs]; headS[bs] = (u16)p;
  }
  for (u32 p = 1; p + 8 <= n; p++) {
    u32 q = lkL[p];
    for (int k = 0; k < 3 && q; k++) { EL[p].q[k] = (u16)q; if (rd64(src + q) == rd64(src + p)) EL[p].fl |= 1u << k; q = lkL[q]; }
    q = lkS[p];
    for (int k = 0; k < 3 && q; k++) { ES[p].q[k] = (u16)q; if (rd32(src + q) == rd32(src + p)) ES[p].fl |= 1u << k; q = lkS[q]; }
  }
  free(headL); free(headS); free(lkL); free(lkS);
}

typedef struct { u32 cand; int hit, dep, deep; u32 last; } wres;
static inline int bit_get(const u64* bm, u32 p) { return (int)((bm[p >> 6] >> (p & 63)) & 1); }
static inline void bit_set(u64* bm, u32 p) { bm[p >> 6] |= 1ull << (p & 63); }

/* the lane's walk over its three predecessors: positions < ipNow are decided by the bitmap, positions >= ipNow are pending */
static wres walk3(const ent* E, u32 p, u32 ipNow, u32 base, u32 s, const u64* ins) {
  wres r; r.cand = 0; r.hit = 0; r.dep = -1; r.deep = 0; r.last = 0;
  for (int k = 0; k < 3; k++) {
    u32 q = E[p].q[k];
    if (!q) return r;
    if (q >= ipNow) {
      if (s == 1) { r.cand = q; r.hit = (E[p].fl >> k) & 1; r.dep = (int)(q - base); return r; }
      if ((q - ipNow) % s == 0) { r.cand = q; r.hit = (E[p].fl >> k) & 1; r.dep = (int)((q - ipNow) / s); return r; }
      continue;                                        /* between the lanes of a strided window: never inserted before the window ends */
    }
    if (bit_get(ins, q)) { r.cand = q; r.hit = (E[p].fl >> k) & 1; return r; }
    r.last = q;
  }
  r.deep = 1; r.last = E[p].q[2];
  return r;
}
/* the memory walk of a deep lane (or of a probe position outside the window): exact, serial */
static wres walk_mem(const ent* E, const u8* src, u32 p, u32 from, const u64* ins, int isLong) {
  wres r; r.cand = 0; r.hit = 0; r.dep = -1; r.deep = 0; r.last = 0;
  u32 q = from;                                        /* `from` itself is known not to be inserted */
  { u32 t = from; int any = 0; for (;;) { t = E[t].q[0]; if (!t) break; if (isLong ? rd64(src + t) == rd64(src + p) : rd32(src + t) == rd32(src + p)) { any = 1; break; } } ST.deepAnyEq += any; }
  for (;;) {
    int found = 0;
    for (int k = 0; k < 3; k++) {
      u32 t = E[q].q[k];
      ST.deepSteps++;
      if (!t) return r;
      if (bit_get(ins, t)) { r.cand = t; found = 1; break; }
      if (k == 2) q = t;
    }
    if (found) break;
  }
  r.hit = isLong ? rd64(src + r.cand) == rd64(src + p) : rd32(src + r.cand) == rd32(src + p);
  ST.deepHit += r.hit;
  return r;
}

static size_t model_dfast(cctx* c, const ent* EL, const ent* ES, u64* insL, u64* insS, const u8* src, u32 bs, u32 be, u32 rep[3]) {
  u32 o1 = rep[0], o2 = rep[1], saved;
  u32 anchor = bs, ilimit = be >= 8 ? be - 8 : 0;
  u32 ip = (u32)mf_prologue(c, bs, 0, &o1, &o2, &saved);
  while (ip < ilimit) {
    /* ---------------------------------------------------------------- window build */
    const u32 run = ip - anchor;
    u32 s = 1, base, lo, hi;
    if (run < 256) { base = ip & ~63u; lo = ip - base; hi = 64; if (anchor + 256 - base < hi) hi = anchor + 256 - base; if (ilimit - base < hi) hi = ilimit - base; }
    else {
      s = (run >> 8) + 1; base = ip; lo = 0;
      hi = 64; { u32 a = (256 * s - run + s - 1) / s, b = (ilimit - ip + s - 1) / s; if (a < hi) hi = a; if (b < hi) hi = b; }
      ST.strideWin++;
    }
    ST.windows++; ST.v2Win++;
    wres L[64], S[64]; u32 P[64];
    u64 deepL = 0, deepS = 0, LH = 0, SH = 0, AM = 0;
    for (u32 l = lo; l < hi; l++) {
      P[l] = base + l * s;
      L[l] = walk3(EL, P[l], ip, base, s, insL); S[l] = walk3(ES, P[l], ip, base, s, insS);
      AM |= 1ull << l;
      if (L[l].deep) deepL |= 1ull << l; else if (L[l].hit) LH |= 1ull << l;
      if (S[l].deep) deepS |= 1ull << l; else if (S[l].hit) SH |= 1ull << l;
    }
    if (deepL | deepS) ST.winWithDeep++;
    ST.deepLanes += (u64)__builtin_popcountll(deepL) + (u64)__builtin_popcountll(deepS);
    u64 mkL = 0, mkS = 0;                               /* pending insertions, window-relative (stride 1: bit = lane; strided: kept per position below) */
#define FLUSH() { if (s == 1) { insL[base >> 6] |= mkL; insS[base >> 6] |= mkS; } else { for (u32 l_ = 0; l_ < 64; l_++) { if ((mkL >> l_) & 1) bit_set(insL, base + l_ * s); if ((mkS >> l_) & 1) bit_set(insS, base + l_ * s); } } }
    /* insert position q (>= base): a mask bit while it lies in the stride-1 window's word, the bitmap itself otherwise */
#define INS(q, doL, doS) { const u32 q_ = (q); if (s == 1 && q_ - base < 64) { if (doL) mkL |= 1ull << (q_ - base); if (doS) mkS |= 1ull << (q_ - base); } else { if (doL) bit_set(insL, q_); if (doS) bit_set(insS, q_); } }
    u32 cur = lo;
    int done = 0;
    for (;;) {
      const u64 live = AM & (~0ull << cur);
      u64 RH = 0;
      for (u32 l = cur; l < hi; l++) if (o1 > 0 && rd32(src + P[l] + 1 - o1) == rd32(src + P[l] + 1)) RH |= 1ull << l;
      const u64 hm = (RH | LH | SH | deepL | deepS) & live;
      if (!hm) { mkL |= live; mkS |= live; ST.visited += (u64)__builtin_popcountll(live); ip = base + hi * s; FLUSH(); break; }
      const u32 f = (u32)__builtin_ctzll(hm);
      const u32 top = P[f];
      const int isRep = (int)((RH >> f) & 1);
      if (!isRep && (((deepL | deepS) >> f) & 1)) {
        /* suspect lane reached: settle its chain(s) through memory, then look at the lane again */
        FLUSH();                                        /* (the walk reads the bitmap: pending lanes below f are all inserted — see below) */
        /* lanes cur..f-1 are literal positions: inserted; they are pending in the masks only if not yet marked — mark them now */
        { const u64 vis = live & ((1ull << f) - 1); if (s == 1) { insL[base >> 6] |= vis; insS[base >> 6] |= vis; } else for (u32 l_ = cur; l_ < f; l_++) { bit_set(insL, P[l_]); bit_set(insS, P[l_]); } }
        ST.v2Deep++;
        if ((deepL >> f) & 1) { L[f] = walk_mem(EL, src, top, L[f].last, insL, 1); deepL &= ~(1ull << f); if (L[f].hit) LH |= 1ull << f; ST.deepResolved++; }
        if ((deepS >> f) & 1) { S[f] = walk_mem(ES, src, top, S[f].last, insS, 0); deepS &= ~(1ull << f); if (S[f].hit) SH |= 1ull << f; ST.deepResolved++; }
        continue;
      }
      { const u64 vis = live & ((f == 63 ? 0 : (1ull << (f + 1))) - 1); mkL |= vis; mkS |= vis; ST.visited += (u64)__builtin_popcountll(vis); }
      ip = top;
      u32 m, known, offVal = 1;
      if (isRep) { ip = top + 1; m = ip - o1; known = 4; ST.repSeq++; ST.v2Rep++; }
      else if (((LH >> f) & 1) && !((deepL >> f) & 1)) { m = L[f].cand; known = 8; }
      else {
        /* short hit: long-table probe at top+1 */
        int hit3; u32 m3;
        if (s == 1 && f + 1 < hi && !((deepL >> (f + 1)) & 1)) { hit3 = (int)((LH >> (f + 1)) & 1); m3 = L[f + 1].cand; mkL |= 1ull << (f + 1); }
        else {
          ST.slowProbe++; ST.v2Probe++;
          FLUSH();
          wres r3;
          if (s == 1 && f + 1 < hi) { r3 = walk_mem(EL, src, top + 1, L[f + 1].last, insL, 1); }   /* deep lane f+1: continue its walk */
          else {
            /* position outside the window: its entry is read from memory, then the same walk */
            r3 = walk3(EL, top + 1, top + 1, 0, 1, insL);
            if (r3.deep) r3 = walk_mem(EL, src, top + 1, r3.last, insL, 1);
          }
          hit3 = r3.hit; m3 = r3.cand;
          INS(top + 1, 1, 0);
          if (s == 1 && f + 1 < hi) { deepL &= ~(1ull << (f + 1)); L[f + 1] = r3; if (hit3) LH |= 1ull << (f + 1); }
        }
        if (hit3) { m = m3; ip = top + 1; known = 8; }
        else { m = S[f].cand; known = 4; }
      }
      const u32 off = ip - m;
      u32 ml = known + (u32)count_eq(src, ip + known, m + known, be);
      if (ml > known + 120) ST.longFwd++;
      if (!isRep && ml > known + 14) ST.v2Sat++;
      if (!isRep) {
        u32 back = 0;
        while (ip - back > anchor && m - back > 0 && src[ip - back - 1] == src[m - back - 1]) back++;
        if (back) ST.backAny++;
        if (m > 0 && ip > anchor && src[ip - 1] == src[m - 1]) ST.v2Back++;
        ip -= back; ml += back;
        o2 = o1; o1 = off; offVal = off + 3;
      }
      emit(c, src, anchor, ip - anchor, ml, offVal); ST.seqs++; ST.v2Seqs++;
      ip += ml; anchor = ip;
      if (ip - base >= 64 && o2) ST.v2ImmOut++;
      if (ip > ilimit) { FLUSH(); done = 1; break; }
      INS(top + 2, 1, 1); INS(ip - 2, 1, 0); INS(ip - 1, 0, 1);
      if (ip - base >= 128) ST.tripB++;
      while (ip <= ilimit && o2 > 0 && rd32(src + ip) == rd32(src + ip - o2)) {
        const u32 rl = (u32)count_eq(src, ip + 4, ip + 4 - o2, be) + 4;
        const u32 t = o2; o2 = o1; o1 = t;
        INS(ip, 1, 1);
        emit(c, src, anchor, 0, rl, 1); ST.seqs++; ST.immRep++;
        ip += rl; anchor = ip;
      }
      if (s != 1 || ip >= base + hi || ip >= ilimit) { FLUSH(); break; }
      cur = ip - base;
      /* lanes behind the match whose pending predecessor was skipped (or inserted into the other table only): walk them again */
      {
        u64 bad = 0;
        for (u32 l = cur; l < hi; l++) {
          if (L[l].dep >= 0 && (u32)L[l].dep < cur && !((mkL >> L[l].dep) & 1)) bad |= 1ull << l;
          if (S[l].dep >= 0 && (u32)S[l].dep < cur && !((mkS >> S[l].dep) & 1)) bad |= 1ull << l;
        }
        if (bad) {
          ST.rewalkEvents++; ST.rewalkLanes += (u64)__builtin_popcountll(bad);
          FLUSH();
          for (u32 l = cur; l < hi; l++) if ((bad >> l) & 1) {
            /* satisfied dependencies below cur are inserted positions now: the walk finds them in the bitmap */
            L[l] = walk3(EL, P[l], ip, base, 1, insL); S[l] = walk3(ES, P[l], ip, base, 1, insS);
            LH &= ~(1ull << l); SH &= ~(1ull << l); deepL &= ~(1ull << l); deepS &= ~(1ull << l);
            if (L[l].deep) deepL |= 1ull << l; else if (L[l].hit) LH |= 1ull << l;
            if (S[l].deep) deepS |= 1ull << l; else if (S[l].hit) SH |= 1ull << l;
          }
        }
      }
    }
    if (done) break;
  }
  rep[0] = o1 ? o1 : saved; rep[1] = o2 ? o2 : saved;
  return be - anchor;
}

/* the driver of zo_generate_sequences with the model as dfast match finder (single-block frames of at most 64 KiB) */
static size_t model_sequences(zo_seq* out, size_t cap, const u8* src, size_t n, int level) {
  cctx c; size_t total = 0;
  if (n > 65536) return (size_t)-2;
  if (cctx_init(&c, level, n)) { cctx_free(&c); return (size_t)-1; }
  if (c.cp.strategy != 2) { cctx_free(&c); return (size_t)-2; }
  ent* EL = (ent*)malloc(sizeof(ent) * (n + 8)); ent* ES = (ent*)malloc(sizeof(ent) * (n + 8));
  u64* insL = (u64*)calloc(1024 + 2, 8); u64* insS = (u64*)calloc(1024 + 2, 8);
  if (n >= 7) {
    if (n >= 9) prepass(src, (u32)n, c.cp.hashLog, c.cp.chainLog, c.cp.minMatch, EL, ES);
    c.nbSeq = 0; c.litSize = 0;
    memcpy(c.next.rep, c.prev.rep, sizeof(c.prev.rep));
    size_t lastLL = model_dfast(&c, EL, ES, insL, insS, src, 0, (u32)n, c.next.rep);
    for (size_t i = 0; i < c.nbSeq && total < cap; i++) out[total++] = c.seqs[i];
    if (total < cap) { out[total].litLength = (u32)lastLL; out[total].matchLength = 0; out[total].offsetValue = 0; total++; }
  } else if (total < cap) { out[total].litLength = (u32)n; out[total].matchLength = 0; out[total].offsetValue = 0; total++; }
  ST.frames++;
  free(EL); free(ES); free(insL); free(insS); cctx_free(&c);
  return total;
}

/* ---- inputs */
static u32 rs;
static u32 rnd(void) { rs ^= rs << 13; rs ^= rs >> 17; rs ^= rs << 5; return rs; }
static void gen(u8* b, size_t n, u32 seed) {
  rs = seed * 2654435761u + 12345u; if (!rs) rs = 1;
  static const char* words[] = {"the ", "quick ", "brown ", "fox ", "jumps ", "over ", "lazy ", "dog ", "status=", "value=", "OK\n", "WARN\n", "0123456789", "abcabcabc", "    ", "zra "};
  size_t i = 0;
  u32 mode = rnd() % 7;
  while (i < n) {
    u32 k = rnd() % 100;
    if (mode == 5) k = 50 + k % 40;                      /* mostly incompressible */
    if (mode == 6 && k < 50) k = 70 + k % 30;            /* mostly copies and short periods: long bucket chains */
    if (k < 35) { const char* w = words[rnd() % 16]; size_t L = strlen(w); for (size_t j = 0; j < L && i < n; j++) b[i++] = (u8)w[j]; }
    else if (k < 50) { u32 L = 1 + rnd() % (mode == 1 ? 400 : 40); u8 ch = (u8)rnd(); for (u32 j = 0; j < L && i < n; j++) b[i++] = ch; }
    else if (k < 70) { u32 L = 1 + rnd() % (mode == 2 ? 600 : 24); for (u32 j = 0; j < L && i < n; j++) b[i++] = (u8)(rnd() >> (mode == 3 ? 29 : 24)); }
    else if (k < 90 && i > 8) { u32 d = 1 + rnd() % (u32)(i < 60000 ? i : 60000); if (rnd() & 1) d = 1 + rnd() % (d < 200 ? d : 200); u32 L = 3 + rnd() % (mode == 4 ? 900 : 60); for (u32 j = 0; j < L && i < n; j++) { b[i] = b[i - d]; i++; } }
    else { u32 per = 1 + rnd() % 9, L = 4 + rnd() % 80; for (u32 j = 0; j < L && i < n; j++) { b[i] = i >= per ? b[i - per] : (u8)rnd(); i++; } }
  }
}

static void print_stats(void) {
  double F = (double)(ST.frames ? ST.frames : 1);
  printf("per frame: windows %.0f (strided %.1f, with a deep lane %.0f) seqs %.0f repSeq %.0f immRep %.0f visited %.0f | rewalk events %.1f lanes %.1f | deep lanes %.0f resolved %.1f (chain steps %.1f) slowProbe %.1f | ip beyond base+128 %.1f longFwd %.1f back>0 %.0f\n",
         ST.windows / F, ST.strideWin / F, ST.winWithDeep / F, ST.seqs / F, ST.repSeq / F, ST.immRep / F, ST.visited / F, ST.rewalkEvents / F, ST.rewalkLanes / F,
         ST.deepLanes / F, ST.deepResolved / F, ST.deepSteps / F, ST.slowProbe / F, ST.tripB / F, ST.longFwd / F, ST.backAny / F);
  printf("deep walks: any equal-content predecessor beyond q3 %.1f, ended in a hit %.1f\n", ST.deepAnyEq / F, ST.deepHit / F);
  /* match lengths in the entries (forward extra capped at 14, a "may extend backward" bit), the window resolved on "no repcode hit"
     and validated by one gather, deep chains through a per-frame table of the last inserted position of heavy buckets (one read + one
     verify). Dependent round trips per frame under those rules: */
  { double w = ST.v2Win / F, r = 2.0 * (ST.v2Rep + ST.immRep) / F, sa = ST.v2Sat / F, b = ST.v2Back / F, d = 2.0 * ST.v2Deep / F, pr = ST.v2Probe / F, io = ST.v2ImmOut / F;
    printf("no-load-per-sequence parse: round trips per frame %.0f = windows %.0f + repcode events x2 %.0f + saturated lengths %.0f + backward extensions %.0f + deep x2 %.0f + probes outside the window %.0f + immediate-repcode tests behind the window %.0f   (today: 3 per window + 1 per sequence + out-of-window loads = %.0f)\n",
           w + r + sa + b + d + pr + io, w, r, sa, b, d, pr, io, 3.0 * 1354 + ST.v2Seqs / F + 920); }
}

int main(int argc, char** argv) {
  if (argc > 2 && !strcmp(argv[1], "file")) {
    FILE* f = fopen(argv[2], "rb"); if (!f) return 2;
    size_t fs = argc > 3 ? (size_t)atol(argv[3]) : 65536, nf = argc > 4 ? (size_t)atol(argv[4]) : 64; int level = argc > 5 ? atoi(argv[5]) : 3;
    u8* b = (u8*)malloc(fs + 16); zo_seq* a = (zo_seq*)malloc(sizeof(zo_seq) * (fs / 3 + 16)); zo_seq* m = (zo_seq*)malloc(sizeof(zo_seq) * (fs / 3 + 16));
    size_t frames = 0, bad = 0;
    while (frames < nf && fread(b, 1, fs, f) == fs) {
      size_t na = zo_generate_sequences(a, fs / 3 + 16, b, fs, level), nm = model_sequences(m, fs / 3 + 16, b, fs, level);
      if (na != nm || memcmp(a, m, na * sizeof(zo_seq))) { bad++; if (bad < 4) printf("MISMATCH frame %zu na %zu nm %zu\n", frames, na, nm); }
      frames++;
    }
    printf("frames %zu bad %zu level %d fs %zu\n", frames, bad, level, fs);
    print_stats();
    return bad != 0;
  }
  u32 seed0 = argc > 1 ? (u32)atol(argv[1]) : 1, nseed = argc > 2 ? (u32)atol(argv[2]) : 200;
  static const size_t sizes[] = {7, 8, 9, 15, 16, 17, 63, 64, 65, 100, 127, 128, 129, 200, 255, 256, 257, 300, 511, 512, 513, 1000, 4096, 5000, 16384, 16385, 20000, 32768, 50000, 65535, 65536};
  const size_t cap = 65536 / 3 + 64;
  u8* b = (u8*)malloc(65536 + 16); zo_seq* a = (zo_seq*)malloc(sizeof(zo_seq) * cap); zo_seq* m = (zo_seq*)malloc(sizeof(zo_seq) * cap);
  size_t cases = 0, bad = 0;
  for (u32 sd = seed0; sd < seed0 + nseed; sd++) {
    rs = sd * 977u + 1; size_t n = sizes[rnd() % (sizeof(sizes) / sizeof(sizes[0]))];
    if (rnd() % 4 == 0) n = 7 + rnd() % 65530;
    gen(b, n, sd);
    for (int level = 3; level <= 4; level++) {
      size_t na = zo_generate_sequences(a, cap, b, n, level), nm = model_sequences(m, cap, b, n, level);
      if (nm == (size_t)-2) continue;
      cases++;
      if (na != nm || memcmp(a, m, na * sizeof(zo_seq))) {
        bad++;
        size_t i = 0; while (i < na && i < nm && !memcmp(&a[i], &m[i], sizeof(zo_seq))) i++;
        if (bad < 10) printf("MISMATCH seed %u n %zu level %d: na %zu nm %zu first diff at seq %zu: oracle (%u,%u,%u) model (%u,%u,%u)\n", sd, n, level, na, nm, i,
                             a[i].litLength, a[i].matchLength, a[i].offsetValue, m[i].litLength, m[i].matchLength, m[i].offsetValue);
      }
    }
  }
  printf("cases %zu bad %zu\n", cases, bad);
  print_stats();
  return bad != 0;
}
