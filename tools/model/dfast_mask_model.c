/* BRING-UP MODEL — CPU emulation (lane arrays) of the mask-resolve dfast window parse of zra_amd/csrc/zra_encode_mf.hip.
 *
 * Not product code and not the oracle: it restates the KERNEL's algorithm (64 lanes as arrays, ballots as loops) so that the
 * exactness argument of the kernel can be fuzzed on the CPU against oracle/zo_encode.c (the restatement of zstd 1.4.9's
 * ZSTD_compressBlock_doubleFast, reference call site zra.cpp:219) before a GPU minute is spent. Build + run: tools/model/run.sh
 *
 * What the kernel does differently from round 2's window-resolve parse, and what this file checks:
 *   * windows of stride 1 sit on a fixed 64-position grid (lanes below the parse position are inactive);
 *   * every lane with a table candidate loads 72 candidate bytes ONCE and derives a byte-equality mask E (and the backward
 *     equal count); match lengths, the rep-offset tests of the following positions and the immediate-repcode test are then
 *     bit operations on 128-bit "equality streams" (EQA for offset_1, EQB for offset_2) — no memory round trip per sequence;
 *   * in-window table insertions are collected in two lane masks and stored once per window.
 * Slow paths (counted): a match that outruns its mask, a saturated backward count, streams that do not reach, positions outside
 * the window.
 */
#include "../../oracle/zo_encode.c"
#include <stdio.h>
#include <time.h>

typedef struct { u64 lo, hi; } s128;
static inline int sbit(s128 s, u32 x) { return x < 64 ? (int)((s.lo >> x) & 1) : x < 128 ? (int)((s.hi >> (x - 64)) & 1) : 0; }
static inline void sset(s128* s, u32 x) { if (x < 64) s->lo |= 1ull << x; else if (x < 128) s->hi |= 1ull << (x - 64); }
/* consecutive ones from bit x0 (stops at 128) */
static inline u32 srun(s128 s, u32 x0) { u32 r = 0; while (x0 + r < 128 && sbit(s, x0 + r)) r++; return r; }
static inline int s4(s128 s, u32 x) { return sbit(s, x) && sbit(s, x + 1) && sbit(s, x + 2) && sbit(s, x + 3); }

static struct { u64 windows, seqs, slowFwd, slowBack, slowRep, refresh, oow, probeSlow, pass2, cuts, repSeq, immRep, lanesE, strideWin, barrier, patched, laterSlow; } ST;

typedef struct {
  u32 hlog, clog, mls, ib, tagMask, idxMask;
  u32* HL; u32* HS;
  u32 shL, shS; u32* bmL; u32* bmS;   /* bucket filter, 1 bit per 2^sh buckets */
} MD;

static void md_hash(const MD* M, u64 v, u32* bL, u32* bS, u32* tL, u32* tS) {
  u32 shL = 64 - M->hlog, shS = (M->mls == 4 ? 32 : 64) - M->clog, shT = shL - (32 - M->ib);
  u64 pl = v * 0xCF1BBCDCB7A56463ULL;
  *bL = (u32)(pl >> shL); *tL = (u32)(pl >> shT) << M->ib;
  u32 p4 = (u32)v * 2654435761u;
  *tS = p4 & M->tagMask;
  if (M->mls == 5) *bS = (u32)(((v << 24) * 889523592379ULL) >> shS);
  else if (M->mls == 6) *bS = (u32)(((v << 16) * 227718039650203ULL) >> shS);
  else if (M->mls == 7) *bS = (u32)(((v << 8) * 58295818150454627ULL) >> shS);
  else *bS = p4 >> shS;
}
static void md_markL(MD* M, u32 b) { u32 g = b >> M->shL; M->bmL[g >> 5] |= 1u << (g & 31); }
static void md_markS(MD* M, u32 b) { u32 g = b >> M->shS; M->bmS[g >> 5] |= 1u << (g & 31); }
static void md_insert(MD* M, const u8* src, u32 pos, int doL, int doS) {
  u32 bl, bs, tl, ts; md_hash(M, rd64(src + pos), &bl, &bs, &tl, &ts);
  if (doL) { M->HL[bl] = (pos + 1) | tl; md_markL(M, bl); }
  if (doS) { M->HS[bs] = (pos + 1) | ts; md_markS(M, bs); }
}

/* the kernel's per-lane candidate compare: E = byte-equality mask of src[p..p+72) vs src[c..c+72), in load chunks
 * [0,8) [8,16) [16,32) [32,48) [48,64) [64,72), each loaded iff p + end <= fsize; bits at and beyond min(known, be - p) are 0 */
static void lane_E(const u8* src, u32 fsize, u32 be, u32 p, u32 c, u64* E0, u32* E1, u32* known, u32* bk, u32* bkKnown) {
  static const u32 ends[6] = {8, 16, 32, 48, 64, 72};
  u32 kn = 0;
  for (int i = 0; i < 6; i++) { if (p + ends[i] <= fsize) kn = ends[i]; else break; }
  u64 e0 = 0; u32 e1 = 0;
  u32 lim = kn < be - p ? kn : be - p;
  for (u32 j = 0; j < lim; j++) if (src[p + j] == src[c + j]) { if (j < 64) e0 |= 1ull << j; else e1 |= 1u << (j - 64); }
  *E0 = e0; *E1 = e1; *known = kn;
  /* backward: 8 bytes before both, only when c >= 8 */
  if (c >= 8) { u32 k = 0; while (k < 8 && src[p - 1 - k] == src[c - 1 - k]) k++; *bk = k; *bkKnown = 8; }
  else { *bk = 0; *bkKnown = 0; }
}

static size_t model_dfast(cctx* c, MD* M, const u8* src, u32 fsize, u32 bs, u32 be, u32 rep[3]) {
  u32 o1 = rep[0], o2 = rep[1], saved;
  u32 anchor = bs;
  const u32 ilimit = be >= 8 ? be - 8 : 0;
  u32 ip = (u32)mf_prologue(c, bs, 0, &o1, &o2, &saved);
  u64 pendL = 0, pendS = 0; u32 pendG = 0xFFFFFFFFu;        /* insertions handed to the next window: lanes of chunk pendG */
  while (ip < ilimit) {
    /* ------------------------------------------------ window build */
    ST.windows++;
    const u32 run = ip - anchor;
    u32 s, g, l0, l1;
    if (run < 256) {
      s = 1; g = ip & ~63u; l0 = ip - g; l1 = l0 + (256 - run); if (l1 > 64) l1 = 64; if (l1 > ilimit - g) l1 = ilimit - g;
    } else {
      s = (run >> 8) + 1; g = ip; l0 = 0; l1 = (256 * s - run + s - 1) / s; u32 t = (ilimit - ip + s - 1) / s; if (t < l1) l1 = t; if (l1 > 64) l1 = 64;
      ST.strideWin++;
    }
    if (pendG != g || s != 1) { if (pendL | pendS) { fprintf(stderr, "model: pending insertions for another window\n"); abort(); } }
    u32 p[64], bL[64], bS[64], tL[64], tS[64], mL[64], mS[64], cand[64], E1[64], kn[64], bk[64], bkK[64];
    u64 v8[64], E0[64];
    /* every lane of the chunk whose 8 bytes lie inside the block is hashed (pending lanes sit below l0) */
    u64 HM = 0;
    for (u32 l = 0; l < 64; l++) {
      p[l] = g + l * s;
      if (p[l] + 8 > be || (l < l0 && !(((pendL | pendS) >> l) & 1)) || l >= l1) { bL[l] = bS[l] = 0xFFFFFFFFu; continue; }
      v8[l] = rd64(src + p[l]); md_hash(M, v8[l], &bL[l], &bS[l], &tL[l], &tS[l]); HM |= 1ull << l;
    }
    u64 AM = 0; for (u32 l = l0; l < l1; l++) AM |= 1ull << l;
    /* D: lanes that share a bucket with another hashed lane of the window (the kernel: LDS slot collisions, a superset) */
    u64 D = 0;
    for (u32 j = 0; j < 64; j++) if ((HM >> j) & 1) for (u32 i = 0; i < 64; i++) if (i != j && ((HM >> i) & 1) && (bL[i] == bL[j] || bS[i] == bS[j])) { D |= 1ull << j; break; }
    if (D) ST.cuts++;
    /* pending insertions first (the latest position of a bucket wins), then the gather */
#define STORE_MASKED(mkL_, mkS_) do { u64 a_ = (mkL_), b_ = (mkS_); \
      for (u32 l_ = 0; l_ < 64; l_++) if (((a_ & D) >> l_) & 1) for (u32 k_ = l_ + 1; k_ < 64; k_++) if (((a_ >> k_) & 1) && bL[k_] == bL[l_]) { a_ &= ~(1ull << l_); break; } \
      for (u32 l_ = 0; l_ < 64; l_++) if (((b_ & D) >> l_) & 1) for (u32 k_ = l_ + 1; k_ < 64; k_++) if (((b_ >> k_) & 1) && bS[k_] == bS[l_]) { b_ &= ~(1ull << l_); break; } \
      for (int l_ = 63; l_ >= 0; l_--) { \
        if ((a_ >> l_) & 1) { M->HL[bL[l_]] = (p[l_] + 1) | tL[l_]; md_markL(M, bL[l_]); } \
        if ((b_ >> l_) & 1) { M->HS[bS[l_]] = (p[l_] + 1) | tS[l_]; md_markS(M, bS[l_]); } } } while (0)
    STORE_MASKED(pendL, pendS); pendL = pendS = 0; pendG = 0xFFFFFFFFu;
    for (u32 l = l0; l < l1; l++) {
      u32 gL = bL[l] >> M->shL, gS = bS[l] >> M->shS;
      int needL = (M->bmL[gL >> 5] >> (gL & 31)) & 1, needS = (M->bmS[gS >> 5] >> (gS & 31)) & 1;
      u32 rL = needL ? M->HL[bL[l]] : 0, rS = needS ? M->HS[bS[l]] : 0;
      mL[l] = ((rL & M->tagMask) == tL[l]) ? (rL & M->idxMask) : 0;
      mS[l] = ((rS & M->tagMask) == tS[l]) ? (rS & M->idxMask) : 0;
    }
    u64 LH = 0, SH = 0;
    for (u32 l = l0; l < l1; l++) {
      int prim = mL[l] > 1 ? 2 : mS[l] > 1 ? 1 : 0;
      if (!prim) continue;
      cand[l] = (prim == 2 ? mL[l] : mS[l]) - 1;
      lane_E(src, fsize, be, p[l], cand[l], &E0[l], &E1[l], &kn[l], &bk[l], &bkK[l]); ST.lanesE++;
      if (prim == 2) {
        if ((E0[l] & 0xFF) == 0xFF) { LH |= 1ull << l; continue; }
        if (mS[l] <= 1) continue;
        ST.pass2++;
        cand[l] = mS[l] - 1;
        lane_E(src, fsize, be, p[l], cand[l], &E0[l], &E1[l], &kn[l], &bk[l], &bkK[l]);
      }
      if ((E0[l] & 0xF) == 0xF) SH |= 1ull << l;
    }
    s128 EQA = {0, 0}, EQB = {0, 0}; u32 hiA = 0, hiB = 0;
    u64 RHa = 0, RHb = 0;
    if (s == 1) {
      for (u32 x = 0; x < 128; x++) {
        u32 pos = g + x; if (pos >= be) break;
        if (o1 > 0 && pos >= o1 && src[pos] == src[pos - o1]) sset(&EQA, x);
        if (o2 > 0 && pos >= o2 && src[pos] == src[pos - o2]) sset(&EQB, x);
      }
      hiA = hiB = 128;
      for (u32 l = 0; l < 64; l++) { if (s4(EQA, l + 1)) RHa |= 1ull << l; if (s4(EQB, l + 1)) RHb |= 1ull << l; }
    } else {
      for (u32 l = l0; l < l1; l++) if (o1 > 0 && p[l] + 1 >= o1 && rd32(src + p[l] + 1 - o1) == rd32(src + p[l] + 1)) RHa |= 1ull << l;
    }
    /* ------------------------------------------------ resolve */
    u64 insL = 0, insS = 0, Dres = 0;                            /* Dres: shared-bucket lanes whose candidates are settled */
    u32 patLane = 64, patCand = 0; int patLong = 0;              /* the one lane whose candidate came from an in-window insertion */
    u32 cur = l0;
    /* positions beyond this window that the parse inserts: lanes of the next chunk when that is the next window, else stored now */
    u32 laterPos[8]; int laterL[8], laterS[8]; int nLater = 0;
#define FLUSH() do { STORE_MASKED(insL, insS); insL = insS = 0; } while (0)
    /* settle lane d's table candidates against the insertions this window has made so far (lanes below `lim`) */
#define SETTLE(d, lim, doS_) do { \
      u64 cmL_ = 0, cmS_ = 0; const u64 bel_ = (lim) == 0 ? 0 : (~0ull >> (64 - (lim))); ST.slowBack += 0; \
      for (u32 i_ = 0; i_ < 64; i_++) { if (((insL & bel_) >> i_) & 1 && bL[i_] == bL[d]) cmL_ |= 1ull << i_; if ((doS_) && ((insS & bel_) >> i_) & 1 && bS[i_] == bS[d]) cmS_ |= 1ull << i_; } \
      ST.barrier++; \
      if (cmL_ | cmS_) { ST.patched++; \
        u32 nL_ = mL[d], nS_ = mS[d]; \
        if (cmL_) { u32 i_ = 63 - (u32)__builtin_clzll(cmL_); nL_ = tL[i_] == tL[d] ? p[i_] + 1 : 0; } \
        if (cmS_) { u32 i_ = 63 - (u32)__builtin_clzll(cmS_); nS_ = tS[i_] == tS[d] ? p[i_] + 1 : 0; } \
        /* re-evaluate the lane with the candidates the tables hold now (the kernel: one small round trip) */ \
        const int wasL_ = (LH >> d) & 1; \
        if (cmL_) { LH &= ~(1ull << d); if (nL_ > 1 && rd64(src + nL_ - 1) == v8[d]) { LH |= 1ull << d; patLane = d; patCand = nL_ - 1; patLong = 1; } } \
        if (!((LH >> d) & 1) && (doS_) && (cmS_ || (cmL_ && wasL_))) { SH &= ~(1ull << d); \
          if (nS_ > 1 && rd32(src + nS_ - 1) == (u32)v8[d]) { SH |= 1ull << d; patLane = d; patCand = nS_ - 1; patLong = 0; } } \
      } } while (0)
    for (;;) {
      const u64 live = AM & (~0ull << cur);
      const u64 Dl = D & live & ~Dres;
      const u32 h = Dl ? (u32)__builtin_ctzll(Dl) : 64;
      const u64 below = h >= 64 ? live : live & ((1ull << h) - 1);
      u64 hm = (RHa | LH | SH) & below;
      if (!hm && h < 64) {
        if ((RHa >> h) & 1) hm = 1ull << h;                     /* a repcode hit needs no table */
        else {
          const u64 upto_ = live & ((1ull << h) - 1);          /* the lanes before h are visited without a hit */
          insL |= upto_; insS |= upto_;
          SETTLE(h, h, 1); Dres |= 1ull << h; cur = h;
          continue;
        }
      }
      if (!hm) { insL |= live; insS |= live; FLUSH(); ip = g + l1 * s; break; }
      const u32 f = (u32)__builtin_ctzll(hm);
      const u64 upto = live & ((2ull << f) - 1);                 /* visited lanes cur..f */
      insL |= upto; insS |= upto;
      const u32 top = g + f * s;
      const int isRep = (RHa >> f) & 1, isLong = (LH >> f) & 1;
      u32 ml, offVal = 1;
      ip = top;
      if (isRep) {
        ip = top + 1; ST.repSeq++;
        const u32 x0 = ip - g;
        u32 r = 0; int slow = 1;
        if (s == 1) { r = srun(EQA, x0); slow = (x0 + r >= hiA) && (g + hiA < be); }
        if (slow) { ST.slowRep++; ml = (u32)count_eq(src, ip + 4, ip + 4 - o1, be) + 4; }
        else ml = r;
      } else {
        u32 fE = f; int haveE = !(patLane == f); u32 m = haveE ? 0 : patCand;
        if (!isLong) {
          if (s == 1 && f + 1 < l1) {
            if (((D & ~Dres) >> (f + 1)) & 1) { SETTLE(f + 1, f + 1, 0); Dres |= 1ull << (f + 1); }   /* the probe reads the long table only */
            insL |= 1ull << (f + 1);
            if ((LH >> (f + 1)) & 1) { fE = f + 1; ip = top + 1; haveE = !(patLane == f + 1); if (!haveE) m = patCand; }
          } else {
            /* the probed position is outside the window: commit first, then probe on one lane */
            ST.probeSlow++;
            FLUSH();
            u64 v9 = rd64(src + top + 1); u32 b3, bx, t3, tx; md_hash(M, v9, &b3, &bx, &t3, &tx);
            u32 r3 = M->HL[b3]; u32 m3 = ((r3 & M->tagMask) == t3) ? (r3 & M->idxMask) : 0;
            M->HL[b3] = (top + 2) | t3; md_markL(M, b3);
            if (m3 > 1 && rd64(src + m3 - 1) == v9) { haveE = 0; m = m3 - 1; ip = top + 1; }
          }
        }
        u32 back;
        const u32 known = (isLong || ip != top) ? 8 : 4;
        if (haveE) {
          m = cand[fE];
          u32 r = E0[fE] == ~0ull ? 64 + (u32)__builtin_ctz(~E1[fE]) : (u32)__builtin_ctzll(~E0[fE]);
          const u32 lim = kn[fE] < be - ip ? kn[fE] : be - ip;
          if (r >= lim && ip + lim < be) { ST.slowFwd++; ml = lim + (u32)count_eq(src, ip + lim, m + lim, be); }
          else ml = r;
          const u32 blim = ip - anchor < m ? ip - anchor : m;
          if (bk[fE] == bkK[fE] && blim > bkK[fE]) { ST.slowBack++; back = 0; while (back < blim && src[ip - 1 - back] == src[m - 1 - back]) back++; }
          else back = bk[fE] < blim ? bk[fE] : blim;
        } else {
          ST.slowFwd++;
          ml = known + (u32)count_eq(src, ip + known, m + known, be);
          const u32 blim = ip - anchor < m ? ip - anchor : m;
          back = 0; while (back < blim && src[ip - 1 - back] == src[m - 1 - back]) back++;
        }
        const u32 off = ip - m;
        EQB = EQA; hiB = hiA; RHb = RHa;
        EQA.lo = EQA.hi = 0; hiA = 0; RHa = 0;
        if (s == 1 && haveE) {
          const u32 xE = ip - g;
          const u32 lim = kn[fE] < be - ip ? kn[fE] : be - ip;
          for (u32 j = 0; j < lim; j++) { int b = j < 64 ? (int)((E0[fE] >> j) & 1) : (int)((E1[fE] >> (j - 64)) & 1); if (b) sset(&EQA, xE + j); }
          hiA = (ip + kn[fE] >= be) ? 128 : (xE + kn[fE] < 128 ? xE + kn[fE] : 128);
          for (u32 l = 0; l < 64; l++) if (s4(EQA, l + 1)) RHa |= 1ull << l;
        }
        ip -= back; ml += back;
        o2 = o1; o1 = off; offVal = off + 3;
      }
      emit(c, src, anchor, ip - anchor, ml, offVal); ST.seqs++;
      ip += ml; anchor = ip;
      if (ip > ilimit) { FLUSH(); break; }
      /* complementary insertions (top+2 both tables, ip-2 long, ip-1 short) and the immediate repcode test */
      const u32 x = ip - g;
      const int in2 = s == 1 && f + 2 < l1, inE2 = s == 1 && x - 2 < l1, inE1 = s == 1 && x - 1 < l1;
      if (in2) { insL |= 1ull << (f + 2); insS |= 1ull << (f + 2); } else { laterPos[nLater] = top + 2; laterL[nLater] = 1; laterS[nLater] = 1; nLater++; }
      if (inE2) insL |= 1ull << (x - 2); else { laterPos[nLater] = ip - 2; laterL[nLater] = 1; laterS[nLater] = 0; nLater++; }
      if (inE1) insS |= 1ull << (x - 1); else { laterPos[nLater] = ip - 1; laterL[nLater] = 0; laterS[nLater] = 1; nLater++; }
      int here_eq_there;
      if (s == 1 && (o2 == 0 || x + 4 <= hiB)) here_eq_there = o2 > 0 && s4(EQB, x);
      else { ST.oow++; here_eq_there = o2 > 0 && rd32(src + ip) == rd32(src + ip - o2); }
      while (here_eq_there) {
        ST.immRep++;
        const u32 xx = ip - g;
        u32 rl = 0; int slow = 1;
        if (s == 1 && xx + 4 <= hiB) { u32 r = srun(EQB, xx); slow = (xx + r >= hiB) && (g + hiB < be); rl = r; }
        if (slow) { ST.slowRep++; rl = (u32)count_eq(src, ip + 4, ip + 4 - o2, be) + 4; }
        { u32 t = o2; o2 = o1; o1 = t; s128 ts = EQA; EQA = EQB; EQB = ts; u32 th = hiA; hiA = hiB; hiB = th; u64 tr = RHa; RHa = RHb; RHb = tr; }
        if (s == 1 && xx < l1) { insL |= 1ull << xx; insS |= 1ull << xx; }
        else { laterPos[nLater] = ip; laterL[nLater] = 1; laterS[nLater] = 1; nLater++; }
        emit(c, src, anchor, 0, rl, 1); ST.seqs++;
        ip += rl; anchor = ip;
        if (!(ip <= ilimit && o2 > 0)) break;
        const u32 xn = ip - g;
        if (s == 1 && xn + 4 <= hiB) here_eq_there = s4(EQB, xn);
        else { ST.oow++; here_eq_there = rd32(src + ip) == rd32(src + ip - o2); }
        if (nLater > 4 && here_eq_there) {                     /* keep the list short: store what has piled up */
          FLUSH(); for (int k = 0; k < nLater; k++) md_insert(M, src, laterPos[k], laterL[k], laterS[k]); nLater = 0;
        }
      }
      if (s != 1 || ip >= g + l1 || ip >= ilimit) { FLUSH(); break; }
      cur = ip - g;
      if (nLater) { fprintf(stderr, "model: later insertions although the window goes on\n"); abort(); }
      if (hiA < l1 + 4 && o1 > 0) {
        ST.refresh++;
        EQA.lo = EQA.hi = 0; RHa = 0;
        for (u32 xq = 0; xq < 128; xq++) { u32 pos = g + xq; if (pos >= be) break; if (pos >= o1 && src[pos] == src[pos - o1]) sset(&EQA, xq); }
        hiA = 128;
        for (u32 l = 0; l < 64; l++) if (s4(EQA, l + 1)) RHa |= 1ull << l;
      }
    }
    /* hand the insertions behind the window to the next one, or store them now */
    if (nLater) {
      const int nextS1 = ip < ilimit && ip - anchor < 256;
      const u32 gN = ip & ~63u;
      for (int k = 0; k < nLater; k++) {
        const u32 q = laterPos[k];
        if (nextS1 && q >= gN && q < gN + 64 && q + 8 <= be) { if (laterL[k]) pendL |= 1ull << (q - gN); if (laterS[k]) pendS |= 1ull << (q - gN); pendG = gN; }
        else { ST.laterSlow++; md_insert(M, src, q, laterL[k], laterS[k]); }
      }
    }
  }
  if (pendL | pendS) { fprintf(stderr, "model: pending insertions at the block end\n"); abort(); }
  rep[0] = o1 ? o1 : saved; rep[1] = o2 ? o2 : saved;
  return be - anchor;
}

/* the driver of zo_generate_sequences with the model as dfast match finder */
static size_t model_sequences(zo_seq* out, size_t cap, const u8* src, size_t n, int level) {
  cctx c; size_t total = 0;
  if (cctx_init(&c, level, n)) { cctx_free(&c); return (size_t)-1; }
  if (c.cp.strategy != 2) { cctx_free(&c); return (size_t)-2; }
  size_t blockSize = (size_t)1 << c.cp.windowLog; if (blockSize > (128u << 10)) blockSize = 128u << 10;
  u8* scratch = (u8*)malloc(zo_compress_bound(blockSize) + 64);
  MD M; memset(&M, 0, sizeof(M));
  M.hlog = c.cp.hashLog; M.clog = c.cp.chainLog; M.mls = c.cp.minMatch; M.HL = c.hashTable; M.HS = c.chainTable;
  M.ib = 32 - (u32)__builtin_clz((u32)n - 1); M.tagMask = ~((1u << M.ib) - 1u); M.idxMask = ~M.tagMask;
  M.shL = 1; M.shS = 2;
  M.bmL = (u32*)calloc((((size_t)1 << M.hlog) >> M.shL) / 32 + 1, 4); M.bmS = (u32*)calloc((((size_t)1 << M.clog) >> M.shS) / 32 + 1, 4);
  size_t pos = 0;
  while (pos < n) {
    size_t L = n - pos < blockSize ? n - pos : blockSize;
    if (L >= 7) {
      c.nbSeq = 0; c.litSize = 0;
      memcpy(c.next.rep, c.prev.rep, sizeof(c.prev.rep));
      if (pos > 0) { memset(M.bmL, 0xFF, ((((size_t)1 << M.hlog) >> M.shL) / 32 + 1) * 4); memset(M.bmS, 0xFF, ((((size_t)1 << M.clog) >> M.shS) / 32 + 1) * 4); }
      size_t lastLL = model_dfast(&c, &M, src, (u32)n, (u32)pos, (u32)(pos + L), c.next.rep);
      for (size_t i = 0; i < c.nbSeq && total < cap; i++) out[total++] = c.seqs[i];
      if (total < cap) { out[total].litLength = (u32)lastLL; out[total].matchLength = 0; out[total].offsetValue = 0; total++; }
      memcpy(c.lits + c.litSize, src + pos + L - lastLL, lastLL); c.litSize += lastLL;
      size_t cSize = entropy_compress(&c, scratch, zo_compress_bound(blockSize) + 64);
      if (ZO_ISERR(cSize) || cSize >= L - min_gain(L, c.cp.strategy)) cSize = 0;
      if (cSize > 1) c.prev = c.next;
    } else if (total < cap) { out[total].litLength = (u32)L; out[total].matchLength = 0; out[total].offsetValue = 0; total++; }
    pos += L;
  }
  free(scratch); free(M.bmL); free(M.bmS); cctx_free(&c);
  return total;
}

/* ---- inputs */
static u32 rs;
static u32 rnd(void) { rs ^= rs << 13; rs ^= rs >> 17; rs ^= rs << 5; return rs; }
static void gen(u8* b, size_t n, u32 seed) {
  rs = seed * 2654435761u + 12345u; if (!rs) rs = 1;
  static const char* words[] = {"the ", "quick ", "brown ", "fox ", "jumps ", "over ", "lazy ", "dog ", "status=", "value=", "OK\n", "WARN\n", "0123456789", "abcabcabc", "    ", "zra "};
  size_t i = 0;
  u32 mode = rnd() % 6;
  while (i < n) {
    u32 k = rnd() % 100;
    if (mode == 5) k = 50 + k % 40;                      /* mostly incompressible */
    if (k < 35) { const char* w = words[rnd() % 16]; size_t L = strlen(w); for (size_t j = 0; j < L && i < n; j++) b[i++] = (u8)w[j]; }
    else if (k < 50) { u32 L = 1 + rnd() % (mode == 1 ? 400 : 40); u8 ch = (u8)rnd(); for (u32 j = 0; j < L && i < n; j++) b[i++] = ch; }
    else if (k < 70) { u32 L = 1 + rnd() % (mode == 2 ? 600 : 24); for (u32 j = 0; j < L && i < n; j++) b[i++] = (u8)(rnd() >> (mode == 3 ? 29 : 24)); }
    else if (k < 90 && i > 8) { u32 d = 1 + rnd() % (u32)(i < 60000 ? i : 60000); if (rnd() & 1) d = 1 + rnd() % (d < 200 ? d : 200); u32 L = 3 + rnd() % (mode == 4 ? 900 : 60); for (u32 j = 0; j < L && i < n; j++) { b[i] = b[i - d]; i++; } }
    else { u32 per = 1 + rnd() % 9, L = 4 + rnd() % 80; for (u32 j = 0; j < L && i < n; j++) { b[i] = i >= per ? b[i - per] : (u8)rnd(); i++; } }
  }
}

int main(int argc, char** argv) {
  if (argc > 2 && !strcmp(argv[1], "file")) {
    /* statistics on a real corpus: frames of `fs` bytes of the file */
    FILE* f = fopen(argv[2], "rb"); if (!f) return 2;
    size_t fs = argc > 3 ? (size_t)atol(argv[3]) : 65536, nf = argc > 4 ? (size_t)atol(argv[4]) : 64; int level = argc > 5 ? atoi(argv[5]) : 3;
    u8* b = (u8*)malloc(fs); zo_seq* a = (zo_seq*)malloc(sizeof(zo_seq) * (fs / 3 + 16)); zo_seq* m = (zo_seq*)malloc(sizeof(zo_seq) * (fs / 3 + 16));
    size_t frames = 0, bad = 0;
    while (frames < nf && fread(b, 1, fs, f) == fs) {
      size_t na = zo_generate_sequences(a, fs / 3 + 16, b, fs, level), nm = model_sequences(m, fs / 3 + 16, b, fs, level);
      if (na != nm || memcmp(a, m, na * sizeof(zo_seq))) bad++;
      frames++;
    }
    printf("frames %zu bad %zu | per frame: windows %.0f seqs %.0f repSeq %.0f immRep %.0f lanesE %.0f cuts %.0f | slowFwd %.1f slowBack %.1f slowRep %.1f refresh %.1f oow %.1f probeSlow %.1f pass2 %.2f strideWin %.1f barrier %.1f patched %.1f laterSlow %.1f\n",
           frames, bad, (double)ST.windows / frames, (double)ST.seqs / frames, (double)ST.repSeq / frames, (double)ST.immRep / frames, (double)ST.lanesE / frames, (double)ST.cuts / frames,
           (double)ST.slowFwd / frames, (double)ST.slowBack / frames, (double)ST.slowRep / frames, (double)ST.refresh / frames, (double)ST.oow / frames, (double)ST.probeSlow / frames, (double)ST.pass2 / frames, (double)ST.strideWin / frames, (double)ST.barrier / frames, (double)ST.patched / frames, (double)ST.laterSlow / frames);
    return bad != 0;
  }
  u32 seed0 = argc > 1 ? (u32)atol(argv[1]) : 1, nseed = argc > 2 ? (u32)atol(argv[2]) : 200;
  static const size_t sizes[] = {7, 8, 9, 15, 16, 17, 63, 64, 65, 100, 127, 128, 129, 200, 255, 256, 257, 300, 511, 512, 513, 1000, 4096, 5000, 16384, 16385, 20000, 65535, 65536, 65537, 100000, 131072, 131073, 200000, 262144, 300000};
  const size_t cap = 300000 / 3 + 64;
  u8* b = (u8*)malloc(300000 + 8); zo_seq* a = (zo_seq*)malloc(sizeof(zo_seq) * cap); zo_seq* m = (zo_seq*)malloc(sizeof(zo_seq) * cap);
  size_t cases = 0, bad = 0;
  for (u32 sd = seed0; sd < seed0 + nseed; sd++) {
    rs = sd * 977u + 1; size_t n = sizes[rnd() % (sizeof(sizes) / sizeof(sizes[0]))];
    if (rnd() % 4 == 0) n = 7 + rnd() % 70000;
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
  printf("cases %zu bad %zu | windows %llu seqs %llu slowFwd %llu slowBack %llu slowRep %llu refresh %llu oow %llu probeSlow %llu pass2 %llu cuts %llu\n", cases, bad,
         (unsigned long long)ST.windows, (unsigned long long)ST.seqs, (unsigned long long)ST.slowFwd, (unsigned long long)ST.slowBack, (unsigned long long)ST.slowRep,
         (unsigned long long)ST.refresh, (unsigned long long)ST.oow, (unsigned long long)ST.probeSlow, (unsigned long long)ST.pass2, (unsigned long long)ST.cuts);
  return bad != 0;
}
