/* ORACLE — TEST INFRASTRUCTURE ONLY (see zo_common.h).
 *
 * zstd frame decoder restated from SURVEY.md Appendix A.1-A.3 (RFC 8878 format). Stands in for
 * ZSTD_decompressDCtx at the reference call sites zra.cpp:249,280,289,293,397,406,410,435.
 * Sequential, one byte at a time where that is clearest: this is a checker, not a fast path.
 */
#include "zo_internal.h"
#include <stdlib.h>

const u32 zo_ll_base[36] = {0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,18,20,22,24,28,32,40,48,64,128,256,512,1024,2048,4096,8192,16384,32768,65536};
const u8 zo_ll_bits[36] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16};
const u32 zo_ml_base[53] = {3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,
                            35,37,39,41,43,47,51,59,67,83,99,131,259,515,1027,2051,4099,8195,16387,32771,65539};
const u8 zo_ml_bits[53] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,
                           1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16};
const s16 zo_ll_defnorm[36] = {4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1};
const s16 zo_ml_defnorm[53] = {1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,
                               1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1};
const s16 zo_of_defnorm[29] = {1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1};

typedef struct {
  zo_huf_dtable huf; int hufValid, hufX2;
  zo_fse_dtable ll, of, ml; int llValid, ofValid, mlValid;
  int llRle, ofRle, mlRle; /* RLE tables are stored as 1-cell dtables with tableLog 0 */
  u32 rep[3];
} dctx;

static void rle_dtable(zo_fse_dtable* dt, unsigned sym) { dt->tableLog = 0; dt->sym[0] = (u8)sym; dt->nbBits[0] = 0; dt->base[0] = 0; }

/* HUF_selectDecoder of zstd 1.4.9 (huf_decompress.c algoTime[][]): 0 = single-symbol tables (X1), 1 = double-symbol tables (X2).
   Both regenerate the same bytes from a valid stream; they differ in which damaged streams they still accept (see below), and
   statuses are results. Pinned against the library's exported HUF_selectDecoder in tests/test_oracle.py. */
static int huf_select_decoder(size_t dstSize, size_t cSrcSize) {
  static const u16 t0[16][2] = {{0,0},{0,0},{38,130},{448,128},{556,128},{714,128},{883,128},{897,128},{926,128},{947,128},{1107,128},{1177,128},{1242,128},{1349,128},{1455,128},{722,128}};
  static const u16 t1[16][2] = {{1,1},{1,1},{1313,74},{1353,74},{1353,74},{1418,74},{1437,74},{1515,75},{1613,75},{1729,77},{2083,81},{2379,87},{2415,93},{2644,106},{2422,124},{1891,145}};
  const u32 Q = cSrcSize >= dstSize ? 15 : (u32)(cSrcSize * 16 / dstSize), D256 = (u32)(dstSize >> 8);
  const u32 d0 = t0[Q][0] + t0[Q][1] * D256;
  u32 d1 = t1[Q][0] + t1[Q][1] * D256;
  d1 += d1 >> 3;
  return d1 < d0;
}
int zo_huf_select_decoder(size_t dstSize, size_t cSrcSize) { return huf_select_decoder(dstSize, cSrcSize); }

/* peek `mb` bits ending at bit position `pos` of a backward stream (zeros below bit 0) */
static u32 huf_peek(const u8* src, long pos, unsigned mb) {
  u32 v = 0;
  for (unsigned k = 0; k < mb; k++) {
    long bit = pos - (long)mb + (long)k;
    if (bit >= 0) v |= (u32)((src[bit >> 3] >> (bit & 7)) & 1) << k;
  }
  return v;
}

/* decode `regen` symbols from one backward Huffman stream; 0 ok.
   x2 = 0: HUF_decodeStreamX1 — the stream must end exactly on bit 0.
   x2 = 1: HUF_decodeStreamX2 — a 12-bit lookup returns one symbol or a PAIR (when the two codes fit in 12 bits together). Pairs cover
   two output positions; if the walk ends one position short, HUF_decodeLastSymbolX2 takes the first symbol of the entry under the
   cursor and, for a pair entry, skips the bits of BOTH codes clamped to the end of the stream (and skips nothing when no bit is left).
   A damaged stream can therefore pass here that the single-symbol decoder rejects. */
static int huf_decode_stream(u8* out, size_t regen, const u8* src, size_t n, const zo_huf_dtable* dt, int x2) {
  zo_bitr br;
  if (zo_bitr_init(&br, src, n)) return -1;
  const unsigned mb = dt->maxBits;
  if (!x2) {
    for (size_t i = 0; i < regen; i++) {
      const u32 v = huf_peek(br.src, br.pos, mb);
      out[i] = dt->sym[v];
      br.pos -= dt->nbBits[v];
    }
    return br.pos == 0 ? 0 : -1;
  }
  size_t i = 0;
  while (i + 2 <= regen) {
    if (br.pos <= 0) return -1;                         /* over-read: bitsConsumed can only grow, the end check fails */
    const u32 v1 = huf_peek(br.src, br.pos, mb); const unsigned a = dt->nbBits[v1];
    const u32 v2 = huf_peek(br.src, br.pos - (long)a, mb); const unsigned b = dt->nbBits[v2];
    out[i] = dt->sym[v1];
    if (a + b <= 12) { out[i + 1] = dt->sym[v2]; br.pos -= (long)(a + b); i += 2; }
    else { br.pos -= (long)a; i += 1; }
  }
  if (i < regen) {
    if (br.pos < 0) return -1;
    u32 v1, v2; unsigned a, b;
    if (br.pos > 0) {
      v1 = huf_peek(br.src, br.pos, mb); a = dt->nbBits[v1];
      v2 = huf_peek(br.src, br.pos - (long)a, mb); b = dt->nbBits[v2];
    } else {
      /* nothing left: BIT_lookBitsFast shifts the container by (64 & 63) = 0 and returns its TOP bits again (the container holds
         the first 8 bytes of the stream, or all of it zero-extended) */
      u64 c = 0; for (size_t k = 0; k < 8 && k < n; k++) c |= (u64)src[k] << (8 * k);
      const u32 top12 = (u32)(c >> 52);
      v1 = top12 >> (12 - mb); a = dt->nbBits[v1];
      v2 = ((top12 << a) & 0xFFF) >> (12 - mb); b = dt->nbBits[v2];
    }
    out[i] = dt->sym[v1];
    if (a + b <= 12) { if (br.pos > 0) { br.pos -= (long)(a + b); if (br.pos < 0) br.pos = 0; } }
    else br.pos -= (long)a;
  }
  return br.pos == 0 ? 0 : -1;
}

/* returns bytes of the literals section consumed, or error; literals land in lit[0..*litSize) */
static size_t decode_literals(dctx* d, u8* lit, size_t* litSize, const u8* src, size_t n, size_t blockMax) {
  if (n < 3) return ZO_ERR(ZO_E_CORRUPTION);              /* ZSTD_decodeLiteralsBlock: srcSize < MIN_CBLOCK_SIZE */
  unsigned type = src[0] & 3, sf = (src[0] >> 2) & 3;
  if (type < 2) {
    size_t size, lh;
    if (sf == 0 || sf == 2) { size = src[0] >> 3; lh = 1; }
    else if (sf == 1) { if (n < 2) return ZO_ERR(ZO_E_CORRUPTION); size = rd16(src) >> 4; lh = 2; }
    else { if (n < 3) return ZO_ERR(ZO_E_CORRUPTION); size = rd24(src) >> 4; lh = 3; }
    if (size > blockMax) return ZO_ERR(ZO_E_CORRUPTION);
    if (type == 0) {
      if (lh + size > n) return ZO_ERR(ZO_E_CORRUPTION);
      memcpy(lit, src + lh, size); *litSize = size; return lh + size;
    }
    if (lh + 1 > n) return ZO_ERR(ZO_E_CORRUPTION);
    memset(lit, src[lh], size); *litSize = size; return lh + 1;
  }
  size_t regen, comp, lh; int streams;
  if (type == 3 && !d->hufValid) return ZO_ERR(30);       /* set_repeat without a previous table: dictionary_corrupted, before any size check */
  if (n < 5) return ZO_ERR(ZO_E_CORRUPTION);              /* "here we need up to 5 for case 3", whatever the size format */
  if (sf < 2) { u32 v = rd24(src); regen = (v >> 4) & 0x3FF; comp = v >> 14; lh = 3; streams = sf == 0 ? 1 : 4; }
  else if (sf == 2) { u32 v = rd32(src); regen = (v >> 4) & 0x3FFF; comp = v >> 18; lh = 4; streams = 4; }
  else { u64 v = (u64)rd32(src) | ((u64)src[4] << 32); regen = (v >> 4) & 0x3FFFF; comp = (size_t)(v >> 22); lh = 5; streams = 4; }
  if (regen > blockMax) return ZO_ERR(ZO_E_CORRUPTION);
  if (lh + comp > n) return ZO_ERR(ZO_E_CORRUPTION);
  const u8* p = src + lh; size_t rem = comp;
  if (type == 2) {
    u8 weights[256]; unsigned nSym, maxBits;
    size_t h = zo_huf_read_weights(weights, &nSym, &maxBits, p, rem);
    if (!h) return ZO_ERR(ZO_E_CORRUPTION);
    if (zo_huf_build_dtable(&d->huf, weights, nSym, maxBits)) return ZO_ERR(ZO_E_CORRUPTION);
    d->hufValid = 1;
    if (h >= rem) return ZO_ERR(ZO_E_CORRUPTION);        /* "hSize >= cSrcSize" (srcSize_wrong inside HUF, corruption_detected outside) */
    /* table kind (zstd_decompress_block.c, ZSTD_decodeLiteralsBlock): one stream with a new tree -> X1; four streams -> HUF_selectDecoder
       on (regenerated size, compressed size incl. the tree); treeless blocks reuse whatever the kept table is */
    d->hufX2 = streams == 4 ? huf_select_decoder(regen, comp) : 0;
    p += h; rem -= h;
  }
  if (streams == 1) {
    if (huf_decode_stream(lit, regen, p, rem, &d->huf, d->hufX2)) return ZO_ERR(ZO_E_CORRUPTION);
  } else {
    if (regen == 0 && type == 2) return ZO_ERR(ZO_E_CORRUPTION);   /* HUF_decompress4X_hufOnly_wksp: dstSize == 0 */
    if (rem < 10) return ZO_ERR(ZO_E_CORRUPTION);
    size_t s1 = rd16(p), s2 = rd16(p + 2), s3 = rd16(p + 4);
    if (6 + s1 + s2 + s3 > rem) return ZO_ERR(ZO_E_CORRUPTION);
    size_t s4 = rem - 6 - s1 - s2 - s3;
    size_t seg = (regen + 3) / 4;
    if (seg * 3 > regen) return ZO_ERR(ZO_E_CORRUPTION);
    const u8* q = p + 6;
    if (huf_decode_stream(lit, seg, q, s1, &d->huf, d->hufX2)) return ZO_ERR(ZO_E_CORRUPTION);
    if (huf_decode_stream(lit + seg, seg, q + s1, s2, &d->huf, d->hufX2)) return ZO_ERR(ZO_E_CORRUPTION);
    if (huf_decode_stream(lit + 2 * seg, seg, q + s1 + s2, s3, &d->huf, d->hufX2)) return ZO_ERR(ZO_E_CORRUPTION);
    if (huf_decode_stream(lit + 3 * seg, regen - 3 * seg, q + s1 + s2 + s3, s4, &d->huf, d->hufX2)) return ZO_ERR(ZO_E_CORRUPTION);
  }
  *litSize = regen;
  return lh + comp;
}


/* libzstd's BIT_DStream_t (bitstream.h of 1.4.9), restated field for field: a 64-bit container refilled from the END of the stream
   towards its start. Kept exact because an over-read stream is NOT an error for the sequence decoder of 1.4.9: the wrapped container
   bits it then reads decide what the frame finally returns. */
typedef struct { u64 c; unsigned bc; const u8* ptr; const u8* start; const u8* limit; } zds;
enum { ZDS_UNFINISHED = 0, ZDS_ENDOFBUFFER = 1, ZDS_COMPLETED = 2, ZDS_OVERFLOW = 3 };
static int zds_init(zds* b, const u8* src, size_t n) {
  if (n < 1) return -1;
  b->start = src; b->limit = src + 8;
  if (n >= 8) { b->ptr = src + n - 8; b->c = rd64(b->ptr); }
  else { b->ptr = src; b->c = 0; for (size_t i = 0; i < n; i++) b->c |= (u64)src[i] << (8 * i); }
  const u8 last = src[n - 1];
  if (last == 0) return -1;
  b->bc = 8 - hb32(last);
  if (n < 8) b->bc += (unsigned)(8 - n) * 8;
  return 0;
}
static u32 zds_look(const zds* b, unsigned nb) {           /* BIT_lookBits: BIT_getMiddleBits(container, 64 - bc - nb, nb) */
  const unsigned start = 64u - b->bc - nb;
  return (u32)((b->c >> (start & 63)) & ((nb >= 32) ? 0xFFFFFFFFu : ((1u << nb) - 1)));
}
static u32 zds_read(zds* b, unsigned nb) { const u32 v = zds_look(b, nb); b->bc += nb; return v; }
static u32 zds_read_fast(zds* b, unsigned nb) {            /* BIT_readBitsFast: nb >= 1 */
  const u32 v = (u32)((b->c << (b->bc & 63)) >> ((64 - nb) & 63));
  b->bc += nb; return v;
}
static int zds_reload(zds* b) {
  if (b->bc > 64) return ZDS_OVERFLOW;
  if (b->ptr >= b->limit) { b->ptr -= b->bc >> 3; b->bc &= 7; b->c = rd64(b->ptr); return ZDS_UNFINISHED; }
  if (b->ptr == b->start) return b->bc < 64 ? ZDS_ENDOFBUFFER : ZDS_COMPLETED;
  {
    unsigned nbBytes = b->bc >> 3; int r = ZDS_UNFINISHED;
    if (b->ptr - nbBytes < b->start) { nbBytes = (unsigned)(b->ptr - b->start); r = ZDS_ENDOFBUFFER; }
    b->ptr -= nbBytes; b->bc -= nbBytes * 8; b->c = rd64(b->ptr);
    return r;
  }
}

/* one decoded sequence, and the two halves of the loop body (ZSTD_decodeSequence / ZSTD_execSequence[End]) */
typedef struct { u32 ll, ml, off; } seq3;
typedef struct { dctx* d; zds* br; u32 sLL, sOF, sML; } seqstate;
typedef struct { u8* out; size_t outCap; const u8* frameOut; const u8* lit; size_t litSize, produced, litPos; } execstate;
static seq3 decode_sequence(seqstate* q) {
  dctx* d = q->d; zds* br = q->br;
  const unsigned llc = d->ll.sym[q->sLL], ofc = d->of.sym[q->sOF], mlc = d->ml.sym[q->sML];
  const unsigned llBits = zo_ll_bits[llc], mlBits = zo_ml_bits[mlc], ofBits = ofc;
  seq3 r; r.ll = zo_ll_base[llc]; r.ml = zo_ml_base[mlc];
  if (ofBits > 1) {
    r.off = ((1u << ofc) - 3) + zds_read_fast(br, ofBits);
    d->rep[2] = d->rep[1]; d->rep[1] = d->rep[0]; d->rep[0] = r.off;
  } else {
    const u32 ll0 = (r.ll == 0);                        /* the BASE value: codes 0 with no extra bits */
    if (ofBits == 0) {
      if (!ll0) r.off = d->rep[0];
      else { r.off = d->rep[1]; d->rep[1] = d->rep[0]; d->rep[0] = r.off; }
    } else {
      const u32 idx = 1 + ll0 + zds_read_fast(br, 1);
      u32 temp = idx == 3 ? d->rep[0] - 1 : d->rep[idx];
      temp += !temp;                                    /* "0 is not valid; input is corrupted; force offset to 1" */
      if (idx != 1) d->rep[2] = d->rep[1];
      d->rep[1] = d->rep[0]; d->rep[0] = r.off = temp;
    }
  }
  if (mlBits > 0) r.ml += zds_read_fast(br, mlBits);
  if (llBits + mlBits + ofBits >= 57 - (9 + 9 + 8)) zds_reload(br);
  if (llBits > 0) r.ll += zds_read_fast(br, llBits);
  q->sLL = d->ll.base[q->sLL] + zds_read(br, d->ll.nbBits[q->sLL]);
  q->sML = d->ml.base[q->sML] + zds_read(br, d->ml.nbBits[q->sML]);
  q->sOF = d->of.base[q->sOF] + zds_read(br, d->of.nbBits[q->sOF]);
  return r;
}
/* order of the checks: destination room, literal buffer, (literals are copied), offset */
static int exec_sequence(execstate* x, seq3 s) {
  if ((size_t)s.ll + s.ml > x->outCap - x->produced) return ZO_E_DSTSIZE_TOOSMALL;
  if (s.ll > x->litSize - x->litPos) return ZO_E_CORRUPTION;
  memcpy(x->out + x->produced, x->lit + x->litPos, s.ll); x->litPos += s.ll;
  if (s.off > (size_t)(x->out + x->produced + s.ll - x->frameOut)) return ZO_E_CORRUPTION;
  x->produced += s.ll;
  for (u32 k = 0; k < s.ml; k++) x->out[x->produced + k] = x->out[x->produced + k - s.off];
  x->produced += s.ml;
  return 0;
}

/* one of LL/OF/ML table descriptions; returns bytes consumed or error */
static size_t decode_seq_table(zo_fse_dtable* dt, int* valid, unsigned mode, const u8* src, size_t n,
                               unsigned maxSym, unsigned maxAL, const s16* defNorm, unsigned defMax, unsigned defLog) {
  if (mode == 0) { zo_fse_build_dtable(dt, defNorm, defMax, defLog); *valid = 1; return 0; }
  if (mode == 1) {
    if (n < 1 || src[0] > maxSym) return ZO_ERR(ZO_E_CORRUPTION);
    rle_dtable(dt, src[0]); *valid = 1; return 1;
  }
  if (mode == 2) {
    s16 norm[64]; unsigned ms = maxSym, t;
    memset(norm, 0, sizeof(norm));
    size_t h = zo_fse_read_ncount(norm, &ms, &t, src, n, maxAL);
    if (!h) return ZO_ERR(ZO_E_CORRUPTION);
    if (zo_fse_build_dtable(dt, norm, ms, t)) return ZO_ERR(ZO_E_CORRUPTION);
    *valid = 1; return h;
  }
  if (!*valid) return ZO_ERR(ZO_E_CORRUPTION);
  return 0;
}

/* decompress one compressed block into out (frameOut = start of this frame's output, for offset validation) */
static size_t decode_block(dctx* d, u8* out, size_t outCap, const u8* frameOut, const u8* src, size_t n, size_t blockMax, u64 windowSize) {
  u8* lit = (u8*)malloc(blockMax + 32);
  size_t litSize = 0, produced = 0;
  if (!lit) return ZO_ERR(ZO_E_GENERIC);
  size_t r = decode_literals(d, lit, &litSize, src, n, blockMax);
  if (ZO_ISERR(r)) { free(lit); return r; }
  const u8* p = src + r; size_t rem = n - r;
#define FAIL(code) do { free(lit); return ZO_ERR(code); } while (0)
  if (rem < 1) FAIL(ZO_E_SRCSIZE_WRONG);
  size_t nbSeq = p[0];
  /* ZSTD_decodeSeqHeaders leaves early only for a FIRST BYTE of zero; a count of zero in the two-byte form (0x80 0x00 — never written by an
     encoder, found by the round-6 soak on damaged archives, seeds 145238 / 146031) goes on through the table descriptions — whose errors
     count and whose tables stay for later blocks' repeat modes — and then decodes no sequence and never opens the bitstream */
  const int seqTables = p[0] != 0;
  if (nbSeq == 0) { p++; rem--; if (rem) FAIL(ZO_E_SRCSIZE_WRONG); }   /* ZSTD_decodeSeqHeaders: "srcSize != 1" */
  else if (nbSeq < 128) { p++; rem--; }
  else if (nbSeq < 255) { if (rem < 2) FAIL(ZO_E_SRCSIZE_WRONG); nbSeq = ((nbSeq - 128) << 8) + p[1]; p += 2; rem -= 2; }
  else { if (rem < 3) FAIL(ZO_E_SRCSIZE_WRONG); nbSeq = (size_t)p[1] + ((size_t)p[2] << 8) + 0x7F00; p += 3; rem -= 3; }
  size_t litPos = 0;
  if (seqTables) {
    if (rem < 1) FAIL(ZO_E_SRCSIZE_WRONG);
    unsigned modes = p[0]; p++; rem--;      /* the two reserved bits are not looked at by libzstd 1.4.9 */
    size_t h;
    h = decode_seq_table(&d->ll, &d->llValid, modes >> 6, p, rem, 35, 9, zo_ll_defnorm, 35, 6); if (ZO_ISERR(h)) FAIL(ZO_E_CORRUPTION); p += h; rem -= h;
    h = decode_seq_table(&d->of, &d->ofValid, (modes >> 4) & 3, p, rem, 31, 8, zo_of_defnorm, 28, 5); if (ZO_ISERR(h)) FAIL(ZO_E_CORRUPTION); p += h; rem -= h;
    h = decode_seq_table(&d->ml, &d->mlValid, (modes >> 2) & 3, p, rem, 52, 9, zo_ml_defnorm, 52, 6); if (ZO_ISERR(h)) FAIL(ZO_E_CORRUPTION); p += h; rem -= h;
    /* The sequence loops follow libzstd 1.4.9's control flow exactly, because statuses are results:
       - the bit reader is its BIT_DStream_t (zds), so an over-read stream yields the same wrapped container bits;
       - FSE states are updated after EVERY sequence, the last one included (the bits read there are never used);
       - short loop (ZSTD_decompressSequences_body): decode, execute, reload; the first execution error ends the block; over-reading
         is never looked at inside the loop; after it "stream consumed" = BIT_reloadDStream() >= BIT_DStream_completed (over-read passes);
       - long loop (ZSTD_decompressSequencesLong, chosen by ZSTD_decompressBlock_internal for frames declaring a window above 16 MiB
         whose offset table holds enough long codes): sequences are decoded four ahead of their execution, the loop stops with
         corruption_detected as soon as the stream is over-read, and there is no "stream consumed" check at all. */
    if (nbSeq) {
    zds br;
    if (zds_init(&br, p, rem)) FAIL(ZO_E_CORRUPTION);
    seqstate q; q.d = d; q.br = &br;
    q.sLL = zds_read(&br, d->ll.tableLog); zds_reload(&br);
    q.sOF = zds_read(&br, d->of.tableLog); zds_reload(&br);
    q.sML = zds_read(&br, d->ml.tableLog); zds_reload(&br);
    execstate x; x.out = out; x.outCap = outCap; x.frameOut = frameOut; x.lit = lit; x.litSize = litSize; x.produced = 0; x.litPos = 0;
    int longMode = 0;
    if (windowSize > (1u << 24) && nbSeq > 4) {                     /* ZSTD_getLongOffsetsShare >= 7 (64-bit build) */
      u32 share = 0; const u32 cells = 1u << d->of.tableLog;
      for (u32 u = 0; u < cells; u++) share += d->of.sym[u] > 22;
      share <<= (8 - d->of.tableLog);
      longMode = share >= 7;
    }
    if (!longMode) {
      for (size_t i = 0; i < nbSeq; i++) {
        const seq3 sq = decode_sequence(&q);
        const int e = exec_sequence(&x, sq);
        zds_reload(&br);
        if (e) FAIL(e);
      }
      if (zds_reload(&br) < ZDS_COMPLETED) FAIL(ZO_E_CORRUPTION);
    } else {
      seq3 ring[4]; size_t i = 0;
      for (; zds_reload(&br) <= ZDS_COMPLETED && i < 4; i++) ring[i] = decode_sequence(&q);
      if (i < 4) FAIL(ZO_E_CORRUPTION);
      for (; zds_reload(&br) <= ZDS_COMPLETED && i < nbSeq; i++) {
        const seq3 sq = decode_sequence(&q);
        const int e = exec_sequence(&x, ring[(i - 4) & 3]);
        if (e) FAIL(e);
        ring[i & 3] = sq;
      }
      if (i < nbSeq) FAIL(ZO_E_CORRUPTION);
      for (i -= 4; i < nbSeq; i++) { const int e = exec_sequence(&x, ring[i & 3]); if (e) FAIL(e); }
    }
    produced = x.produced; litPos = x.litPos;
    }
  }
  if (litSize - litPos > outCap - produced) FAIL(ZO_E_DSTSIZE_TOOSMALL);
  memcpy(out + produced, lit + litPos, litSize - litPos);
  produced += litSize - litPos;
#undef FAIL
  free(lit);
  return produced;
}

/* parse the frame header; returns header size or error. */
static size_t parse_frame_header(const u8* src, size_t n, size_t* blockMax, int* checksum, u64* contentSize, u64* windowSize) {
  /* check order of ZSTD_decompressFrame + ZSTD_getFrameHeader_advanced (1.4.9): the size checks come before the magic number */
  if (n < 6 + 3) return ZO_ERR(ZO_E_SRCSIZE_WRONG);
  unsigned fhd = src[4], did = fhd & 3, ss = (fhd >> 5) & 1, fcs = fhd >> 6;
  static const unsigned didSize[4] = {0, 1, 2, 4};
  size_t fcsSize = fcs == 0 ? ss : fcs == 1 ? 2 : fcs == 2 ? 4 : 8;
  size_t hs = 5 + !ss + didSize[did] + fcsSize;
  if (n < hs + 3) return ZO_ERR(ZO_E_SRCSIZE_WRONG);
  if (rd32(src) != 0xFD2FB528u) return ZO_ERR(ZO_E_PREFIX_UNKNOWN);
  if (fhd & 8) return ZO_ERR(ZO_E_FRAMEPARAM_UNSUPPORTED);
  u64 window;
  if (!ss) {
    unsigned b = src[5], wl = 10 + (b >> 3);
    if (wl > 31) return ZO_ERR(ZO_E_WINDOW_TOO_LARGE);   /* windowLog > ZSTD_WINDOWLOG_MAX; the one-shot decoder has no other window limit */
    window = (1ULL << wl) + ((1ULL << wl) >> 3) * (b & 7);
  } else {
    const u8* q = src + 5 + didSize[did];
    window = fcs == 0 ? q[0] : fcs == 1 ? (u64)rd16(q) + 256 : fcs == 2 ? rd32(q) : rd64(q);
  }
  {
    /* frame content size (RFC 8878 3.1.1.1.4): absent (-1) or 1/2/4/8 bytes, the 2-byte form is biased by 256; a non-zero
       dictionary id cannot be honoured (the reference never loads one): dictionary_wrong, as ZSTD_decompressFrame reports it */
    const u8* q = src + 5 + !ss;
    u32 dict = did == 0 ? 0 : did == 1 ? q[0] : did == 2 ? rd16(q) : rd32(q);
    if (dict) return ZO_ERR(ZO_E_DICT_WRONG);
    q += didSize[did];
    *contentSize = fcsSize == 0 ? (u64)-1 : fcsSize == 1 ? q[0] : fcsSize == 2 ? (u64)rd16(q) + 256 : fcsSize == 4 ? rd32(q) : rd64(q);
  }
  *windowSize = window;
  *blockMax = window < (128u << 10) ? (size_t)window : (128u << 10);
  *checksum = (fhd >> 2) & 1;
  return hs;
}

static size_t decode_frame(u8* dst, size_t cap, const u8* src, size_t n, size_t* consumed) {
  size_t blockMax; int checksum; u64 contentSize, windowSize;
  size_t hs = parse_frame_header(src, n, &blockMax, &checksum, &contentSize, &windowSize);
  if (ZO_ISERR(hs)) return hs;
  (void)blockMax;
  const u8* p = src + hs; size_t rem = n - hs, produced = 0;
  dctx* d = (dctx*)calloc(1, sizeof(dctx));
  if (!d) return ZO_ERR(ZO_E_GENERIC);
  d->rep[0] = 1; d->rep[1] = 4; d->rep[2] = 8;
  for (;;) {
    if (rem < 3) { free(d); return ZO_ERR(ZO_E_SRCSIZE_WRONG); }
    u32 bh = rd24(p); p += 3; rem -= 3;
    unsigned last = bh & 1, type = (bh >> 1) & 3; size_t bs = bh >> 3;
    size_t r;
    if (type == 3) { free(d); return ZO_ERR(ZO_E_CORRUPTION); }
    if (type == 0) {
      if (bs > rem) { free(d); return ZO_ERR(ZO_E_SRCSIZE_WRONG); }
      if (bs > cap - produced) { free(d); return ZO_ERR(ZO_E_DSTSIZE_TOOSMALL); }
      memcpy(dst + produced, p, bs); r = bs; p += bs; rem -= bs;
    } else if (type == 1) {
      if (rem < 1) { free(d); return ZO_ERR(ZO_E_SRCSIZE_WRONG); }
      if (bs > cap - produced) { free(d); return ZO_ERR(ZO_E_DSTSIZE_TOOSMALL); }
      memset(dst + produced, p[0], bs); r = bs; p += 1; rem -= 1;
    } else {
      if (bs > rem) { free(d); return ZO_ERR(ZO_E_SRCSIZE_WRONG); }
      if (bs >= (128u << 10)) { free(d); return ZO_ERR(ZO_E_SRCSIZE_WRONG); } /* single-pass decoder checks the constant, not the window */
      r = decode_block(d, dst + produced, cap - produced, dst, p, bs, 128u << 10, windowSize);
      if (ZO_ISERR(r)) { free(d); return r; }
      p += bs; rem -= bs;
    }
    produced += r;
    if (last) break;
  }
  free(d);
  if (contentSize != (u64)-1 && contentSize != produced) return ZO_ERR(ZO_E_CORRUPTION);   /* declared size first, then the checksum */
  if (checksum) {
    if (rem < 4) return ZO_ERR(ZO_E_CHECKSUM_WRONG);
    if (rd32(p) != (u32)zo_xxh64(dst, produced, 0)) return ZO_ERR(ZO_E_CHECKSUM_WRONG);
    p += 4;
  }
  *consumed = (size_t)(p - src);
  return produced;
}

size_t zo_decompress(void* dstv, size_t cap, const void* srcv, size_t n) {
  u8* dst = (u8*)dstv; const u8* src = (const u8*)srcv;
  size_t total = 0; int more = 0;
  while (n >= 5) {
    u32 magic = rd32(src);
    if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {
      if (n < 8) return ZO_ERR(ZO_E_SRCSIZE_WRONG);
      size_t skip = (size_t)rd32(src + 4) + 8;
      if (skip > n) return ZO_ERR(ZO_E_SRCSIZE_WRONG);
      src += skip; n -= skip; continue;
    }
    size_t consumed = 0;
    size_t r = decode_frame(dst + total, cap - total, src, n, &consumed);
    if (ZO_ISERR(r)) {
      if (ZO_ERRCODE(r) == ZO_E_PREFIX_UNKNOWN && more) return ZO_ERR(ZO_E_SRCSIZE_WRONG);
      return r;
    }
    total += r; src += consumed; n -= consumed; more = 1;
  }
  if (n) return ZO_ERR(ZO_E_SRCSIZE_WRONG);
  return total;
}

/* ZSTD_findFrameCompressedSize: walk block headers only */
size_t zo_find_frame_size(const void* srcv, size_t n) {
  const u8* src = (const u8*)srcv;
  if (n >= 8 && (rd32(src) & 0xFFFFFFF0u) == 0x184D2A50u) {
    size_t skip = (size_t)rd32(src + 4) + 8;
    return skip > n ? ZO_ERR(ZO_E_SRCSIZE_WRONG) : skip;
  }
  size_t blockMax; int checksum; u64 contentSize, windowSize;
  size_t hs = parse_frame_header(src, n, &blockMax, &checksum, &contentSize, &windowSize);
  if (ZO_ISERR(hs)) return hs;
  const u8* p = src + hs; size_t rem = n - hs;
  for (;;) {
    if (rem < 3) return ZO_ERR(ZO_E_SRCSIZE_WRONG);
    u32 bh = rd24(p); p += 3; rem -= 3;
    unsigned type = (bh >> 1) & 3; size_t bs = type == 1 ? 1 : (bh >> 3);
    if (type == 3) return ZO_ERR(ZO_E_CORRUPTION);
    if (bs > rem) return ZO_ERR(ZO_E_SRCSIZE_WRONG);
    p += bs; rem -= bs;
    if (bh & 1) break;
  }
  if (checksum) { if (rem < 4) return ZO_ERR(ZO_E_SRCSIZE_WRONG); p += 4; }
  return (size_t)(p - src);
}
