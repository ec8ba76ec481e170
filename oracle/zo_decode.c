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
  zo_huf_dtable huf; int hufValid;
  zo_fse_dtable ll, of, ml; int llValid, ofValid, mlValid;
  int llRle, ofRle, mlRle; /* RLE tables are stored as 1-cell dtables with tableLog 0 */
  u32 rep[3];
} dctx;

static void rle_dtable(zo_fse_dtable* dt, unsigned sym) { dt->tableLog = 0; dt->sym[0] = (u8)sym; dt->nbBits[0] = 0; dt->base[0] = 0; }

/* decode `regen` symbols from one backward Huffman stream; 0 ok */
static int huf_decode_stream(u8* out, size_t regen, const u8* src, size_t n, const zo_huf_dtable* dt) {
  zo_bitr br;
  if (zo_bitr_init(&br, src, n)) return -1;
  unsigned mb = dt->maxBits;
  for (size_t i = 0; i < regen; i++) {
    /* peek maxBits (zeros below bit 0) */
    u32 v = 0;
    for (unsigned k = 0; k < mb; k++) {
      long bit = br.pos - (long)mb + (long)k;
      if (bit >= 0) v |= (u32)((br.src[bit >> 3] >> (bit & 7)) & 1) << k;
    }
    out[i] = dt->sym[v];
    br.pos -= dt->nbBits[v];
  }
  return br.pos == 0 ? 0 : -1;
}

/* returns bytes of the literals section consumed, or error; literals land in lit[0..*litSize) */
static size_t decode_literals(dctx* d, u8* lit, size_t* litSize, const u8* src, size_t n, size_t blockMax) {
  if (n < 1) return ZO_ERR(ZO_E_CORRUPTION);
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
  if (n < 5 && !(n >= 3 && sf < 2) && !(n >= 4 && sf == 2)) return ZO_ERR(ZO_E_CORRUPTION);
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
    p += h; rem -= h;
  } else if (!d->hufValid) return ZO_ERR(30); /* dictionary_corrupted: treeless without a previous table */
  if (streams == 1) {
    if (huf_decode_stream(lit, regen, p, rem, &d->huf)) return ZO_ERR(ZO_E_CORRUPTION);
  } else {
    if (rem < 10) return ZO_ERR(ZO_E_CORRUPTION);
    size_t s1 = rd16(p), s2 = rd16(p + 2), s3 = rd16(p + 4);
    if (6 + s1 + s2 + s3 > rem) return ZO_ERR(ZO_E_CORRUPTION);
    size_t s4 = rem - 6 - s1 - s2 - s3;
    size_t seg = (regen + 3) / 4;
    if (seg * 3 > regen) return ZO_ERR(ZO_E_CORRUPTION);
    const u8* q = p + 6;
    if (huf_decode_stream(lit, seg, q, s1, &d->huf)) return ZO_ERR(ZO_E_CORRUPTION);
    if (huf_decode_stream(lit + seg, seg, q + s1, s2, &d->huf)) return ZO_ERR(ZO_E_CORRUPTION);
    if (huf_decode_stream(lit + 2 * seg, seg, q + s1 + s2, s3, &d->huf)) return ZO_ERR(ZO_E_CORRUPTION);
    if (huf_decode_stream(lit + 3 * seg, regen - 3 * seg, q + s1 + s2 + s3, s4, &d->huf)) return ZO_ERR(ZO_E_CORRUPTION);
  }
  *litSize = regen;
  return lh + comp;
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
static size_t decode_block(dctx* d, u8* out, size_t outCap, const u8* frameOut, const u8* src, size_t n, size_t blockMax) {
  u8* lit = (u8*)malloc(blockMax + 32);
  size_t litSize = 0, produced = 0;
  if (!lit) return ZO_ERR(ZO_E_GENERIC);
  size_t r = decode_literals(d, lit, &litSize, src, n, blockMax);
  if (ZO_ISERR(r)) { free(lit); return r; }
  const u8* p = src + r; size_t rem = n - r;
#define FAIL(code) do { free(lit); return ZO_ERR(code); } while (0)
  if (rem < 1) FAIL(ZO_E_SRCSIZE_WRONG);
  size_t nbSeq = p[0];
  if (nbSeq == 0) { p++; rem--; if (rem) FAIL(ZO_E_CORRUPTION); }
  else if (nbSeq < 128) { p++; rem--; }
  else if (nbSeq < 255) { if (rem < 2) FAIL(ZO_E_SRCSIZE_WRONG); nbSeq = ((nbSeq - 128) << 8) + p[1]; p += 2; rem -= 2; }
  else { if (rem < 3) FAIL(ZO_E_SRCSIZE_WRONG); nbSeq = (size_t)p[1] + ((size_t)p[2] << 8) + 0x7F00; p += 3; rem -= 3; }
  size_t litPos = 0;
  if (nbSeq) {
    if (rem < 1) FAIL(ZO_E_SRCSIZE_WRONG);
    unsigned modes = p[0]; p++; rem--;
    if (modes & 3) FAIL(ZO_E_CORRUPTION);
    size_t h;
    h = decode_seq_table(&d->ll, &d->llValid, modes >> 6, p, rem, 35, 9, zo_ll_defnorm, 35, 6); if (ZO_ISERR(h)) FAIL(ZO_E_CORRUPTION); p += h; rem -= h;
    h = decode_seq_table(&d->of, &d->ofValid, (modes >> 4) & 3, p, rem, 31, 8, zo_of_defnorm, 28, 5); if (ZO_ISERR(h)) FAIL(ZO_E_CORRUPTION); p += h; rem -= h;
    h = decode_seq_table(&d->ml, &d->mlValid, (modes >> 2) & 3, p, rem, 52, 9, zo_ml_defnorm, 52, 6); if (ZO_ISERR(h)) FAIL(ZO_E_CORRUPTION); p += h; rem -= h;
    zo_bitr br;
    if (zo_bitr_init(&br, p, rem)) FAIL(ZO_E_CORRUPTION);
    u32 sLL = zo_bitr_read(&br, d->ll.tableLog), sOF = zo_bitr_read(&br, d->of.tableLog), sML = zo_bitr_read(&br, d->ml.tableLog);
    if (br.pos < 0) FAIL(ZO_E_CORRUPTION);
    for (size_t i = 0; i < nbSeq; i++) {
      unsigned llc = d->ll.sym[sLL], ofc = d->of.sym[sOF], mlc = d->ml.sym[sML];
      if (ofc > 31 || llc > 35 || mlc > 52) FAIL(ZO_E_CORRUPTION);
      u32 offVal = (1u << ofc) + zo_bitr_read(&br, ofc);
      u32 ml = zo_ml_base[mlc] + zo_bitr_read(&br, zo_ml_bits[mlc]);
      u32 ll = zo_ll_base[llc] + zo_bitr_read(&br, zo_ll_bits[llc]);
      if (i + 1 < nbSeq) {
        sLL = d->ll.base[sLL] + zo_bitr_read(&br, d->ll.nbBits[sLL]);
        sML = d->ml.base[sML] + zo_bitr_read(&br, d->ml.nbBits[sML]);
        sOF = d->of.base[sOF] + zo_bitr_read(&br, d->of.nbBits[sOF]);
      }
      if (br.pos < 0) FAIL(ZO_E_CORRUPTION);
      u32 off;
      if (offVal > 3) { off = offVal - 3; d->rep[2] = d->rep[1]; d->rep[1] = d->rep[0]; d->rep[0] = off; }
      else {
        u32 idx = offVal + (ll == 0);
        if (idx == 1) off = d->rep[0];
        else {
          off = idx == 2 ? d->rep[1] : idx == 3 ? d->rep[2] : d->rep[0] - 1;
          off += !off;                                  /* zstd 1.4.9: "offset == 0 means corruption, but force offset to 1" (no error) */
          if (idx != 2) d->rep[2] = d->rep[1];
          d->rep[1] = d->rep[0]; d->rep[0] = off;
        }
      }
      /* ZSTD_execSequenceEnd order: destination room first, then the literal buffer, then the offset */
      if ((size_t)ll + ml > outCap - produced) FAIL(ZO_E_DSTSIZE_TOOSMALL);
      if (ll > litSize - litPos) FAIL(ZO_E_CORRUPTION);
      memcpy(out + produced, lit + litPos, ll); produced += ll; litPos += ll;
      if (off > (size_t)(out + produced - frameOut)) FAIL(ZO_E_CORRUPTION);
      for (u32 k = 0; k < ml; k++) out[produced + k] = out[produced + k - off];
      produced += ml;
    }
    if (br.pos != 0) FAIL(ZO_E_CORRUPTION);
  }
  if (litSize - litPos > outCap - produced) FAIL(ZO_E_DSTSIZE_TOOSMALL);
  memcpy(out + produced, lit + litPos, litSize - litPos);
  produced += litSize - litPos;
#undef FAIL
  free(lit);
  return produced;
}

/* parse the frame header; returns header size or error. */
static size_t parse_frame_header(const u8* src, size_t n, size_t* blockMax, int* checksum, u64* contentSize) {
  if (n < 5) return ZO_ERR(ZO_E_SRCSIZE_WRONG);
  if (rd32(src) != 0xFD2FB528u) return ZO_ERR(ZO_E_PREFIX_UNKNOWN);
  unsigned fhd = src[4], did = fhd & 3, ss = (fhd >> 5) & 1, fcs = fhd >> 6;
  static const unsigned didSize[4] = {0, 1, 2, 4};
  size_t fcsSize = fcs == 0 ? ss : fcs == 1 ? 2 : fcs == 2 ? 4 : 8;
  size_t hs = 5 + !ss + didSize[did] + fcsSize;
  if (n < hs) return ZO_ERR(ZO_E_SRCSIZE_WRONG);
  if (fhd & 8) return ZO_ERR(ZO_E_FRAMEPARAM_UNSUPPORTED);
  u64 window;
  if (!ss) {
    unsigned b = src[5], wl = 10 + (b >> 3);
    if (wl > 31) return ZO_ERR(ZO_E_FRAMEPARAM_UNSUPPORTED);
    window = (1ULL << wl) + ((1ULL << wl) >> 3) * (b & 7);
  } else {
    const u8* q = src + 5 + didSize[did];
    window = fcs == 0 ? q[0] : fcs == 1 ? (u64)rd16(q) + 256 : fcs == 2 ? rd32(q) : rd64(q);
  }
  if (window > (1ULL << 27) + 1 && !ss) return ZO_ERR(ZO_E_WINDOW_TOO_LARGE);
  {
    /* frame content size (RFC 8878 3.1.1.1.4): absent (-1) or 1/2/4/8 bytes, the 2-byte form is biased by 256; a non-zero
       dictionary id cannot be honoured (the reference never loads one): dictionary_wrong, as ZSTD_decompressFrame reports it */
    const u8* q = src + 5 + !ss;
    u32 dict = did == 0 ? 0 : did == 1 ? q[0] : did == 2 ? rd16(q) : rd32(q);
    if (dict) return ZO_ERR(ZO_E_DICT_WRONG);
    q += didSize[did];
    *contentSize = fcsSize == 0 ? (u64)-1 : fcsSize == 1 ? q[0] : fcsSize == 2 ? (u64)rd16(q) + 256 : fcsSize == 4 ? rd32(q) : rd64(q);
  }
  *blockMax = window < (128u << 10) ? (size_t)window : (128u << 10);
  *checksum = (fhd >> 2) & 1;
  return hs;
}

static size_t decode_frame(u8* dst, size_t cap, const u8* src, size_t n, size_t* consumed) {
  size_t blockMax; int checksum; u64 contentSize;
  size_t hs = parse_frame_header(src, n, &blockMax, &checksum, &contentSize);
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
      r = decode_block(d, dst + produced, cap - produced, dst, p, bs, 128u << 10);
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
  size_t blockMax; int checksum; u64 contentSize;
  size_t hs = parse_frame_header(src, n, &blockMax, &checksum, &contentSize);
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
