/* ORACLE — TEST INFRASTRUCTURE ONLY (see zo_common.h). Internal helpers shared by the oracle's C files. */
#ifndef ZO_INTERNAL_H
#define ZO_INTERNAL_H
#include "zo_common.h"
#include <string.h>

#define ZO_MAXSYM 255

static inline u16 rd16(const void* p) { u16 v; memcpy(&v, p, 2); return v; }
static inline u32 rd32(const void* p) { u32 v; memcpy(&v, p, 4); return v; }
static inline u64 rd64(const void* p) { u64 v; memcpy(&v, p, 8); return v; }
static inline u32 rd24(const u8* p) { return (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16); }
static inline void wr16(void* p, u16 v) { memcpy(p, &v, 2); }
static inline void wr24(u8* p, u32 v) { p[0] = (u8)v; p[1] = (u8)(v >> 8); p[2] = (u8)(v >> 16); }
static inline void wr32(void* p, u32 v) { memcpy(p, &v, 4); }
static inline u32 hb32(u32 x) { return x ? 31u - (u32)__builtin_clz(x) : 0; }

/* forward LSB-first bit writer (BIT_CStream): bits appended low-to-high, bytes little-endian */
typedef struct { u8* start; u8* ptr; u8* end; u64 acc; int n; int overflow; } zo_bitw;
static inline void zo_bitw_init(zo_bitw* b, u8* dst, size_t cap) { b->start = b->ptr = dst; b->end = dst + cap; b->acc = 0; b->n = 0; b->overflow = 0; }
static inline void zo_bitw_add(zo_bitw* b, u64 v, unsigned nb) {
  if (nb == 0) return;
  v &= (nb >= 64) ? ~0ULL : ((1ULL << nb) - 1);
  b->acc |= v << b->n;
  int total = b->n + (int)nb;
  while (total >= 8) {
    if (b->ptr < b->end) *b->ptr++ = (u8)b->acc; else b->overflow = 1;
    b->acc >>= 8; total -= 8;
  }
  b->n = total;
}
/* closing 1-bit then flush; returns byte size, 0 on overflow */
static inline size_t zo_bitw_close(zo_bitw* b) {
  zo_bitw_add(b, 1, 1);
  if (b->n > 0) { if (b->ptr < b->end) *b->ptr++ = (u8)b->acc; else b->overflow = 1; b->n = 0; }
  if (b->overflow) return 0;
  return (size_t)(b->ptr - b->start);
}

/* backward bitstream reader (A.2 convention): pos = number of unread bits; may go negative (zeros are read) */
typedef struct { const u8* src; long pos; } zo_bitr;
static inline int zo_bitr_init(zo_bitr* b, const u8* src, size_t n) {
  if (n == 0 || src[n - 1] == 0) return -1;
  b->src = src;
  b->pos = (long)(n - 1) * 8 + (long)hb32(src[n - 1]);
  return 0;
}
static inline u32 zo_bitr_read(zo_bitr* b, unsigned nb) {
  u32 v = 0;
  if (nb == 0) return 0;
  b->pos -= (long)nb;
  for (unsigned i = 0; i < nb; i++) {
    long bit = b->pos + (long)i;
    if (bit >= 0) v |= (u32)((b->src[bit >> 3] >> (bit & 7)) & 1) << i;
  }
  return v;
}

/* forward LSB-first peek of up to 32 bits at bit offset `bitpos` (zeros past the end) */
static inline u64 zo_peek_fwd(const u8* src, size_t n, size_t bitpos) {
  u64 v = 0;
  size_t byte = bitpos >> 3;
  for (int i = 0; i < 6; i++) if (byte + (size_t)i < n) v |= (u64)src[byte + (size_t)i] << (8 * i);
  return v >> (bitpos & 7);
}

typedef struct {
  unsigned tableLog, maxSym; int rle;
  u16 stateTable[1 << 9];
  u32 deltaNbBits[53];
  int deltaFindState[53];
} zo_fse_ctable;
typedef struct { unsigned tableLog; u8 sym[1 << 9]; u8 nbBits[1 << 9]; u16 base[1 << 9]; } zo_fse_dtable;
typedef struct { unsigned maxSym, tableLog; u8 nbBits[256]; u16 val[256]; } zo_huf_ctable;
typedef struct { unsigned maxBits; u8 sym[1 << 12]; u8 nbBits[1 << 12]; } zo_huf_dtable;   /* HUF_TABLELOG_MAX = 12 */

unsigned zo_fse_optimal_tablelog(unsigned maxLog, size_t n, unsigned maxSym, unsigned minus);
int zo_fse_normalize(s16* norm, unsigned t, const u32* cnt, size_t total, unsigned maxSym, int useLowProb);
size_t zo_fse_write_ncount(u8* out, size_t cap, const s16* norm, unsigned maxSym, unsigned t);
size_t zo_fse_read_ncount(s16* norm, unsigned* maxSymPtr, unsigned* tPtr, const u8* src, size_t n, unsigned maxAL);
int zo_fse_build_ctable(zo_fse_ctable* ct, const s16* norm, unsigned maxSym, unsigned t);
void zo_fse_build_ctable_rle(zo_fse_ctable* ct, unsigned sym);
u32 zo_fse_init_state(const zo_fse_ctable* ct, unsigned sym);
u32 zo_fse_encode(const zo_fse_ctable* ct, u32* state, unsigned sym, u32* bits);
int zo_fse_build_dtable(zo_fse_dtable* dt, const s16* norm, unsigned maxSym, unsigned t);
unsigned zo_huf_build(zo_huf_ctable* ct, const u32* count, unsigned maxSym, unsigned maxNbBits);
size_t zo_huf_write_ctable(u8* dst, size_t cap, const zo_huf_ctable* ct);
size_t zo_huf_read_weights(u8* weights, unsigned* nSymPtr, unsigned* maxBitsPtr, const u8* src, size_t n);
int zo_huf_build_dtable(zo_huf_dtable* dt, const u8* weights, unsigned nSym, unsigned maxBits);

/* code tables (A.3) */
extern const u32 zo_ll_base[36]; extern const u8 zo_ll_bits[36];
extern const u32 zo_ml_base[53]; extern const u8 zo_ml_bits[53];
extern const s16 zo_ll_defnorm[36]; extern const s16 zo_ml_defnorm[53]; extern const s16 zo_of_defnorm[29];
#endif
