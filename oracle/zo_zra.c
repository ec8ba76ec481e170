/* ORACLE — TEST INFRASTRUCTURE ONLY (see zo_common.h).
 *
 * ZRA container restated from the reference source/zra.cpp:
 *   FixedHeader / Entry / CalculateHash  zra.cpp:88-139
 *   Header parse + validation            zra.cpp:141-171
 *   GetOutputBufferSize                  zra.cpp:189-192
 *   CompressBuffer                       zra.cpp:194-234
 *   DecompressBuffer                     zra.cpp:243-250
 *   DecompressRA (3-phase)               zra.cpp:258-296
 * Two codec backends sit under the same container code:
 *   backend 0 = the restatement in zo_encode.c / zo_decode.c
 *   backend 1 = the real dependency, libzstd 1.4.9, dlopen()ed from the image (pins the restatement
 *               and is the timed CPU baseline of bench.py; see oracle/README.md)
 */
#include "zo_internal.h"
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>

enum { ZRA_Success, ZRA_ZStdError, ZRA_VersionLow, ZRA_HeaderInvalid, ZRA_HeaderIncomplete, ZRA_OutOfBounds, ZRA_OutputTooSmall, ZRA_CompressedTooLarge, ZRA_FrameSizeMismatch };

/* ---------------------------------------------------------------- libzstd backend (dlopen) */
typedef size_t (*fn_compress2)(void*, void*, size_t, const void*, size_t);
typedef size_t (*fn_setparam)(void*, int, int);
typedef void* (*fn_create)(void);
typedef size_t (*fn_free)(void*);
typedef size_t (*fn_decompress)(void*, void*, size_t, const void*, size_t);
typedef unsigned (*fn_iserr)(size_t);
typedef int (*fn_errcode)(size_t);
static struct {
  void* h; fn_compress2 compress2; fn_setparam setparam; fn_create createC, createD; fn_free freeC, freeD;
  fn_decompress decompress; fn_iserr iserr; fn_errcode errcode; const char* (*version)(void);
} Z;

int zo_libzstd_load(const char* path) {
  if (Z.h) return 0;
  const char* cands[] = {path, "/opt/conda/lib/libzstd.so.1.4.9", "libzstd.so.1.4.9", "/usr/lib/x86_64-linux-gnu/libzstd.so.1.4.8", "libzstd.so.1", NULL};
  for (int i = 0; i < 5 && !Z.h; i++) if (cands[i]) Z.h = dlopen(cands[i], RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);   /* DEEPBIND: a profiler's preloaded zstd must not interpose on 1.4.9's own calls */
  if (!Z.h) return -1;
  Z.compress2 = (fn_compress2)dlsym(Z.h, "ZSTD_compress2");
  Z.setparam = (fn_setparam)dlsym(Z.h, "ZSTD_CCtx_setParameter");
  Z.createC = (fn_create)dlsym(Z.h, "ZSTD_createCCtx"); Z.freeC = (fn_free)dlsym(Z.h, "ZSTD_freeCCtx");
  Z.createD = (fn_create)dlsym(Z.h, "ZSTD_createDCtx"); Z.freeD = (fn_free)dlsym(Z.h, "ZSTD_freeDCtx");
  Z.decompress = (fn_decompress)dlsym(Z.h, "ZSTD_decompressDCtx");
  Z.iserr = (fn_iserr)dlsym(Z.h, "ZSTD_isError"); Z.errcode = (fn_errcode)dlsym(Z.h, "ZSTD_getErrorCode");
  Z.version = (const char* (*)(void))dlsym(Z.h, "ZSTD_versionString");
  if (!Z.compress2 || !Z.setparam || !Z.createC || !Z.decompress || !Z.errcode) { dlclose(Z.h); Z.h = NULL; return -1; }
  return 0;
}
const char* zo_libzstd_version(void) { return (Z.h && Z.version) ? Z.version() : ""; }
/* any other export of the loaded dependency (tests pin internals such as HUF_selectDecoder and ZSTD_getErrorString through it) */
void* zo_libzstd_symbol(const char* name) { return (zo_libzstd_load(NULL) == 0) ? dlsym(Z.h, name) : NULL; }

/* a per-call codec context, mirroring ZCCtx / ZDCtx (zra.cpp:25-44) */
typedef struct { int backend; void* cctx; void* dctx; int level, checksum; } codec;
/* side: 1 = a compression context only (ZCCtx, zra.cpp:209), 2 = a decompression context only (ZDCtx, zra.cpp:248, 271) — the
   reference creates exactly one of them per call, and bench.py's CPU baseline times these calls */
static int codec_open(codec* k, int backend, int level, int checksum, int side) {
  k->backend = backend; k->cctx = k->dctx = NULL; k->level = level; k->checksum = checksum;
  if (backend == 1) {
    if (zo_libzstd_load(NULL)) return -1;
    if (side & 1) {
      k->cctx = Z.createC();
      /* zra.cpp:210-213: ZSTD_c_compressionLevel=100, contentSizeFlag=200, checksumFlag=201, dictIDFlag=202 */
      Z.setparam(k->cctx, 100, level); Z.setparam(k->cctx, 200, 0); Z.setparam(k->cctx, 201, checksum); Z.setparam(k->cctx, 202, 0);
    }
    if (side & 2) k->dctx = Z.createD();
  }
  return 0;
}
static void codec_close(codec* k) { if (k->backend == 1) { if (k->cctx) Z.freeC(k->cctx); if (k->dctx) Z.freeD(k->dctx); } }
/* both return a size or set *err to the zstd error code */
static size_t codec_compress(codec* k, void* dst, size_t cap, const void* src, size_t n, int* err) {
  *err = 0;
  if (k->backend == 1) { size_t r = Z.compress2(k->cctx, dst, cap, src, n); if (Z.iserr(r)) { *err = Z.errcode(r); return 0; } return r; }
  size_t r = zo_compress_frame(dst, cap, src, n, k->level, k->checksum);
  if (ZO_ISERR(r)) { *err = ZO_ERRCODE(r); return 0; }
  return r;
}
static size_t codec_decompress(codec* k, void* dst, size_t cap, const void* src, size_t n, int* err) {
  *err = 0;
  if (k->backend == 1) { size_t r = Z.decompress(k->dctx, dst, cap, src, n); if (Z.iserr(r)) { *err = Z.errcode(r); return 0; } return r; }
  size_t r = zo_decompress(dst, cap, src, n);
  if (ZO_ISERR(r)) { *err = ZO_ERRCODE(r); return 0; }
  return r;
}

/* ---------------------------------------------------------------- container */
#define FIXED 38 /* sizeof(FixedHeader), zra.cpp:111-126 */
static zo_status st(int zra, int zstd) { zo_status s; s.zra = zra; s.zstd = (int)(int8_t)zstd; return s; } /* i8 narrowing, zra.cpp:440 */
static u64 entry_get(const u8* p) { return (u64)rd32(p) | ((u64)p[4] << 32); }          /* zra.cpp:96-107 */
static void entry_put(u8* p, u64 v) { wr32(p, (u32)v); p[4] = (u8)(v >> 32); }

size_t zo_zra_output_bound(size_t inputSize, u32 frameSize, u32 metaSize) {
  u32 tableSize = (u32)(inputSize / frameSize) + ((inputSize % frameSize) ? 2 : 1);
  return FIXED + metaSize + (size_t)tableSize * 5 + zo_compress_bound(frameSize) * (size_t)(tableSize - 1);
}

static void write_fixed(u8* h, u64 origSize, u32 tableSize, u32 frameSize, u32 metaSize) {
  wr32(h, 0x184D2A50u); wr32(h + 4, FIXED + metaSize + tableSize * 5 - 8); wr32(h + 8, 0x3041525Au); wr16(h + 12, 1);
  wr32(h + 14, 0); memcpy(h + 18, &origSize, 8); wr32(h + 26, tableSize); wr32(h + 30, frameSize); wr32(h + 34, metaSize);
}
/* CalculateHash zra.cpp:128-133: [0,14) || [18,38) || rest (headerSize - 38 + 8 bytes starting at `rest`) */
static u32 header_hash(const u8* h, const u8* rest) {
  u32 crc = zo_crc32(0, h, 14);
  crc = zo_crc32(crc, h + 18, 20);
  return zo_crc32(crc, rest, rd32(h + 4) - FIXED + 8);
}

static zo_status compress_buffer(int backend, const u8* in, size_t n, u8* out, size_t outCap, size_t* outSize,
                                 int level, u32 frameSize, int checksum, size_t metaSize) {
  u32 tableSize = (u32)(n / frameSize) + ((n % frameSize) ? 2 : 1);
  size_t need = FIXED + (size_t)tableSize * 5 + zo_compress_bound(frameSize) * (size_t)(tableSize - 1);
  if (outCap < need) return st(ZRA_OutputTooSmall, 0);
  write_fixed(out, n, tableSize, frameSize, (u32)metaSize);
  size_t off = FIXED;
  u8* entry = out + off;                 /* NB: table at +38 even when metaSize != 0 — reference quirk zra.cpp:205 */
  off += (size_t)tableSize * 5;
  size_t bodyStart = off, remaining = n;
  codec k;
  if (codec_open(&k, backend, level, checksum, 1)) return st(ZRA_ZStdError, ZO_E_GENERIC);
  size_t fs = frameSize;
  while (remaining) {
    if (fs > remaining) fs = remaining;
    int err; size_t c = codec_compress(&k, out + off, outCap - off, in + (n - remaining), fs, &err);
    if (err) { codec_close(&k); return st(ZRA_ZStdError, err); }
    entry_put(entry, off - bodyStart); entry += 5;
    off += c; remaining -= fs;
  }
  codec_close(&k);
  if (off >= (1ULL << 40)) return st(ZRA_CompressedTooLarge, 0);
  entry_put(entry, off - bodyStart);
  wr32(out + 14, header_hash(out, out + FIXED));
  *outSize = off;
  return st(ZRA_Success, 0);
}

typedef struct { u16 version; u32 size; u64 uncompressedSize; u32 frameSize, metaOffset, metaSize, seekTableOffset, seekTableSize; } hdr;
/* Header(BufferView) zra.cpp:141-171, including the ">=" bounds quirk of the lambda (:166) */
static int parse_header(hdr* h, const u8* in, size_t n) {
  if (0 + FIXED >= n) return ZRA_OutOfBounds;
  if (rd32(in + 8) != 0x3041525Au || rd16(in + 12) > 1) return ZRA_HeaderInvalid;
  h->version = rd16(in + 12);
  h->size = rd32(in + 4) + 8;
  memcpy(&h->uncompressedSize, in + 18, 8);
  h->frameSize = rd32(in + 30); h->metaOffset = FIXED; h->metaSize = rd32(in + 34);
  h->seekTableOffset = FIXED + h->metaSize; h->seekTableSize = rd32(in + 26) * 5;
  if (h->version != 1) return ZRA_VersionLow;
  if (n < h->size) return ZRA_OutOfBounds;
  return 0;
}

/* bytes the last successful decompress_buffer regenerated on this thread: the reference call returns void and promises
   uncompressedSize bytes; when a damaged header makes the two differ, only this prefix is defined (tests compare just that) */
static __thread size_t g_last_produced;
size_t zo_zra_last_produced(void) { return g_last_produced; }

static zo_status decompress_buffer(int backend, const u8* in, size_t n, u8* out, size_t outCap) {
  hdr h; int e = parse_header(&h, in, n);
  if (e) return st(e, 0);
  if (outCap < h.uncompressedSize) return st(ZRA_OutputTooSmall, 0);
  codec k; int err;
  if (codec_open(&k, backend, 0, 0, 2)) return st(ZRA_ZStdError, ZO_E_GENERIC);
  g_last_produced = 0;
  size_t r = codec_decompress(&k, out, outCap, in + h.size, n - h.size, &err);
  if (!err) g_last_produced = r;
  codec_close(&k);
  return err ? st(ZRA_ZStdError, err) : st(ZRA_Success, 0);
}

static zo_status decompress_ra(int backend, const u8* in, size_t n, u8* out, size_t outCap, size_t offset, size_t size) {
  hdr h; int e = parse_header(&h, in, n);
  if (e) return st(e, 0);
  if (offset + size >= h.uncompressedSize) return st(ZRA_OutOfBounds, 0);   /* ">=" quirk, zra.cpp:260 */
  if (outCap < size) return st(ZRA_OutputTooSmall, 0);
  u64 foq = offset / h.frameSize, forem = offset % h.frameSize;
  u64 fzq = (forem + size) / h.frameSize, fzrem = (forem + size) % h.frameSize;
  const u8* first = in + h.seekTableOffset + foq * 5;
  const u8* last = first + (fzq + (fzrem ? 1 : 0)) * 5;
  codec k; int err = 0;
  if (codec_open(&k, backend, 0, 0, 2)) return st(ZRA_ZStdError, ZO_E_GENERIC);
  u8* fb = NULL;
  if (forem || fzrem) fb = (u8*)calloc(h.frameSize ? h.frameSize : 1, 1);
  const u8* contents = in + h.size;
  size_t outOff = 0;
  if (forem) {
    codec_decompress(&k, fb, h.frameSize, contents + entry_get(first), entry_get(first + 5) - entry_get(first), &err);
    if (err) goto done;
    size_t m = h.frameSize - forem; if (size < m) m = size;
    memcpy(out, fb + forem, m); outOff += m; first += 5;
  }
  if (outOff < size) {
    size_t csz = entry_get(fzrem ? last - 5 : last) - entry_get(first);
    outOff += codec_decompress(&k, out + outOff, outCap - outOff, contents + entry_get(first), csz, &err);
    if (err) goto done;
  }
  if (outOff < size && fzrem) {
    codec_decompress(&k, fb, h.frameSize, contents + entry_get(last - 5), entry_get(last) - entry_get(last - 5), &err);
    if (err) goto done;
    memcpy(out + outOff, fb, size - outOff);
  }
done:
  free(fb); codec_close(&k);
  return err ? st(ZRA_ZStdError, err) : st(ZRA_Success, 0);
}

/* ---------------------------------------------------------------- exported: restatement backend */
zo_status zo_zra_compress_buffer(const void* in, size_t n, void* out, size_t outCap, size_t* outSize, int level, u32 frameSize, int checksum, size_t metaSize) {
  return compress_buffer(0, (const u8*)in, n, (u8*)out, outCap, outSize, level, frameSize, checksum, metaSize);
}
zo_status zo_zra_decompress_buffer(const void* in, size_t n, void* out, size_t outCap) { return decompress_buffer(0, (const u8*)in, n, (u8*)out, outCap); }
zo_status zo_zra_decompress_ra(const void* in, size_t n, void* out, size_t outCap, size_t offset, size_t size) { return decompress_ra(0, (const u8*)in, n, (u8*)out, outCap, offset, size); }

/* ---------------------------------------------------------------- exported: libzstd backend ("zl_") */
zo_status zl_zra_compress_buffer(const void* in, size_t n, void* out, size_t outCap, size_t* outSize, int level, u32 frameSize, int checksum, size_t metaSize) {
  return compress_buffer(1, (const u8*)in, n, (u8*)out, outCap, outSize, level, frameSize, checksum, metaSize);
}
zo_status zl_zra_decompress_buffer(const void* in, size_t n, void* out, size_t outCap) { return decompress_buffer(1, (const u8*)in, n, (u8*)out, outCap); }
zo_status zl_zra_decompress_ra(const void* in, size_t n, void* out, size_t outCap, size_t offset, size_t size) { return decompress_ra(1, (const u8*)in, n, (u8*)out, outCap, offset, size); }
/* nq DecompressRA calls of `size` bytes in one C loop (bench.py's CPU baseline: no per-query interpreter overhead in the timed span);
   returns the first failing status */
static zo_status ra_many(int backend, const void* in, size_t n, void* out, size_t size, const u64* offs, size_t nq) {
  for (size_t q = 0; q < nq; q++) {
    zo_status s = decompress_ra(backend, (const u8*)in, n, (u8*)out, size, (size_t)offs[q], size);
    if (s.zra) return s;
  }
  return st(ZRA_Success, 0);
}
zo_status zo_zra_decompress_ra_many(const void* in, size_t n, void* out, size_t size, const u64* offs, size_t nq) { return ra_many(0, in, n, out, size, offs, nq); }
zo_status zl_zra_decompress_ra_many(const void* in, size_t n, void* out, size_t size, const u64* offs, size_t nq) { return ra_many(1, in, n, out, size, offs, nq); }

/* raw per-frame access to the dependency, for pinning zo_compress_frame / zo_decompress */
size_t zl_compress_frame(void* dst, size_t cap, const void* src, size_t n, int level, int checksum) {
  codec k; int err;
  if (codec_open(&k, 1, level, checksum, 1)) return ZO_ERR(ZO_E_GENERIC);
  size_t r = codec_compress(&k, dst, cap, src, n, &err);
  codec_close(&k);
  return err ? ZO_ERR(err) : r;
}
size_t zl_decompress(void* dst, size_t cap, const void* src, size_t n) {
  codec k; int err;
  if (codec_open(&k, 1, 0, 0, 2)) return ZO_ERR(ZO_E_GENERIC);
  size_t r = codec_decompress(&k, dst, cap, src, n, &err);
  codec_close(&k);
  return err ? ZO_ERR(err) : r;
}
