/* ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under zra_amd/ may include, link or call this.
 *
 * Shared declarations of the CPU restatement of the reference hot path
 * (zraorg/ZRA source/zra.cpp + its un-vendored dependency facebook/zstd, pinned to 1.4.9
 *  semantics — see oracle/README.md for the pin and how it was validated).
 */
#ifndef ZO_COMMON_H
#define ZO_COMMON_H
#include <stddef.h>
#include <stdint.h>

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int16_t s16;

/* zstd error codes that the reference surfaces through ZraStatus.zstd (zra.cpp:19-23). */
enum {
  ZO_OK = 0,
  ZO_E_GENERIC = 1,
  ZO_E_PREFIX_UNKNOWN = 10,
  ZO_E_FRAMEPARAM_UNSUPPORTED = 14,
  ZO_E_WINDOW_TOO_LARGE = 16,
  ZO_E_CORRUPTION = 20,
  ZO_E_CHECKSUM_WRONG = 22,
  ZO_E_DICT_CORRUPTED = 30,
  ZO_E_DICT_WRONG = 32,
  ZO_E_PARAM_UNSUPPORTED = 40,
  ZO_E_DSTSIZE_TOOSMALL = 70,
  ZO_E_SRCSIZE_WRONG = 72,
};
#define ZO_ERR(code) ((size_t)0 - (size_t)(code))
#define ZO_ISERR(x) ((x) > (size_t)0 - (size_t)120)
#define ZO_ERRCODE(x) ((int)((size_t)0 - (x)))

typedef struct {
  unsigned windowLog, chainLog, hashLog, searchLog, minMatch, targetLength, strategy;
} zo_cparams;

/* one sequence as the match finder emits it: offsetValue 1..3 = repcode slot, else offset+3 */
typedef struct {
  u32 litLength, matchLength, offsetValue;
} zo_seq;

#ifdef __cplusplus
extern "C" {
#endif

/* --- primitives --- */
u32 zo_crc32(u32 crc, const void* data, size_t n);          /* CRCpp CRC_32() == zlib crc32 */
u64 zo_xxh64(const void* data, size_t n, u64 seed);          /* zstd content checksum (low 32 bits stored) */

/* --- zstd 1.4.9 codec restatement --- */
size_t zo_compress_bound(size_t n);                          /* ZSTD_compressBound   (zra.cpp:191,196,316) */
int zo_get_cparams(int level, size_t srcSize, zo_cparams* out); /* ZSTD_getCParams row + adjust; 0 ok, else unsupported */
size_t zo_compress_frame(void* dst, size_t cap, const void* src, size_t n, int level, int checksum); /* ZSTD_compress2 (zra.cpp:219,331) */
size_t zo_decompress(void* dst, size_t cap, const void* src, size_t n);  /* ZSTD_decompressDCtx: 0..n concatenated frames (zra.cpp:249,...) */
size_t zo_find_frame_size(const void* src, size_t n);        /* compressed size of the first frame */
/* match-finder staging (G2): sequences of ONE frame, all blocks concatenated with {ll,0,0} block delimiters */
size_t zo_generate_sequences(zo_seq* out, size_t cap, const void* src, size_t n, int level);

/* --- ZRA container restatement (zra.cpp:88-302) --- */
typedef struct { int zra; int zstd; } zo_status;
size_t zo_zra_output_bound(size_t inputSize, u32 frameSize, u32 metaSize);   /* zra.cpp:189-192 */
zo_status zo_zra_compress_buffer(const void* in, size_t n, void* out, size_t outCap, size_t* outSize,
                                 int level, u32 frameSize, int checksum, size_t metaSize);  /* zra.cpp:194-234 */
zo_status zo_zra_decompress_buffer(const void* in, size_t n, void* out, size_t outCap);      /* zra.cpp:243-250 */
zo_status zo_zra_decompress_ra(const void* in, size_t n, void* out, size_t outCap, size_t offset, size_t size); /* zra.cpp:258-296 */

#ifdef __cplusplus
}
#endif
#endif
