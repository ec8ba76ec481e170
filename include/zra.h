/* zra_amd — C ABI of the MI355X-native ZRA engine (drop-in for the reference's libzra).
 *
 * Every declaration below binds to the same symbol name, argument order and ZraStatus convention as the
 * reference's C interface, so a program built against the reference header links against libzra_amd.so
 * unchanged. Each entry cites the reference interface it replaces (file:line in zraorg/ZRA).
 * Behaviour notes that differ from "what one would expect" are the reference's own quirks, kept on purpose
 * (see DESIGN.md §boundary): RA bound is ">=", in-memory CompressBuffer ignores the metadata payload,
 * the header CRC is written but never verified.
 *
 * The per-frame zstd work behind these calls runs as hand-written HIP kernels on gfx950; there is no CPU
 * codec in this library. If no GPU is usable the compute entry points return {ZStdError, 1 (GENERIC)}.
 */
#ifndef ZRA_AMD_ZRA_H
#define ZRA_AMD_ZRA_H

#ifndef ZRA_EXPORT
#if defined(_WIN32)
#define ZRA_EXPORT __declspec(dllimport)
#else
#define ZRA_EXPORT __attribute__((visibility("default")))
#endif
#endif

#ifdef __cplusplus
#include <cstddef>
#include <cstdint>
extern "C" {
#else
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#endif

/* Status codes, same numeric order as the reference enum (zra.h:31-41 == zra.hpp:58-68). */
typedef enum ZraStatusCode {
  Success = 0,
  ZStdError = 1,              /* codec failure; ZraStatus.zstd carries the zstd error code */
  ZraVersionLow = 2,
  HeaderInvalid = 3,
  HeaderIncomplete = 4,
  OutOfBoundsAccess = 5,
  OutputBufferTooSmall = 6,
  CompressedSizeTooLarge = 7, /* body >= 2^40 bytes */
  InputFrameSizeMismatch = 8
} ZraStatusCode;

/* Returned by value, 8 bytes (zra.h:46-49). */
typedef struct ZraStatus {
  ZraStatusCode zra;
  int zstd;
} ZraStatus;

/* read callback used by the streaming objects: fill `buffer` with `size` bytes found at `offset` of the archive.
 * No user-data argument, exactly like the reference (zra.h:72,215,244). */
typedef void(ZraReadFunction)(size_t offset, size_t size, void* buffer);

/* ---- library ------------------------------------------------------------------------------------ */
ZRA_EXPORT uint16_t ZraGetVersion(void);                          /* zra.h:56  / zra.cpp:448 -> 1 */
ZRA_EXPORT const char* ZraGetErrorString(ZraStatus status);       /* zra.h:62  / zra.cpp:452 */

/* ---- header object (zra.h:65-117, zra.cpp:456-505) ----------------------------------------------- */
typedef struct ZraHeader ZraHeader;
ZRA_EXPORT ZraStatus ZraCreateHeader(ZraHeader** header, ZraReadFunction* readFunction);
ZRA_EXPORT ZraStatus ZraCreateHeader2(ZraHeader** header, void* buffer, size_t size);
ZRA_EXPORT void ZraDeleteHeader(ZraHeader* header);
ZRA_EXPORT size_t ZraGetVersionWithHeader(ZraHeader* header);
ZRA_EXPORT size_t ZraGetHeaderSizeWithHeader(ZraHeader* header);
ZRA_EXPORT size_t ZraGetUncompressedSizeWithHeader(ZraHeader* header);
ZRA_EXPORT size_t ZraGetFrameSizeWithHeader(ZraHeader* header);
ZRA_EXPORT size_t ZraGetMetadataSize(ZraHeader* header);
ZRA_EXPORT void ZraGetMetadata(ZraHeader* header, void* buffer);

/* ---- in-memory calls (zra.h:119-156, zra.cpp:507-533) — the hot path ------------------------------ */
ZRA_EXPORT size_t ZraGetCompressedOutputBufferSize(size_t inputSize, size_t frameSize);
ZRA_EXPORT ZraStatus ZraCompressBuffer(void* inputBuffer, size_t inputSize, void* outputBuffer, size_t* outputSize,
                                       int8_t compressionLevel, uint32_t frameSize, bool checksum, void* metaBuffer, size_t metaSize);
ZRA_EXPORT ZraStatus ZraDecompressBuffer(void* inputBuffer, size_t inputSize, void* outputBuffer);
ZRA_EXPORT ZraStatus ZraDecompressRA(void* inputBuffer, size_t inputSize, void* outputBuffer, size_t offset, size_t size);

/* ---- streaming compressor (zra.h:158-206, zra.cpp:535-574) ---------------------------------------- */
typedef struct ZraCompressor ZraCompressor;
ZRA_EXPORT ZraStatus ZraCreateCompressor(ZraCompressor** compressor, size_t size, int8_t compressionLevel, uint32_t frameSize,
                                         bool checksum, void* metaBuffer, size_t metaSize);
ZRA_EXPORT void ZraDeleteCompressor(ZraCompressor* compressor);
ZRA_EXPORT size_t ZraGetOutputBufferSizeWithCompressor(ZraCompressor* compressor, size_t inputSize);
ZRA_EXPORT ZraStatus ZraCompressWithCompressor(ZraCompressor* compressor, void* inputBuffer, size_t inputSize, void* outputBuffer, size_t* outputSize);
ZRA_EXPORT size_t ZraGetHeaderSizeWithCompressor(ZraCompressor* compressor);
ZRA_EXPORT ZraStatus ZraGetHeaderWithCompressor(ZraCompressor* compressor, void* outputBuffer);

/* ---- streaming random-access decompressor (zra.h:208-234, zra.cpp:576-600) ------------------------- */
typedef struct ZraDecompressor ZraDecompressor;
ZRA_EXPORT ZraStatus ZraCreateDecompressor(ZraDecompressor** decompressor, ZraReadFunction* readFunction, size_t maxCacheSize);
ZRA_EXPORT void ZraDeleteDecompressor(ZraDecompressor* decompressor);
ZRA_EXPORT ZraHeader* ZraGetHeaderWithDecompressor(ZraDecompressor* decompressor);   /* borrowed pointer */
ZRA_EXPORT ZraStatus ZraDecompressWithDecompressor(ZraDecompressor* decompressor, size_t offset, size_t size, void* outputBuffer);

/* ---- streaming full decompressor (zra.h:236-263, zra.cpp:602-626) ---------------------------------- */
typedef struct ZraFullDecompressor ZraFullDecompressor;
ZRA_EXPORT ZraStatus ZraCreateFullDecompressor(ZraFullDecompressor** decompressor, ZraReadFunction* readFunction, size_t maxCacheSize);
ZRA_EXPORT void ZraDeleteFullDecompressor(ZraFullDecompressor* decompressor);
ZRA_EXPORT ZraHeader* ZraGetHeaderWithFullDecompressor(ZraFullDecompressor* decompressor);   /* borrowed pointer */
ZRA_EXPORT ZraStatus ZraDecompressWithFullDecompressor(ZraFullDecompressor* decompressor, void* outputBuffer, size_t outputCapacity, size_t* outputSize);

#ifdef __cplusplus
}
#endif
#endif /* ZRA_AMD_ZRA_H */
