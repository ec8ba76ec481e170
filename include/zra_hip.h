/* zra_amd — additive device-side C ABI of the MI355X-native ZRA engine.
 *
 * NOT part of the reference. The 29 reference entry points (include/zra.h) keep their host-pointer
 * semantics; one kernel launch per call cannot serve BASELINE config C3 (1M random-access queries), and
 * callers that already hold data in HBM should not bounce it through the host. These calls take DEVICE
 * pointers (hipMalloc / torch.cuda tensors: plain addresses, no torch types) and are what bench.py times.
 *
 * Reference interfaces they accelerate:
 *   ZraHipCompressBuffer      <- zra::CompressBuffer    (zra.cpp:194-234, zra.h:138)
 *   ZraHipDecompressBuffer    <- zra::DecompressBuffer  (zra.cpp:243-250, zra.h:146)
 *   ZraHipDecompressRABatch   <- zra::DecompressRA      (zra.cpp:258-296, zra.h:156), batched
 *   ZraHipCompressFrames / ZraHipStitch <- the per-frame loop zra.cpp:216-225 split for multi-GPU sharding
 */
#ifndef ZRA_HIP_H
#define ZRA_HIP_H
#include "zra.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ZraHipEngine ZraHipEngine;

/** Number of visible HIP devices (0 when there is no GPU / no driver). Never throws. */
ZRA_EXPORT int ZraHipDeviceCount(void);

/** Creates an engine bound to `device` (its own stream + scratch pool). Fails with ZStdError/GENERIC(1) when no GPU is usable. */
ZRA_EXPORT ZraStatus ZraHipCreateEngine(ZraHipEngine** engine, int device);
ZRA_EXPORT void ZraHipDestroyEngine(ZraHipEngine* engine);
/** Blocks until all work queued on the engine's stream is complete. */
ZRA_EXPORT ZraStatus ZraHipSynchronize(ZraHipEngine* engine);
/** STREAM ORDERING. The engine works on two private non-blocking HIP streams; they are not ordered against the caller's
 *  streams (nor against the null stream). Inputs (dIn / the archive / device-resident query arrays) must be complete before a
 *  ZraHip* compute call starts reading them: either synchronise the producing stream, or call ZraHipWaitStream(engine, stream)
 *  first — it records an event on `producerStream` (a hipStream_t; NULL = the null stream) and makes the engine's streams wait
 *  for it, with no host synchronisation. Every ZraHip* compute call is host-synchronous on return: its outputs are complete and
 *  visible to any stream. */
ZRA_EXPORT ZraStatus ZraHipWaitStream(ZraHipEngine* engine, void* producerStream);
/** Returns the engine's scratch allocations to the device. Scratch is grow-only between calls (a level-9 compression of a large
 *  buffer keeps tens of GiB for its hash-chain tables); a long-running host that is done with such a phase can hand it back. */
ZRA_EXPORT ZraStatus ZraHipReleaseScratch(ZraHipEngine* engine);
/** The engine's hipStream_t, as an opaque pointer (for event timing on the stream kernels run on). */
ZRA_EXPORT void* ZraHipGetStream(ZraHipEngine* engine);

/** CompressBuffer with device-resident input/output. dOut must hold ZraGetCompressedOutputBufferSize(inSize, frameSize) bytes.
 *  Output is byte-identical to the reference at the same level (zstd 1.4.9 semantics). Synchronous. */
ZRA_EXPORT ZraStatus ZraHipCompressBuffer(ZraHipEngine* engine, const void* dIn, size_t inSize, void* dOut, size_t* outSize,
                                          int8_t compressionLevel, uint32_t frameSize, bool checksum);

/** DecompressBuffer with a device-resident archive; dOut must hold the archive's uncompressedSize bytes. Synchronous.
 *  Frame boundaries come from the archive's own seek table (the body stays on the device, nothing walks it on the host): on a valid
 *  archive the result is the reference's; on a damaged one the frames' own errors are reported like libzstd's, but a seek table
 *  that does not match the frames is an error here (srcSize_wrong / corruption_detected) where zra::DecompressBuffer — one
 *  multi-frame zstd call that never looks at the table, zra.cpp:249 — may still succeed. The host-pointer ZraDecompressBuffer /
 *  ZraDecompressRA of include/zra.h reproduce the reference on such archives too. The same holds for ZraHipDecompressRABatch. */
ZRA_EXPORT ZraStatus ZraHipDecompressBuffer(ZraHipEngine* engine, const void* dIn, size_t inSize, void* dOut, size_t outCapacity);

/** Batched DecompressRA: query i returns bytes [hOffsets[i], hOffsets[i]+hSizes[i]) of the original data at dOut + hOutOffsets[i].
 *  hOffsets/hSizes/hOutOffsets are HOST arrays of nQueries entries. Bounds rule per query is the reference's
 *  (offset+size >= uncompressedSize -> OutOfBoundsAccess, zra.cpp:260). Synchronous.
 *  The frames a batch touches are found and scheduled on the device from the archive's own seek table; every touched frame is
 *  decoded once, and only as far as the last byte any query needs from it (the reference decodes whole frames, zra.cpp:279-295).
 *  For a valid archive the bytes are identical. What differs is damage detection: bytes of a frame behind the last one needed, and
 *  the frame's content checksum, are not looked at. ZRA_HIP_OPT_RA_WHOLE_FRAMES restores whole-frame decode + checksum. */
ZRA_EXPORT ZraStatus ZraHipDecompressRABatch(ZraHipEngine* engine, const void* dIn, size_t inSize, void* dOut,
                                             const uint64_t* hOffsets, const uint64_t* hSizes, const uint64_t* hOutOffsets, size_t nQueries);

/* ---- sharded compression (one process per GPU; frames [firstFrame, firstFrame+nFrames) of a larger input) ---- */
/** Compresses nFrames frames of frameSize bytes (last may be shorter: inSize bytes total) from dIn into a packed body at dBody
 *  (capacity nFrames*ZSTD_compressBound(frameSize)); writes the nFrames local frame sizes (u64, device) to dSizes and the
 *  local body size to *bodySize. No header is produced. */
ZRA_EXPORT ZraStatus ZraHipCompressFrames(ZraHipEngine* engine, const void* dIn, size_t inSize, void* dBody, uint64_t* dSizes,
                                          size_t* bodySize, int8_t compressionLevel, uint32_t frameSize, bool checksum);

/** Builds the complete ZRA header (38 bytes + 5*tableSize) on the HOST from all frame sizes (rank-ordered concatenation),
 *  including the CRC-32 (zra.cpp:128-133). hHeader must hold 38 + 5*(nFramesTotal+1) bytes. */
ZRA_EXPORT ZraStatus ZraHipStitchHeader(const uint64_t* hFrameSizes, size_t nFramesTotal, uint64_t uncompressedSize,
                                        uint32_t frameSize, void* hHeader, size_t* headerSize);

/* ---- distributed archive: one process per GPU, frames sharded by index (zra_amd/csrc/zra_comm.hip) ----
 * The reference has no multi-device mode; its frame loop (zra.cpp:216-225) and its lookup (zra.cpp:265-269) are what is split here.
 * Rank r of `world` owns frames [F*r/world, F*(r+1)/world). All ZraHipComm* calls are COLLECTIVE: every rank of the communicator
 * makes the same call, and every rank returns the same status (the first failing rank's). */
typedef struct ZraHipComm ZraHipComm;
typedef struct ZraHipShard ZraHipShard;

/** Frames [*lo, *hi) owned by `rank`. Pure arithmetic. */
ZRA_EXPORT void ZraHipShardRange(uint64_t nFrames, int rank, int world, uint64_t* lo, uint64_t* hi);
/** The rank that owns `frame`. */
ZRA_EXPORT int ZraHipOwnerOfFrame(uint64_t nFrames, int world, uint64_t frame);

/** One piece of a query after it has been cut at ownership boundaries. */
typedef struct ZraHipSlice {
  uint32_t owner;      /* rank that holds the frames of this piece */
  uint64_t query;      /* index of the query it belongs to */
  uint64_t offset;     /* first uncompressed byte (offset into the whole archive's content) */
  uint64_t size;       /* bytes */
  uint64_t within;     /* where the piece starts inside the query's answer */
} ZraHipSlice;
/** Query router: cuts (offset, size) queries at ownership boundaries; slices are written grouped by owner, query order inside an
 *  owner, and counted per owner in perOwnerCount[world] (may be NULL). Bounds as DecompressRA: any query with
 *  offset + size >= uncompressedSize fails the call with OutOfBoundsAccess (zra.cpp:260). OutputBufferTooSmall: *nSlices tells the
 *  capacity needed. Pure host arithmetic — no engine, no GPU. */
ZRA_EXPORT ZraStatus ZraHipRouteQueries(uint64_t uncompressedSize, uint32_t frameSize, int world, const uint64_t* offsets, const uint64_t* sizes,
                                        size_t nQueries, ZraHipSlice* slices, size_t sliceCapacity, size_t* nSlices, uint64_t* perOwnerCount);

/** RCCL transport (ncclAllGather, grouped ncclSend/ncclRecv over xGMI). Rank 0 creates the 128-byte
 *  id and hands it to the other ranks by the host program's own means. */
ZRA_EXPORT ZraStatus ZraHipCommGetUniqueId(void* id128);
ZRA_EXPORT ZraStatus ZraHipCommCreateRccl(ZraHipComm** comm, ZraHipEngine* engine, const void* id128, int rank, int world);
/** Host transport: the exchange steps are handed to two callbacks of the host program (MPI, sockets, ...) on HOST buffers; device
 *  data is staged. Return 0 for success.
 *    allgather: every rank contributes `bytes` bytes, recv gets world*bytes in rank order.
 *    exchange:  one point-to-point round — all sends and receives of this rank; returns when all of them are complete.
 *  engine may be NULL for a communicator that only stitches (ZraHipCommStitchSizes). */
typedef struct ZraHipHostTransport {
  void* user;
  int (*allgather)(void* user, const void* send, void* recv, size_t bytes);
  int (*exchange)(void* user, int nSend, const int* sendPeer, const void* const* sendBuf, const size_t* sendBytes,
                  int nRecv, const int* recvPeer, void* const* recvBuf, const size_t* recvBytes);
} ZraHipHostTransport;
ZRA_EXPORT ZraStatus ZraHipCommCreateHost(ZraHipComm** comm, ZraHipEngine* engine, const ZraHipHostTransport* transport, int rank, int world);
/** Diagnostic (RCCL transport): bytes from dSrc to dDst through the point-to-point path, this rank sending to itself — what a one-rank
 *  run can exercise of the transport's message chunking (pieces of at most 1 GiB per ncclSend / ncclRecv; ZRA_COMM_CHUNK_MIB). */
ZRA_EXPORT ZraStatus ZraHipCommLoopback(ZraHipComm* comm, const void* dSrc, void* dDst, size_t bytes);
ZRA_EXPORT void ZraHipCommDestroy(ZraHipComm* comm);

/** Sharded CompressBuffer (zra.cpp:194-235): dLocal = the uncompressed bytes of this rank's frames, i.e. bytes
 *  [lo*frameSize, min(totalBytes, hi*frameSize)) of the input with [lo, hi) = ZraHipShardRange(ceil(totalBytes/frameSize), rank, world)
 *  (InputFrameSizeMismatch if localBytes is not exactly that). Only the frame sizes travel (8 bytes per frame, all-gather): every rank
 *  ends up with the complete header + seek table and the compressed body of its own frames — a shard. Byte-identical to the archive
 *  one GPU would write. */
ZRA_EXPORT ZraStatus ZraHipCommCompress(ZraHipComm* comm, const void* dLocal, size_t localBytes, uint64_t totalBytes, int8_t compressionLevel,
                                        uint32_t frameSize, bool checksum, ZraHipShard** shard);
/** The size exchange and stitch alone (zra.cpp:216-230), for frames that were compressed elsewhere: hLocalSizes = the compressed sizes
 *  of this rank's frames [lo, hi) (nLocal = hi - lo, InputFrameSizeMismatch otherwise). Collective; every rank gets the complete
 *  header + seek table (ZraHipShardGetHeader), the shard holds no body (ZraHipShardGetBody: NULL, 0 bytes). Works on a communicator
 *  created without an engine. Engine-less communicators and stitch-only shards must be that on EVERY rank: ZraHipCommCompress /
 *  GatherArchive / Serve return parameter_unsupported on them after one agreement round, and ranks that hold an engine or a body would
 *  go on to larger exchanges. More than 2^32 - 2 frames, or hLocalSizes == NULL with nLocal > 0, are rejected on every rank. */
ZRA_EXPORT ZraStatus ZraHipCommStitchSizes(ZraHipComm* comm, const uint64_t* hLocalSizes, size_t nLocal, uint64_t totalBytes, uint32_t frameSize,
                                           ZraHipShard** shard);
ZRA_EXPORT void ZraHipShardDestroy(ZraHipShard* shard);
ZRA_EXPORT size_t ZraHipShardHeaderSize(const ZraHipShard* shard);
/** Header + seek table of the WHOLE archive (host copy, CRC-32 set). */
ZRA_EXPORT void ZraHipShardGetHeader(const ZraHipShard* shard, void* hHeader);
/** Size of the whole archive (header + all ranks' frames). */
ZRA_EXPORT uint64_t ZraHipShardArchiveSize(const ZraHipShard* shard);
/** This rank's frames: device pointer, their offset inside the archive's body, their length (NULL and 0 for a stitch-only shard). */
ZRA_EXPORT void ZraHipShardGetBody(const ZraHipShard* shard, const void** dBody, uint64_t* bodyBase, uint64_t* bodyBytes);
/** The archive in one piece in dArchive on rank `root` (other ranks pass NULL / 0): world-1 inbound point-to-point messages in one group. */
ZRA_EXPORT ZraStatus ZraHipCommGatherArchive(ZraHipComm* comm, const ZraHipShard* shard, int root, void* dArchive, size_t archiveCapacity,
                                             size_t* archiveSize);
/** The gather beside other work (round 6): a SECOND communicator of the process gets a stream of its own (UseOwnStream); Begin starts
 *  ZraHipCommGatherArchive on it (a worker thread drives the collective; every rank calls Begin and End) and returns at once, so that the
 *  ranks can serve queries from their shards on the first communicator meanwhile (ZraHipCommServe works on the shards, not on the gathered
 *  archive); End joins and returns the gather's status and, on the root, the archive's size. */
ZRA_EXPORT ZraStatus ZraHipCommUseOwnStream(ZraHipComm* comm);
ZRA_EXPORT ZraStatus ZraHipCommGatherArchiveBegin(ZraHipComm* comm, const ZraHipShard* shard, int root, void* dArchive, size_t archiveCapacity);
ZRA_EXPORT ZraStatus ZraHipCommGatherArchiveEnd(ZraHipComm* comm, size_t* archiveSize);
/** Sharded serving (BASELINE config C5): every rank passes its own queries over the WHOLE uncompressed range; slices go to the ranks
 *  that own the frames, are decoded there (ZraHipDecompressRABatch semantics on the shard) and come back; answer q lands at
 *  dOut + hOutOffsets[q]. Bounds as DecompressRA (zra.cpp:260). */
ZRA_EXPORT ZraStatus ZraHipCommServe(ZraHipComm* comm, const ZraHipShard* shard, const uint64_t* hOffsets, const uint64_t* hSizes,
                                     const uint64_t* hOutOffsets, size_t nQueries, void* dOut);

/** Last per-call timing of the dominant kernel on the engine's stream, measured with HIP events (milliseconds; 0 if none). */
ZRA_EXPORT double ZraHipLastKernelMs(ZraHipEngine* engine);
/** HIP-event timings (ms) and launch counts of the kernels of the LAST call on this engine, measured on the engine's stream:
 *  out6 = {match-finder ms, launches, entropy-stage ms, launches, decode ms, launches}. */
ZRA_EXPORT void ZraHipGetKernelStats(ZraHipEngine* engine, double* out6);
/** Decode stages of the LAST decode / random-access call on this engine (HIP events on the engine's stream, summed over its rounds):
 *  out8 = {parse ms, Huffman ms, sequence-chain ms, execute ms, rounds, one-launch small-batch kernel ms, its launches, 0}. */
ZRA_EXPORT void ZraHipGetDecodeStageStats(ZraHipEngine* engine, double* out8);
/** Launch telemetry of the LAST persistent level-3/4 compress on this engine, written by every wave of the match-finder launch:
 *  u64 words [0] shader cycles summed over waves, [1] ticks of the constant 100 MHz clock summed over waves (cycles / ticks x 100 = the
 *  effective shader MHz of the launch), [2] waves, [3] longest wave (ticks), [4] ~earliest wave start, [5] latest wave end, [6] latest
 *  wave start, [7] ~earliest wave end, [8..15] waves per XCD, [16..23] frames taken per XCD, [24..31] ticks per XCD, [32 + k] waves on
 *  compute unit k = xcc << 8 | se << 5 | sh << 4 | cu (2048 words, each four 16-bit counts: the unit's four SIMDs); then the entropy stage's persistent workgroups: [2080] workgroups
 *  that took a frame, [2081] their resident ticks, [2082] ticks they spent waiting for a frame or a slot, [2083] frames, [2088 + k]
 *  workgroups on compute unit k. Returns the words written (0: the call took another path). */
ZRA_EXPORT size_t ZraHipGetLaunchTelemetry(ZraHipEngine* engine, uint64_t* out, size_t capWords);

/* ---- opt-in integrity options (default 0: bit- and error-compatible with the reference, quirks included) ---- */
#define ZRA_HIP_OPT_VERIFY_HEADER_CRC 1u     /* Header constructors check the CRC-32 the reference writes but never reads (zra.cpp:128-133) */
#define ZRA_HIP_OPT_INCLUSIVE_RA_BOUND 2u    /* DecompressRA accepts offset+size == uncompressedSize (reference: '>=', zra.cpp:260) */
#define ZRA_HIP_OPT_STORE_META_IN_MEMORY 4u  /* in-memory CompressBuffer stores `meta` like the streaming Compressor (reference: zra.cpp:202-205) */
#define ZRA_HIP_OPT_RA_WHOLE_FRAMES 8u      /* ZraHipDecompressRABatch decodes every touched frame in full and verifies its checksum */
/** Process-wide; affects the zra.h / zra.hpp entry points (and, for the last flag, ZraHipDecompressRABatch). */
ZRA_EXPORT void ZraHipSetOptions(uint32_t mask);
ZRA_EXPORT uint32_t ZraHipGetOptions(void);

/** Bring-up aid (not a product entry point): the match finder's sequences {litLength | matchLength<<20 | offsetValue<<40} left in
 *  scratch for frame `frame` of the last compress call's last batch (last block of the frame); meta3 = {nbSeq, lastLL, skip}. */
ZRA_EXPORT uint32_t ZraHipDebugReadSeqs(ZraHipEngine* engine, uint32_t frame, uint64_t* out, uint32_t cap, uint32_t* meta3);

#ifdef __cplusplus
}
#endif
#endif
