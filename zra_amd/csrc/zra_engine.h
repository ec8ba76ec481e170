// zra_amd — host engine: owns one HIP device, one stream and a grow-only scratch pool, and drives the
// decode / encode kernels. Internal C++ interface used by the C ABI (zra_capi.cpp) and zra_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstddef>
#include <vector>
#include "zra_kernels.h"

namespace zra_eng {

struct Status { int zra; int zstd; };   // mirrors ZraStatus (kept free of the public header here)
enum : int { kSuccess = 0, kZStdError = 1, kVersionLow = 2, kHeaderInvalid = 3, kHeaderIncomplete = 4, kOutOfBounds = 5,
             kOutputTooSmall = 6, kCompressedTooLarge = 7, kFrameSizeMismatch = 8 };
inline Status ok() { return {kSuccess, 0}; }
inline Status zerr(int code) { return {kZStdError, (int)(int8_t)code}; }   // i8 narrowing like zra.cpp:440

// grow-only device allocation
struct DevBuf {
  void* p = nullptr; size_t cap = 0;
  bool reserve(size_t n);
  void release();
  template <typename T> T* as() const { return (T*)p; }
};

// device scratch held by all engines of the process (DevBuf reservations), in bytes
uint64_t scratch_bytes_in_use();

// parsed fixed header (host side; zra.cpp:141-163 semantics)
struct HeaderInfo {
  uint16_t version; uint32_t size; uint64_t uncompressedSize; uint32_t frameSize, metaOffset, metaSize, seekTableOffset, seekTableSize;
  uint32_t frames() const { return seekTableSize / 5 ? seekTableSize / 5 - 1 : 0; }
};
// parse the 38 fixed bytes; returns a ZRA status code (0 ok)
int parse_fixed_header(const uint8_t* fixed38, HeaderInfo* h);
// chunk length of the pipelined host-pointer calls (zra_hostpipe.hip); ZRA_HOST_CHUNK_MIB, default 1024
size_t host_chunk_bytes();

class Engine {
 public:
  static Status create(Engine** out, int device);
  ~Engine();
  hipStream_t stream() const { return stream_; }
  Status sync();
  // stream-ordering contract of the device-pointer API: the engine runs on two private non-blocking streams, which do not order
  // themselves against the caller's streams. Work queued on `producer` before this call completes before anything the engine
  // launches afterwards (an event wait, no host synchronisation). Outputs are complete when an engine call returns.
  Status wait_stream(hipStream_t producer);
  // gives every scratch allocation of this engine back to the device (they are grow-only otherwise and can reach tens of GiB after a
  // large level-9 compression); the next call allocates again what it needs
  Status release_scratch();
  double last_kernel_ms() const { return lastKernelMs_; }
  // launch telemetry of the last persistent dfast compress (ZraEncArgs::mfTele); empty when the call took another path
  // (fetched from the device here, on demand: the two blocking copies used to be paid by every persistent call, the small ones included)
  size_t launch_telemetry(uint64_t* out, size_t cap);
  // after a whole-archive decode_host(): bytes regenerated when frames had to be packed one after the other (a frame regenerated
  // another size than its slot), ~0 when every frame filled exactly its slot
  uint64_t last_produced_total() const { return lastProducedTotal_; }
  // HIP-event timings of the last call on the engine's stream: {mf ms, mf launches, entropy ms, entropy launches, decode ms, decode launches}
  void kernel_stats(double out[6]) const { for (int i = 0; i < 6; i++) out[i] = kstats_[i]; }
  // decode stages of the last call: {parse ms, Huffman ms, sequence-chain ms, execute ms, rounds, one-launch kernel ms, its launches, pipelined first rounds}
  void decode_stage_stats(double out[8]) const { for (int i = 0; i < 8; i++) out[i] = dstats_[i]; }
  // bring-up: sequences {ll | ml<<20 | offVal<<40} the match finder left in scratch context 0 for frame `frame` of the LAST batch
  // (last block of the frame); returns the count, meta = {nbSeq, lastLL, skip}
  uint32_t debug_read_seqs(uint32_t frame, uint64_t* out, uint32_t cap, uint32_t meta[3]);

  // ---- decode
  // Decode nFrames frames described by device job arrays. Synchronises and returns the first failing frame's code.
  // maxFrameBytes: upper bound of what one frame regenerates (sizes the per-pass scratch); ra: optional random-access extras
  // (limit / pieceBase / pieces / raOut of ZraDecodeArgs, indexed like the job arrays)
  Status decode_jobs(const uint8_t* dBody, uint64_t bodySize, const uint64_t* dFrameOff, uint8_t* dOut,
                     const uint64_t* dOutOff, const uint32_t* dExpect, uint32_t nFrames, uint32_t maxFrameBytes, uint32_t offStride = 1,
                     uint64_t seqTotal = 0, const struct ZraDecodeArgs* ra = nullptr);
  // one pass: jobs [0, a.nFrames) of the arrays in `a` through the parse / chain / execute rounds + frame-end checks;
  // *res = min over failing jobs of ((jobBase + job) << 8 | code), untouched when none fails
  Status decode_launch(const struct ZraDecodeArgs& a, const uint32_t* dExpect, uint32_t maxFrameBytes, uint32_t jobBase, unsigned long long* hResult);
  // the same for few jobs: one launch (zra_ra_small_kernel), one synchronisation; *bailed = jobs that need decode_launch after all
  Status decode_small(const struct ZraDecodeArgs& a, const uint32_t* dExpect, uint32_t maxFrameBytes, uint32_t jobBase, unsigned long long* hResult, uint32_t* bailed);
  Status decode_scratch(struct ZraDecodeArgs& a, uint32_t maxFrameBytes);
  // Whole archive resident on the device (header + body), output on the device.
  Status decompress_device(const uint8_t* dArc, size_t arcSize, uint8_t* dOut, size_t outCap);
  // Batched random access, archive + output on the device, query arrays on the host.
  Status decompress_ra_batch(const uint8_t* dArc, size_t arcSize, uint8_t* dOut, const uint64_t* hOff, const uint64_t* hSize,
                             const uint64_t* hOutOff, size_t nq);
  Status decompress_ra_batch_shard(const uint8_t* dArc, size_t arcSize, const uint8_t* dBody, uint64_t bodyBytes, uint64_t bodyBase, uint8_t* dOut,
                                   const uint64_t* hOff, const uint64_t* hSize, const uint64_t* hOutOff, size_t nq);
  // Host-walked frame list (reference semantics of DecompressBuffer: seek table not consulted). hFrameOff has nFrames+1 entries
  // relative to dBody; frames are assumed to regenerate frameSize bytes each (last: the remainder of total).
  Status decompress_frames_host_list(const uint8_t* dBody, uint64_t bodySize, const std::vector<uint64_t>& hFrameOff,
                                     uint8_t* dOut, uint64_t total, uint32_t frameSize);

  // ---- encode (zra_encode.hip)
  // Compress frames of `frameSize` from dIn (inSize bytes) into a packed body at dBody; per-frame sizes (u64) to dSizes.
  Status compress_frames(const uint8_t* dIn, size_t inSize, uint8_t* dBody, uint64_t* dSizes, size_t* bodySize,
                         int level, uint32_t frameSize, bool checksum);
  Status compress_persistent(const uint8_t* dIn, size_t inSize, uint8_t* dBody, uint64_t bodyBase0, uint8_t* dEntries, uint64_t* dSizes,
                             size_t* bodySize, uint32_t frameSize, bool checksum, const struct ZraEncParams& full, const struct ZraEncParams& tail,
                             uint64_t tableWords, uint64_t seqStride, uint64_t litStride, uint64_t slotStride);
  Status compress_impl(const uint8_t* dIn, size_t inSize, uint8_t* dBody, uint64_t bodyBase0, uint8_t* dEntries, uint64_t* dSizes,
                       size_t* bodySize, int level, uint32_t frameSize, bool checksum);
  Status compress_impl_body(const uint8_t* dIn, size_t inSize, uint8_t* dBody, uint64_t bodyBase0, uint8_t* dEntries, uint64_t* dSizes,
                            size_t* bodySize, int level, uint32_t frameSize, bool checksum);
  void drain_after_error();
  // Full archive (header + table + body) on the device.
  Status compress_device(const uint8_t* dIn, size_t inSize, uint8_t* dOut, size_t* outSize, int level, uint32_t frameSize, bool checksum);

  // ---- host-pointer helpers (H2D -> kernels -> D2H) behind the reference-compatible C/C++ API
  Status compress_host(const uint8_t* hIn, size_t n, uint8_t* hOut, size_t* outSize, int level, uint32_t frameSize, bool checksum);
  Status compress_frames_host(const uint8_t* hIn, size_t n, uint8_t* hBody, std::vector<uint64_t>& sizes, size_t* bodySize,
                              int level, uint32_t frameSize, bool checksum);
  // frames given as (start,end) pairs inside hSpan; frame i regenerates min(frameSize, total - i*frameSize) bytes;
  // bytes [skip, skip+size) of the concatenated output are returned in hOut
  Status decode_host(const uint8_t* hSpan, size_t spanSize, const std::vector<uint64_t>& starts, const std::vector<uint64_t>& ends,
                     uint32_t frameSize, uint64_t total, uint8_t* hOut, size_t skip, size_t size, bool wholeArchive = false);
  // whole archive in ~1 GiB chunks, copies beside the kernels (zra_hostpipe.hip); *fallBack: a frame failed, take decode_host
  Status decode_host_pipelined(const uint8_t* hSpan, const std::vector<uint64_t>& starts, const std::vector<uint64_t>& ends,
                               uint32_t frameSize, uint64_t total, uint8_t* hOut, bool* fallBack);

  // batched random access: false (default) = a frame is decoded up to the last byte a query needs, so damage behind that byte
  // and the frame's content checksum go unnoticed; true = whole frames + checksums, the reference's error behaviour
  void set_ra_verify_whole_frames(bool on) { raVerifyWholeFrames_ = on; }
  int device() const { return device_; }
  int num_cus() const { return numCUs_; }

 private:
  Engine() = default;
  int device_ = 0, numCUs_ = 0;
  hipStream_t stream_ = nullptr;
  hipEvent_t ev0_ = nullptr, ev1_ = nullptr, evWait_ = nullptr;
  double lastKernelMs_ = 0;
  std::vector<uint64_t> mfTele_;
  const uint64_t* mfTeleDev_ = nullptr;   // the last persistent launch's telemetry block on the device (inside encScan_), not fetched yet; nullptr: none / fetched
  double kstats_[6] = {0, 0, 0, 0, 0, 0};
  double dstats_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  std::vector<hipEvent_t> stageEv_; size_t stageEvNext_ = 0;
  hipEvent_t stage_event();
  hipEvent_t evR_[17] = {nullptr};   // per-round events of one encode batch: e[2r] before mf, e[2r+1] between, e[2r+2] after entropy
  // decode scratch
  DevBuf decFrames_, decTables_, decLists_, decCounters_, decLits_, decSeqs_, roundN_;
  DevBuf decBlkRecs_, decBlkTables_, decBlkLists_;   // block-parallel pass (frames of several blocks): per-block records, tables, job lists
  DevBuf raPlan_, raLimit_, raPieceBase_, raPieces_;
  uint8_t* pinSmall_ = nullptr;             // page-locked staging of small host-pointer decodes (decode_host: job arrays + compressed span in, answer out)
  uint64_t* pinQ_ = nullptr; size_t pinQCap_ = 0;   // page-locked query tuples of the running batch (host side of an asynchronous copy)
  int decOccParse_ = 0, decOccExec_ = 0, decOccHuf_ = 0; // resident workgroups per CU of the parse / execute kernels
  bool raVerifyWholeFrames_ = false;     // batched random access decodes every touched frame in full and checks its checksum
  DevBuf status_, produced_, frameMeta_, frameOff_, outOff_, expect_, result_, temp_, qmeta_;
  // encode scratch (see zra_encode.hip)
  struct EncCtx { DevBuf tables, seqs, lits, work, slots, misc, ck, sizes, rec; };   // rec: the split entropy stage's per-frame records (ZraEntRec)
  EncCtx encCtx_[2];
  DevBuf encScan_;
  DevBuf mfFlags_;                         // bucket-flag masks of the dfast match finder: one slot per resident wave (df_later_flags)
  int lsAttr_ = 0;                  // zra_mf_dfast_ls_kernel's dynamic LDS limit raised: 1 yes, -1 refused
  hipStream_t stream2_ = nullptr;          // entropy stage / gather stream (overlaps the match finder on stream_)
  bool decCountersClean_ = false;          // the decoder's round counters are known to be zero (zeroed behind the last one-launch decode)
  int chainLdsAttr_ = 0;                   // zra_dec_chain_lds_kernel's dynamic LDS size: 0 not asked yet, 1 granted, -1 refused
  hipStream_t pipeStreams_[3] = {nullptr, nullptr, nullptr};   // decode stage pipeline: Huffman, chain, execute (parse runs on stream_)
  std::vector<hipEvent_t> evPool_;
  DevBuf hostIn_, hostOut_, seqScratch_;
  uint64_t dbgSeqStride_ = 0; uint32_t dbgB_ = 0;
  uint64_t lastProducedTotal_ = ~0ull;     // whole-archive decode that fell back to the sequential tail: bytes actually regenerated
  void* encCounters_ = nullptr; size_t encCountersBytes_ = 0;   // sub-batch counters stream B may be waiting on (drain_after_error)
  int waitValueOk_ = 0;                    // 0 unknown, 1 hipStreamWaitValue32 works on device memory, -1 it does not (batch path)
  friend struct EncodeImpl;
};

}  // namespace zra_eng
