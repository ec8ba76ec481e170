// zra_amd — host side of the ENCODE path: zstd 1.4.9 parameter selection, scratch layout in HBM, the
// block-round driver (A.4.2) over batches of frames, and the seek-table build (device scan + gather).
// Replaces the loop body of the reference's CompressBuffer / Compressor::Compress (zra.cpp:216-225, 329-338).
#include <cstdio>
#include <unistd.h>
#include "zra_engine.h"
#include "zra_dev.h"
#include "zra_format.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" __global__ void zra_mf_kernel(ZraEncArgs a, uint32_t block, uint32_t only, uint32_t onlySlot, uint32_t perWave);
extern "C" __global__ void zra_mf_opt_kernel(ZraEncArgs a, uint32_t block, uint32_t only, uint32_t onlySlot, uint32_t perWave);
extern "C" __global__ void zra_mf_dfast_ls_kernel(ZraEncArgs a, uint32_t block, uint32_t only, uint32_t onlySlot);
extern "C" __global__ void zra_mf_hc_kernel(ZraEncArgs a, uint32_t block, uint32_t only, uint32_t onlySlot);
extern "C" __global__ void zra_mf_fast_kernel(ZraEncArgs a, uint32_t block, uint32_t perWave);
extern "C" __global__ void zra_mf_dfast_kernel(ZraEncArgs a, uint32_t block, uint32_t only, uint32_t onlySlot);
extern "C" __global__ void zra_mf_dfast_fl_kernel(ZraEncArgs a, ZraFlagArgs g, uint32_t block, uint32_t only, uint32_t onlySlot);
extern "C" __global__ void zra_entropy_kernel(ZraEncArgs a, uint32_t block);
extern "C" __global__ void zra_entropy_front_kernel(ZraEncArgs a, uint32_t block);
extern "C" __global__ void zra_entropy_back_kernel(ZraEncArgs a, uint32_t block);
extern "C" __global__ void zra_ent_chain_kernel(ZraEncArgs a);
extern "C" __global__ void zra_ent_chain5_kernel(ZraEncArgs a);
extern "C" __global__ void zra_ent_chain3_kernel(ZraEncArgs a);
extern "C" __global__ void zra_ent_chain2_kernel(ZraEncArgs a);
extern "C" __global__ void zra_ent_chain1_kernel(ZraEncArgs a);

using namespace zra_dev;

namespace {

// ---- content checksum of every frame of a batch: 4 lanes per frame
__global__ void zra_content_ck_kernel(const u8* in, u64 inSize, u32 frameSize, u32 firstFrame, u32 nFrames, u32* ck) {
  const u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
  const u32 f = gid >> 2; const int j = gid & 3;
  const bool active = f < nFrames;
  const u64 start = active ? (u64)(firstFrame + f) * frameSize : 0;
  const u32 n = active ? (u32)min((u64)frameSize, inSize - start) : 0;
  const u64 h = zra_xxh64_quad(in + start, n, j);
  if (active && j == 0) ck[f] = (u32)h;
}

// ---- exclusive scan of the batch's frame sizes: ONE wave (a 1024-thread workgroup cannot be placed while the persistent match
//      finder holds wave slots on every SIMD; a single wave fits anywhere). The running body offset lives on the device
//      (`running`), so batches chain without a host round trip: offsets[i] = *running + local prefix; *running += total.
__global__ void __launch_bounds__(64) zra_scan_sizes_kernel(const u64* sizes, u32 n, u64* offsets, u64* running, const u32* abortFlag = nullptr) {
  const int lane = threadIdx.x;
  if (abortFlag && *abortFlag) return;                 // (persistent pipeline gave up: the sizes of this sub-batch are not all written)
  u64 carry = *running;
  for (u32 base = 0; base < n; base += 64) {
    const u32 i = base + lane;
    const u64 v = i < n ? sizes[i] : 0;
    u64 inc = v;
    for (int d = 1; d < 64; d <<= 1) { const u64 t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
    if (i < n) offsets[i] = carry + inc - v;
    carry += __shfl(inc, 63, 64);
  }
  if (lane == 0) *running = carry;
}

// ---- gather: frame f of the batch moves from its slot to body + bodyBase + offsets[f] (one workgroup per frame);
//      optionally writes the 5-byte seek-table entry and/or the u64 size
__global__ void zra_gather_frames_kernel(const u8* slots, u64 slotStride, const u64* sizes, const u64* offsets, u8* body, u64 bodyBase,
                                         u8* entries, u32 firstFrame, u64* sizesOut, const u32* abortFlag = nullptr) {
  const u32 f = blockIdx.x;
  if (abortFlag && *abortFlag) return;
  const u64 n = sizes[f], off = offsets[f];   // offsets are absolute within the body (running offset folded in by the scan)
  (void)bodyBase;
  const u8* s = slots + (size_t)f * slotStride; u8* d = body + off;
  // slots are 16-byte aligned; destination is arbitrary: align on the source, let the stores be unaligned
  const u64 n16 = n >> 4;
  for (u64 i = threadIdx.x; i < n16; i += blockDim.x) {
    const uint4 v = ((const uint4*)s)[i];
    st64(d + 16 * i, (u64)v.x | ((u64)v.y << 32)); st64(d + 16 * i + 8, (u64)v.z | ((u64)v.w << 32));
  }
  for (u64 i = (n16 << 4) + threadIdx.x; i < n; i += blockDim.x) d[i] = s[i];
  if (threadIdx.x == 0) {
    if (entries) { u8* e = entries + (size_t)(firstFrame + f) * 5; st32(e, (u32)off); e[4] = (u8)(off >> 32); }
    if (sizesOut) sizesOut[firstFrame + f] = n;
  }
}

// ---- zstd 1.4.9 parameter rows (dumped there from the dependency): [wlog clog hlog slog mml tlen strat]
const uint16_t kCP16[23][7] = {{14,14,15,2,4,0,2},{14,14,15,1,5,0,1},{14,14,15,1,4,0,1},{14,14,15,2,4,0,2},{14,14,14,4,4,2,3},{14,14,14,3,4,4,4},
  {14,14,14,4,4,8,5},{14,14,14,6,4,8,5},{14,14,14,8,4,8,5},{14,15,14,5,4,8,6},{14,15,14,9,4,8,6},{14,15,14,3,4,12,7},{14,15,14,4,3,24,7},
  {14,15,14,5,3,32,8},{14,15,15,6,3,64,8},{14,15,15,7,3,256,8},{14,15,15,5,3,48,9},{14,15,15,6,3,128,9},{14,15,15,7,3,256,9},{14,15,15,8,3,256,9},{14,15,15,8,3,512,9},{14,15,15,9,3,512,9},{14,15,15,10,3,999,9}};
const uint16_t kCP128[23][7] = {{17,15,16,2,5,0,2},{17,12,13,1,6,0,1},{17,13,15,1,5,0,1},{17,15,16,2,5,0,2},{17,17,17,2,4,0,2},{17,16,17,3,4,2,3},
  {17,17,17,3,4,4,4},{17,17,17,3,4,8,5},{17,17,17,4,4,8,5},{17,17,17,5,4,8,5},{17,17,17,6,4,8,5},{17,17,17,5,4,8,6},{17,18,17,7,4,12,6},
  {17,18,17,3,4,12,7},{17,18,17,4,3,32,7},{17,18,17,6,3,256,7},{17,18,17,6,3,128,8},{17,18,17,8,3,256,8},{17,18,17,10,3,512,8},{17,18,17,5,3,256,9},{17,18,17,7,3,512,9},{17,18,17,9,3,512,9},{17,18,17,11,3,999,9}};
const uint16_t kCP256[23][7] = {{18,16,16,1,4,0,2},{18,13,14,1,6,0,1},{18,14,14,1,5,0,2},{18,16,16,1,4,0,2},{18,16,17,2,5,2,3},{18,18,18,3,5,2,3},
  {18,18,19,3,5,4,4},{18,18,19,4,4,4,4},{18,18,19,4,4,8,5},{18,18,19,5,4,8,5},{18,18,19,6,4,8,5},{18,18,19,5,4,12,6},{18,19,19,7,4,12,6},
  {18,18,19,4,4,16,7},{18,18,19,4,3,32,7},{18,18,19,6,3,128,7},{18,19,19,6,3,128,8},{18,19,19,8,3,256,8},{18,19,19,6,3,128,9},{18,19,19,8,3,256,9},{18,19,19,10,3,512,9},{18,19,19,12,3,512,9},{18,19,19,13,3,999,9}};

inline uint32_t hbit(uint32_t x) { return 31u - (uint32_t)__builtin_clz(x); }

// row 0 of ZSTD_getCParams_internal's level tables, "base for negative levels": strategy fast, the level becomes the acceleration
// (targetLength = -level), and literals are then stored raw (ZSTD_compressLiterals' disableLiteralCompression)
const uint16_t kCPNeg16[7] = {14, 12, 13, 1, 5, 1, 1}, kCPNeg128[7] = {17, 12, 12, 1, 5, 1, 1}, kCPNeg256[7] = {18, 12, 13, 1, 5, 1, 1};

// the "default" table (srcSize > 256 KB), levels 0..15 (13-15: btlazy2)
const uint16_t kCPDef[23][7] = {{21,16,17,1,5,0,2},{19,13,14,1,7,0,1},{20,15,16,1,6,0,1},{21,16,17,1,5,0,2},{21,18,18,1,5,0,2},{21,18,19,2,5,2,3},
  {21,19,19,3,5,4,3},{21,19,19,3,5,8,4},{21,19,19,3,5,16,5},{21,19,20,4,5,16,5},{22,20,21,4,5,16,5},{22,21,22,4,5,16,5},{22,21,22,5,5,16,5},
  {22,21,22,5,5,32,6},{22,22,23,5,5,32,6},{22,23,23,6,5,32,6},
  {22,22,22,5,5,48,7},{23,23,22,5,4,64,7},{23,23,22,6,3,64,8},{23,24,22,7,3,256,9},{25,25,23,7,3,256,9},{26,26,24,7,3,512,9},{27,27,25,9,3,999,9}};
const uint16_t kCPNegDef[7] = {19, 12, 13, 1, 6, 1, 1};

// parameters of ZSTD_getCParams(level, S) for every level (-128..22); false only for S == 0
bool get_params(int level, size_t S, ZraEncParams* p) {
  if (level == 0) level = 3;
  if (S == 0) return false;
  if (level > 22) level = 22;                                                       // ZSTD_maxCLevel
  const uint16_t* r = level < 0 ? (S <= (16u << 10) ? kCPNeg16 : S <= (128u << 10) ? kCPNeg128 : S <= (256u << 10) ? kCPNeg256 : kCPNegDef)
                               : (S <= (16u << 10) ? kCP16[level] : S <= (128u << 10) ? kCP128[level] : S <= (256u << 10) ? kCP256[level] : kCPDef[level]);
  // (a frame larger than 2^windowLog is parsed by the serial finders with the sliding-window rules: compress_impl_body, serialAll)
  p->windowLog = r[0]; p->chainLog = r[1]; p->hashLog = r[2]; p->searchLog = r[3]; p->minMatch = r[4];
  p->targetLength = level < 0 ? (uint32_t)(-level) : r[5]; p->strategy = r[6];
  const uint32_t srcLog = S < 64 ? 6 : hbit((uint32_t)S - 1) + 1;
  if (p->windowLog > srcLog) p->windowLog = srcLog;
  if (p->hashLog > p->windowLog + 1) p->hashLog = p->windowLog + 1;
  const uint32_t cycleLog = p->chainLog - (p->strategy >= 6);
  if (cycleLog > p->windowLog) p->chainLog -= cycleLog - p->windowLog;
  if (p->windowLog < 10) p->windowLog = 10;
  p->blockSize = std::min<uint32_t>(128u << 10, 1u << p->windowLog);
  return true;
}

}  // namespace

namespace zra_eng {

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return zerr(1); } while (0)

Status Engine::compress_frames(const uint8_t* dIn, size_t inSize, uint8_t* dBody, uint64_t* dSizes, size_t* bodySize,
                               int level, uint32_t frameSize, bool checksum) {
  return compress_impl(dIn, inSize, dBody, 0, nullptr, dSizes, bodySize, level, frameSize, checksum);
}

// Shared driver. Frames are gathered to dBody + bodyBase0 + running offset; seek-table entries (5 B, offsets relative to
// bodyBase0) are written to dEntries when non-null; u64 sizes to dSizes when non-null.
Status Engine::compress_impl(const uint8_t* dIn, size_t inSize, uint8_t* dBody, uint64_t bodyBase0, uint8_t* dEntries, uint64_t* dSizes,
                             size_t* bodySize, int level, uint32_t frameSize, bool checksum) {
  encCounters_ = nullptr; encCountersBytes_ = 0;
  mfTele_.clear(); mfTeleDev_ = nullptr;               // (a call that takes another path, or fails, leaves no telemetry of an earlier launch behind)
  Status s = compress_impl_body(dIn, inSize, dBody, bodyBase0, dEntries, dSizes, bodySize, level, frameSize, checksum);
  if (s.zra) drain_after_error();
  return s;
}

// Every error exit of the encode driver funnels through here: nothing of the failed call may still run (the caller is free to
// release dIn / dOut, and the next call reuses the same scratch and counters). Stream A's persistent match finder ends on its own
// once its frame queue is exhausted; stream B may sit in a wait-value on a sub-batch counter that will never be reached, so the
// counters are raised past every target before it is drained.
void Engine::drain_after_error() {
  (void)hipGetLastError();
  (void)hipStreamSynchronize(stream_);
  if (encCounters_ && encCountersBytes_) (void)hipMemsetAsync(encCounters_, 0x7F, encCountersBytes_, stream_);
  (void)hipStreamSynchronize(stream_);
  (void)hipStreamSynchronize(stream2_);
  // the batch path's second match-finder stream (and whatever else runs on the side streams) reads dIn and writes the contexts' scratch too
  for (auto st : pipeStreams_) if (st) (void)hipStreamSynchronize(st);
  (void)hipGetLastError();
}

Status Engine::compress_impl_body(const uint8_t* dIn, size_t inSize, uint8_t* dBody, uint64_t bodyBase0, uint8_t* dEntries, uint64_t* dSizes,
                                  size_t* bodySize, int level, uint32_t frameSize, bool checksum) {
  HIPCHK(hipSetDevice(device_));
  *bodySize = 0;
  if (inSize == 0) return ok();
  if (frameSize == 0) return zerr(42);
  const uint64_t nFramesTotal = (inSize + frameSize - 1) / frameSize;
  const size_t tailSize = inSize % frameSize;
  ZraEncParams full{}, tail{};
  if (nFramesTotal > 1 || !tailSize) { if (!get_params(level, frameSize, &full)) return zerr(40); } // parameter_unsupported: no CPU fallback
  if (tailSize) { if (!get_params(level, tailSize, &tail)) return zerr(40); }
  if (!(nFramesTotal > 1 || !tailSize)) full = tail;
  if (!tailSize) tail = full;
  // the generic (one lane per frame) finder exists with and without the optimal parsers: only levels 13-22 pay for their registers
  const auto mfGeneric = (full.strategy >= 7 || tail.strategy >= 7) ? zra_mf_opt_kernel : zra_mf_kernel;

  // per-frame table slot: hash table + chain table / tree; the optimal parsers add a 3-byte hash table and their state (ZraOptState)
  auto slotWords = [](const ZraEncParams& q) -> uint64_t {
    uint64_t w = (1ull << q.hashLog) + (1ull << q.chainLog);
    if (q.strategy >= 3 && q.strategy <= 5) w += 3ull << q.chainLog;   // the wave-cooperative hash-chain finder keeps four links per slot
    if (q.strategy >= 7) w += (q.minMatch == 3 ? 1ull << std::min(17u, q.windowLog) : 0) + (sizeof(ZraOptState) + 3) / 4 + 16;
    return w;
  };
  const uint64_t tableWords = (std::max(slotWords(full), slotWords(tail)) + 3) & ~3ull;   // 16-byte slots
  const uint32_t maxBlock = std::max(full.blockSize, tail.blockSize);
  const uint64_t seqStride = (maxBlock / 4 + 16 + 7) & ~7ull;   // (a multiple of 8: the entropy stage's u16 chain output sits behind 3 * seqStride code bytes, and the chain kernel moves 8 bytes at a time)
  const uint64_t litStride = ((uint64_t)maxBlock + 64 + 15) & ~15ull;
  // blocks per frame, per size class (a short last frame has its own, smaller, block size but is still one block)
  const uint64_t fullFrame = std::min<uint64_t>(frameSize, inSize);
  const uint32_t maxBlocksPerFrame = (uint32_t)std::max<uint64_t>(
      (nFramesTotal > 1 || !tailSize) ? (fullFrame + full.blockSize - 1) / full.blockSize : 0,
      tailSize ? (tailSize + tail.blockSize - 1) / tail.blockSize : 0);
  const uint64_t slotStride = (zra_fmt::compress_bound(frameSize) + 1024 + 4 * (uint64_t)maxBlocksPerFrame + 15) & ~15ull;
  const uint64_t perFrame = tableWords * 4 + seqStride * 8 + litStride + slotStride + sizeof(ZraEncFrameState) + sizeof(ZraEncBlockOut) + 64;
  // ---------------------------------------------------------------------------------------------------------------------
  // Persistent pipeline (dfast, single-block frames: levels 3-4 up to 128 KiB frames — the headline configuration).
  // ONE match-finder launch per super-batch: `nSlots` resident waves pull frames from a queue, each wave owns one hash-table
  // slot (tables are per resident wave, not per frame). Finished frames are counted per sub-batch; the entropy stage, scan and
  // gather of a sub-batch are released on stream B by hipStreamWaitValue32 on that counter — no per-batch launch tails in the
  // DRAM-bound match finder, and the entropy stage runs under it in small pieces.
  {
    bool persist = maxBlocksPerFrame == 1 && full.strategy == 2 && !std::getenv("ZRA_MF_NOPERSIST");   // (one block: the frame fits any window)
    if (persist && waitValueOk_ == 0) {
      // the pipeline needs stream memory operations on plain device memory; probe once (a wait that is already satisfied), and
      // use the batch path below on runtimes without them
      if (!encScan_.reserve(64)) return zerr(64);
      bool okProbe = hipMemsetAsync(encScan_.p, 0, 64, stream2_) == hipSuccess &&
                     hipStreamWaitValue32(stream2_, encScan_.p, 0, hipStreamWaitValueGte, 0xFFFFFFFFu) == hipSuccess &&
                     hipStreamSynchronize(stream2_) == hipSuccess;
      if (!okProbe) (void)hipGetLastError();
      waitValueOk_ = okProbe ? 1 : -1;
    }
    if (waitValueOk_ < 0) persist = false;
    if (persist)
      return compress_persistent(dIn, inSize, dBody, bodyBase0, dEntries, dSizes, bodySize, frameSize, checksum, full, tail,
                                 tableWords, seqStride, litStride, slotStride);
  }
  // Two scratch contexts: the match finder of batch k+1 (stream A) overlaps the entropy stage + gather of batch k (stream B);
  // both kernels are latency-bound, so they share the CUs almost for free. 8 GiB of scratch per context.
  // scratch per context: enough for one frame per resident wave (32 per CU) of the one-lane finders, whose throughput is frames in
  // flight / frame latency (level 9 @ 256 KiB needs 6 MiB of tables per frame); bring-up knob ZRA_ENC_BUDGET_GIB
  uint64_t budget = 8ull << 30;
  if (const char* e = std::getenv("ZRA_ENC_BUDGET_GIB")) budget = (uint64_t)std::atoi(e) << 30;
  else if (full.strategy != 2) {
    // measured at level 9 @ 256 KiB (4 GiB input): 8 GiB -> 0.33 GiB/s, 16 -> 0.58, 32 -> 0.71, 64 -> 0.93 (all frames resident)
    size_t freeB = 0, totalB = 0;
    // (what this engine's contexts hold already counts as free: a second call must not get smaller batches than the first)
    uint64_t mine = 0; for (auto& x : encCtx_) mine += x.tables.cap + x.seqs.cap + x.lits.cap + x.slots.cap;
    if (hipMemGetInfo(&freeB, &totalB) == hipSuccess) budget = std::min<uint64_t>(64ull << 30, std::max<uint64_t>(budget, (uint64_t)(((uint64_t)freeB + mine) * 0.25)));
  }
  uint32_t B = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>({nFramesTotal, budget / perFrame, full.strategy == 2 ? 16384ull : 65536ull}));
  if (B > 1024) B &= ~1023u;
  // Frames of several blocks whose batch would hold the whole call: two half batches instead. Block b + 1's match finder waits for block
  // b's entropy stage (the confirmed repcodes: a block stored raw does not move them), so ONE batch runs finder and entropy stage strictly
  // in turns; two contexts fill each other's gaps — as long as half a batch still fills the chip (16 frames per CU). Measured on one box
  // (profiles/r05_experiments.md §9; log-like data, 256 KiB frames): 8,192 frames at level 9 3.69-3.80 -> 3.89-3.98 GiB/s, at level 5
  // 8.14-8.18 -> 8.94-8.99; 4,096 frames at level 3 12.8 -> 10.6 (hence the bound). ZRA_ENC_SPLIT=0: off.
  // And batches of EQUAL size whenever there are several: a call of 8,192 frames with room for 7,168 ran as 7,168 + 1,024 (the bench's
  // c4_share, where the timed buffers leave less free memory than a lone call finds), and the small batch left its context idle.
  { static const bool split = !(std::getenv("ZRA_ENC_SPLIT") && std::atoi(std::getenv("ZRA_ENC_SPLIT")) == 0);
    uint64_t nBatches = (nFramesTotal + B - 1) / B;
    if (split && maxBlocksPerFrame > 1 && nBatches == 1 && nFramesTotal >= 32ull * (uint64_t)numCUs_) nBatches = 2;
    if (split && nBatches > 1) B = (uint32_t)std::min<uint64_t>(B, ((nFramesTotal + nBatches - 1) / nBatches + 63) & ~63ull); }
  // the entropy stage: workgroups that take the batch's frames from a queue — as many as the device holds at once (5 waves per SIMD by
  // registers), each with its own literal buffer and sequence work area
  const uint32_t entGridB = (uint32_t)std::min<uint64_t>(B, (uint64_t)numCUs_ * 5);
  const uint64_t entWorkStride = (9 * seqStride + 255) & ~255ull;
  const int nCtx = nFramesTotal > B ? 2 : 1;
  for (int c = 0; c < nCtx; c++) {
    EncCtx& x = encCtx_[c];
    if (!x.tables.reserve(B * tableWords * 4) || !x.seqs.reserve(B * seqStride * 8) || !x.lits.reserve((size_t)entGridB * litStride) ||
        !x.work.reserve((size_t)entGridB * entWorkStride) ||
        !x.slots.reserve(B * slotStride) || !x.misc.reserve(B * (sizeof(ZraEncFrameState) + sizeof(ZraEncBlockOut))) ||
        !x.ck.reserve((size_t)B * 4) || !x.sizes.reserve((size_t)B * 16))
      return zerr(64);
  }
  // [u64 running offset][pad][one frame queue of the entropy stage per launch, 4 bytes each, cleared here]
  const size_t nEntLaunches = (size_t)((nFramesTotal + B - 1) / B) * maxBlocksPerFrame + 4;
  if (!encScan_.reserve(64 + 4 * nEntLaunches)) return zerr(64);
  uint64_t* dRunning = encScan_.as<uint64_t>();
  uint32_t* dEntQueues = (uint32_t*)(encScan_.as<uint8_t>() + 64);
  HIPCHK(hipMemsetAsync(encScan_.p, 0, 64 + 4 * nEntLaunches, stream2_));
  size_t entLaunch = 0;

  ZraEncArgs base{};
  base.in = dIn; base.inSize = inSize; base.frameSize = frameSize; base.checksum = checksum ? 1 : 0;
  base.full = full; base.tail = tail;
  dbgSeqStride_ = seqStride; dbgB_ = (uint32_t)B;
  base.tableStride = tableWords; base.seqStride = seqStride; base.litStride = litStride; base.slotStride = slotStride;
  base.entWorkStride = entWorkStride; base.slotRing = B; base.entSubFrames = B; base.entPrio = 3;

  // event pool: [2 per mf launch on stream A] [2 per entropy launch on stream B]; dependencies mfDone / entDone per context
  size_t evNext = 0;
  auto ev = [&]() -> hipEvent_t {
    if (evNext == evPool_.size()) { hipEvent_t e = nullptr; if (hipEventCreate(&e) != hipSuccess) return nullptr; evPool_.push_back(e); }
    return evPool_[evNext++];
  };
  std::vector<std::pair<hipEvent_t, hipEvent_t>> mfSpans, entSpans;
  hipEvent_t entDone[2] = {nullptr, nullptr};
  // the second context's match-finder stream (ZRA_MF_ONE_STREAM: bring-up knob, everything on stream A as before)
  hipStream_t mfStream2 = nullptr;
  if (nCtx == 2 && !std::getenv("ZRA_MF_ONE_STREAM")) {
    if (!pipeStreams_[0] && hipStreamCreateWithFlags(&pipeStreams_[0], hipStreamNonBlocking) != hipSuccess) { pipeStreams_[0] = nullptr; (void)hipGetLastError(); }
    mfStream2 = pipeStreams_[0];
  }
  // make stream B's first use wait for whatever the caller queued on stream A (inputs produced on the engine stream)
  { hipEvent_t e0 = ev(); if (!e0) return zerr(1); HIPCHK(hipEventRecord(e0, stream_)); HIPCHK(hipStreamWaitEvent(stream2_, e0, 0));
    if (mfStream2) HIPCHK(hipStreamWaitEvent(mfStream2, e0, 0)); }

  uint64_t batchIdx = 0;
  for (uint64_t f0 = 0; f0 < nFramesTotal; f0 += B, batchIdx++) {
    const uint32_t nb = (uint32_t)std::min<uint64_t>(B, nFramesTotal - f0);
    const int c = (int)(batchIdx % nCtx);
    EncCtx& x = encCtx_[c];
    ZraEncArgs a = base;
    a.firstFrame = (uint32_t)f0; a.nFrames = nb;
    a.tables = x.tables.as<uint32_t>(); a.seqs = x.seqs.as<uint64_t>(); a.lits = x.lits.as<uint8_t>(); a.slots = x.slots.as<uint8_t>();
    a.state = x.misc.as<ZraEncFrameState>();
    a.blockOut = (ZraEncBlockOut*)(x.misc.as<uint8_t>() + (size_t)B * sizeof(ZraEncFrameState));
    a.contentCk = x.ck.as<uint32_t>();
    a.sizes = x.sizes.as<uint64_t>();
    uint64_t* dOffsets = x.sizes.as<uint64_t>() + B;
    if (checksum)
      hipLaunchKernelGGL(zra_content_ck_kernel, dim3((nb * 4 + 255) / 256), dim3(256), 0, stream2_, dIn, (u64)inSize, frameSize, (u32)f0, nb, a.contentCk);
    // frames of this batch: how many block rounds? (a short last frame may need fewer)
    const uint64_t firstFrameSize = std::min<uint64_t>(frameSize, inSize - f0 * frameSize);
    const ZraEncParams& P0 = firstFrameSize == frameSize ? full : tail;
    const uint32_t rounds = (uint32_t)((firstFrameSize + P0.blockSize - 1) / P0.blockSize);
    // Match-finder launches of the two contexts go to two streams: a batch's launches wait for each other through the entropy stage
    // (block b + 1 needs block b's confirmed state) and every launch ends in a tail of a few slow frames — with one stream the next
    // batch's first launch sat behind all of that; on its own stream it fills the CUs the other context leaves idle (round 4: 8 GiB at
    // level 9 / 256 KiB frames, profiles/r04_experiments.md §10).
    hipStream_t sA = (c == 1 && mfStream2) ? mfStream2 : stream_;
    // the context's scratch is free once the entropy stage + gather of its previous batch are done
    if (entDone[c]) HIPCHK(hipStreamWaitEvent(sA, entDone[c], 0));
    // test knob: the table scratch filled with a pattern before a batch — whatever a finder does not clear itself (the wave-cooperative
    // hash-chain finder leaves its chain slots alone) must not matter (tests/test_gpu_parity.py::test_opt_in_kernels_are_bit_exact_too)
    { static const bool poison = std::getenv("ZRA_ENC_POISON") != nullptr;
      if (poison) HIPCHK(hipMemsetAsync(x.tables.p, 0xA5, (size_t)nb * tableWords * 4, sA)); }
    for (uint32_t blk = 0; blk < rounds; blk++) {
      hipEvent_t m0 = ev(), m1 = ev(), e1 = ev();
      if (!m0 || !m1 || !e1) return zerr(1);
      HIPCHK(hipEventRecord(m0, sA));
      {
        // occupancy experiment knob (bring-up): extra dynamic LDS per workgroup caps the frames in flight per CU
        static const int dynLds = std::getenv("ZRA_MF_LDS") ? std::atoi(std::getenv("ZRA_MF_LDS")) : 0;
        // dfast batches (levels 3-4) run the lean window-resolve kernel; a short last frame whose cparams select another
        // strategy is parsed by the generic kernel in a second single-frame launch
        const bool hasTail = tailSize && f0 + nb == nFramesTotal;
        // LDS geometry of the dfast kernel's bucket filter (see compress_persistent)
        uint32_t shL = 1, shS = 2, dupLog = 8;
        if (const char* f = std::getenv("ZRA_MF_FILTER")) { unsigned x = 1, y = 1, z = 8; if (std::sscanf(f, "%u,%u,%u", &x, &y, &z) >= 1) { shL = x & 15; shS = y & 15; dupLog = z & 15; } }
        a.mfFilter = shL | (shS << 4) | (dupLog << 8);
        const uint32_t hl = std::max(full.hashLog, tail.hashLog), cl = std::max(full.chainLog, tail.chainLog);
        const size_t filterBytes = (8u << dupLog) + (((size_t)1 << hl) >> shL) / 8 + (((size_t)1 << cl) >> shS) / 8 + 64;
        // frames larger than the level's window (0.5 - 4 MiB and more): the sliding-window rules live in the serial finders only
        const bool serialAll = std::min<uint64_t>(frameSize, inSize) > (1ull << full.windowLog);
        a.serialAll = serialAll ? 1u : 0u;
        const bool oddTail = !serialAll && hasTail && (tail.strategy == 2) != (full.strategy == 2);
        // a lone short last frame that is not dfast: the wave-cooperative hash-chain kernel for greedy / lazy / lazy2, else the generic one
        auto launchLone = [&](uint32_t only, uint32_t slot) {
          if (tail.strategy >= 3 && tail.strategy <= 5) hipLaunchKernelGGL(zra_mf_hc_kernel, dim3(1), dim3(64), 0, sA, a, blk, only, slot);
          else hipLaunchKernelGGL(mfGeneric, dim3(1), dim3(64), 0, sA, a, blk, only, slot, 1u);
        };
        if (full.strategy == 2 && !serialAll) {
          hipLaunchKernelGGL(zra_mf_dfast_kernel, dim3(nb), dim3(64), filterBytes + dynLds, sA, a, blk, 0xFFFFFFFFu, 0u);
          if (oddTail) launchLone((uint32_t)(nb - 1), (uint32_t)(nb - 1));
        } else {
          // frames per wave: as many as it takes to have every frame of the batch resident at once (32 waves per CU)
          static const int pwEnv = std::getenv("ZRA_MF_PERWAVE") ? std::atoi(std::getenv("ZRA_MF_PERWAVE")) : 0;
          // hash-chain strategies (greedy / lazy / lazy2): one frame per wave, the wave-cooperative finder; fast gains 9.4 -> 14.5 from 8
          const bool hashChain = full.strategy >= 3 && full.strategy <= 5 && !serialAll;
          uint32_t perWave = pwEnv > 0 ? (uint32_t)pwEnv : hashChain ? 1u
                           : std::min<uint32_t>(8u, std::max<uint32_t>(1u, (nb + (uint32_t)numCUs_ * 32 - 1) / ((uint32_t)numCUs_ * 32)));
          // the hash-chain kernel and the fast kernel hold their own finder only (lean register budgets); a short last frame with any
          // other strategy goes to the generic kernel (or the dfast kernel) in a second, single-frame launch
          const bool lean = pwEnv <= 0 && (hashChain || full.strategy == 1);
          // resident waves of the hash-chain finder: 32 per CU (all the hardware holds) — except for the deep searches over big tables
          // (search log >= 5 with >= 4 MiB of tables per frame: levels 9-10 at 256 KiB frames), which run faster with 20, five per SIMD
          // (level 9 @ 256 KiB: 3.53 GiB/s at 32, 3.48 at 22, 3.74 at 20, 3.70 at 16, 3.21 at 14; level 10: 2.11 -> 2.29 at 16; level 6 @
          // 256 KiB with the same tables and search log 3: 7.16 -> 6.55 at 16; level 9 @ 64 KiB: 5.20 -> 4.53 at 16 — profiles/
          // r04_experiments.md §9). Padding each wave's LDS (3 KiB of its own) to 8 KiB sets it.
          uint32_t hcPad = (uint32_t)dynLds;
          if (!std::getenv("ZRA_MF_LDS") && hashChain && full.searchLog >= 5 && tableWords * 4 >= (4ull << 20)) hcPad = 5120;
          if (hashChain && pwEnv <= 0) hipLaunchKernelGGL(zra_mf_hc_kernel, dim3(nb), dim3(64), hcPad, sA, a, blk, 0xFFFFFFFFu, 0u);
          else if (lean) hipLaunchKernelGGL(zra_mf_fast_kernel, dim3((nb + perWave - 1) / perWave), dim3(64), dynLds, sA, a, blk, perWave);
          else hipLaunchKernelGGL(mfGeneric, dim3((nb + perWave - 1) / perWave), dim3(64), dynLds, sA, a, blk, 0xFFFFFFFFu, 0u, perWave);
          if (oddTail) hipLaunchKernelGGL(zra_mf_dfast_kernel, dim3(1), dim3(64), filterBytes, sA, a, blk, (uint32_t)(nb - 1), (uint32_t)(nb - 1));
          else if (lean && hasTail && (hashChain ? (tail.strategy < 3 || tail.strategy > 5) : tail.strategy != 1))
            launchLone((uint32_t)(nb - 1), (uint32_t)(nb - 1));
        }
      }
      HIPCHK(hipEventRecord(m1, sA));
      mfSpans.push_back({m0, m1});
      HIPCHK(hipStreamWaitEvent(stream2_, m1, 0));
      hipEvent_t e0 = ev(); if (!e0) return zerr(1);
      HIPCHK(hipEventRecord(e0, stream2_));
      { ZraEncArgs ae = a; ae.entQueue = dEntQueues + entLaunch++; ae.entWork = x.work.as<uint8_t>();
        hipLaunchKernelGGL(zra_entropy_kernel, dim3(std::min<uint32_t>(nb, entGridB)), dim3(256), 0, stream2_, ae, blk); }
      HIPCHK(hipEventRecord(e1, stream2_));
      entSpans.push_back({e0, e1});
      if (blk + 1 < rounds) HIPCHK(hipStreamWaitEvent(sA, e1, 0));   // next block's match finder needs the confirmed state
    }
    hipLaunchKernelGGL(zra_scan_sizes_kernel, dim3(1), dim3(64), 0, stream2_, a.sizes, nb, dOffsets, dRunning, (const u32*)nullptr);
    hipLaunchKernelGGL(zra_gather_frames_kernel, dim3(nb), dim3(256), 0, stream2_, a.slots, slotStride, a.sizes, dOffsets, dBody,
                       bodyBase0, dEntries ? dEntries : nullptr, (u32)f0, dSizes, (const u32*)nullptr);
    hipEvent_t done = ev(); if (!done) return zerr(1);
    HIPCHK(hipEventRecord(done, stream2_));
    entDone[c] = done;
    // test hook: give up behind batch ZRA_ENC_FAIL_BATCH with its kernels in flight (the error exit must drain every stream: tests/test_gpu_parity.py)
    if (const char* fb = std::getenv("ZRA_ENC_FAIL_BATCH")) if ((uint64_t)std::atoll(fb) == batchIdx) return zerr(1);
    { static const bool serial = std::getenv("ZRA_ENC_SERIAL") != nullptr;   // bring-up knob: no mf/entropy overlap (per-kernel timing in isolation)
      if (serial) { HIPCHK(hipStreamWaitEvent(stream_, done, 0)); if (mfStream2) HIPCHK(hipStreamWaitEvent(mfStream2, done, 0)); } }
  }
  uint64_t total = 0;
  HIPCHK(hipMemcpyAsync(&total, dRunning, 8, hipMemcpyDeviceToHost, stream2_));
  HIPCHK(hipStreamSynchronize(stream2_));
  HIPCHK(hipStreamSynchronize(stream_));
  if (mfStream2) HIPCHK(hipStreamSynchronize(mfStream2));
  HIPCHK(hipGetLastError());
  double kernelMs = 0;
  kstats_[0] = kstats_[1] = kstats_[2] = kstats_[3] = 0;
  for (auto& sp : mfSpans) { float m = 0; if (hipEventElapsedTime(&m, sp.first, sp.second) == hipSuccess) { kstats_[0] += m; kstats_[1] += 1; kernelMs += m; } }
  for (auto& sp : entSpans) { float m = 0; if (hipEventElapsedTime(&m, sp.first, sp.second) == hipSuccess) { kstats_[2] += m; kstats_[3] += 1; } }
  lastKernelMs_ = kernelMs;
  *bodySize = total;
  return ok();
}

Status Engine::compress_persistent(const uint8_t* dIn, size_t inSize, uint8_t* dBody, uint64_t bodyBase0, uint8_t* dEntries, uint64_t* dSizes,
                                   size_t* bodySize, uint32_t frameSize, bool checksum, const ZraEncParams& full, const ZraEncParams& tail,
                                   uint64_t tableWords, uint64_t seqStride, uint64_t litStride, uint64_t slotStride) {
  const uint64_t nFramesTotal = (inSize + frameSize - 1) / frameSize;
  const size_t tailSize = inSize % frameSize;
  const auto mfGeneric = (full.strategy >= 7 || tail.strategy >= 7) ? zra_mf_opt_kernel : zra_mf_kernel;
  // How the entropy stage shares the device with the match finder (ZRA_PIPE; one box, 16 GiB, round 4's library 997-999 ms in its slower
  // state on that box — profiles/r05_experiments.md):
  //   1 (default)  round 4's overlap: per sub-batch of 8192 frames one entropy launch + scan + gather on stream B, released by the
  //                finder's count of finished frames; 18 finder waves per CU leave room for one entropy workgroup.  938 ms
  //   0            the entropy stage behind the whole finder launch: 22 finder waves per CU (842 ms) + 190 ms.       1038 ms
  //   2            the entropy stage resident beside the finder, one queue-driven workgroup per CU, scanning and gathering itself:
  //                the finder loses what the stage gains.                                                      988-1075 ms
  // With the bucket flags fewer table requests queue up and more finder waves pay again (alone: 18 waves 880-897 ms, 20 waves 855-862,
  // 22 and 24 waves 824-846; without the flags 18 = 24 waves, round 4) — but the entropy stage needs its room: at 20 waves it falls behind.
  static const int pipeMode = std::getenv("ZRA_PIPE") ? std::atoi(std::getenv("ZRA_PIPE")) : 1;
  // (round 6: 19 beside the entropy stage — its workgroup takes 14 of the CU's 128 LDS pieces since the histograms share storage with the
  //  tables, a finder wave 6: 19 x 6 + 14 = 128; it was 18 x 6 + 19)
  static const uint32_t wavesPerCU = std::getenv("ZRA_MF_WAVES") ? (uint32_t)std::atoi(std::getenv("ZRA_MF_WAVES")) : (pipeMode == 0 ? 22u : 19u);
  static const uint32_t SB = std::getenv("ZRA_ENC_SUB") ? (uint32_t)std::atoi(std::getenv("ZRA_ENC_SUB")) : 8192u;   // frames per sub-batch
  const uint32_t nSlots = (uint32_t)std::min<uint64_t>((uint64_t)numCUs_ * wavesPerCU, nFramesTotal);
  // per-frame scratch that lives from the match finder to the entropy stage: sequences + block record + checksum + size/offset
  const uint64_t perFrame = seqStride * 8 + sizeof(ZraEncFrameState) + sizeof(ZraEncBlockOut) + 4 + 16 + 8;
  // per context (two of them); every launch boundary costs the pipeline about 7 ms (A/B on one box: 8 launches instead of 5 per 16 GiB
  // = -2 %), so the launches are as long as a scratch budget allows; bring-up knob ZRA_ENC_PBUDGET_GIB
  // Round 3: the boundary costs more than that once everything else is tuned — at 16 GiB one launch instead of five is 11-13 % of the
  // match finder's time (A/B on one box: 5 x 200 ms vs 1 x 867-897 ms; every launch ends with a tail of straggling frames and starts
  // with all waves in step). So the budget is what the device can spare: a third of its free memory, between 8 and 64 GiB (46 GiB of
  // sequence scratch take 16 GiB of 64 KiB frames through in one launch; the scratch is grow-only and given back by
  // ZraHipReleaseScratch / the engine pool's cap).
  uint64_t budget;
  if (const char* e = std::getenv("ZRA_ENC_PBUDGET_GIB")) budget = (uint64_t)std::max(1, std::atoi(e)) << 30;
  else {
    size_t freeB = 0, totalB = 0;
    if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) freeB = 24ull << 30;
    // what this engine already holds for the purpose counts as free (the reservation below reuses it)
    uint64_t mine = 0; for (auto& x : encCtx_) mine += x.seqs.cap;
    budget = std::min<uint64_t>(64ull << 30, std::max<uint64_t>(8ull << 30, ((uint64_t)freeB + mine) / 3));
  }
  // the budget is a wish (other engines of the pool, other ranks on the device and the caller's own buffers read the same free-memory
  // figure): a reservation that fails is tried again with half of it, down to 1 GiB, before the call gives up with memory_allocation
  uint64_t SBIG = 0, nSuper = 0, subsPerSuper = 0; int nCtx = 1;
  EncCtx& sh = encCtx_[0];                       // shared: table slots, the entropy workgroups' literal buffers, the ring of encoded-frame slots
  // entropy stage: queue-driven workgroups. Behind the match finder (pipeMode 0, the default): as many as the device holds, five per CU
  // by registers. Resident beside it (ZRA_PIPE=2): ONE per CU — what fits next to 18 match-finder waves; workgroups that find no room
  // would keep the launch's hardware queue busy until the finder leaves.
  // Round 5 measured both on one box (16 GiB, profiles/r05_experiments.md): beside the finder the stage keeps up (0.92-0.97 ms per frame
  // and workgroup), and slows the finder from 846-970 ms to 1020-1075 ms: its work is ALU and LDS work, not idle latency, and costs about
  // what it costs alone. Behind the finder: 846 + 187 ms.
  static const uint32_t entPerCU = std::getenv("ZRA_ENT_WGS") ? (uint32_t)std::max(1, std::atoi(std::getenv("ZRA_ENT_WGS"))) : (pipeMode == 0 ? 5u : pipeMode == 1 ? 8u : 1u);
  const uint32_t entGrid = (uint32_t)std::min<uint64_t>((uint64_t)numCUs_ * entPerCU, nFramesTotal);
  const uint64_t entWorkStride = (9 * seqStride + 255) & ~255ull;
  // round 6: the split entropy stage (pipeMode 1 only): per sub-batch a FRONT launch (literals, tables), zra_ent_chain_kernel (the state chains,
  // lane = (frame, stream)), a BACK launch (sequence bitstream, block / frame end). The work area and a record are then per FRAME of the
  // sub-batch. ZRA_ENT_SPLIT: 0 never, 1 (default) calls of at least 256 frames, 2 always.
  static const int splitEnv = std::getenv("ZRA_ENT_SPLIT") ? std::atoi(std::getenv("ZRA_ENT_SPLIT")) : 1;
  const bool entSplit = pipeMode == 1 && (splitEnv >= 2 || (splitEnv == 1 && nFramesTotal >= 256));
  static const uint32_t ringSubsEnv = std::getenv("ZRA_ENC_RING") ? (uint32_t)std::max(2, std::atoi(std::getenv("ZRA_ENC_RING"))) : 4u;   // sub-batches the slot ring holds
  uint64_t slotRing = 0;
  for (;; budget /= 2) {
    SBIG = std::max<uint64_t>(1, std::min<uint64_t>(nFramesTotal, budget / perFrame));
    if (const char* e = std::getenv("ZRA_ENC_SUPER")) SBIG = std::max<uint64_t>(1, std::min<uint64_t>(SBIG, (uint64_t)std::atoll(e)));   // bring-up knob
    if (SBIG > SB) SBIG -= SBIG % SB;
    nCtx = nFramesTotal > SBIG ? 2 : 1;
    nSuper = (nFramesTotal + SBIG - 1) / SBIG;
    subsPerSuper = (SBIG + SB - 1) / SB;
    slotRing = std::min<uint64_t>((uint64_t)ringSubsEnv * SB, subsPerSuper * (uint64_t)SB);
    if (SBIG <= SB) slotRing = SBIG;
    const uint64_t workUnits = entSplit ? std::min<uint64_t>(SB, nFramesTotal) : entGrid;
    bool okR = sh.tables.reserve((size_t)nSlots * tableWords * 4) && sh.lits.reserve((size_t)entGrid * litStride) && sh.work.reserve((size_t)workUnits * entWorkStride) &&
               sh.slots.reserve(slotRing * slotStride) && (!entSplit || sh.rec.reserve((size_t)workUnits * sizeof(ZraEntRec)));
    for (int c = 0; c < nCtx && okR; c++) {
      EncCtx& x = encCtx_[c];
      okR = x.seqs.reserve(SBIG * seqStride * 8) && x.misc.reserve(SBIG * (sizeof(ZraEncFrameState) + sizeof(ZraEncBlockOut))) &&
            x.ck.reserve((size_t)SBIG * 4) && x.sizes.reserve((size_t)SBIG * 16);
    }
    if (okR) break;
    if (budget <= (1ull << 30) || SBIG <= SB) return zerr(64);
    (void)hipGetLastError();
  }
  // counters: [u64 running offset][pad][abort + 3 pad][per super-batch 8: mf queue, mf started, entropy queue, scanned, handing out,
  // gathered, 2 pad][per sub-batch 4: encoded, handed out for copying, copied, finished by the match finder]
  const size_t nCnt = 4 + 8 * (size_t)nSuper + 4 * (size_t)(nSuper * subsPerSuper) + 4;
  const size_t cntCore = (16 + 4 * nCnt + 15) & ~(size_t)15;
  const size_t cntBytes = cntCore + 8 * (size_t)ZRA_TELE_WORDS;      // + the launch telemetry (ZraEncArgs::mfTele)
  if (!encScan_.reserve(cntBytes)) return zerr(64);
  uint64_t* dRunning = encScan_.as<uint64_t>();
  uint32_t* dCnt = (uint32_t*)(encScan_.as<uint8_t>() + 16);
  uint32_t* dAbort = dCnt;                       // [0]
  uint32_t* dPerSuper = dCnt + 4;                // 8 words per super-batch
  uint32_t* dPerSub = dPerSuper + 8 * nSuper;    // 4 x subsPerSuper words per super-batch
  HIPCHK(hipMemsetAsync(encScan_.as<uint8_t>(), 0, cntBytes, stream_));
  encCounters_ = dCnt; encCountersBytes_ = 4 * (4 + 8 * (size_t)nSuper + 4 * (size_t)(nSuper * subsPerSuper));   // (... the per-sub-batch words too: stream B waits on the finder's mfDone counts)   // (an error exit fills them with 0x7F: the abort word is set, the queues are past their ends, every wait of the two kernels and of stream B ends)

  ZraEncArgs base{};
  base.in = dIn; base.inSize = inSize; base.frameSize = frameSize; base.checksum = checksum ? 1 : 0;
  base.full = full; base.tail = tail;
  dbgSeqStride_ = seqStride; dbgB_ = (uint32_t)SBIG;
  base.tableStride = tableWords; base.seqStride = seqStride; base.litStride = litStride; base.slotStride = slotStride;
  base.mfTele = (uint64_t*)(encScan_.as<uint8_t>() + cntCore);
  base.entWorkStride = entWorkStride;
  // issue priority of the entropy stage's waves beside the finder: 1 (one box, alternating processes: priority 3 -> 903-911 ms, 1 -> 879-880,
  // 0 -> 919 with the stage falling behind)
  { static const int ep = std::getenv("ZRA_ENT_PRIO") ? std::atoi(std::getenv("ZRA_ENT_PRIO")) : 1; base.entPrio = (uint32_t)ep; }
  base.pipeAbort = dAbort; base.entSubFrames = SB; base.slotRing = (uint32_t)slotRing; base.readyStamp = 1u;
  base.running = dRunning; base.gBody = dBody + bodyBase0; base.gEntries = dEntries; base.gSizesOut = dSizes;
  // LDS geometry of the match finder's wave: the bucket filter — 1 bit per 2^shL long-table buckets, 1 bit per 2^shS short-table buckets —
  // and the duplicate-detection slots: 2 KiB + 4 KiB + 0.5 KiB = 6.5 KiB at hashLog 16 / chainLog 15 (one bit per 2 long buckets, per 8
  // short buckets). The flag sweep ahead of a frame's parse (df_later_flags) runs over the same bytes. What fits a CU beside the entropy
  // stage's workgroup is decided by LDS in pieces of 1,280 bytes (below): 18 waves of <= 7,680 bytes, or 20 of <= 6,400 — and 20 were
  // measured slower whichever part paid for it (128 duplicate slots: + 9 %; 1 filter bit per 16 short buckets: + 3 %).
  uint32_t shL = 1, shS = 3, dupLog = 8;
  if (const char* f = std::getenv("ZRA_MF_FILTER")) { unsigned x = 1, y = 1, z = 8; if (std::sscanf(f, "%u,%u,%u", &x, &y, &z) >= 1) { shL = x & 15; shS = y & 15; dupLog = z & 15; } }
  base.mfFilter = shL | (shS << 4) | (dupLog << 8);
  const uint32_t hl = std::max(full.hashLog, tail.hashLog), cl = std::max(full.chainLog, tail.chainLog);
  const size_t filterBytes = (8u << dupLog) + (((size_t)1 << hl) >> shL) / 8 + (((size_t)1 << cl) >> shS) / 8;

  // ---- latency mode (round 4): calls of at most ZRA_MF_LS_MAX frames (default: two per CU, what LDS holds at 64 KiB) run the dfast parse over
  // a copy of the frame in LDS (zra_mf_dfast_ls_kernel). ZRA_MF_LS=0 turns it off.
  static const int lsEnv = std::getenv("ZRA_MF_LS") ? std::atoi(std::getenv("ZRA_MF_LS")) : 1;
  const uint32_t lsBytes = (uint32_t)((std::min<uint64_t>(frameSize, inSize) + 64 + 63) & ~63ull);
  const uint32_t lsPerCu = (uint32_t)((160u << 10) / (lsBytes + filterBytes));
  static const int lsMaxEnv = std::getenv("ZRA_MF_LS_MAX") ? std::atoi(std::getenv("ZRA_MF_LS_MAX")) : -1;
  bool useLs = lsEnv != 0 && full.strategy == 2 && lsPerCu >= 1 &&
               nFramesTotal <= (lsMaxEnv >= 0 ? (uint64_t)lsMaxEnv : (uint64_t)lsPerCu * (uint64_t)numCUs_);
  if (useLs && lsAttr_ == 0) {
    const bool okA = hipFuncSetAttribute((const void*)zra_mf_dfast_ls_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10) == hipSuccess;
    if (!okA) (void)hipGetLastError();
    lsAttr_ = okA ? 1 : -1;
  }
  if (lsAttr_ < 0) useLs = false;

  // ---- round 5: bucket flags computed by the match finder's own waves, ahead of each frame's parse (df_later_flags): one flag slot per
  // resident wave. ZRA_MF_FLAGS=0 turns them off (bring-up A/B).
  static const int flEnv = std::getenv("ZRA_MF_FLAGS") ? std::atoi(std::getenv("ZRA_MF_FLAGS")) : 1;
  const bool useFlagsWave = flEnv != 0 && !useLs && full.strategy == 2;
  ZraFlagArgs fw{};
  if (useFlagsWave) {
    fw.flagStride = ((((uint64_t)std::min<uint64_t>(frameSize, inSize) + 511) / 512) * 128 + 255) & ~255ull;   // 16 bytes per window of 64 positions, whole blocks of 8 windows
    if (!mfFlags_.reserve((size_t)nSlots * fw.flagStride)) return zerr(64);
    fw.flags = mfFlags_.as<uint8_t>(); fw.ldsWords = (uint32_t)(filterBytes / 4);
    // the parse's source span in LDS behind the filter (mf_dfast_lean): 768 bytes + their flags = 992 bytes per wave. LDS is handed out
    // in pieces of 1,280 bytes on this chip (measured, profiles/r05_experiments.md §6: 21 waves of 6.5 KiB fit a CU, not 24), so a wave
    // holds 7,680 bytes with or without the span, and 18 waves + the entropy stage's workgroup (24,320) fill the CU's 163,840 to within
    // 1,280 bytes. ZRA_MF_SPAN=<bytes> (multiple of 64, <= 1024; 0: none)
    static const int spEnv = std::getenv("ZRA_MF_SPAN") ? std::atoi(std::getenv("ZRA_MF_SPAN")) : 768;
    fw.spanBytes = (uint32_t)std::min(1024, std::max(0, spEnv)) & ~63u;
    // (advisor, round 5: the parse needs 15 bytes of alignment slack + a window's 64 + 7 bytes in the span; a shorter one could never be filled)
    if (fw.spanBytes < 128) fw.spanBytes = 0;
    // round 6: epoch bits in the cells' tag field — a wave clears its 384 KiB table slot once per 2^epochBits frames instead of per frame.
    // The tag keeps >= 10 hash bits (position + 1 takes 16-17 bits of a cell at 64-128 KiB frames). ZRA_MF_EPOCH=<bits> (0: clear per frame)
    static const int epEnv = std::getenv("ZRA_MF_EPOCH") ? std::atoi(std::getenv("ZRA_MF_EPOCH")) : 4;
    const uint32_t ibMax = 32u - (uint32_t)__builtin_clz((uint32_t)std::max<uint64_t>(2, std::min<uint64_t>(frameSize, inSize)) - 1u);
    fw.epochBits = (uint32_t)std::max(0, std::min(epEnv, (int)(32u - ibMax) - 10));   // (the kernel holds ZRA_DF_EPOCH_BITS = 4 of them)
  }
  const size_t spanLds = fw.spanBytes ? fw.spanBytes + 16 + (fw.spanBytes / 64 + 1) * 16 : 0;

  size_t evNext = 0;
  auto ev = [&]() -> hipEvent_t {
    if (evNext == evPool_.size()) { hipEvent_t e = nullptr; if (hipEventCreate(&e) != hipSuccess) return nullptr; evPool_.push_back(e); }
    return evPool_[evNext++];
  };
  std::vector<std::pair<hipEvent_t, hipEvent_t>> mfSpans, entSpans;
  static const bool traceOn = std::getenv("ZRA_ENC_TRACE") != nullptr;   // bring-up: timeline of the launches on stderr
  hipEvent_t superDone[2] = {nullptr, nullptr};
  // stream B starts after the counters are cleared and after whatever the caller queued on the engine stream
  { hipEvent_t e0 = ev(); if (!e0) return zerr(1); HIPCHK(hipEventRecord(e0, stream_)); HIPCHK(hipStreamWaitEvent(stream2_, e0, 0)); }

  for (uint64_t S = 0; S < nSuper; S++) {
    const uint64_t F0 = S * SBIG;
    const uint32_t n = (uint32_t)std::min<uint64_t>(SBIG, nFramesTotal - F0);
    const int c = (int)(S % nCtx);
    EncCtx& x = encCtx_[c];
    ZraEncArgs a = base;
    a.firstFrame = (uint32_t)F0; a.nFrames = n;
    a.tables = sh.tables.as<uint32_t>(); a.lits = sh.lits.as<uint8_t>(); a.slots = sh.slots.as<uint8_t>(); a.entWork = sh.work.as<uint8_t>();
    a.seqs = x.seqs.as<uint64_t>();
    a.state = x.misc.as<ZraEncFrameState>();
    a.blockOut = (ZraEncBlockOut*)(x.misc.as<uint8_t>() + (size_t)SBIG * sizeof(ZraEncFrameState));
    a.contentCk = x.ck.as<uint32_t>();
    a.sizes = x.sizes.as<uint64_t>();
    a.mfQueue = dPerSuper + 8 * S; a.mfStarted = a.mfQueue + 1; a.entQueue = a.mfQueue + 2; a.scanDone = a.mfQueue + 3; a.gatherJ = a.mfQueue + 4; a.gatherDone = a.mfQueue + 5;
    a.entDone = dPerSub + 4 * S * subsPerSuper; a.gQueue = a.entDone + subsPerSuper; a.gCopied = a.gQueue + subsPerSuper;
    a.mfDone = pipeMode == 1 ? a.gCopied + subsPerSuper : nullptr;
    a.offsets = x.sizes.as<uint64_t>() + SBIG;
    // the context's per-frame scratch is free once the last sub-batch that used it has been gathered
    if (superDone[c]) HIPCHK(hipStreamWaitEvent(stream_, superDone[c], 0));
    // no frame of this launch is published yet (the stamps of an earlier call may still sit in the block records)
    HIPCHK(hipMemsetAsync(a.blockOut, 0, (size_t)n * sizeof(ZraEncBlockOut), stream_));
    hipEvent_t m0 = ev(), m1 = ev();
    if (!m0 || !m1) return zerr(1);
    HIPCHK(hipEventRecord(m0, stream_));
    const uint32_t mfGrid = std::min<uint32_t>(n, nSlots);
    if (useFlagsWave) hipLaunchKernelGGL(zra_mf_dfast_fl_kernel, dim3(mfGrid), dim3(64), filterBytes + spanLds, stream_, a, fw, 0u, 0xFFFFFFFFu, 0u);
    else if (useLs) {
      ZraEncArgs al = a; al.mfFilter |= (lsBytes / 64) << 16;
      hipLaunchKernelGGL(zra_mf_dfast_ls_kernel, dim3(mfGrid), dim3(64), lsBytes + filterBytes, stream_, al, 0u, 0xFFFFFFFFu, 0u);
    } else hipLaunchKernelGGL(zra_mf_dfast_kernel, dim3(mfGrid), dim3(64), filterBytes, stream_, a, 0u, 0xFFFFFFFFu, 0u);
    HIPCHK(hipGetLastError());                        // (a launch that failed would leave stream B waiting for waves that never start)
    HIPCHK(hipEventRecord(m1, stream_));
    mfSpans.push_back({m0, m1});
    // a short last frame whose cparams select another strategy: parsed by the generic kernel (table slot 0 is free by then), then published
    const bool oddTail = tailSize && F0 + n == nFramesTotal && tail.strategy != 2;
    if (oddTail) {
      ZraEncArgs at = a; at.mfQueue = nullptr;
      if (tail.strategy >= 3 && tail.strategy <= 5) hipLaunchKernelGGL(zra_mf_hc_kernel, dim3(1), dim3(64), 0, stream_, at, 0u, (uint32_t)(n - 1), 0u);
      else hipLaunchKernelGGL(mfGeneric, dim3(1), dim3(64), 0, stream_, at, 0u, (uint32_t)(n - 1), 0u, 1u);
      HIPCHK(hipStreamWriteValue32(stream_, &a.blockOut[n - 1].ready, a.readyStamp, 0));
    }
    if (pipeMode == 1) {
      // ---- round 4's overlap, with this round's kernels: per sub-batch of 8192 frames, released by the match finder's own count of
      // finished frames, one entropy launch + scan + gather on stream B. The stage's workgroups come and go: they sit where a CU has
      // room beside the finder's waves, and nowhere when there is nothing to do.
      // (round 6: the checksum kernel — 4 x n lanes, ~2 ms of the whole chip — only once every wave of the finder is resident. Queued at once it
      //  could reach the CUs first, the dispatcher then placed the finder's waves unevenly (20 on most CUs, 12-18 on the ones the checksum
      //  blocks were leaving), and a CU with 20 has no room for the entropy workgroup: one process of the round ended 31 ms behind that way,
      //  profiles/r06_experiments.md §0)
      // (no wait of the finder depends on stream B inside a super-batch, so this cannot hang; with a bring-up ZRA_MF_WAVES beyond what a CU
      //  holds — 21 by LDS — the last waves start when the first ones leave, and the stage then runs BEHIND the finder instead of beside it)
      HIPCHK(hipStreamWaitValue32(stream2_, a.mfStarted, mfGrid, hipStreamWaitValueGte, 0xFFFFFFFFu));
      if (checksum)
        hipLaunchKernelGGL(zra_content_ck_kernel, dim3((n * 4 + 255) / 256), dim3(256), 0, stream2_, dIn, (u64)inSize, frameSize, (u32)F0, n, a.contentCk);
      const uint32_t nSub1 = (n + SB - 1) / SB;
      hipEvent_t eLast = nullptr;
      for (uint32_t j = 0; j < nSub1; j++) {
        const uint32_t j0 = j * SB, nbj = std::min<uint32_t>(SB, n - j0);
        const bool hasOdd = oddTail && j == nSub1 - 1;
        HIPCHK(hipStreamWaitValue32(stream2_, a.mfDone + j, nbj - (hasOdd ? 1u : 0u), hipStreamWaitValueGte, 0xFFFFFFFFu));
        if (hasOdd) { hipEvent_t te = ev(); if (!te) return zerr(1); HIPCHK(hipEventRecord(te, stream_)); HIPCHK(hipStreamWaitEvent(stream2_, te, 0)); }
        ZraEncArgs aj = a;
        aj.firstFrame = (uint32_t)(F0 + j0); aj.nFrames = nbj;
        aj.seqs = a.seqs + (size_t)j0 * seqStride; aj.state = a.state + j0; aj.blockOut = a.blockOut + j0;
        aj.contentCk = a.contentCk + j0; aj.sizes = a.sizes + j0;
        aj.readyStamp = 0; aj.slotRing = (uint32_t)std::min<uint64_t>(SB, slotRing); aj.entSubFrames = SB;
        aj.entQueue = a.entDone + j;                     // (a zeroed word per sub-batch: the launch's frame queue)
        uint64_t* dOffsets = x.sizes.as<uint64_t>() + SBIG + j0;
        hipEvent_t e0 = ev(), e1 = ev(); if (!e0 || !e1) return zerr(1);
        HIPCHK(hipEventRecord(e0, stream2_));
        if (entSplit) {
          aj.entRec = sh.rec.as<ZraEntRec>();
          hipLaunchKernelGGL(zra_entropy_front_kernel, dim3(std::min<uint32_t>(nbj, entGrid)), dim3(256), 0, stream2_, aj, 0u);
          // frames per wave of the chain kernel (6 / 3 / 2 / 1: the same LDS per CU as 1 / 2 / 3 / 6 waves)
          // (5 by default: 17,640 B of LDS, what is left of a CU beside 19 finder waves; 6 needs the room 18 waves leave)
          static const uint32_t chainG = std::getenv("ZRA_CHAIN_G") ? (uint32_t)std::atoi(std::getenv("ZRA_CHAIN_G")) : (wavesPerCU >= 19 ? 5u : 6u);
          const uint32_t cg = chainG <= 1 ? 1u : chainG == 2 ? 2u : chainG <= 4 ? 3u : chainG == 5 ? 5u : 6u;
          const auto chainK = cg == 1 ? zra_ent_chain1_kernel : cg == 2 ? zra_ent_chain2_kernel : cg == 3 ? zra_ent_chain3_kernel : cg == 5 ? zra_ent_chain5_kernel : zra_ent_chain_kernel;
          hipLaunchKernelGGL(chainK, dim3((nbj + cg - 1) / cg), dim3(64), 0, stream2_, aj);
          ZraEncArgs ab = aj; ab.entQueue = a.gQueue + j;   // (another zeroed word of the sub-batch: the BACK launch's frame queue)
          hipLaunchKernelGGL(zra_entropy_back_kernel, dim3(std::min<uint32_t>(nbj, entGrid)), dim3(256), 0, stream2_, ab, 0u);
        } else
          hipLaunchKernelGGL(zra_entropy_kernel, dim3(std::min<uint32_t>(nbj, entGrid)), dim3(256), 0, stream2_, aj, 0u);
        HIPCHK(hipEventRecord(e1, stream2_));
        entSpans.push_back({e0, e1});
        hipLaunchKernelGGL(zra_scan_sizes_kernel, dim3(1), dim3(64), 0, stream2_, aj.sizes, nbj, dOffsets, dRunning, (const u32*)nullptr);
        hipLaunchKernelGGL(zra_gather_frames_kernel, dim3(nbj), dim3(256), 0, stream2_, aj.slots, slotStride, aj.sizes, dOffsets, dBody,
                           bodyBase0, dEntries ? dEntries : nullptr, (u32)(F0 + j0), dSizes, (const u32*)nullptr);
        eLast = e1;
      }
      hipEvent_t done = ev(); if (!done) return zerr(1);
      HIPCHK(hipEventRecord(done, stream2_));
      superDone[c] = done;
      (void)eLast;
      continue;
    }
    // stream B: once every wave of the match finder is resident (they are placed first, side by side: what is left of each CU is one
    // contiguous piece), the content checksums, then the queue-driven entropy stage. pipeMode 0: only behind the whole match-finder
    // launch, with as many workgroups as the device holds; pipeMode 2: at once, resident beside the finder
    if (pipeMode == 0) HIPCHK(hipStreamWaitEvent(stream2_, m1, 0));
    else HIPCHK(hipStreamWaitValue32(stream2_, a.mfStarted, mfGrid, hipStreamWaitValueGte, 0xFFFFFFFFu));
    if (checksum)
      hipLaunchKernelGGL(zra_content_ck_kernel, dim3((n * 4 + 255) / 256), dim3(256), 0, stream2_, dIn, (u64)inSize, frameSize, (u32)F0, n, a.contentCk);
    hipEvent_t e0 = ev(), e1 = ev(); if (!e0 || !e1) return zerr(1);
    HIPCHK(hipEventRecord(e0, stream2_));
    hipLaunchKernelGGL(zra_entropy_kernel, dim3(std::min<uint32_t>(n, entGrid)), dim3(256), 0, stream2_, a, 0u);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(e1, stream2_));
    entSpans.push_back({e0, e1});
    // (the entropy launch ends when every frame of the super-batch sits in the archive: the context's scratch and the slot ring are free)
    superDone[c] = e1;
  }
  uint64_t total = 0;
  HIPCHK(hipMemcpyAsync(&total, dRunning, 8, hipMemcpyDeviceToHost, stream2_));
  HIPCHK(hipStreamSynchronize(stream2_));
  HIPCHK(hipStreamSynchronize(stream_));
  HIPCHK(hipGetLastError());
  {
    uint32_t aborted = 0;
    HIPCHK(hipMemcpy(&aborted, dAbort, 4, hipMemcpyDeviceToHost));
    if (aborted) return zerr(1);                      // a wait between the two persistent kernels ran out of patience
  }
  double kernelMs = 0;
  kstats_[0] = kstats_[1] = kstats_[2] = kstats_[3] = 0;
  for (auto& sp : mfSpans) { float m = 0; if (hipEventElapsedTime(&m, sp.first, sp.second) == hipSuccess) { kstats_[0] += m; kstats_[1] += 1; kernelMs += m; } }
  for (auto& sp : entSpans) { float m = 0; if (hipEventElapsedTime(&m, sp.first, sp.second) == hipSuccess) { kstats_[2] += m; kstats_[3] += 1; } }
  lastKernelMs_ = kernelMs;
  // (the head — sums, per-XCD and per-CU counts of the match finder — then the entropy stage's block; the per-wave records stay on the device)
  mfTeleDev_ = base.mfTele;                            // fetched by launch_telemetry() if anybody asks
  if (traceOn) {                                      // bring-up: timeline relative to the first match-finder launch
    for (auto& sp : mfSpans) { float a0 = 0, a1 = 0; (void)hipEventElapsedTime(&a0, mfSpans[0].first, sp.first); (void)hipEventElapsedTime(&a1, mfSpans[0].first, sp.second); std::fprintf(stderr, "mf  %8.2f .. %8.2f ms\n", a0, a1); }
    for (auto& sp : entSpans) { float a0 = 0, a1 = 0; (void)hipEventElapsedTime(&a0, mfSpans[0].first, sp.first); (void)hipEventElapsedTime(&a1, mfSpans[0].first, sp.second); std::fprintf(stderr, "ent %8.2f .. %8.2f ms\n", a0, a1); }
  }
  *bodySize = total;
  return ok();
}

size_t Engine::launch_telemetry(uint64_t* out, size_t cap) {
  if (mfTeleDev_) {
    // (the head — sums, per-XCD and per-CU counts of the match finder — then the entropy stage's block; the per-wave records stay on the device)
    (void)hipSetDevice(device_);
    mfTele_.resize(ZRA_TELE_HEAD + 8 + ZRA_TELE_CUKEYS);
    const bool okT = hipMemcpy(mfTele_.data(), mfTeleDev_, 8 * (size_t)ZRA_TELE_HEAD, hipMemcpyDeviceToHost) == hipSuccess &&
                     hipMemcpy(mfTele_.data() + ZRA_TELE_HEAD, mfTeleDev_ + ZRA_TELE_ENT, 8 * (size_t)(8 + ZRA_TELE_CUKEYS), hipMemcpyDeviceToHost) == hipSuccess;
    if (!okT) { (void)hipGetLastError(); mfTele_.clear(); }
    mfTeleDev_ = nullptr;
  }
  const size_t n = mfTele_.size() < cap ? mfTele_.size() : cap;
  for (size_t i = 0; i < n; i++) out[i] = mfTele_[i];
  return n;
}

uint32_t Engine::debug_read_seqs(uint32_t frame, uint64_t* out, uint32_t cap, uint32_t meta[3]) {
  if (!dbgB_ || frame >= dbgB_) return 0;
  (void)hipSetDevice(device_);
  (void)hipDeviceSynchronize();
  EncCtx& x = encCtx_[0];
  ZraEncBlockOut bo{};
  (void)hipMemcpy(&bo, x.misc.as<uint8_t>() + (size_t)dbgB_ * sizeof(ZraEncFrameState) + (size_t)frame * sizeof(ZraEncBlockOut), sizeof(bo), hipMemcpyDeviceToHost);
  meta[0] = bo.nbSeq; meta[1] = bo.lastLL; meta[2] = bo.skip;
  const uint32_t n = bo.nbSeq < cap ? bo.nbSeq : cap;
  if (n) (void)hipMemcpy(out, x.seqs.as<uint64_t>() + (size_t)frame * dbgSeqStride_, (size_t)n * 8, hipMemcpyDeviceToHost);
  return n;
}

Status Engine::compress_device(const uint8_t* dIn, size_t inSize, uint8_t* dOut, size_t* outSize, int level, uint32_t frameSize, bool checksum) {
  HIPCHK(hipSetDevice(device_));
  if (frameSize == 0) return zerr(42);
  const uint32_t tableSize = zra_fmt::table_size(inSize, frameSize);
  const size_t headerSize = zra_fmt::kFixedSize + (size_t)tableSize * zra_fmt::kEntrySize;
  size_t bodySize = 0;
  // body is gathered straight behind the (not yet written) header; entries are offsets relative to the body start
  Status s = compress_impl(dIn, inSize, dOut + headerSize, 0, dOut + zra_fmt::kFixedSize, nullptr, &bodySize, level, frameSize, checksum);
  if (s.zra) return s;
  if (headerSize + bodySize >= zra_fmt::kMaxCompressedSize) return {kCompressedTooLarge, 0};   // zra.cpp:227
  // end sentinel + CRC-32 on the host (the table is 5 B/frame: 1.3 MiB at 16 GiB / 64 KiB)
  std::vector<uint8_t> hdr(headerSize);
  if (tableSize > 1)
    HIPCHK(hipMemcpyAsync(hdr.data() + zra_fmt::kFixedSize, dOut + zra_fmt::kFixedSize, (size_t)(tableSize - 1) * 5, hipMemcpyDeviceToHost, stream_));
  HIPCHK(hipStreamSynchronize(stream_));
  zra_fmt::write_fixed(hdr.data(), inSize, tableSize, frameSize, 0);
  zra_fmt::entry_put(hdr.data() + zra_fmt::kFixedSize + (size_t)(tableSize - 1) * 5, bodySize);
  zra_fmt::wr32(hdr.data() + 14, zra_fmt::header_hash(hdr.data(), hdr.data() + zra_fmt::kFixedSize));
  HIPCHK(hipMemcpyAsync(dOut, hdr.data(), zra_fmt::kFixedSize, hipMemcpyHostToDevice, stream_));
  HIPCHK(hipMemcpyAsync(dOut + zra_fmt::kFixedSize + (size_t)(tableSize - 1) * 5, hdr.data() + zra_fmt::kFixedSize + (size_t)(tableSize - 1) * 5, 5,
                        hipMemcpyHostToDevice, stream_));
  HIPCHK(hipStreamSynchronize(stream_));
  *outSize = headerSize + bodySize;
  return ok();
}

// ------------------------------------------------------------------ host-pointer helpers used by the C/C++ API
// compress_host: zra_hostpipe.hip
Status Engine::compress_frames_host(const uint8_t* hIn, size_t n, uint8_t* hBody, std::vector<uint64_t>& sizes, size_t* bodySize,
                                    int level, uint32_t frameSize, bool checksum) {
  HIPCHK(hipSetDevice(device_));
  const size_t nFrames = (n + frameSize - 1) / frameSize;
  const size_t cap = zra_fmt::compress_bound(frameSize) * nFrames;
  if (!hostIn_.reserve(n + 64) || !hostOut_.reserve(cap + nFrames * 8 + 64)) return zerr(64);
  HIPCHK(hipMemcpyAsync(hostIn_.p, hIn, n, hipMemcpyHostToDevice, stream_));
  uint64_t* dSizes = (uint64_t*)(hostOut_.as<uint8_t>() + ((cap + 15) & ~(size_t)15));
  if (!hostOut_.reserve(((cap + 15) & ~(size_t)15) + nFrames * 8 + 64)) return zerr(64);
  dSizes = (uint64_t*)(hostOut_.as<uint8_t>() + ((cap + 15) & ~(size_t)15));
  Status s = compress_frames(hostIn_.as<uint8_t>(), n, hostOut_.as<uint8_t>(), dSizes, bodySize, level, frameSize, checksum);
  if (s.zra) return s;
  sizes.resize(nFrames);
  HIPCHK(hipMemcpyAsync(hBody, hostOut_.p, *bodySize, hipMemcpyDeviceToHost, stream_));
  HIPCHK(hipMemcpyAsync(sizes.data(), dSizes, nFrames * 8, hipMemcpyDeviceToHost, stream_));
  HIPCHK(hipStreamSynchronize(stream_));
  return ok();
}

Status Engine::decode_host(const uint8_t* hSpan, size_t spanSize, const std::vector<uint64_t>& starts, const std::vector<uint64_t>& ends,
                           uint32_t frameSize, uint64_t total, uint8_t* hOut, size_t skip, size_t size, bool wholeArchive) {
  HIPCHK(hipSetDevice(device_));
  const uint32_t nFrames = (uint32_t)starts.size();
  if (nFrames == 0) return ok();
  std::vector<uint64_t> se((size_t)nFrames * 2), oo(nFrames);
  std::vector<uint32_t> ex(nFrames);
  for (uint32_t i = 0; i < nFrames; i++) {
    se[2 * (size_t)i] = starts[i]; se[2 * (size_t)i + 1] = ends[i];
    const uint64_t o = (uint64_t)i * frameSize;
    oo[i] = wholeArchive ? std::min<uint64_t>(o, total) : o;        // a slot past the declared size has no room and no address of its own
    ex[i] = o >= total ? 0 : (uint32_t)std::min<uint64_t>(frameSize, total - o);
  }
  if (wholeArchive && skip == 0 && frameSize && (size_t)nFrames >= 2 * std::max<size_t>(1, host_chunk_bytes() / frameSize) && size == std::min<uint64_t>(total, (uint64_t)nFrames * frameSize)) {
    bool fallBack = false;
    Status s = decode_host_pipelined(hSpan, starts, ends, frameSize, total, hOut, &fallBack);
    if (!fallBack) return s;
  }
  // ---- small calls (round 6; the reference's own calling convention, one query of a few KiB through ZraDecompressRA): the four pageable
  // host-to-device copies (each staged and waited for by the runtime), the synchronisation behind them and the pageable copy back were
  // ~40 % of such a call. Here the job arrays and the compressed span travel in ONE copy from page-locked memory, the kernel is queued
  // straight behind it, and the answer comes back through page-locked memory: one copy in, one launch, one copy out, one wait.
  constexpr size_t kSmallSpan = 768u << 10, kSmallOut = 1u << 20, kSmallJobs = 16;
  static const bool smallHostOff = std::getenv("ZRA_HOST_SMALL") && std::atoi(std::getenv("ZRA_HOST_SMALL")) == 0;
  if (!smallHostOff && !wholeArchive && nFrames <= kSmallJobs && spanSize <= kSmallSpan && size <= kSmallOut && (uint64_t)nFrames * frameSize <= (64ull << 20)) {
    const size_t metaBytes = (size_t)kSmallJobs * (16 + 8 + 4 + 4);                   // frameOff pairs, outOff, expect (padded)
    const size_t inBytes = metaBytes + ((spanSize + 63) & ~(size_t)63);
    if (!pinSmall_) {
      void* pq = nullptr;
      if (hipHostMalloc(&pq, metaBytes + kSmallSpan + 64 + kSmallOut + 64, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); pq = nullptr; }
      pinSmall_ = (uint8_t*)pq;
    }
    if (pinSmall_ && hostIn_.reserve(metaBytes + kSmallSpan + 128) && hostOut_.reserve((size_t)nFrames * frameSize + 64)) {
      uint8_t* const hp = pinSmall_;
      std::memcpy(hp, se.data(), se.size() * 8);
      std::memcpy(hp + kSmallJobs * 16, oo.data(), (size_t)nFrames * 8);
      std::memcpy(hp + kSmallJobs * 24, ex.data(), (size_t)nFrames * 4);
      std::memcpy(hp + metaBytes, hSpan, spanSize);
      uint8_t* const dIn = hostIn_.as<uint8_t>();
      HIPCHK(hipMemcpyAsync(dIn, hp, inBytes, hipMemcpyHostToDevice, stream_));
      Status s = decode_jobs(dIn + metaBytes, spanSize, (const uint64_t*)dIn, hostOut_.as<uint8_t>(), (const uint64_t*)(dIn + kSmallJobs * 16),
                             (const uint32_t*)(dIn + kSmallJobs * 24), nFrames, frameSize, 2, 0);
      if (s.zra) return s;
      if (skip + size > (uint64_t)nFrames * frameSize) return {kOutOfBounds, 0};
      if (size) {
        uint8_t* const ho = pinSmall_ + metaBytes + kSmallSpan + 64;
        HIPCHK(hipMemcpyAsync(ho, hostOut_.as<uint8_t>() + skip, size, hipMemcpyDeviceToHost, stream_));
        HIPCHK(hipStreamSynchronize(stream_));
        std::memcpy(hOut, ho, size);
      }
      return ok();
    }
  }
  // whole-archive mode never writes at or beyond `total` (slots past it have no room), whatever the header's frameSize claims
  if (!hostIn_.reserve(spanSize + 64) || !hostOut_.reserve((wholeArchive ? (size_t)total : (size_t)nFrames * frameSize) + 64) || !frameOff_.reserve(se.size() * 8) ||
      !outOff_.reserve((size_t)nFrames * 8) || !expect_.reserve((size_t)nFrames * 4))
    return zerr(64);
  HIPCHK(hipMemcpyAsync(hostIn_.p, hSpan, spanSize, hipMemcpyHostToDevice, stream_));
  HIPCHK(hipMemcpyAsync(frameOff_.p, se.data(), se.size() * 8, hipMemcpyHostToDevice, stream_));
  HIPCHK(hipMemcpyAsync(outOff_.p, oo.data(), (size_t)nFrames * 8, hipMemcpyHostToDevice, stream_));
  HIPCHK(hipMemcpyAsync(expect_.p, ex.data(), (size_t)nFrames * 4, hipMemcpyHostToDevice, stream_));
  HIPCHK(hipStreamSynchronize(stream_));
  Status s = decode_jobs(hostIn_.as<uint8_t>(), spanSize, frameOff_.as<uint64_t>(), hostOut_.as<uint8_t>(), outOff_.as<uint64_t>(),
                         expect_.as<uint32_t>(), nFrames, frameSize, 2, wholeArchive ? total : 0);
  if (s.zra) return s;
  if (wholeArchive && lastProducedTotal_ != ~0ull) {
    // frames that regenerated another size than the header's frameSize (corrupted or foreign archive): the sequential tail has packed
    // them back to back like the reference's one multi-frame call (zra.cpp:249) — what it wrote, less or MORE than the nominal slots
    // add up to (a frameSize field damaged downwards), is what reaches the caller; never more than the declared size
    const uint64_t have = std::min<uint64_t>(total, lastProducedTotal_);
    size = have > skip ? (size_t)(have - skip) : 0;
  } else if (skip + size > (uint64_t)nFrames * frameSize) return {kOutOfBounds, 0};
  if (size) HIPCHK(hipMemcpyAsync(hOut, hostOut_.as<uint8_t>() + skip, size, hipMemcpyDeviceToHost, stream_));
  HIPCHK(hipStreamSynchronize(stream_));
  return ok();
}

}  // namespace zra_eng
