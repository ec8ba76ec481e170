// zra_amd — host engine implementation (decode side + shared utility kernels).
#include "zra_engine.h"
#include "zra_dev.h"
#include "zra_format.h"
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

extern "C" __global__ void zra_dec_parse_kernel(ZraDecodeArgs a);
extern "C" __global__ void zra_dec_huf_kernel(ZraDecodeArgs a);
extern "C" __global__ void zra_dec_chain_kernel(ZraDecodeArgs a);
extern "C" __global__ void zra_dec_chain_lds_kernel(ZraDecodeArgs a);
extern "C" __global__ void zra_dec_exec_kernel(ZraDecodeArgs a);
extern "C" __global__ void zra_dec_parse_all_kernel(ZraDecodeArgs a);
extern "C" __global__ void zra_dec_exec_all_kernel(ZraDecodeArgs a);
extern "C" __global__ void zra_ra_small_kernel(ZraDecodeArgs a, uint32_t* bail, const uint32_t* expect, uint32_t jobBase, unsigned long long* result);

using namespace zra_dev;

namespace zra_fmt {
uint32_t crc32(uint32_t crc, const void* data, size_t n) {
  static uint32_t T[8][256];
  static bool init = false;
  if (!init) {
    for (uint32_t i = 0; i < 256; i++) {
      uint32_t c = i;
      for (int k = 0; k < 8; k++) c = (c & 1) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
      T[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; i++)
      for (int t = 1; t < 8; t++) T[t][i] = (T[t - 1][i] >> 8) ^ T[0][T[t - 1][i] & 0xFF];
    init = true;
  }
  const uint8_t* p = (const uint8_t*)data;
  crc = ~crc;
  while (n >= 8) {
    uint32_t a = rd32(p) ^ crc, b = rd32(p + 4);
    crc = T[7][a & 0xFF] ^ T[6][(a >> 8) & 0xFF] ^ T[5][(a >> 16) & 0xFF] ^ T[4][a >> 24] ^
          T[3][b & 0xFF] ^ T[2][(b >> 8) & 0xFF] ^ T[1][(b >> 16) & 0xFF] ^ T[0][b >> 24];
    p += 8; n -= 8;
  }
  while (n--) crc = T[0][(crc ^ *p++) & 0xFF] ^ (crc >> 8);
  return ~crc;
}
}  // namespace zra_fmt

// =================================================================================================
// utility kernels
// =================================================================================================
namespace {

// seek table (5-byte entries inside the archive) -> u64 frame offsets + trivial output layout
__global__ void zra_jobs_from_seektable_kernel(const u8* table, u32 nFrames, u32 frameSize, u64 total,
                                               u64* frameOff, u64* outOff, u32* expect) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= nFrames) {
    const u8* e = table + (size_t)i * 5;
    frameOff[i] = (u64)ld32(e) | ((u64)e[4] << 32);
  }
  if (i < nFrames) {
    // a frame whose slot starts at or beyond the declared size gets no room at all (an inflated tableSize or a shrunk
    // uncompressedSize in a crafted header): the decoder then reports dstSize_tooSmall for it, like the reference's single
    // multi-frame call running out of destination (zra.cpp:249), and nothing is written past `total`
    u64 o = (u64)i * frameSize;
    outOff[i] = o < total ? o : total;
    expect[i] = o >= total ? 0u : (u32)(total - o < frameSize ? total - o : frameSize);
  }
}

// content-checksum verification of decoded frames: 4 lanes per frame (16 frames per wave)
// Frame end, in the order of ZSTD_decompressFrame: content checksum over the bytes the frame actually regenerated, then (ours, the
// frames being decoded side by side into fixed slots) the regenerated size against the slot: ZE_SIZE_MISMATCH is not a zstd code —
// the caller either re-decodes sequentially from that frame (whole-archive decode: one multi-frame zstd call in the reference packs
// frames back to back whatever they regenerate) or reports corruption_detected.
constexpr u32 ZE_SIZE_MISMATCH = 255;
__global__ void zra_xxh64_verify_kernel(const u8* out, const u64* outOff, const u32* expect, const u32* produced, const u32* frameMeta,
                                        u32* status, u32 nFrames) {
  const u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
  const u32 f = gid >> 2; const int j = gid & 3;
  const bool live = f < nFrames && status[f] == 0 && frameMeta[2 * (size_t)f] != 2;   // 2: stopped early (random access), nothing to check
  const bool active = live && frameMeta[2 * (size_t)f] == 1;
  const u8* p = active ? out + outOff[f] : out;
  const u32 n = active ? produced[f] : 0;
  const u64 h = zra_xxh64_quad(p, n, j);
  if (live && j == 0) {
    if (active && (u32)h != frameMeta[2 * (size_t)f + 1]) status[f] = ZE_CHECKSUM_WRONG;
    else if (expect && produced[f] != expect[f]) status[f] = ZE_SIZE_MISMATCH;
  }
}

// result[0] = min over failing frames of (frame << 8 | code); ~0 when all succeeded
__global__ void zra_first_error_kernel(const u32* status, u32 nFrames, u32 jobBase, unsigned long long* result) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nFrames && status[i]) atomicMin(result, ((unsigned long long)(jobBase + i) << 8) | (status[i] & 0xFF));
}

// ---- batched random access: the jobs of a batch are built on the device from the query arrays and the archive's own seek table
// (zra.cpp:265-269 per query: first frame offset / frameSize, frames touched, head skip, tail length)
struct RaPlan {            // device scratch of one batch
  u32* cnt;                // [nFrames] slices that touch the frame
  u32* need;               // [nFrames] bytes of the frame the batch needs (max over its slices of the slice end)
  u32* slot;               // [nFrames] dense number of the frame among the touched ones
  u32* cursor;             // [nFrames] fill cursor of the frame's slice list
  u32* totals;             // {touched frames, slices}
};
// pass 1: every query marks the frames it touches
__global__ void zra_ra_count_kernel(const u64* q, u32 nq, u64 fs, RaPlan P) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  const u64 off = q[4 * (size_t)i], size = q[4 * (size_t)i + 1];
  if (!size) return;
  const u64 f0 = off / fs, f1 = (off + size - 1) / fs;
  for (u64 f = f0; f <= f1; f++) {
    atomicAdd(&P.cnt[f], 1u);
    const u32 end = f == f1 ? (u32)((off + size - 1) % fs) + 1 : (u32)fs;
    atomicMax(&P.need[f], end);
  }
}
// pass 2 (one workgroup): exclusive scans over the frames -> dense slots + slice-list bases, and the decode job of every touched
// frame: compressed span from the 5-byte seek-table entries, destination slot inside the pass-sized scratch window, bytes to produce
// compressed span of frame f from the 5-byte entries, relative to the body bytes this device holds ([bodyBase, ...) of the archive's
// body: a shard of a distributed archive holds its own frames only); a span that starts before them comes out inverted (refused as
// srcSize_wrong by the decoder, like any span outside the buffer)
__device__ __forceinline__ void ra_frame_span(const u8* table, u64 f, u64 bodyBase, u64* so, u64* se) {
  const u8* e = table + (size_t)f * 5;
  const u64 a = (u64)ld32(e) | ((u64)e[4] << 32), b = (u64)ld32(e + 5) | ((u64)e[9] << 32);
  if (a < bodyBase || b < bodyBase) { *so = 1; *se = 0; }
  else { *so = a - bodyBase; *se = b - bodyBase; }
}
__global__ void __launch_bounds__(1024) zra_ra_plan_kernel(RaPlan P, u32 nFrames, const u8* table, u64 bodyBase, u64 fs, u64 total, u32 passSlots, u32 fullFrames,
                                                           u64* frameOff, u64* outOff, u32* outCap, u32* limit, u32* pieceBase) {
  __shared__ u32 sT[1024], sP[1024];
  const u32 tid = threadIdx.x;
  const u32 per = (nFrames + 1023) / 1024;
  const u32 b0 = tid * per, b1 = min(nFrames, b0 + per);
  u32 t = 0, p = 0;
  for (u32 f = b0; f < b1; f++) { const u32 c = P.cnt[f]; t += c != 0; p += c; }
  sT[tid] = t; sP[tid] = p;
  __syncthreads();
  for (u32 d = 1; d < 1024; d <<= 1) {                     // Hillis-Steele inclusive scan of the 1024 partials
    const u32 xt = tid >= d ? sT[tid - d] : 0, xp = tid >= d ? sP[tid - d] : 0;
    __syncthreads();
    sT[tid] += xt; sP[tid] += xp;
    __syncthreads();
  }
  u32 st = sT[tid] - t, sp = sP[tid] - p;                  // exclusive
  for (u32 f = b0; f < b1; f++) {
    const u32 c = P.cnt[f];
    if (!c) continue;
    P.slot[f] = st;
    ra_frame_span(table, f, bodyBase, &frameOff[2 * (size_t)st], &frameOff[2 * (size_t)st + 1]);
    const u64 o = (u64)f * fs;
    const u32 expect = o >= total ? 0u : (u32)(total - o < fs ? total - o : fs);
    outOff[st] = (u64)(st % passSlots) * fs;
    outCap[st] = expect;
    limit[st] = fullFrames ? expect : min(P.need[f], expect);
    pieceBase[st] = sp;
    st++; sp += c;
  }
  if (tid == 1023) { P.totals[0] = sT[1023]; P.totals[1] = sP[1023]; pieceBase[sT[1023]] = sP[1023]; }
}
// pass 3: every query writes its slices into the lists of the frames it touches
__global__ void zra_ra_fill_kernel(const u64* q, u32 nq, u64 fs, RaPlan P, const u32* pieceBase, ZraRaPiece* pieces) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  const u64 off = q[4 * (size_t)i], size = q[4 * (size_t)i + 1], dst = q[4 * (size_t)i + 2];
  if (!size) return;
  const u64 f0 = off / fs, f1 = (off + size - 1) / fs;
  u64 done = 0;
  for (u64 f = f0; f <= f1; f++) {
    const u32 srcOff = f == f0 ? (u32)(off % fs) : 0u;
    const u64 len = min<u64>(fs - srcOff, size - done);
    const u32 at = pieceBase[P.slot[f]] + atomicAdd(&P.cursor[f], 1u);
    ZraRaPiece pc; pc.dstOff = dst + done; pc.srcOff = srcOff; pc.len = (u32)len;
    pieces[at] = pc;
    done += len;
  }
}

// small batches (far fewer slices than the archive has frames): one decode job per slice, built from the query alone — no pass over
// the frames of the archive, no count/scan, nothing read back. A frame two slices share is decoded once per slice.
__global__ void zra_ra_direct_kernel(const u64* q, u32 nq, u32 nPieces, u64 fs, u64 total, const u8* table, u64 bodyBase, u32 fullFrames, u64* frameOff, u64* outOff,
                                     u32* outCap, u32* limit, u32* pieceBase, ZraRaPiece* pieces) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) pieceBase[nPieces] = nPieces;
  if (i >= nq) return;
  const u64 off = q[4 * (size_t)i], size = q[4 * (size_t)i + 1], dst = q[4 * (size_t)i + 2];
  if (!size) return;
  const u64 f0 = off / fs, f1 = (off + size - 1) / fs;
  u32 st = (u32)q[4 * (size_t)i + 3];
  u64 done = 0;
  for (u64 f = f0; f <= f1; f++, st++) {
    ra_frame_span(table, f, bodyBase, &frameOff[2 * (size_t)st], &frameOff[2 * (size_t)st + 1]);
    const u64 o = f * fs;
    const u32 expect = o >= total ? 0u : (u32)(total - o < fs ? total - o : fs);
    const u32 srcOff = f == f0 ? (u32)(off % fs) : 0u;
    const u64 len = min<u64>(fs - srcOff, size - done);
    outOff[st] = (u64)st * fs;
    outCap[st] = expect;
    limit[st] = fullFrames ? expect : min((u32)(srcOff + len), expect);
    pieceBase[st] = st;
    ZraRaPiece pc; pc.dstOff = dst + done; pc.srcOff = srcOff; pc.len = (u32)len;
    pieces[st] = pc;
    done += len;
  }
}

}  // namespace

// =================================================================================================
namespace zra_eng {

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { last_hip_error = e_; return zerr(1); } } while (0)
static thread_local hipError_t last_hip_error = hipSuccess;

// bytes of device scratch all engines of the process hold (what the engine pool's cap looks at)
static std::atomic<uint64_t> g_scratchBytes{0};
uint64_t scratch_bytes_in_use() { return g_scratchBytes.load(); }

bool DevBuf::reserve(size_t n) {
  if (n <= cap) return true;
  if (p) { (void)hipFree(p); g_scratchBytes -= cap; p = nullptr; cap = 0; }
  size_t want = n + n / 8 + 256;
  // bring-up / test knob: ZRA_ALLOC_LIMIT_MIB makes any single reservation above the limit fail (memory_allocation paths without a full device)
  static const uint64_t limit = std::getenv("ZRA_ALLOC_LIMIT_MIB") ? (uint64_t)std::atoll(std::getenv("ZRA_ALLOC_LIMIT_MIB")) << 20 : ~0ull;
  // bring-up knob ZRA_ALLOC_MODE (round 5, the launch-time states of the match finder): 1 = hipDeviceMallocUncached, 2 = fine-grained
  static const int mode = std::getenv("ZRA_ALLOC_MODE") ? std::atoi(std::getenv("ZRA_ALLOC_MODE")) : 0;
  const hipError_t ea = want > limit ? hipErrorOutOfMemory : mode == 1 ? hipExtMallocWithFlags(&p, want, hipDeviceMallocUncached)
                        : mode == 2 ? hipExtMallocWithFlags(&p, want, hipDeviceMallocFinegrained) : hipMalloc(&p, want);
  if (ea != hipSuccess) { p = nullptr; cap = 0; (void)hipGetLastError(); return false; }
  cap = want; g_scratchBytes += cap;
  return true;
}
void DevBuf::release() { if (p) { (void)hipFree(p); g_scratchBytes -= cap; } p = nullptr; cap = 0; }

int parse_fixed_header(const uint8_t* b, HeaderInfo* h) {
  using namespace zra_fmt;
  if (rd32(b + 8) != kZraMagic || rd16(b + 12) > kVersion) return kHeaderInvalid;   // zra.cpp:144-145
  h->version = rd16(b + 12);
  h->size = rd32(b + 4) + 8;
  h->uncompressedSize = rd64(b + 18);
  h->frameSize = rd32(b + 30);
  h->metaOffset = (uint32_t)kFixedSize;
  h->metaSize = rd32(b + 34);
  h->seekTableOffset = h->metaOffset + h->metaSize;
  h->seekTableSize = rd32(b + 26) * (uint32_t)kEntrySize;
  if (h->version != 1) return kVersionLow;                                           // zra.cpp:156-162
  return 0;
}

Status Engine::create(Engine** out, int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return zerr(1);
  HIPCHK(hipSetDevice(device));
  Engine* e = new Engine();
  e->device_ = device;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) { delete e; return zerr(1); }
  e->numCUs_ = prop.multiProcessorCount;
  if (hipStreamCreateWithFlags(&e->stream_, hipStreamNonBlocking) != hipSuccess) { delete e; return zerr(1); }
  {
    // Stream B carries the persistent entropy stage (which also scans and gathers) beside the persistent match finder of stream A.
    // The runtime multiplexes streams onto a handful of hardware queues per priority class, and two streams that land on one queue run
    // their kernels one after the other — two persistent kernels that wait for each other must not: B gets the highest priority class,
    // a queue that stream A (normal priority) is never put on.
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi != lo) {
      if (hipStreamCreateWithPriority(&e->stream2_, hipStreamNonBlocking, hi) != hipSuccess) { delete e; return zerr(1); }
    } else if (hipStreamCreateWithFlags(&e->stream2_, hipStreamNonBlocking) != hipSuccess) { delete e; return zerr(1); }
  }
  if (hipEventCreate(&e->ev0_) != hipSuccess || hipEventCreate(&e->ev1_) != hipSuccess ||
      hipEventCreateWithFlags(&e->evWait_, hipEventDisableTiming) != hipSuccess) { delete e; return zerr(1); }
  for (auto& ev : e->evR_) if (hipEventCreate(&ev) != hipSuccess) { delete e; return zerr(1); }
  *out = e;
  return ok();
}

Status Engine::release_scratch() {
  HIPCHK(hipSetDevice(device_));
  HIPCHK(hipStreamSynchronize(stream_));
  HIPCHK(hipStreamSynchronize(stream2_));
  for (DevBuf* b : {&raPlan_, &raLimit_, &raPieceBase_, &raPieces_, &decFrames_, &decTables_, &decLists_, &decCounters_, &decLits_, &decSeqs_, &roundN_, &status_,
                    &produced_, &frameMeta_, &frameOff_, &outOff_, &expect_, &result_, &temp_, &qmeta_, &encScan_, &hostIn_, &hostOut_, &seqScratch_, &mfFlags_, &decBlkRecs_, &decBlkTables_, &decBlkLists_})
    b->release();
  mfTeleDev_ = nullptr;                                // (lived inside encScan_)
  for (auto& x : encCtx_) for (DevBuf* b : {&x.tables, &x.seqs, &x.lits, &x.work, &x.slots, &x.misc, &x.ck, &x.sizes, &x.rec}) b->release();
  decCountersClean_ = false;
  return ok();
}

Engine::~Engine() {
  (void)hipSetDevice(device_);
  if (stream_) (void)hipStreamSynchronize(stream_);
  for (DevBuf* b : {&raPlan_, &raLimit_, &raPieceBase_, &raPieces_, &decFrames_, &decTables_, &decLists_, &decCounters_, &decLits_, &decSeqs_, &roundN_, &status_, &produced_, &frameMeta_, &frameOff_, &outOff_, &expect_, &result_, &temp_, &qmeta_,
                    &encScan_, &hostIn_, &hostOut_, &seqScratch_, &mfFlags_, &decBlkRecs_, &decBlkTables_, &decBlkLists_})
    b->release();
  for (auto& x : encCtx_) for (DevBuf* b : {&x.tables, &x.seqs, &x.lits, &x.work, &x.slots, &x.misc, &x.ck, &x.sizes, &x.rec}) b->release();
  for (auto ev : evPool_) (void)hipEventDestroy(ev);
  for (auto ev : stageEv_) (void)hipEventDestroy(ev);
  if (stream2_) { (void)hipStreamSynchronize(stream2_); (void)hipStreamDestroy(stream2_); }
  for (auto st : pipeStreams_) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
  if (pinQ_) (void)hipHostFree(pinQ_);
  if (pinSmall_) (void)hipHostFree(pinSmall_);
  for (auto& ev : evR_) if (ev) (void)hipEventDestroy(ev);
  if (ev0_) (void)hipEventDestroy(ev0_);
  if (ev1_) (void)hipEventDestroy(ev1_);
  if (evWait_) (void)hipEventDestroy(evWait_);
  if (stream_) (void)hipStreamDestroy(stream_);
}

Status Engine::sync() { HIPCHK(hipSetDevice(device_)); HIPCHK(hipStreamSynchronize(stream_)); HIPCHK(hipStreamSynchronize(stream2_)); return ok(); }

Status Engine::wait_stream(hipStream_t producer) {
  HIPCHK(hipSetDevice(device_));
  HIPCHK(hipEventRecord(evWait_, producer));
  HIPCHK(hipStreamWaitEvent(stream_, evWait_, 0));
  HIPCHK(hipStreamWaitEvent(stream2_, evWait_, 0));
  return ok();
}

// One pass of the decoder over the jobs of `a0`: rounds of parse -> chain -> execute (a round = one compressed block of every
// unfinished frame) until no frame is left, then the content checksums and the first-error reduction.
// scratch of one decode pass over a.nFrames jobs (decode records, table slots, stage lists, literal / sequence scratch, result words)
Status Engine::decode_scratch(ZraDecodeArgs& a, uint32_t maxFrameBytes) {
  const uint32_t n = a.nFrames;
  // scratch of a round: Huffman-decoded literals and decoded sequences of one block per frame, bump-allocated on the device
  const uint64_t perFrame = std::min<uint64_t>((uint64_t)maxFrameBytes + 16, (128u << 10) + 16);
  // sized for what data needs, not for the worst case (a frame whose allocation does not fit simply takes the next round):
  // Huffman-coded literals rarely exceed half of the output, sequences (8 bytes each) three quarters of it
  const uint64_t litCap = std::max<uint64_t>((uint64_t)n * perFrame / 2, 1u << 20);
  const uint64_t seqCap = std::max<uint64_t>((uint64_t)n * perFrame * 3 / 32, 1u << 20);               // entries of 8 bytes
  if (!decFrames_.reserve((size_t)n * sizeof(ZraDecFrame)) || !decTables_.reserve((size_t)n * ZRA_DEC_TBL_WORDS * 4) ||
      !decLists_.reserve((size_t)n * 16 + 64) || !decCounters_.reserve(ZRA_DC_WORDS * 4 + 64) || !decLits_.reserve(litCap + 64) ||
      !decSeqs_.reserve(seqCap * 8 + 64) || !status_.reserve((size_t)n * 4) || !produced_.reserve((size_t)n * 4) ||
      !frameMeta_.reserve((size_t)n * 8) || !result_.reserve(64))
    return zerr(64 /* memory_allocation */);
  uint32_t* listA = decLists_.as<uint32_t>();
  a.pending = listA + 2 * (size_t)n; a.hufJobs = listA + 3 * (size_t)n;
  a.counters = decCounters_.as<uint32_t>();
  a.frames = decFrames_.as<ZraDecFrame>(); a.tables = decTables_.as<uint32_t>();
  a.lits = decLits_.as<uint8_t>(); a.litCap = litCap; a.seqs = decSeqs_.as<uint64_t>(); a.seqCap = seqCap;
  a.status = status_.as<uint32_t>(); a.produced = produced_.as<uint32_t>(); a.frameMeta = frameMeta_.as<uint32_t>();
  return ok();
}

// Small batches (random access): every job in ONE launch of zra_ra_small_kernel, frame-end checks behind it, one synchronisation.
// *bailed: jobs the kernel handed back (damaged or unusual frames): the caller takes the batch through decode_launch.
Status Engine::decode_small(const ZraDecodeArgs& a0, const uint32_t* dExpect, uint32_t maxFrameBytes, uint32_t jobBase, unsigned long long* hResult, uint32_t* bailed) {
  ZraDecodeArgs a = a0;
  const uint32_t n = a.nFrames;
  static const bool trace = std::getenv("ZRA_RA_TRACE") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto mark = [&](const char* what) {
    if (!trace) return;
    const auto t = std::chrono::steady_clock::now();
    std::fprintf(stderr, "    small %-12s %7.1f us\n", what, std::chrono::duration<double, std::micro>(t - t_last).count());
    t_last = t;
  };
  { Status st = decode_scratch(a, maxFrameBytes); if (st.zra) return st; }
  mark("scratch");
  a.active = nullptr; a.nActive = n; a.round = 0; a.nextActive = decLists_.as<uint32_t>();
  { static const int skip = std::getenv("ZRA_DEC_SKIP") ? std::atoi(std::getenv("ZRA_DEC_SKIP")) : 0; a.debugSkip = (uint32_t)skip; }
  // the round counters are zero whenever this path finds them (zeroed behind the previous use, off the caller's wait); the result
  // word and the kernel's bail counter (counting down) were preset together by decode_jobs' one memset; frame-end checks and the
  // first-error reduction happen inside the kernel: one launch, one copy back, one synchronisation
  if (!decCountersClean_) HIPCHK(hipMemsetAsync(a.counters, 0, ZRA_DC_WORDS * 4, stream_));
  decCountersClean_ = false;
  uint32_t* dBail = (uint32_t*)(result_.as<uint8_t>() + 8);
  HIPCHK(hipEventRecord(ev0_, stream_));
  hipLaunchKernelGGL(zra_ra_small_kernel, dim3(n), dim3(192), 0, stream_, a, dBail, dExpect, jobBase, result_.as<unsigned long long>());
  HIPCHK(hipEventRecord(ev1_, stream_));
  mark("launched");
  unsigned long long two[2] = {~0ull, ~0ull};
  HIPCHK(hipMemcpyAsync(two, result_.p, 16, hipMemcpyDeviceToHost, stream_));
  mark("queued rest");
  HIPCHK(hipStreamSynchronize(stream_));
  mark("sync");
  HIPCHK(hipGetLastError());
  if (hipMemsetAsync(a.counters, 0, ZRA_DC_WORDS * 4, stream_) == hipSuccess) decCountersClean_ = true;     // (for the next call; nobody waits for it)
  two[1] = 0xFFFFFFFFull - (two[1] & 0xFFFFFFFFull);
  *bailed = (uint32_t)two[1];
  if (!*bailed) *hResult = two[0];
  float ms = 0;
  if (hipEventElapsedTime(&ms, ev0_, ev1_) == hipSuccess) { lastKernelMs_ = ms; kstats_[4] += ms; kstats_[5] += 1; dstats_[5] += ms; dstats_[6] += 1; }
  return ok();
}

hipEvent_t Engine::stage_event() {
  if (stageEvNext_ == stageEv_.size()) { hipEvent_t e = nullptr; if (hipEventCreate(&e) != hipSuccess) return nullptr; stageEv_.push_back(e); }
  return stageEv_[stageEvNext_++];
}

// One pass of the decoder over the jobs of `a0`: rounds of parse -> chain -> execute (a round = one compressed block of every
// unfinished frame) until no frame is left, then the content checksums and the first-error reduction.
Status Engine::decode_launch(const ZraDecodeArgs& a0, const uint32_t* dExpect, uint32_t maxFrameBytes, uint32_t jobBase, unsigned long long* hResult) {
  ZraDecodeArgs a = a0;
  const uint32_t n = a.nFrames;
  { Status st = decode_scratch(a, maxFrameBytes); if (st.zra) return st; }
  decCountersClean_ = false;
  uint32_t* listA = decLists_.as<uint32_t>(); uint32_t* listB = listA + n;
  static const int wavesCap = std::getenv("ZRA_DEC_WAVES") ? std::atoi(std::getenv("ZRA_DEC_WAVES")) : 0;   // bring-up: occupancy sweep
  if (!decOccParse_) {
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&decOccParse_, zra_dec_parse_kernel, 64, 0));
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&decOccExec_, zra_dec_exec_kernel, 64, 0));
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&decOccHuf_, zra_dec_huf_kernel, 64, 0));
    decOccParse_ = std::max(1, decOccParse_); decOccExec_ = std::max(1, decOccExec_); decOccHuf_ = std::max(1, decOccHuf_);
  }
  int perCUParse = decOccParse_, perCUExec = decOccExec_;
  if (wavesCap > 0) { perCUParse = std::min(perCUParse, wavesCap); perCUExec = std::min(perCUExec, wavesCap); }
  { static const int skip = std::getenv("ZRA_DEC_SKIP") ? std::atoi(std::getenv("ZRA_DEC_SKIP")) : 0; a.debugSkip = (uint32_t)skip; }
  HIPCHK(hipEventRecord(ev0_, stream_));
  static const uint32_t chainWaves = std::getenv("ZRA_DEC_CHAIN_WAVES") ? (uint32_t)std::atoi(std::getenv("ZRA_DEC_CHAIN_WAVES")) : 2u;
  // (ZRA_DEC_CHAIN_LDS_MIN: jobs from which the LDS-table chain kernel runs beside the other one; the tests set it to 1)
  static const uint32_t chainLdsMin = std::getenv("ZRA_DEC_CHAIN_LDS_MIN") ? (uint32_t)std::atoi(std::getenv("ZRA_DEC_CHAIN_LDS_MIN")) : (uint32_t)numCUs_ * 96u;
  static const int chainLdsMode = std::getenv("ZRA_DEC_CHAIN_LDS") ? std::atoi(std::getenv("ZRA_DEC_CHAIN_LDS")) : 1;   // 0: without the LDS-table kernel; 2 (bring-up): that kernel alone
  static const bool chainLdsOn = chainLdsMode != 0;
  static const uint32_t chainGrid = std::getenv("ZRA_DEC_CHAIN_GRID") ? (uint32_t)std::atoi(std::getenv("ZRA_DEC_CHAIN_GRID")) : 0u;        // bring-up: absolute wave count
  // the rounds of one set of jobs, one stage after the other on the engine's stream (resident waves per CU of the chain kernel — lane =
  // frame, 64 frames' tables per wave: fewer frames in flight keep more of their table cells in the caches; A/B on one box, round 3,
  // 8 GiB decode, chain stage 2 / 3 / 4 / 6 / 8 waves per CU -> 26.0 / 26.6 / 30.1 / 35.2 / 32.9 ms; 16 GiB, 1 / 1.5 / 2 / 2.5: 74.1 / 58.2 / 51.9 / 53.0 ms)
  // Rounds are enqueued `ahead` at a time without a host synchronisation in between (round 4): a frame of B blocks needs B rounds, and the
  // job count of round r+1 is round r's counter — copied aside on the device and read by the parse kernel (the other stages always
  // counted on the device). Grids are sized by the previous burst's job count (an upper bound: jobs only drop out). One copy back and
  // one synchronisation per burst; frames that still go on afterwards (foreign archives with more, smaller blocks; scratch deferrals)
  // take further bursts of one round.
  static const uint32_t aheadCap = std::getenv("ZRA_DEC_AHEAD") ? (uint32_t)std::max(1, std::atoi(std::getenv("ZRA_DEC_AHEAD"))) : 16u;
  if (!roundN_.reserve(4 * (aheadCap + 2) + 64)) return zerr(64);
  auto run_rounds = [&](ZraDecodeArgs& x, uint32_t nActive, const uint32_t* active, uint32_t round, uint32_t* lA, uint32_t* lB, uint32_t ahead) -> Status {
    uint32_t* const dN = roundN_.as<uint32_t>();
    while (nActive) {
      const uint32_t burst = std::max(1u, std::min(ahead, aheadCap));
      std::vector<hipEvent_t> evs;
      for (uint32_t bi = 0; bi < burst; bi++) {
        HIPCHK(hipMemsetAsync(x.counters, 0, ZRA_DC_WORDS * 4, stream_));
        x.active = active; x.nActive = nActive; x.round = round + bi;
        x.nActivePtr = bi ? dN + bi : nullptr;
        x.nextActive = (active == lA) ? lB : lA;
        const uint32_t gridParse = (uint32_t)std::min<uint64_t>(nActive, (uint64_t)numCUs_ * perCUParse);
        const uint32_t gridChain = (uint32_t)std::min<uint64_t>((nActive + 63) / 64, chainGrid ? chainGrid : (uint64_t)numCUs_ * chainWaves);
        const uint32_t gridExec = (uint32_t)std::min<uint64_t>(nActive, (uint64_t)numCUs_ * perCUExec);
        // per-stage spans (HIP events on the engine's stream; summed into dstats_ once the burst has synchronised)
        hipEvent_t se[5];
        for (auto& e : se) { e = stage_event(); if (!e) return zerr(1); evs.push_back(e); }
        HIPCHK(hipEventRecord(se[0], stream_));
        hipLaunchKernelGGL(zra_dec_parse_kernel, dim3(gridParse), dim3(64), 0, stream_, x);
        HIPCHK(hipEventRecord(se[1], stream_));
        hipLaunchKernelGGL(zra_dec_huf_kernel, dim3((uint32_t)std::min<uint64_t>((nActive + ZRA_HUF_FRAMES - 1) / ZRA_HUF_FRAMES, (uint64_t)numCUs_ * decOccHuf_)), dim3(64), 0, stream_, x);
        HIPCHK(hipEventRecord(se[2], stream_));
        // beside the lane-per-frame chain kernel (tables in HBM scratch, two waves per CU) one workgroup per CU with its frames' tables in
        // LDS, on another stream, pulling from the same queue
        bool forked = false;
        if (chainLdsOn && nActive >= chainLdsMin) {
          if (!pipeStreams_[1]) { if (hipStreamCreateWithFlags(&pipeStreams_[1], hipStreamNonBlocking) != hipSuccess) { pipeStreams_[1] = nullptr; (void)hipGetLastError(); } }
          const size_t ldsBytes = (128 + (size_t)ZRA_CHAIN_LDS_FRAMES * (ZRA_DEC_TBL_WORDS / 2 + ZRA_CHAIN_RING_WORDS)) * 4;   // two-byte cells + a 144-byte bitstream ring per frame
          if (pipeStreams_[1] && !chainLdsAttr_) {
            if (hipFuncSetAttribute((const void*)zra_dec_chain_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes) == hipSuccess) chainLdsAttr_ = 1;
            else { chainLdsAttr_ = -1; (void)hipGetLastError(); }
          }
          if (pipeStreams_[1] && chainLdsAttr_ > 0) {
            hipEvent_t eJoin = stage_event(); if (!eJoin) return zerr(1);
            HIPCHK(hipStreamWaitEvent(pipeStreams_[1], se[2], 0));
            hipLaunchKernelGGL(zra_dec_chain_lds_kernel, dim3((uint32_t)numCUs_), dim3(64), ldsBytes, pipeStreams_[1], x);
            HIPCHK(hipEventRecord(eJoin, pipeStreams_[1]));
            if (chainLdsMode != 2) hipLaunchKernelGGL(zra_dec_chain_kernel, dim3(gridChain), dim3(64), 0, stream_, x);
            HIPCHK(hipStreamWaitEvent(stream_, eJoin, 0));
            forked = true;
          }
        }
        if (!forked) hipLaunchKernelGGL(zra_dec_chain_kernel, dim3(gridChain), dim3(64), 0, stream_, x);
        HIPCHK(hipEventRecord(se[3], stream_));
        hipLaunchKernelGGL(zra_dec_exec_kernel, dim3(gridExec), dim3(64), 0, stream_, x);
        HIPCHK(hipEventRecord(se[4], stream_));
        // the next round's job count stays on the device (the counters are cleared before it starts)
        HIPCHK(hipMemcpyAsync(dN + bi + 1, x.counters + ZRA_DC_NNEXT, 4, hipMemcpyDeviceToDevice, stream_));
        active = x.nextActive;
      }
      uint32_t next = 0;
      HIPCHK(hipMemcpyAsync(&next, dN + burst, 4, hipMemcpyDeviceToHost, stream_));
      HIPCHK(hipStreamSynchronize(stream_));
      HIPCHK(hipGetLastError());
      for (uint32_t bi = 0; bi < burst; bi++)
        for (int k = 0; k < 4; k++) { float m = 0; if (hipEventElapsedTime(&m, evs[5 * bi + k], evs[5 * bi + k + 1]) == hipSuccess) dstats_[k] += m; }
      dstats_[4] += burst;
      stageEvNext_ = 0;
      x.nActivePtr = nullptr;
      nActive = next; round += burst; ahead = 1;
      if (round > (1u << 20)) return zerr(1);          // cannot happen: every round finishes at least one block of some frame
    }
    return ok();
  };
  // Stage pipeline (round 3, opt-in: ZRA_DEC_PIPE=K). With the jobs cut into K slices, slice k's parse, Huffman + chain, and execute
  // kernels run on four streams with grids that leave room for each other, so that slice k executes while slice k+1 decodes its literals
  // and chains and slice k+2 parses. Same kernels, same per-job results; only the first block round is pipelined (frames with more blocks
  // finish slice by slice in the classic way afterwards). Measured, 8 GiB decode on one box: off 105 GiB/s, K = 4 / 8 / 16: 108 / 96 / 66 —
  // the stages do not hide each other (they queue on the same request path), and slices below one chain grid (32 Ki frames) starve
  // the lane-per-frame chain kernel. Not the default.
  static const uint32_t pipeK = std::getenv("ZRA_DEC_PIPE") ? (uint32_t)std::atoi(std::getenv("ZRA_DEC_PIPE")) : 0u;
  static const uint32_t pipeMin = std::getenv("ZRA_DEC_PIPE_MIN") ? (uint32_t)std::atoi(std::getenv("ZRA_DEC_PIPE_MIN")) : 16384u;
  bool piped = false;
  if (pipeK >= 2 && n >= pipeMin && n >= pipeK * 1024u) {
    for (auto& st : pipeStreams_) if (!st && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { st = nullptr; (void)hipGetLastError(); }   // (each on its own: the encode side creates [0], the LDS chain kernel [1])
    if (pipeStreams_[0] && pipeStreams_[1] && pipeStreams_[2] && decCounters_.reserve((size_t)(pipeK + 1) * ZRA_DC_WORDS * 4 + 64)) {
      piped = true;
      a.counters = decCounters_.as<uint32_t>();
      static const uint32_t wP = std::getenv("ZRA_DEC_PIPE_PARSE") ? (uint32_t)std::atoi(std::getenv("ZRA_DEC_PIPE_PARSE")) : 8u;
      static const uint32_t wH = std::getenv("ZRA_DEC_PIPE_HUF") ? (uint32_t)std::atoi(std::getenv("ZRA_DEC_PIPE_HUF")) : 8u;
      static const uint32_t wX = std::getenv("ZRA_DEC_PIPE_EXEC") ? (uint32_t)std::atoi(std::getenv("ZRA_DEC_PIPE_EXEC")) : 12u;
      const uint32_t B = ((n + pipeK - 1) / pipeK + 63u) & ~63u;
      const uint32_t K = (n + B - 1) / B;
      hipStream_t sP = stream_, sH = pipeStreams_[0], sC = pipeStreams_[1], sX = pipeStreams_[2];
      HIPCHK(hipMemsetAsync(a.counters, 0, (size_t)(K + 1) * ZRA_DC_WORDS * 4, stream_));
      hipEvent_t e0 = stage_event(); if (!e0) return zerr(1);
      HIPCHK(hipEventRecord(e0, stream_));
      HIPCHK(hipStreamWaitEvent(sH, e0, 0)); HIPCHK(hipStreamWaitEvent(sC, e0, 0)); HIPCHK(hipStreamWaitEvent(sX, e0, 0));
      std::vector<ZraDecodeArgs> sub(K);
      hipEvent_t eX = nullptr;
      for (uint32_t k = 0; k < K; k++) {
        const uint32_t j0 = k * B, nb = std::min(B, n - j0);
        ZraDecodeArgs x = a;
        x.nFrames = nb;
        x.frameOff = a.frameOff + (size_t)j0 * a.offStride; x.outOff = a.outOff + j0; x.outCap = a.outCap + j0;
        if (a.limit) x.limit = a.limit + j0;
        if (a.pieceBase) x.pieceBase = a.pieceBase + j0;
        x.frames = a.frames + j0; x.tables = a.tables + (size_t)j0 * ZRA_DEC_TBL_WORDS;
        x.status = a.status + j0; x.produced = a.produced + j0; x.frameMeta = a.frameMeta + 2 * (size_t)j0;
        x.pending = a.pending + j0; x.hufJobs = a.hufJobs + j0;
        x.counters = a.counters + (size_t)(k + 1) * ZRA_DC_WORDS;
        const uint64_t litShare = (a.litCap / K) & ~255ull;
        x.lits = a.lits + litShare * k; x.litCap = litShare;
        x.seqs = a.seqs + (a.seqCap / K) * k; x.seqCap = a.seqCap / K;
        x.active = nullptr; x.nActive = nb; x.round = 0; x.nextActive = listA + j0;
        sub[k] = x;
        hipEvent_t eP = stage_event(), eH = stage_event(), eC = stage_event(); eX = stage_event();
        if (!eP || !eH || !eC || !eX) return zerr(1);
        hipLaunchKernelGGL(zra_dec_parse_kernel, dim3((uint32_t)std::min<uint64_t>(nb, (uint64_t)numCUs_ * wP)), dim3(64), 0, sP, x);
        HIPCHK(hipEventRecord(eP, sP));
        HIPCHK(hipStreamWaitEvent(sH, eP, 0));
        hipLaunchKernelGGL(zra_dec_huf_kernel, dim3((uint32_t)std::min<uint64_t>((nb + ZRA_HUF_FRAMES - 1) / ZRA_HUF_FRAMES, (uint64_t)numCUs_ * wH)), dim3(64), 0, sH, x);
        HIPCHK(hipEventRecord(eH, sH));
        HIPCHK(hipStreamWaitEvent(sC, eP, 0));
        hipLaunchKernelGGL(zra_dec_chain_kernel, dim3((uint32_t)std::min<uint64_t>((nb + 63) / 64, chainGrid ? chainGrid : (uint64_t)numCUs_ * chainWaves)), dim3(64), 0, sC, x);
        HIPCHK(hipEventRecord(eC, sC));
        HIPCHK(hipStreamWaitEvent(sX, eH, 0)); HIPCHK(hipStreamWaitEvent(sX, eC, 0));
        hipLaunchKernelGGL(zra_dec_exec_kernel, dim3((uint32_t)std::min<uint64_t>(nb, (uint64_t)numCUs_ * wX)), dim3(64), 0, sX, x);
        HIPCHK(hipEventRecord(eX, sX));
      }
      // everything of the first round is behind the last execute kernel (stream order + its waits); the engine's stream takes over
      HIPCHK(hipStreamWaitEvent(stream_, eX, 0));
      std::vector<uint32_t> hc((size_t)(K + 1) * ZRA_DC_WORDS);
      HIPCHK(hipMemcpyAsync(hc.data(), a.counters, hc.size() * 4, hipMemcpyDeviceToHost, stream_));
      HIPCHK(hipStreamSynchronize(stream_));
      HIPCHK(hipGetLastError());
      dstats_[4] += 1; dstats_[7] += 1;
      stageEvNext_ = 0;
      // frames that go on (more blocks, or no scratch in their slice's share): slice by slice, the classic rounds
      for (uint32_t k = 0; k < K; k++) {
        const uint32_t next = hc[(size_t)(k + 1) * ZRA_DC_WORDS + ZRA_DC_NNEXT];
        if (!next) continue;
        const uint32_t j0 = k * B;
        Status st = run_rounds(sub[k], next, listA + j0, 1, listA + j0, listB + j0, 1);
        if (st.zra) return st;
      }
    }
  }
  // ---- Block-parallel pass (round 6) for frames of several blocks. The rounds below take one block of every frame per round, and their
  // chain stage runs one LANE per frame: at 256 KiB frames a pass of 8 GiB has 32 Ki lanes walking a 128 KiB block each, twice in a row (47 of
  // 76 ms), at 2 MiB frames 4 Ki lanes, sixteen times (168 of 208 ms). Here every compressed block of every frame is a job of the Huffman
  // and chain stages at once — one parse launch that walks all blocks of a frame, one Huffman launch, one chain launch over the blocks
  // (their initial repeat offsets as markers), one execute launch that walks a frame's blocks in order and puts the markers' values in.
  // A frame that is not a clean frame of zstd's own making (an error anywhere, a compressed block that regenerates something else than
  // 128 KiB, the long-offset mode) comes back on a list and takes the rounds below, where every status of the reference is reproduced.
  // ZRA_DEC_FMB=0 turns the pass off.
  static const int fmbEnv = std::getenv("ZRA_DEC_FMB") ? std::atoi(std::getenv("ZRA_DEC_FMB")) : 1;
  static const uint32_t fmbMin = std::getenv("ZRA_DEC_FMB_MIN") ? (uint32_t)std::atoi(std::getenv("ZRA_DEC_FMB_MIN")) : 64u;
  const uint32_t bpf = (uint32_t)(((uint64_t)maxFrameBytes + ZRA_FMB_BLOCK - 1) / ZRA_FMB_BLOCK);
  bool fmb = fmbEnv != 0 && !piped && bpf >= 2 && maxFrameBytes <= (64u << 20) && n >= fmbMin && (uint64_t)n * bpf < (1ull << 30);
  uint32_t nRest = n;
  if (fmb) {
    const uint64_t nb = (uint64_t)n * bpf;
    // scratch for every block of every frame at once (the rounds size theirs for one block per frame)
    const uint64_t litAll = std::max<uint64_t>((uint64_t)n * ((uint64_t)maxFrameBytes + 16) / 2, 1u << 20);
    const uint64_t seqAll = std::max<uint64_t>((uint64_t)n * ((uint64_t)maxFrameBytes + 16) * 3 / 32, 1u << 20);
    if (!decBlkRecs_.reserve((size_t)nb * sizeof(ZraDecFrame)) || !decBlkTables_.reserve((size_t)nb * ZRA_DEC_TBL_WORDS * 4) ||
        !decBlkLists_.reserve((size_t)(2 * nb + n) * 4 + 64) || !decLits_.reserve(litAll + 64) || !decSeqs_.reserve(seqAll * 8 + 64)) {
      (void)hipGetLastError();
      fmb = false;                                      // (no memory for it: the rounds)
    }
    a.lits = decLits_.as<uint8_t>(); a.seqs = decSeqs_.as<uint64_t>();       // (the buffers may have moved)
    if (fmb) {
      ZraDecodeArgs x = a;
      x.bpf = bpf; x.blkRecs = decBlkRecs_.as<ZraDecFrame>(); x.blkTables = decBlkTables_.as<uint32_t>();
      uint32_t* bl = decBlkLists_.as<uint32_t>();
      x.pending = bl; x.hufJobs = bl + nb; x.execList = bl + 2 * nb;
      x.litCap = litAll; x.seqCap = seqAll;
      x.active = nullptr; x.nActive = n; x.round = 0; x.nActivePtr = nullptr; x.nextActive = listA;
      HIPCHK(hipMemsetAsync(x.counters, 0, ZRA_DC_WORDS * 4, stream_));
      hipEvent_t se[5];
      for (auto& e : se) { e = stage_event(); if (!e) return zerr(1); }
      HIPCHK(hipEventRecord(se[0], stream_));
      hipLaunchKernelGGL(zra_dec_parse_all_kernel, dim3((uint32_t)std::min<uint64_t>(n, (uint64_t)numCUs_ * perCUParse)), dim3(64), 0, stream_, x);
      HIPCHK(hipEventRecord(se[1], stream_));
      ZraDecodeArgs y = x; y.frames = x.blkRecs; y.tables = x.blkTables;      // the Huffman and chain stages: jobs are blocks
      hipLaunchKernelGGL(zra_dec_huf_kernel, dim3((uint32_t)std::min<uint64_t>((nb + ZRA_HUF_FRAMES - 1) / ZRA_HUF_FRAMES, (uint64_t)numCUs_ * decOccHuf_)), dim3(64), 0, stream_, y);
      HIPCHK(hipEventRecord(se[2], stream_));
      const uint32_t gridChain = (uint32_t)std::min<uint64_t>((nb + 63) / 64, chainGrid ? chainGrid : (uint64_t)numCUs_ * chainWaves);
      bool forked = false;
      if (chainLdsOn && nb >= chainLdsMin) {
        if (!pipeStreams_[1]) { if (hipStreamCreateWithFlags(&pipeStreams_[1], hipStreamNonBlocking) != hipSuccess) { pipeStreams_[1] = nullptr; (void)hipGetLastError(); } }
        const size_t ldsBytes = (128 + (size_t)ZRA_CHAIN_LDS_FRAMES * (ZRA_DEC_TBL_WORDS / 2 + ZRA_CHAIN_RING_WORDS)) * 4;
        if (pipeStreams_[1] && !chainLdsAttr_) {
          if (hipFuncSetAttribute((const void*)zra_dec_chain_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes) == hipSuccess) chainLdsAttr_ = 1;
          else { chainLdsAttr_ = -1; (void)hipGetLastError(); }
        }
        if (pipeStreams_[1] && chainLdsAttr_ > 0) {
          hipEvent_t eJoin = stage_event(); if (!eJoin) return zerr(1);
          HIPCHK(hipStreamWaitEvent(pipeStreams_[1], se[2], 0));
          hipLaunchKernelGGL(zra_dec_chain_lds_kernel, dim3((uint32_t)numCUs_), dim3(64), ldsBytes, pipeStreams_[1], y);
          HIPCHK(hipEventRecord(eJoin, pipeStreams_[1]));
          if (chainLdsMode != 2) hipLaunchKernelGGL(zra_dec_chain_kernel, dim3(gridChain), dim3(64), 0, stream_, y);
          HIPCHK(hipStreamWaitEvent(stream_, eJoin, 0));
          forked = true;
        }
      }
      if (!forked) hipLaunchKernelGGL(zra_dec_chain_kernel, dim3(gridChain), dim3(64), 0, stream_, y);
      HIPCHK(hipEventRecord(se[3], stream_));
      hipLaunchKernelGGL(zra_dec_exec_all_kernel, dim3((uint32_t)std::min<uint64_t>(n, (uint64_t)numCUs_ * perCUExec)), dim3(64), 0, stream_, x);
      HIPCHK(hipEventRecord(se[4], stream_));
      uint32_t back = 0;
      HIPCHK(hipMemcpyAsync(&back, x.counters + ZRA_DC_NNEXT, 4, hipMemcpyDeviceToHost, stream_));
      HIPCHK(hipStreamSynchronize(stream_));
      HIPCHK(hipGetLastError());
      for (int k = 0; k < 4; k++) { float m = 0; if (hipEventElapsedTime(&m, se[k], se[k + 1]) == hipSuccess) dstats_[k] += m; }
      dstats_[4] += 1;
      stageEvNext_ = 0;
      nRest = back;
    }
  }
  if (!piped) {
    Status st = fmb ? run_rounds(a, nRest, listA, 0, listA, listB, 1)
                    : run_rounds(a, n, nullptr, 0, listA, listB, (maxFrameBytes + (128u << 10) - 1) / (128u << 10));
    if (st.zra) return st;
  }
  HIPCHK(hipEventRecord(ev1_, stream_));
  const uint32_t tb = 256;
  hipLaunchKernelGGL(zra_xxh64_verify_kernel, dim3((n * 4 + tb - 1) / tb), dim3(tb), 0, stream_, a.out, a.outOff, dExpect,
                     produced_.as<uint32_t>(), frameMeta_.as<uint32_t>(), status_.as<uint32_t>(), n);
  hipLaunchKernelGGL(zra_first_error_kernel, dim3((n + tb - 1) / tb), dim3(tb), 0, stream_, status_.as<uint32_t>(), n, jobBase,
                     result_.as<unsigned long long>());
  HIPCHK(hipMemcpyAsync(hResult, result_.p, 8, hipMemcpyDeviceToHost, stream_));
  HIPCHK(hipStreamSynchronize(stream_));
  HIPCHK(hipGetLastError());
  float ms = 0;
  if (hipEventElapsedTime(&ms, ev0_, ev1_) == hipSuccess) { lastKernelMs_ = ms; kstats_[4] += ms; kstats_[5] += 1; }
  return ok();
}

// seqTotal == 0: every frame owns its slot (random access: the reference decodes the touched frames into frameSize buffers).
// seqTotal != 0: whole-archive semantics of ONE multi-frame zstd call over `seqTotal` bytes of destination (zra.cpp:249): frames are
// decoded side by side into their nominal slots; if one regenerates another size than its slot (only possible for a corrupted or
// foreign archive) everything from that frame on is re-decoded one frame at a time, packed back to back, exactly as the reference
// would — first error in frame order, dstSize_tooSmall against the whole destination.
Status Engine::decode_jobs(const uint8_t* dBody, uint64_t bodySize, const uint64_t* dFrameOff, uint8_t* dOut,
                           const uint64_t* dOutOff, const uint32_t* dExpect, uint32_t nFrames, uint32_t maxFrameBytes, uint32_t offStride,
                           uint64_t seqTotal, const ZraDecodeArgs* ra) {
  lastProducedTotal_ = ~0ull;
  if (nFrames == 0) return ok();
  HIPCHK(hipSetDevice(device_));
  if (!result_.reserve(64)) return zerr(64);
  HIPCHK(hipMemsetAsync(result_.p, 0xFF, 64, stream_));
  ZraDecodeArgs a{};
  if (ra) a = *ra;
  a.body = dBody; a.bodySize = bodySize; a.out = dOut; a.offStride = offStride;
  // passes: the chain kernel runs one LANE per frame, so a pass wants hundreds of thousands of frames; its per-round scratch
  // (literals + sequences of one block per frame, ~1.25 bytes per output byte) is what bounds it — 16 GiB of output per pass
  const uint64_t perFrame = std::min<uint64_t>((uint64_t)maxFrameBytes + 16, (128u << 10) + 16);
  static const uint64_t passBytes = std::getenv("ZRA_DEC_PASS_MIB") ? (uint64_t)std::atoll(std::getenv("ZRA_DEC_PASS_MIB")) << 20 : 16ull << 30;
  // equal passes (a remainder pass of a few frames would cost a whole latency-bound round)
  const uint64_t inFlight = std::max<uint64_t>(1, passBytes / perFrame);
  const uint32_t nPass = (uint32_t)((nFrames + inFlight - 1) / inFlight);
  const uint32_t passFrames = (nFrames + nPass - 1) / nPass;
  // (Measured and dropped: two passes in flight on the engine's two streams, each driven by its own host thread, with full or with
  // halved grids — 52.1 vs 53.0 ms per 4 GiB. The four stages do not hide each other: they queue on the same L2 / fabric request path.)
  unsigned long long res = ~0ull;
  for (uint32_t p0 = 0; p0 < nFrames; p0 += passFrames) {
    ZraDecodeArgs b = a;
    b.nFrames = std::min(passFrames, nFrames - p0);
    b.frameOff = dFrameOff + (size_t)p0 * offStride; b.outOff = dOutOff + p0; b.outCap = dExpect + p0;
    if (ra) { if (ra->limit) b.limit = ra->limit + p0; if (ra->pieceBase) b.pieceBase = ra->pieceBase + p0; }
    // few jobs: the one-launch kernel (latency path); a job it hands back (damaged / unusual frame) sends the pass through the
    // four-kernel pipeline, where every status of the reference is reproduced
    static const uint32_t smallMax = std::getenv("ZRA_DEC_SMALL_MAX") ? (uint32_t)std::atoi(std::getenv("ZRA_DEC_SMALL_MAX")) : 1024u;
    bool done = false;
    if (b.nFrames <= smallMax) {
      uint32_t bailed = 0;
      unsigned long long r2 = res;
      Status st = decode_small(b, dExpect + p0, maxFrameBytes, p0, &r2, &bailed);
      if (st.zra) return st;
      if (!bailed) { res = r2; done = true; }
      else if (res == ~0ull) HIPCHK(hipMemsetAsync(result_.p, 0xFF, 64, stream_));   // (what the handed-back pass left in the result word)
      else { unsigned long long keep = res; HIPCHK(hipMemsetAsync(result_.p, 0xFF, 64, stream_)); HIPCHK(hipMemcpyAsync(result_.p, &keep, 8, hipMemcpyHostToDevice, stream_)); HIPCHK(hipStreamSynchronize(stream_)); }
    }
    if (!done) {
      Status st = decode_launch(b, dExpect + p0, maxFrameBytes, p0, &res);
      if (st.zra) return st;
    }
  }
  if (res == ~0ull) return ok();
  const uint32_t code = (uint32_t)(res & 0xFF), first = (uint32_t)(res >> 8);
  const bool resize = code == 255 /* ZE_SIZE_MISMATCH */ || code == 70;
  if (!seqTotal || !resize) return zerr(code == 255 ? 20 : (int)code);
  // ---- sequential tail from frame `first`
  uint64_t cur = 0;
  HIPCHK(hipMemcpyAsync(&cur, dOutOff + first, 8, hipMemcpyDeviceToHost, stream_));
  HIPCHK(hipStreamSynchronize(stream_));
  // One frame at a time is what the reference's multi-frame call does, and what decides a frame's status (its room is what the frames
  // before it left) — but an archive whose frames simply regenerate another size than the header says (a damaged or foreign frameSize)
  // would take one four-kernel pass per frame that way: hundreds of thousands of them. So after a frame has regenerated `got` bytes
  // the frames behind it are decoded side by side, on the guess that they regenerate the same (each at its guessed place, with exactly
  // that room); the longest run for which the guess held — no status, that size — stands as decoded; the first frame that deviates is
  // decoded again alone with its true room, which decides what it reports, and gives the next guess.
  constexpr uint32_t kSpec = 1u << 16;
  if (!seqScratch_.reserve(64 + (size_t)kSpec * 12)) return zerr(64);
  uint64_t* dCur = seqScratch_.as<uint64_t>(); uint32_t* dCap = (uint32_t*)(seqScratch_.as<uint8_t>() + 16);
  uint64_t* dSpecOff = (uint64_t*)(seqScratch_.as<uint8_t>() + 64); uint32_t* dSpecCap = (uint32_t*)(seqScratch_.as<uint8_t>() + 64 + (size_t)kSpec * 8);
  std::vector<uint64_t> hOff; std::vector<uint32_t> hCap, hProduced, hStatus;
  uint32_t guess = 0;
  // the side-by-side batch grows geometrically while the guess holds and starts small again after a miss: an archive whose frame sizes
  // alternate would otherwise decode up to 64 Ki frames to advance by one (quadratic work an untrusted archive could ask for)
  uint32_t specB = 16;
  for (uint32_t f = first; f < nFrames;) {
    if (guess && nFrames - f >= 2 && seqTotal > cur) {
      const uint32_t B = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(std::min<uint64_t>(nFrames - f, kSpec), specB), (seqTotal - cur) / guess);
      if (B >= 2) {
        hOff.resize(B); hCap.assign(B, guess); hProduced.resize(B); hStatus.resize(B);
        for (uint32_t i = 0; i < B; i++) hOff[i] = cur + (uint64_t)i * guess;
        HIPCHK(hipMemcpyAsync(dSpecOff, hOff.data(), (size_t)B * 8, hipMemcpyHostToDevice, stream_));
        HIPCHK(hipMemcpyAsync(dSpecCap, hCap.data(), (size_t)B * 4, hipMemcpyHostToDevice, stream_));
        HIPCHK(hipMemsetAsync(result_.p, 0xFF, 64, stream_));
        ZraDecodeArgs b = a;
        b.frameOff = dFrameOff + (size_t)f * offStride; b.outOff = dSpecOff; b.outCap = dSpecCap; b.nFrames = B;
        b.limit = nullptr; b.pieceBase = nullptr; b.pieces = nullptr;
        unsigned long long r2 = ~0ull;
        // (scratch sized for real frames, not for a tiny guess in front of full-size ones: a block is at most 128 KiB)
        Status st = decode_launch(b, nullptr, std::max<uint32_t>(guess, std::min<uint32_t>(maxFrameBytes ? maxFrameBytes : (128u << 10), 128u << 10)), f, &r2);
        if (st.zra) return st;
        HIPCHK(hipMemcpyAsync(hProduced.data(), produced_.p, (size_t)B * 4, hipMemcpyDeviceToHost, stream_));
        HIPCHK(hipMemcpyAsync(hStatus.data(), status_.p, (size_t)B * 4, hipMemcpyDeviceToHost, stream_));
        HIPCHK(hipStreamSynchronize(stream_));
        uint32_t k = 0;
        while (k < B && hStatus[k] == 0 && hProduced[k] == guess) k++;
        cur += (uint64_t)k * guess; f += k;
        specB = k == B ? (uint32_t)std::min<uint64_t>((uint64_t)specB * 4, kSpec) : 16u;
        if (f >= nFrames) break;
      }
    }
    const uint32_t cap = (uint32_t)std::min<uint64_t>(seqTotal > cur ? seqTotal - cur : 0, 0xFFFFFF00u);
    HIPCHK(hipMemcpyAsync(dCur, &cur, 8, hipMemcpyHostToDevice, stream_));
    HIPCHK(hipMemcpyAsync(dCap, &cap, 4, hipMemcpyHostToDevice, stream_));
    HIPCHK(hipMemsetAsync(result_.p, 0xFF, 64, stream_));
    ZraDecodeArgs b = a;
    b.frameOff = dFrameOff + (size_t)f * offStride; b.outOff = dCur; b.outCap = dCap; b.nFrames = 1;
    b.limit = nullptr; b.pieceBase = nullptr; b.pieces = nullptr;
    Status st = decode_launch(b, nullptr, (uint32_t)std::min<uint64_t>(cap, 0x7FFFFFFFu), f, &res);
    if (st.zra) return st;
    uint32_t got = 0;
    HIPCHK(hipMemcpyAsync(&got, produced_.p, 4, hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    if (res != ~0ull) return zerr((int)(res & 0xFF));
    cur += got; guess = got; f++;
  }
  lastProducedTotal_ = cur;
  return ok();
}

Status Engine::decompress_device(const uint8_t* dArc, size_t arcSize, uint8_t* dOut, size_t outCap) {
  HIPCHK(hipSetDevice(device_));
  kstats_[4] = kstats_[5] = 0; for (auto& d : dstats_) d = 0;
  if (arcSize <= zra_fmt::kFixedSize) return {kOutOfBounds, 0};          // BufferView reader quirk, zra.cpp:166
  uint8_t fixed[zra_fmt::kFixedSize];
  HIPCHK(hipMemcpyAsync(fixed, dArc, sizeof(fixed), hipMemcpyDeviceToHost, stream_));
  HIPCHK(hipStreamSynchronize(stream_));
  HeaderInfo h;
  if (int e = parse_fixed_header(fixed, &h)) return {e, 0};
  if (arcSize < h.size) return {kOutOfBounds, 0};                          // zra.cpp:169-170
  if (outCap < h.uncompressedSize) return {kOutputTooSmall, 0};            // zra.cpp:245-246
  const uint32_t nFrames = h.frames();
  if (nFrames == 0 || h.frameSize == 0) return ok();
  if ((uint64_t)h.seekTableOffset + h.seekTableSize > h.size) return {kHeaderInvalid, 0};
  // Frames beyond the declared size get a zero-capacity slot in the job kernel (dstSize_tooSmall, like the reference's
  // sequential call running out of destination); nothing can be written at or beyond dOut + uncompressedSize <= dOut + outCap.
  if (!frameOff_.reserve(((size_t)nFrames + 1) * 8) || !outOff_.reserve((size_t)nFrames * 8) || !expect_.reserve((size_t)nFrames * 4))
    return zerr(64);
  hipLaunchKernelGGL(zra_jobs_from_seektable_kernel, dim3((nFrames + 256) / 256), dim3(256), 0, stream_, dArc + h.seekTableOffset,
                     nFrames, h.frameSize, h.uncompressedSize, frameOff_.as<uint64_t>(), outOff_.as<uint64_t>(), expect_.as<uint32_t>());
  return decode_jobs(dArc + h.size, arcSize - h.size, frameOff_.as<uint64_t>(), dOut, outOff_.as<uint64_t>(), expect_.as<uint32_t>(), nFrames, h.frameSize, 1, outCap);
}

Status Engine::decompress_frames_host_list(const uint8_t* dBody, uint64_t bodySize, const std::vector<uint64_t>& hFrameOff,
                                           uint8_t* dOut, uint64_t total, uint32_t frameSize) {
  HIPCHK(hipSetDevice(device_));
  const uint32_t nFrames = (uint32_t)(hFrameOff.size() - 1);
  if (nFrames == 0) return ok();
  std::vector<uint64_t> oo(nFrames); std::vector<uint32_t> ex(nFrames);
  for (uint32_t i = 0; i < nFrames; i++) {
    uint64_t o = (uint64_t)i * frameSize;
    oo[i] = o;
    ex[i] = o >= total ? 0 : (uint32_t)std::min<uint64_t>(frameSize, total - o);
  }
  if (!frameOff_.reserve(((size_t)nFrames + 1) * 8) || !outOff_.reserve((size_t)nFrames * 8) || !expect_.reserve((size_t)nFrames * 4))
    return zerr(64);
  HIPCHK(hipMemcpyAsync(frameOff_.p, hFrameOff.data(), ((size_t)nFrames + 1) * 8, hipMemcpyHostToDevice, stream_));
  HIPCHK(hipMemcpyAsync(outOff_.p, oo.data(), (size_t)nFrames * 8, hipMemcpyHostToDevice, stream_));
  HIPCHK(hipMemcpyAsync(expect_.p, ex.data(), (size_t)nFrames * 4, hipMemcpyHostToDevice, stream_));
  HIPCHK(hipStreamSynchronize(stream_));   // host vectors go out of scope
  return decode_jobs(dBody, bodySize, frameOff_.as<uint64_t>(), dOut, outOff_.as<uint64_t>(), expect_.as<uint32_t>(), nFrames, frameSize);
}

Status Engine::decompress_ra_batch(const uint8_t* dArc, size_t arcSize, uint8_t* dOut, const uint64_t* hOff, const uint64_t* hSize,
                                   const uint64_t* hOutOff, size_t nq) {
  return decompress_ra_batch_shard(dArc, arcSize, nullptr, 0, 0, dOut, hOff, hSize, hOutOff, nq);
}

// dBody == nullptr: a whole archive at dArc (header, table, body). Otherwise dArc holds header + table only and dBody the bytes
// [bodyBase, bodyBase + bodyBytes) of the archive's body — the frames one rank of a distributed archive owns (zra_comm.hip).
Status Engine::decompress_ra_batch_shard(const uint8_t* dArc, size_t arcSize, const uint8_t* dBody, uint64_t bodyBytes, uint64_t bodyBase, uint8_t* dOut,
                                         const uint64_t* hOff, const uint64_t* hSize, const uint64_t* hOutOff, size_t nq) {
  // bring-up: ZRA_RA_TRACE=1 prints the host microseconds between the marks of a call
  static const bool trace = std::getenv("ZRA_RA_TRACE") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto mark = [&](const char* what) {
    if (!trace) return;
    const auto t = std::chrono::steady_clock::now();
    std::fprintf(stderr, "  ra %-14s %7.1f us\n", what, std::chrono::duration<double, std::micro>(t - t_last).count());
    t_last = t;
  };
  HIPCHK(hipSetDevice(device_));
  kstats_[4] = kstats_[5] = 0; for (auto& d : dstats_) d = 0;
  if (arcSize <= zra_fmt::kFixedSize) return {kOutOfBounds, 0};
  uint8_t fixed[zra_fmt::kFixedSize];
  HIPCHK(hipMemcpyAsync(fixed, dArc, sizeof(fixed), hipMemcpyDeviceToHost, stream_));
  HIPCHK(hipStreamSynchronize(stream_));
  mark("header read");
  HeaderInfo h;
  if (int e = parse_fixed_header(fixed, &h)) return {e, 0};
  if (arcSize < h.size) return {kOutOfBounds, 0};
  if (!dBody) { dBody = dArc + h.size; bodyBytes = arcSize - h.size; bodyBase = 0; }
  const uint32_t nFrames = h.frames();
  const uint64_t fs = h.frameSize, U = h.uncompressedSize;
  // the reference indexes the table with offset / frameSize without looking at tableSize (zra.cpp:265-268); a header whose fields
  // disagree (size beyond what the table covers, table outside the header) would send it out of bounds — here it is HeaderInvalid
  if ((uint64_t)h.seekTableOffset + h.seekTableSize > h.size) return {kHeaderInvalid, 0};
  if (fs && U && (U + fs - 1) / fs != nFrames) return {kHeaderInvalid, 0};
  if (nq == 0) return ok();
  if (nq > 0xFFFFFFF0ull) return zerr(64);
  // one walk over the queries: the reference's bound (offset + size >= uncompressedSize is refused: the ">=" quirk, zra.cpp:260;
  // overflow-safe), the slices (one per frame a query touches) and the (offset, size, destination, first slice) tuples the device
  // kernels read — written straight into page-locked memory, so that their copy runs at bus speed beside the launches that follow
  if (pinQCap_ < 4 * nq) {
    if (pinQ_) (void)hipHostFree(pinQ_);
    pinQ_ = nullptr; pinQCap_ = 0;
    void* pq = nullptr;
    const size_t cap = std::max<size_t>(4 * nq, 4096);
    if (hipHostMalloc(&pq, cap * 8 + 64, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return zerr(64); }
    pinQ_ = (uint64_t*)pq; pinQCap_ = cap;
  }
  uint64_t* const hq = pinQ_;
  uint64_t maxPieces = 0;
  const bool pow2 = fs && !(fs & (fs - 1));
  const unsigned fsLog = pow2 ? (unsigned)__builtin_ctzll(fs) : 0u;
  if (fs == 0 || nFrames == 0) {
    for (size_t q = 0; q < nq; q++) if (hSize[q] >= U || hOff[q] >= U - hSize[q]) return {kOutOfBounds, 0};
    return ok();
  }
  if (!qmeta_.reserve(4 * nq * 8 + 64)) return zerr(64);
  constexpr size_t kChunk = 1u << 17;                   // tuples go to the device while the next ones are being written
  for (size_t q0 = 0; q0 < nq; q0 += kChunk) {
    const size_t q1 = std::min(nq, q0 + kChunk);
    for (size_t q = q0; q < q1; q++) {
      const uint64_t o = hOff[q], z = hSize[q];
      if (z >= U || o >= U - z) { (void)hipStreamSynchronize(stream_); return {kOutOfBounds, 0}; }
      hq[4 * q] = o; hq[4 * q + 1] = z; hq[4 * q + 2] = hOutOff[q]; hq[4 * q + 3] = maxPieces;
      if (z) maxPieces += pow2 ? ((o + z - 1) >> fsLog) - (o >> fsLog) + 1 : (o + z - 1) / fs - o / fs + 1;
    }
    HIPCHK(hipMemcpyAsync(qmeta_.as<uint64_t>() + 4 * q0, hq + 4 * q0, (q1 - q0) * 32, hipMemcpyHostToDevice, stream_));
  }
  if (maxPieces == 0) { HIPCHK(hipStreamSynchronize(stream_)); return ok(); }
  mark("queries");
  const uint64_t tempBudget = 16ull << 30;
  const uint32_t passSlots = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(nFrames, tempBudget / fs));
  const bool direct = maxPieces * 8 <= nFrames && maxPieces <= passSlots;
  const size_t nJobsMax = direct ? (size_t)maxPieces : (size_t)nFrames;
  if (!frameOff_.reserve((nJobsMax + 1) * 16) || !outOff_.reserve(nJobsMax * 8) || !expect_.reserve(nJobsMax * 4) ||
      !raLimit_.reserve(nJobsMax * 4) || !raPieceBase_.reserve((nJobsMax + 1) * 4) || !raPieces_.reserve((size_t)maxPieces * sizeof(ZraRaPiece) + 64))
    return zerr(64);
  uint32_t touched = 0;
  if (direct) {
    hipLaunchKernelGGL(zra_ra_direct_kernel, dim3((uint32_t)((nq + 255) / 256)), dim3(256), 0, stream_, qmeta_.as<uint64_t>(), (u32)nq, (u32)maxPieces, (u64)fs,
                       (u64)U, dArc + h.seekTableOffset, (u64)bodyBase, raVerifyWholeFrames_ ? 1u : 0u, frameOff_.as<uint64_t>(), outOff_.as<uint64_t>(),
                       expect_.as<uint32_t>(), raLimit_.as<uint32_t>(), raPieceBase_.as<uint32_t>(), raPieces_.as<ZraRaPiece>());
    touched = (uint32_t)maxPieces;
    mark("jobs queued");
  } else {
    const size_t planWords = 4 * (size_t)nFrames + 16;
    if (!raPlan_.reserve(planWords * 4)) return zerr(64);
    HIPCHK(hipMemsetAsync(raPlan_.p, 0, planWords * 4, stream_));
    RaPlan P;
    P.cnt = raPlan_.as<uint32_t>(); P.need = P.cnt + nFrames; P.slot = P.need + nFrames; P.cursor = P.slot + nFrames; P.totals = P.cursor + nFrames;
    const uint64_t* dQ = qmeta_.as<uint64_t>();
    hipLaunchKernelGGL(zra_ra_count_kernel, dim3((uint32_t)((nq + 255) / 256)), dim3(256), 0, stream_, dQ, (u32)nq, (u64)fs, P);
    hipLaunchKernelGGL(zra_ra_plan_kernel, dim3(1), dim3(1024), 0, stream_, P, nFrames, dArc + h.seekTableOffset, (u64)bodyBase, (u64)fs, (u64)U, passSlots,
                       raVerifyWholeFrames_ ? 1u : 0u, frameOff_.as<uint64_t>(), outOff_.as<uint64_t>(), expect_.as<uint32_t>(), raLimit_.as<uint32_t>(),
                       raPieceBase_.as<uint32_t>());
    hipLaunchKernelGGL(zra_ra_fill_kernel, dim3((uint32_t)((nq + 255) / 256)), dim3(256), 0, stream_, dQ, (u32)nq, (u64)fs, P, raPieceBase_.as<uint32_t>(),
                       raPieces_.as<ZraRaPiece>());
    uint32_t totals[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(totals, P.totals, 8, hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    HIPCHK(hipGetLastError());
    touched = totals[0];
    mark("plan");
  }
  if (!touched) return ok();
  // decode the touched frames, a scratch window of passSlots frames at a time (only frames that are decoded in full — or larger
  // than what the decoder needs as its match window — actually write there); slices leave for dOut as each frame finishes
  if (!temp_.reserve((size_t)std::min<uint64_t>(touched, passSlots) * fs + 64)) return zerr(64);
  ZraDecodeArgs ra{};
  ra.pieces = raPieces_.as<ZraRaPiece>(); ra.raOut = dOut;
  for (uint32_t s0 = 0; s0 < touched; s0 += passSlots) {
    const uint32_t n = std::min(passSlots, touched - s0);
    ra.limit = raLimit_.as<uint32_t>() + s0; ra.pieceBase = raPieceBase_.as<uint32_t>() + s0;
    Status st = decode_jobs(dBody, bodyBytes, frameOff_.as<uint64_t>() + 2 * (size_t)s0, temp_.as<uint8_t>(), outOff_.as<uint64_t>() + s0,
                            expect_.as<uint32_t>() + s0, n, (uint32_t)std::min<uint64_t>(fs, 0xFFFFFFFFu), 2, 0, &ra);
    if (st.zra) return st;
  }
  mark("decode");
  return ok();
}

}  // namespace zra_eng
