// zra_amd — several GPUs, one process per GPU (include/zra_hip.h, "distributed archive" section).
//
// Frames are independent (zra.cpp:216-225), so the reference's frame loop splits by frame index: rank r of W owns the contiguous
// range [F*r/W, F*(r+1)/W). Nothing is exchanged while compressing or decoding. What is exchanged:
//   * after compression, the per-frame compressed sizes (8 bytes per frame, all-gather) — every rank then builds the whole seek table
//     (exclusive scan) and knows where its frames lie in the body; the bodies only move if one rank wants the archive in one piece;
//   * for serving, query slices: a query is cut at ownership boundaries (the lookup of zra.cpp:265-269, per owner), every owner
//     decodes the slices of the frames it holds, the bytes travel back to the rank that asked.
// Transport: RCCL called directly (ncclAllGather / grouped ncclSend + ncclRecv — point-to-point over xGMI, one message per peer), or
// two callbacks of the host program (hosts that already have MPI or sockets; the 2-rank tests of this repo).
#include "zra_engine.h"
#include "zra_format.h"
#include "zra_dev.h"
#include "zra_hip.h"
#include <rccl/rccl.h>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <thread>
#include <vector>

using namespace zra_dev;

namespace {

struct RaSlice { u64 src, dst, len; };
// bytes of decoded slices from a staging buffer to their places in the caller's output
__global__ void zra_scatter_slices_kernel(const u8* staging, u8* out, const RaSlice* sl, u32 n) {
  for (u32 i = blockIdx.x; i < n; i += gridDim.x) {
    const RaSlice s = sl[i];
    const u8* p = staging + s.src; u8* q = out + s.dst;
    for (u64 k = threadIdx.x; k < s.len; k += blockDim.x) q[k] = p[k];
  }
}

// ---- RCCL entry points used (librccl is a link-time dependency of the library)
struct Rccl {
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = ncclGetUniqueId;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = ncclCommInitRank;
  ncclResult_t (*CommDestroy)(ncclComm_t) = ncclCommDestroy;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = ncclAllGather;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = ncclSend;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = ncclRecv;
  ncclResult_t (*GroupStart)() = ncclGroupStart;
  ncclResult_t (*GroupEnd)() = ncclGroupEnd;
  bool ok = true;
};
Rccl& rccl() { static Rccl R; return R; }

struct Xfer { int peer; void* buf; size_t bytes; };

}  // namespace

using zra_eng::DevBuf;
using zra_eng::Engine;
using zra_eng::Status;

struct ZraHipEngine { Engine* e; };        // same layout as in zra_capi.cpp
bool zra_ra_whole_frames_option();         // zra_capi.cpp: bit ZRA_HIP_OPT_RA_WHOLE_FRAMES of ZraHipSetOptions

struct ZraHipComm {
  Engine* eng = nullptr;
  int rank = 0, world = 1;
  ncclComm_t nccl = nullptr;               // RCCL transport
  ZraHipHostTransport host{};              // callback transport
  bool useHost = false;
  DevBuf stage, stage2, served, received, slices;   // grow-only scratch
  std::vector<uint8_t> hstage;
  // round 6: a communicator may run its exchanges on a stream of its own (ZraHipCommUseOwnStream) — the second communicator of a process,
  // whose archive gather then runs beside the first one's serving (ZraHipCommGatherArchiveBegin / End: a worker thread drives it)
  hipStream_t own = nullptr;
  hipStream_t st() const { return own ? own : eng->stream(); }
  std::thread worker; ZraStatus asyncStatus{Success, 0}; size_t asyncSize = 0;

  // every rank contributes `bytes` bytes of host memory; `recv` gets world * bytes, rank order
  bool allgather(const void* send, void* recv, size_t bytes) {
    if (useHost) { if (world == 1) { std::memcpy(recv, send, bytes); return true; } return host.allgather(host.user, send, recv, bytes) == 0; }
    hipStream_t st = this->st();
    if (!stage.reserve(bytes + 64) || !stage2.reserve(bytes * world + 64)) return false;
    if (hipMemcpyAsync(stage.p, send, bytes, hipMemcpyHostToDevice, st) != hipSuccess) return false;
    if (rccl().AllGather(stage.p, stage2.p, bytes, ncclUint8, nccl, st) != ncclSuccess) return false;
    if (hipMemcpyAsync(recv, stage2.p, bytes * world, hipMemcpyDeviceToHost, st) != hipSuccess) return false;
    return hipStreamSynchronize(st) == hipSuccess;
  }
  // one point-to-point round: all sends and receives of this rank at once (RCCL: one group, so the transfers to different peers run
  // side by side, each on its own xGMI link). onDevice: the buffers are device memory.
  bool exchange(const std::vector<Xfer>& sends, const std::vector<Xfer>& recvs, bool onDevice) {
    if (sends.empty() && recvs.empty()) return true;
    hipStream_t st = this->st();
    if (!useHost) {
      std::vector<Xfer> s = sends, r = recvs;
      if (!onDevice) {                       // host buffers ride through device scratch
        size_t tot = 0;
        for (auto& x : sends) tot += (x.bytes + 63) & ~(size_t)63;
        for (auto& x : recvs) tot += (x.bytes + 63) & ~(size_t)63;
        if (!stage.reserve(tot + 64)) return false;
        size_t o = 0;
        for (auto& x : s) { if (x.bytes && hipMemcpyAsync(stage.as<uint8_t>() + o, x.buf, x.bytes, hipMemcpyHostToDevice, st) != hipSuccess) return false; x.buf = stage.as<uint8_t>() + o; o += (x.bytes + 63) & ~(size_t)63; }
        for (auto& x : r) { x.buf = stage.as<uint8_t>() + o; o += (x.bytes + 63) & ~(size_t)63; }
      }
      Rccl& R = rccl();
      // Every message goes as pieces of at most 1 GiB (ZRA_COMM_CHUNK_MIB; the tests use 1 MiB and smaller): at the headline configuration
      // a rank's frames are 5.6 GiB in ONE send to the root, and nothing says a single ncclSend takes that. Both sides cut a message
      // the same way (they know its length), all pieces of all peers inside the one group: transfers to different peers still run side
      // by side, each on its own xGMI link.
      static const size_t chunk = (std::getenv("ZRA_COMM_CHUNK_MIB") ? (size_t)std::max(1, std::atoi(std::getenv("ZRA_COMM_CHUNK_MIB"))) : (size_t)1024) << 20;
      static const size_t chunkB = std::getenv("ZRA_COMM_CHUNK_BYTES") ? (size_t)std::max(64, std::atoi(std::getenv("ZRA_COMM_CHUNK_BYTES"))) : chunk;   // (test hook: pieces of a few KiB)
      if (R.GroupStart() != ncclSuccess) return false;
      bool good = true;
      for (auto& x : r) for (size_t o = 0; o < x.bytes; o += chunkB) good &= R.Recv((uint8_t*)x.buf + o, std::min(chunkB, x.bytes - o), ncclUint8, x.peer, nccl, st) == ncclSuccess;
      for (auto& x : s) for (size_t o = 0; o < x.bytes; o += chunkB) good &= R.Send((const uint8_t*)x.buf + o, std::min(chunkB, x.bytes - o), ncclUint8, x.peer, nccl, st) == ncclSuccess;
      if (R.GroupEnd() != ncclSuccess || !good) return false;
      if (!onDevice)
        for (size_t i = 0; i < r.size(); i++)
          if (r[i].bytes && hipMemcpyAsync(recvs[i].buf, r[i].buf, r[i].bytes, hipMemcpyDeviceToHost, st) != hipSuccess) return false;
      return hipStreamSynchronize(st) == hipSuccess;
    }
    // callbacks work on host memory: device buffers are staged
    std::vector<Xfer> s = sends, r = recvs;
    if (onDevice) {
      size_t tot = 0;
      for (auto& x : sends) tot += x.bytes;
      for (auto& x : recvs) tot += x.bytes;
      hstage.resize(tot + 1);
      size_t o = 0;
      for (auto& x : s) { if (x.bytes && hipMemcpyAsync(hstage.data() + o, x.buf, x.bytes, hipMemcpyDeviceToHost, st) != hipSuccess) return false; x.buf = hstage.data() + o; o += x.bytes; }
      for (auto& x : r) { x.buf = hstage.data() + o; o += x.bytes; }
      if (hipStreamSynchronize(st) != hipSuccess) return false;
    }
    std::vector<int> sp, rp; std::vector<const void*> sb; std::vector<void*> rb; std::vector<size_t> sn, rn;
    for (auto& x : s) if (x.bytes) { sp.push_back(x.peer); sb.push_back(x.buf); sn.push_back(x.bytes); }
    for (auto& x : r) if (x.bytes) { rp.push_back(x.peer); rb.push_back(x.buf); rn.push_back(x.bytes); }
    if (host.exchange(host.user, (int)sp.size(), sp.data(), sb.data(), sn.data(), (int)rp.size(), rp.data(), rb.data(), rn.data()) != 0) return false;
    if (onDevice) {
      for (size_t i = 0; i < r.size(); i++)
        if (r[i].bytes && hipMemcpyAsync(recvs[i].buf, r[i].buf, r[i].bytes, hipMemcpyHostToDevice, st) != hipSuccess) return false;
      return hipStreamSynchronize(st) == hipSuccess;
    }
    return true;
  }
  // all ranks leave a collective call with the same status: the first failing rank's
  ZraStatus agree(Status mine) {
    int32_t v[2] = {mine.zra, mine.zstd};
    std::vector<int32_t> all(2 * (size_t)world);
    if (!allgather(v, all.data(), sizeof(v))) return ZraStatus{ZStdError, 1};
    for (int r = 0; r < world; r++) if (all[2 * r]) return ZraStatus{(ZraStatusCode)all[2 * r], all[2 * r + 1]};
    return ZraStatus{Success, 0};
  }
};

// frames [lo, hi) of this rank's part of a distributed archive: the complete header + seek table, and the body bytes of the own frames
struct ZraHipShard {
  int device = 0;
  uint64_t nFrames = 0, lo = 0, hi = 0, total = 0, bodyBase = 0, bodyBytes = 0, bodyTotal = 0;
  uint32_t frameSize = 0;
  std::vector<uint8_t> header;             // host copy (fixed header + seek table, CRC-32 set)
  std::vector<uint64_t> bodyBaseOf;        // [world + 1] body offset at which each rank's frames start
  DevBuf dev;                              // device: [header | table][own body]
  ~ZraHipShard() { if (dev.p) { (void)hipSetDevice(device); dev.release(); } }   // (error paths of ZraHipCommCompress delete a half-built shard)
};

namespace {
ZraStatus mk(int z, int zstd = 0) { return ZraStatus{(ZraStatusCode)z, (int)(int8_t)zstd}; }
uint64_t range_lo(uint64_t nFrames, int r, int world) { return (uint64_t)((unsigned __int128)nFrames * (unsigned)r / (unsigned)world); }
}  // namespace

extern "C" {

void ZraHipShardRange(uint64_t nFrames, int rank, int world, uint64_t* lo, uint64_t* hi) {
  *lo = range_lo(nFrames, rank, world); *hi = range_lo(nFrames, rank + 1, world);
}

int ZraHipOwnerOfFrame(uint64_t nFrames, int world, uint64_t frame) {
  // the r with lo(r) <= frame < lo(r+1); lo(r) = floor(F*r/W) is monotone, so a guess from the inverse is off by one at most
  if (nFrames == 0 || frame >= nFrames) return world - 1;
  int r = (int)std::min<uint64_t>((uint64_t)world - 1, (uint64_t)(((unsigned __int128)frame * (unsigned)world) / nFrames));
  while (r + 1 < world && range_lo(nFrames, r + 1, world) <= frame) r++;
  while (r > 0 && range_lo(nFrames, r, world) > frame) r--;
  return r;
}

// Cuts queries at ownership boundaries. Per slice: owner, query index, offset (global, uncompressed), size, offset of the slice inside
// the query's answer. Slices come out grouped by owner (stable inside an owner: query order). Pure host arithmetic.
ZraStatus ZraHipRouteQueries(uint64_t uncompressedSize, uint32_t frameSize, int world, const uint64_t* offs, const uint64_t* sizes, size_t nq,
                             ZraHipSlice* slices, size_t sliceCap, size_t* nSlices, uint64_t* perOwnerCount) {
  if (!frameSize || world < 1) return mk(ZStdError, 42);
  const uint64_t F = (uncompressedSize + frameSize - 1) / frameSize;
  for (size_t q = 0; q < nq; q++)
    if (sizes[q] >= uncompressedSize || offs[q] >= uncompressedSize - sizes[q]) return mk(OutOfBoundsAccess);   // zra.cpp:260 (">=" kept)
  std::vector<uint64_t> cnt((size_t)world + 1, 0);
  auto walk = [&](bool emit, std::vector<uint64_t>& cursor) {
    for (size_t q = 0; q < nq; q++) {
      uint64_t o = offs[q], left = sizes[q], done = 0;
      while (left) {
        const int r = ZraHipOwnerOfFrame(F, world, o / frameSize);
        const uint64_t ownEnd = std::min<uint64_t>(uncompressedSize, range_lo(F, r + 1, world) * (uint64_t)frameSize);
        const uint64_t len = std::min<uint64_t>(left, ownEnd - o);
        if (emit) { ZraHipSlice& s = slices[cursor[r]++]; s.owner = (uint32_t)r; s.query = (uint64_t)q; s.offset = o; s.size = len; s.within = done; }
        else cnt[r]++;
        o += len; left -= len; done += len;
      }
    }
  };
  std::vector<uint64_t> cursor;
  walk(false, cursor);
  uint64_t tot = 0;
  cursor.assign((size_t)world, 0);
  for (int r = 0; r < world; r++) { cursor[r] = tot; tot += cnt[r]; if (perOwnerCount) perOwnerCount[r] = cnt[r]; }
  *nSlices = (size_t)tot;
  if (tot > sliceCap) return mk(OutputBufferTooSmall);
  walk(true, cursor);
  return mk(Success);
}

ZraStatus ZraHipCommGetUniqueId(void* id128) {
  Rccl& R = rccl();
  if (!R.ok) return mk(ZStdError, 1);
  ncclUniqueId id;
  if (R.GetUniqueId(&id) != ncclSuccess) return mk(ZStdError, 1);
  static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
  std::memcpy(id128, &id, 128);
  return mk(Success);
}

ZraStatus ZraHipCommCreateRccl(ZraHipComm** comm, ZraHipEngine* engine, const void* id128, int rank, int world) {
  *comm = nullptr;
  if (!engine || world < 1 || rank < 0 || rank >= world) return mk(ZStdError, 42);
  Rccl& R = rccl();
  if (!R.ok) return mk(ZStdError, 1);
  if (hipSetDevice(engine->e->device()) != hipSuccess) return mk(ZStdError, 1);
  ncclUniqueId id; std::memcpy(&id, id128, 128);
  ZraHipComm* c = new ZraHipComm();
  c->eng = engine->e; c->rank = rank; c->world = world;
  if (R.CommInitRank(&c->nccl, world, id, rank) != ncclSuccess) { delete c; return mk(ZStdError, 1); }
  *comm = c;
  return mk(Success);
}

ZraStatus ZraHipCommCreateHost(ZraHipComm** comm, ZraHipEngine* engine, const ZraHipHostTransport* t, int rank, int world) {
  *comm = nullptr;
  // engine == NULL: a communicator for the size exchange alone (ZraHipCommStitchSizes; hosts that compress elsewhere, and the CPU tests)
  if (!t || !t->allgather || !t->exchange || world < 1 || rank < 0 || rank >= world) return mk(ZStdError, 42);
  ZraHipComm* c = new ZraHipComm();
  c->eng = engine ? engine->e : nullptr; c->rank = rank; c->world = world; c->host = *t; c->useHost = true;
  *comm = c;
  return mk(Success);
}

// Diagnostic: `bytes` bytes from dSrc to dDst through the transport's point-to-point path, this rank sending to ITSELF (RCCL transport
// only: a send and a receive of one rank inside one group). With one rank nothing else exercises that path — the gather and the serve
// exchange have no peer — and this is what lets a single-GPU test walk the message chunking (ZRA_COMM_CHUNK_BYTES).
ZraStatus ZraHipCommLoopback(ZraHipComm* c, const void* dSrc, void* dDst, size_t bytes) {
  if (!c || !c->eng || c->useHost) return mk(ZStdError, 42);
  std::vector<Xfer> sends{{c->rank, const_cast<void*>(dSrc), bytes}}, recvs{{c->rank, dDst, bytes}};
  return c->exchange(sends, recvs, true) ? mk(Success) : mk(ZStdError, 1);
}

void ZraHipCommDestroy(ZraHipComm* c) {
  if (!c) return;
  if (c->eng) { (void)hipSetDevice(c->eng->device()); (void)hipStreamSynchronize(c->eng->stream()); }
  if (c->worker.joinable()) c->worker.join();
  if (c->own) { (void)hipStreamSynchronize(c->own); (void)hipStreamDestroy(c->own); }
  if (c->nccl) (void)rccl().CommDestroy(c->nccl);
  for (DevBuf* b : {&c->stage, &c->stage2, &c->served, &c->received, &c->slices}) b->release();
  delete c;
}

// The size exchange + stitch of the sharded CompressBuffer (zra.cpp:216-230: the running offset that becomes the seek table): every rank
// contributes its status and the sizes of its frames (8 bytes per frame, one all-gather), and builds the complete header + seek table.
// Fills sh->header, bodyBaseOf, bodyBase, bodyTotal. All ranks return the same status (the first failing rank's).
static ZraStatus comm_stitch(ZraHipComm* c, ZraHipShard* sh, Status st, const std::vector<uint64_t>& mySizes, uint64_t F, uint64_t totalBytes,
                             uint32_t frameSize, size_t headerSize) {
  // sizes of everybody's frames: ranks hold floor/ceil(F / W) frames, padded to the largest share
  const size_t maxLocal = (size_t)((F + c->world - 1) / c->world) + 1;
  // Every allocation of the stitch happens HERE, before the first collective, and its outcome travels with the rank's status: a rank that
  // runs out of memory must not leave the others inside an exchange it never joins (allocation failure is per rank, not per call).
  std::vector<uint64_t> send, all, sizes;
  try {
    send.assign(maxLocal + 2, 0); all.resize((maxLocal + 2) * (size_t)c->world); sizes.reserve(F);
    if (sh) { sh->bodyBaseOf.assign((size_t)c->world + 1, 0); sh->header.resize(headerSize); }
  } catch (const std::exception&) { if (!st.zra) st = zra_eng::zerr(64); }
  if (!sh && !st.zra) st = zra_eng::zerr(64);
  { const ZraStatus pre = c->agree(st); if (pre.zra) return pre; }
  send[0] = 0; send[1] = mySizes.size();
  std::copy(mySizes.begin(), mySizes.end(), send.begin() + 2);
  if (!c->allgather(send.data(), all.data(), send.size() * 8)) { return mk(ZStdError, 1); }
  uint64_t run = 0;
  for (int r = 0; r < c->world; r++) {
    const uint64_t* p = &all[(maxLocal + 2) * (size_t)r];
    sh->bodyBaseOf[r] = run;
    if (p[1] > maxLocal || sizes.size() + p[1] > F) { return mk(ZStdError, 1); }      // (nothing below grows a vector)
    for (uint64_t i = 0; i < p[1]; i++) { sizes.push_back(p[2 + i]); run += p[2 + i]; }
  }
  sh->bodyBaseOf[c->world] = run;
  if (sizes.size() != F) { return mk(ZStdError, 1); }
  size_t hs = 0;
  ZraStatus zs = ZraHipStitchHeader(sizes.data(), F, totalBytes, frameSize, sh->header.data(), &hs);     // same on every rank
  if (zs.zra) { return zs; }
  sh->bodyBase = sh->bodyBaseOf[c->rank]; sh->bodyTotal = run;
  return mk(Success);
}

// The sharded CompressBuffer (zra.cpp:194-235 with its frame loop split by rank). dLocal: the uncompressed bytes of this rank's frames
// [lo, hi) = ZraHipShardRange(ceil(total / frameSize), rank, world), i.e. bytes [lo * frameSize, min(total, hi * frameSize)).
ZraStatus ZraHipCommCompress(ZraHipComm* c, const void* dLocal, size_t localBytes, uint64_t totalBytes, int8_t level, uint32_t frameSize, bool checksum,
                             ZraHipShard** shardOut) {
  *shardOut = nullptr;
  if (!frameSize || !c->eng) return c->agree(zra_eng::zerr(42));       // (a communicator without an engine only stitches)
  const uint64_t F = (totalBytes + frameSize - 1) / frameSize;
  uint64_t lo, hi; ZraHipShardRange(F, c->rank, c->world, &lo, &hi);
  const uint64_t b0 = std::min<uint64_t>(totalBytes, lo * frameSize), b1 = std::min<uint64_t>(totalBytes, hi * frameSize);
  Status st = zra_eng::ok();
  ZraHipShard* sh = new ZraHipShard();
  sh->device = c->eng->device(); sh->nFrames = F; sh->lo = lo; sh->hi = hi; sh->total = totalBytes; sh->frameSize = frameSize;
  const size_t nLocal = (size_t)(hi - lo);
  const size_t headerSize = zra_fmt::kFixedSize + (size_t)(F + 1) * zra_fmt::kEntrySize;
  const size_t bodyCap = zra_fmt::compress_bound(frameSize) * nLocal;
  const size_t sizesOff = (headerSize + bodyCap + 63) & ~(size_t)63;
  std::vector<uint64_t> mySizes;
  size_t bodyBytes = 0;
  if (localBytes != b1 - b0) st = Status{zra_eng::kFrameSizeMismatch, 0};
  else if (hipSetDevice(sh->device) != hipSuccess || !sh->dev.reserve(sizesOff + nLocal * 8 + 64)) st = zra_eng::zerr(64);
  else if (nLocal) {
    uint64_t* dSizes = (uint64_t*)(sh->dev.as<uint8_t>() + sizesOff);
    st = c->eng->compress_frames((const uint8_t*)dLocal, localBytes, sh->dev.as<uint8_t>() + headerSize, dSizes, &bodyBytes, level, frameSize, checksum);
    if (!st.zra) {
      mySizes.resize(nLocal);
      if (hipMemcpy(mySizes.data(), dSizes, nLocal * 8, hipMemcpyDeviceToHost) != hipSuccess) st = zra_eng::zerr(1);
    }
  }
  ZraStatus zs = comm_stitch(c, sh, st, mySizes, F, totalBytes, frameSize, headerSize);
  if (zs.zra) { delete sh; return zs; }
  sh->bodyBytes = bodyBytes;
  if (hipMemcpy(sh->dev.p, sh->header.data(), headerSize, hipMemcpyHostToDevice) != hipSuccess) { delete sh; return mk(ZStdError, 1); }
  // the shard keeps what it holds, not the compress-bound sized work buffer (a third of it at ratio 3)
  if (headerSize + bodyBytes + (64u << 20) < sh->dev.cap) {
    DevBuf exact;
    if (exact.reserve(headerSize + bodyBytes + 64) &&
        hipMemcpy(exact.p, sh->dev.p, headerSize + bodyBytes, hipMemcpyDeviceToDevice) == hipSuccess) {
      sh->dev.release();
      sh->dev = exact;
    } else exact.release();
  }
  *shardOut = sh;
  return mk(Success);
}

// The stitch alone, from frame sizes the caller already holds on the host (frames compressed elsewhere; also what the CPU tests drive with
// eight ranks): collective like ZraHipCommCompress, the shard it returns holds the complete header + seek table and no body.
ZraStatus ZraHipCommStitchSizes(ZraHipComm* c, const uint64_t* hLocalSizes, size_t nLocal, uint64_t totalBytes, uint32_t frameSize, ZraHipShard** shardOut) {
  *shardOut = nullptr;
  if (!frameSize) return c->agree(zra_eng::zerr(42));
  const uint64_t F = (totalBytes + frameSize - 1) / frameSize;
  // the format's table holds a u32 count of entries (frames + 1, zra.cpp:118); beyond it the header cannot be written, and the vectors
  // below would be sized from a number nobody checked
  if (F + 1 > 0xFFFFFFFFull) return c->agree(zra_eng::zerr(42));
  uint64_t lo, hi; ZraHipShardRange(F, c->rank, c->world, &lo, &hi);
  ZraHipShard* sh = nullptr;
  Status st = zra_eng::ok();
  std::vector<uint64_t> mySizes;
  const size_t headerSize = zra_fmt::kFixedSize + (size_t)(F + 1) * zra_fmt::kEntrySize;
  // (this rank's own allocations; a failure becomes its status, which comm_stitch agrees on before any exchange — the stitch itself is
  //  collective and stays outside the try: a handler that starts another collective after some of the stitch's ran would mis-pair the ranks)
  try {
    sh = new ZraHipShard();
    sh->device = c->eng ? c->eng->device() : 0; sh->nFrames = F; sh->lo = lo; sh->hi = hi; sh->total = totalBytes; sh->frameSize = frameSize;
    if (nLocal != (size_t)(hi - lo) || (nLocal && !hLocalSizes)) st = Status{zra_eng::kFrameSizeMismatch, 0};
    else mySizes.assign(hLocalSizes, hLocalSizes + nLocal);
  } catch (const std::exception&) { st = zra_eng::zerr(64); mySizes.clear(); }
  { ZraStatus zs = comm_stitch(c, sh, st, mySizes, F, totalBytes, frameSize, headerSize);
    if (zs.zra) { delete sh; return zs; } }
  for (uint64_t v : mySizes) sh->bodyBytes += v;
  *shardOut = sh;
  return mk(Success);
}

void ZraHipShardDestroy(ZraHipShard* s) {
  if (!s) return;
  if (s->dev.p) { (void)hipSetDevice(s->device); s->dev.release(); }
  delete s;
}
size_t ZraHipShardHeaderSize(const ZraHipShard* s) { return s->header.size(); }
void ZraHipShardGetHeader(const ZraHipShard* s, void* hHeader) { std::memcpy(hHeader, s->header.data(), s->header.size()); }
uint64_t ZraHipShardArchiveSize(const ZraHipShard* s) { return s->header.size() + s->bodyTotal; }
void ZraHipShardGetBody(const ZraHipShard* s, const void** dBody, uint64_t* bodyBase, uint64_t* bodyBytes) {
  // a stitch-only shard (ZraHipCommStitchSizes) holds the header and no frames: no address, no bytes to copy
  if (!s->dev.p) { *dBody = nullptr; *bodyBase = s->bodyBase; *bodyBytes = 0; return; }
  *dBody = s->dev.as<uint8_t>() + s->header.size(); *bodyBase = s->bodyBase; *bodyBytes = s->bodyBytes;
}

// The archive in one piece on `root` (what CompressBuffer returns): header from the root's own copy, every rank's frames to their
// place in the body — W-1 inbound messages in one group.
ZraStatus ZraHipCommGatherArchive(ZraHipComm* c, const ZraHipShard* s, int root, void* dArchive, size_t archiveCap, size_t* archiveSize) {
  if (!c->eng || !s->dev.p) return c->agree(zra_eng::zerr(42));           // (nothing to gather from a stitch-only communicator or shard)
  Status st = zra_eng::ok();
  const size_t hs = s->header.size();
  std::vector<Xfer> sends, recvs;
  if (c->rank == root) {
    if (archiveCap < hs + s->bodyTotal) st = Status{zra_eng::kOutputTooSmall, 0};
    else {
      uint8_t* d = (uint8_t*)dArchive;
      if (hipMemcpyAsync(d, s->dev.p, hs + 0, hipMemcpyDeviceToDevice, c->st()) != hipSuccess ||
          (s->bodyBytes && hipMemcpyAsync(d + hs + s->bodyBase, s->dev.as<uint8_t>() + hs, s->bodyBytes, hipMemcpyDeviceToDevice, c->st()) != hipSuccess))
        st = zra_eng::zerr(1);
      for (int r = 0; r < c->world; r++)
        if (r != root) recvs.push_back({r, d + hs + s->bodyBaseOf[r], (size_t)(s->bodyBaseOf[r + 1] - s->bodyBaseOf[r])});
      if (archiveSize) *archiveSize = hs + s->bodyTotal;
    }
  } else sends.push_back({root, (void*)(s->dev.as<uint8_t>() + hs), (size_t)s->bodyBytes});
  ZraStatus agreed = c->agree(st);                     // nobody starts a transfer the root cannot take
  if (agreed.zra) return agreed;
  if (!c->exchange(sends, recvs, true)) st = zra_eng::zerr(1);
  if (hipStreamSynchronize(c->st()) != hipSuccess) st = zra_eng::zerr(1);
  return c->agree(st);
}

// The gather beside other work of the process (round 6: in a sharded step the archive's bodies travel to the root WHILE the ranks serve
// queries from their shards — ZraHipCommServe on another communicator; the serve works on the shards, not on the gathered archive).
// `c` must be a communicator with a stream of its own (ZraHipCommUseOwnStream) that no other call uses until End returns; Begin returns
// at once, a worker thread runs ZraHipCommGatherArchive (collective: every rank calls Begin and End); End joins it and returns its status.
ZraStatus ZraHipCommUseOwnStream(ZraHipComm* c) {
  if (!c || !c->eng) return mk(ZStdError, 42);
  if (c->own) return mk(Success);
  if (hipSetDevice(c->eng->device()) != hipSuccess || hipStreamCreateWithFlags(&c->own, hipStreamNonBlocking) != hipSuccess) { c->own = nullptr; (void)hipGetLastError(); return mk(ZStdError, 1); }
  return mk(Success);
}
ZraStatus ZraHipCommGatherArchiveBegin(ZraHipComm* c, const ZraHipShard* s, int root, void* dArchive, size_t archiveCap) {
  if (!c || !c->eng || !c->own || c->worker.joinable()) return mk(ZStdError, 42);
  // what the caller queued on the engine's stream (the shard's frames, the destination buffer) is done before the side stream touches it
  if (hipSetDevice(c->eng->device()) != hipSuccess || hipStreamSynchronize(c->eng->stream()) != hipSuccess) return mk(ZStdError, 1);
  c->asyncStatus = mk(Success); c->asyncSize = 0;
  try {
    c->worker = std::thread([c, s, root, dArchive, archiveCap]() {
      (void)hipSetDevice(c->eng->device());
      c->asyncStatus = ZraHipCommGatherArchive(c, s, root, dArchive, archiveCap, &c->asyncSize);
    });
  } catch (const std::exception&) { return mk(ZStdError, 64); }
  return mk(Success);
}
ZraStatus ZraHipCommGatherArchiveEnd(ZraHipComm* c, size_t* archiveSize) {
  if (!c || !c->worker.joinable()) return mk(ZStdError, 42);
  c->worker.join();
  if (archiveSize) *archiveSize = c->asyncSize;
  return c->asyncStatus;
}

// (A transport failure — an allgather or exchange that returns false — leaves the ranks out of step: the call returns {ZStdError, 1}
// on the rank that saw it and the communicator must not be used again; statuses of the ARCHIVE are agreed on and identical everywhere.)
// Collective random access over a distributed archive: every rank passes its own queries (offset, size over the WHOLE uncompressed
// range; any rank may ask for any byte) and gets its answers in dOut at dstOffs[q]. Bounds as DecompressRA (zra.cpp:260).
ZraStatus ZraHipCommServe(ZraHipComm* c, const ZraHipShard* s, const uint64_t* offs, const uint64_t* sizes, const uint64_t* dstOffs, size_t nq, void* dOut) {
  if (!c->eng || !s->dev.p) return c->agree(zra_eng::zerr(42));           // (a stitch-only communicator or shard holds no frames to serve)
  const int W = c->world, me = c->rank;
  Status st = zra_eng::ok();
  // 1. cut my queries at ownership boundaries, grouped by owner
  size_t nSl = 0;
  std::vector<ZraHipSlice> sl;
  std::vector<uint64_t> outCnt((size_t)W, 0);
  {
    size_t cap = nq + (size_t)W + 16;
    for (int pass = 0; pass < 2; pass++) {
      sl.resize(cap);
      ZraStatus z = ZraHipRouteQueries(s->total, s->frameSize, W, offs, sizes, nq, sl.data(), cap, &nSl, outCnt.data());
      if (z.zra == OutputBufferTooSmall) { cap = nSl; continue; }
      if (z.zra) st = Status{(int)z.zra, z.zstd};
      break;
    }
    if (st.zra) { nSl = 0; std::fill(outCnt.begin(), outCnt.end(), 0); }
    sl.resize(nSl);
  }
  std::vector<uint64_t> outBytes((size_t)W, 0);
  for (auto& x : sl) outBytes[x.owner] += x.size;
  // 2. who asks whom for how much
  std::vector<uint64_t> mine(2 * (size_t)W), all(2 * (size_t)W * W);
  for (int r = 0; r < W; r++) { mine[2 * r] = outCnt[r]; mine[2 * r + 1] = outBytes[r]; }
  if (!c->allgather(mine.data(), all.data(), mine.size() * 8)) return mk(ZStdError, 1);
  std::vector<uint64_t> inCnt((size_t)W), inBytes((size_t)W);
  uint64_t inSl = 0, inTot = 0;
  for (int r = 0; r < W; r++) { inCnt[r] = all[2 * (size_t)W * r + 2 * me]; inBytes[r] = all[2 * (size_t)W * r + 2 * me + 1]; inSl += inCnt[r]; inTot += inBytes[r]; }
  // 3. slice descriptors (offset, size) to their owners
  std::vector<uint64_t> outDesc(2 * nSl), inDesc(2 * (size_t)inSl);
  for (size_t i = 0; i < nSl; i++) { outDesc[2 * i] = sl[i].offset; outDesc[2 * i + 1] = sl[i].size; }
  {
    std::vector<Xfer> sends, recvs;
    size_t so = 0, ro = 0;
    for (int r = 0; r < W; r++) {
      if (r == me) std::memcpy(inDesc.data() + 2 * ro, outDesc.data() + 2 * so, (size_t)outCnt[r] * 16);
      else {
        if (outCnt[r]) sends.push_back({r, outDesc.data() + 2 * so, (size_t)outCnt[r] * 16});
        if (inCnt[r]) recvs.push_back({r, inDesc.data() + 2 * ro, (size_t)inCnt[r] * 16});
      }
      so += outCnt[r]; ro += inCnt[r];
    }
    if (!c->exchange(sends, recvs, false)) return mk(ZStdError, 1);
  }
  // 4. decode what I own: answers packed by asking rank
  uint64_t outTot = 0;
  for (int r = 0; r < W; r++) outTot += outBytes[r];
  if (hipSetDevice(s->device) != hipSuccess || !c->served.reserve(inTot + 64) || !c->received.reserve(outTot + 64) || !c->slices.reserve(nSl * sizeof(RaSlice) + 64)) st = zra_eng::zerr(64);
  if (!st.zra && inSl) {
    std::vector<uint64_t> qo(inSl), qs(inSl), qd(inSl);
    uint64_t run = 0;
    const uint64_t ownLo = s->lo * (uint64_t)s->frameSize, ownHi = std::min<uint64_t>(s->total, s->hi * (uint64_t)s->frameSize);
    for (uint64_t i = 0; i < inSl; i++) {
      qo[i] = inDesc[2 * i]; qs[i] = inDesc[2 * i + 1]; qd[i] = run; run += qs[i];
      if (qo[i] < ownLo || qs[i] > ownHi - std::min(ownHi, qo[i])) st = Status{zra_eng::kOutOfBounds, 0};     // not mine: a router bug, never decoded
    }
    if (!st.zra && run != inTot) st = zra_eng::zerr(1);
    if (!st.zra) {
      // (a query never reaches the archive's last byte — the ">=" rule of zra.cpp:260, applied to the whole query by the router — so
      // the same rule inside the batch call cannot fire for a slice)
      c->eng->set_ra_verify_whole_frames(zra_ra_whole_frames_option());      // ZRA_HIP_OPT_RA_WHOLE_FRAMES, as in ZraHipDecompressRABatch
      st = c->eng->decompress_ra_batch_shard(s->dev.as<uint8_t>(), s->header.size(), s->dev.as<uint8_t>() + s->header.size(), s->bodyBytes, s->bodyBase,
                                             c->served.as<uint8_t>(), qo.data(), qs.data(), qd.data(), (size_t)inSl);
    }
  }
  // 5. answers back to the ranks that asked, theirs to me
  std::vector<uint64_t> recvBase((size_t)W + 1, 0);
  {
    std::vector<Xfer> sends, recvs;
    uint64_t so = 0, ro = 0;
    for (int r = 0; r < W; r++) {
      recvBase[r] = ro;
      if (r != me) {
        if (inBytes[r]) sends.push_back({r, c->served.as<uint8_t>() + so, (size_t)inBytes[r]});
        if (outBytes[r]) recvs.push_back({r, c->received.as<uint8_t>() + ro, (size_t)outBytes[r]});
      } else if (inBytes[r] && hipMemcpyAsync(c->received.as<uint8_t>() + ro, c->served.as<uint8_t>() + so, inBytes[r], hipMemcpyDeviceToDevice, c->eng->stream()) != hipSuccess) st = zra_eng::zerr(1);
      so += inBytes[r]; ro += outBytes[r];
    }
    if (!c->exchange(sends, recvs, true)) return mk(ZStdError, 1);
  }
  // 6. slices to their places in my output
  if (!st.zra && nSl) {
    std::vector<RaSlice> rs(nSl);
    std::vector<uint64_t> cur(recvBase.begin(), recvBase.end());
    for (size_t i = 0; i < nSl; i++) { rs[i].src = cur[sl[i].owner]; cur[sl[i].owner] += sl[i].size; rs[i].dst = dstOffs[sl[i].query] + sl[i].within; rs[i].len = sl[i].size; }
    hipStream_t stream = c->eng->stream();
    if (hipMemcpyAsync(c->slices.p, rs.data(), nSl * sizeof(RaSlice), hipMemcpyHostToDevice, stream) != hipSuccess) st = zra_eng::zerr(1);
    else {
      hipLaunchKernelGGL(zra_scatter_slices_kernel, dim3((u32)std::min<size_t>(nSl, 65535)), dim3(256), 0, stream, c->received.as<u8>(), (u8*)dOut, c->slices.as<RaSlice>(), (u32)nSl);
      if (hipStreamSynchronize(stream) != hipSuccess) st = zra_eng::zerr(1);
    }
  }
  if (st.zra && std::getenv("ZRA_COMM_TRACE")) std::fprintf(stderr, "ZraHipCommServe: rank %d local status {%d, %d}\n", me, st.zra, st.zstd);
  return c->agree(st);
}

}  // extern "C"
