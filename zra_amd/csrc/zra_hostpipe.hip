// zra_amd — host-pointer calls with the PCIe copies running beside the kernels.
//
// The reference's in-memory calls hand over host buffers (zra.cpp:194-256, C ABI zra.cpp:508-533). Large inputs are cut into chunks of
// whole frames; three threads work on neighbouring chunks at once:
//     uploader:   host -> device copy of chunk k+1 (and k+2)
//     this thread: kernels of chunk k (the engine's ordinary device-pointer calls)
//     downloader: device -> host copy of the result of chunk k-1
// Plain pageable memory is enough: on this platform a blocking hipMemcpy of pageable memory runs at the link rate (55 GB/s per
// direction measured, tools/pcie_probe.cpp) and only blocks the thread that issued it, so nothing is staged or registered.
// Frames are independent (zra.cpp:216-225), so the chunked result is byte-identical to the one-piece result.
#include "zra_engine.h"
#include "zra_format.h"
#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace zra_eng {

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return zerr(1); } while (0)

namespace {

// progress counters of the three stages (chunks finished by each)
struct Pipe {
  std::mutex m;
  std::condition_variable cv;
  size_t uploaded = 0, computed = 0, downloaded = 0;
  bool failed = false;
  void bump(size_t& c) { { std::lock_guard<std::mutex> g(m); c++; } cv.notify_all(); }
  void fail() { { std::lock_guard<std::mutex> g(m); failed = true; } cv.notify_all(); }
  // waits until `c` >= want; false when another stage failed
  bool wait(const size_t& c, size_t want) {
    std::unique_lock<std::mutex> g(m);
    cv.wait(g, [&] { return failed || c >= want; });
    return !failed;
  }
};

}  // namespace

size_t host_chunk_bytes() {
  const char* e = std::getenv("ZRA_HOST_CHUNK_MIB");              // read per call: tests shrink it to reach the chunked path with small inputs
  const size_t mib = e ? (size_t)std::atoll(e) : 1024;
  return std::max<size_t>(1, mib) << 20;
}

// ------------------------------------------------------------------------------------------------
// CompressBuffer on host memory. Small inputs: one piece. Otherwise chunks of ~1 GiB of whole frames through compress_frames();
// the seek table and the fixed header are put together on the host from the per-frame sizes (they are 5 bytes per frame).
Status Engine::compress_host(const uint8_t* hIn, size_t n, uint8_t* hOut, size_t* outSize, int level, uint32_t frameSize, bool checksum) {
  HIPCHK(hipSetDevice(device_));
  if (frameSize == 0) return zerr(42);
  const uint32_t tableSize = zra_fmt::table_size(n, frameSize);
  const size_t nFrames = tableSize - 1;
  const size_t headerSize = zra_fmt::kFixedSize + (size_t)tableSize * zra_fmt::kEntrySize;
  const size_t chunkFrames = std::max<size_t>(1, host_chunk_bytes() / frameSize);
  if (nFrames < 2 * chunkFrames) {
    const size_t cap = headerSize + zra_fmt::compress_bound(frameSize) * nFrames;
    if (!hostIn_.reserve(n + 64) || !hostOut_.reserve(cap + 64)) return zerr(64);
    if (n) HIPCHK(hipMemcpyAsync(hostIn_.p, hIn, n, hipMemcpyHostToDevice, stream_));
    Status s = compress_device(hostIn_.as<uint8_t>(), n, hostOut_.as<uint8_t>(), outSize, level, frameSize, checksum);
    if (s.zra) { (void)hipStreamSynchronize(stream_); return s; }
    HIPCHK(hipMemcpyAsync(hOut, hostOut_.p, *outSize, hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    return ok();
  }
  const size_t chunkBytes = chunkFrames * frameSize;
  const size_t nChunks = (n + chunkBytes - 1) / chunkBytes;
  const size_t bodyCap = zra_fmt::compress_bound(frameSize) * chunkFrames;
  const size_t sizesOff = (bodyCap + 63) & ~(size_t)63;
  const size_t outSlot = sizesOff + chunkFrames * 8 + 64;
  // three input slots (one being filled, one queued, one being compressed), two output slots (one being written, one on its way back)
  if (!hostIn_.reserve(3 * chunkBytes + 64) || !hostOut_.reserve(2 * outSlot + 64)) return zerr(64);
  uint8_t* const dIn = hostIn_.as<uint8_t>();
  uint8_t* const dOut = hostOut_.as<uint8_t>();
  std::vector<uint64_t> sizes(nFrames);
  std::vector<uint64_t> bodyOff(nChunks + 1, 0);          // where a chunk's frames start inside the archive body
  std::vector<size_t> bodyLen(nChunks, 0);
  Pipe P;
  const int dev = device_;
  std::thread up([&] {
    if (hipSetDevice(dev) != hipSuccess) { P.fail(); return; }
    for (size_t k = 0; k < nChunks; k++) {
      if (k >= 3 && !P.wait(P.computed, k - 2)) return;                     // slot k%3 was chunk k-3's
      const size_t o = k * chunkBytes, len = std::min(chunkBytes, n - o);
      if (hipMemcpy(dIn + (k % 3) * chunkBytes, hIn + o, len, hipMemcpyHostToDevice) != hipSuccess) { P.fail(); return; }
      P.bump(P.uploaded);
    }
  });
  std::thread down([&] {
    if (hipSetDevice(dev) != hipSuccess) { P.fail(); return; }
    for (size_t k = 0; k < nChunks; k++) {
      if (!P.wait(P.computed, k + 1)) return;
      const uint8_t* slot = dOut + (k % 2) * outSlot;
      const size_t f0 = k * chunkFrames, nf = std::min(chunkFrames, nFrames - f0);
      if ((bodyLen[k] && hipMemcpy(hOut + headerSize + bodyOff[k], slot, bodyLen[k], hipMemcpyDeviceToHost) != hipSuccess) ||
          hipMemcpy(sizes.data() + f0, slot + sizesOff, nf * 8, hipMemcpyDeviceToHost) != hipSuccess) { P.fail(); return; }
      P.bump(P.downloaded);
    }
  });
  Status st = ok();
  for (size_t k = 0; k < nChunks; k++) {
    if (!P.wait(P.uploaded, k + 1) || (k >= 2 && !P.wait(P.downloaded, k - 1))) { st = zerr(1); break; }
    const size_t o = k * chunkBytes, len = std::min(chunkBytes, n - o);
    uint8_t* slot = dOut + (k % 2) * outSlot;
    size_t body = 0;
    st = compress_frames(dIn + (k % 3) * chunkBytes, len, slot, (uint64_t*)(slot + sizesOff), &body, level, frameSize, checksum);
    if (st.zra) { P.fail(); break; }
    bodyLen[k] = body; bodyOff[k + 1] = bodyOff[k] + body;
    P.bump(P.computed);
  }
  up.join(); down.join();
  if (!st.zra && P.failed) st = zerr(1);
  if (st.zra) { (void)hipDeviceSynchronize(); return st; }
  const uint64_t bodySize = bodyOff[nChunks];
  if (headerSize + bodySize >= zra_fmt::kMaxCompressedSize) return {kCompressedTooLarge, 0};   // zra.cpp:227
  zra_fmt::write_fixed(hOut, n, tableSize, frameSize, 0);
  uint8_t* e = hOut + zra_fmt::kFixedSize;
  uint64_t run = 0;
  for (size_t f = 0; f < nFrames; f++) { zra_fmt::entry_put(e + f * 5, run); run += sizes[f]; }
  zra_fmt::entry_put(e + nFrames * 5, run);
  if (run != bodySize) return zerr(1);
  zra_fmt::wr32(hOut + 14, zra_fmt::header_hash(hOut, hOut + zra_fmt::kFixedSize));
  *outSize = headerSize + bodySize;
  return ok();
}

// ------------------------------------------------------------------------------------------------
// DecompressBuffer on host memory (frames walked on the host, zra_capi.cpp). Chunks of whole frames: compressed span up, decoded
// frames down. The chunked path only ever finishes a clean archive; at the first frame that fails (or regenerates another size than
// its slot) the caller is told to take the one-piece path, which reproduces the reference's error and partial-output behaviour.
Status Engine::decode_host_pipelined(const uint8_t* hSpan, const std::vector<uint64_t>& starts, const std::vector<uint64_t>& ends,
                                     uint32_t frameSize, uint64_t total, uint8_t* hOut, bool* fallBack) {
  *fallBack = false;
  HIPCHK(hipSetDevice(device_));
  const size_t nFrames = starts.size();
  const size_t chunkFrames = std::max<size_t>(1, host_chunk_bytes() / frameSize);
  const size_t nChunks = (nFrames + chunkFrames - 1) / chunkFrames;
  size_t maxSpan = 0;
  for (size_t k = 0; k < nChunks; k++) {
    const size_t f0 = k * chunkFrames, f1 = std::min(nFrames, f0 + chunkFrames);
    if (ends[f1 - 1] < starts[f0]) { *fallBack = true; return ok(); }
    maxSpan = std::max<size_t>(maxSpan, ends[f1 - 1] - starts[f0]);
  }
  if (maxSpan > (4ull << 30)) { *fallBack = true; return ok(); }
  const size_t inSlot = (maxSpan + 64 + 63) & ~(size_t)63, outSlot = chunkFrames * (size_t)frameSize;
  std::vector<uint64_t> se(nFrames * 2), oo(nFrames);
  std::vector<uint32_t> ex(nFrames);
  for (size_t i = 0; i < nFrames; i++) {
    const size_t f0 = i / chunkFrames * chunkFrames;
    if (ends[i] < starts[i] || starts[i] < starts[f0]) { *fallBack = true; return ok(); }
    se[2 * i] = starts[i] - starts[f0]; se[2 * i + 1] = ends[i] - starts[f0];
    oo[i] = (uint64_t)(i - f0) * frameSize;
    const uint64_t o = (uint64_t)i * frameSize;
    ex[i] = o >= total ? 0 : (uint32_t)std::min<uint64_t>(frameSize, total - o);
  }
  if (!hostIn_.reserve(3 * inSlot + 64) || !hostOut_.reserve(2 * outSlot + 64) || !frameOff_.reserve(se.size() * 8) ||
      !outOff_.reserve(nFrames * 8) || !expect_.reserve(nFrames * 4))
    return zerr(64);
  HIPCHK(hipMemcpyAsync(frameOff_.p, se.data(), se.size() * 8, hipMemcpyHostToDevice, stream_));
  HIPCHK(hipMemcpyAsync(outOff_.p, oo.data(), nFrames * 8, hipMemcpyHostToDevice, stream_));
  HIPCHK(hipMemcpyAsync(expect_.p, ex.data(), nFrames * 4, hipMemcpyHostToDevice, stream_));
  HIPCHK(hipStreamSynchronize(stream_));
  uint8_t* const dIn = hostIn_.as<uint8_t>();
  uint8_t* const dOut = hostOut_.as<uint8_t>();
  Pipe P;
  const int dev = device_;
  std::thread up([&] {
    if (hipSetDevice(dev) != hipSuccess) { P.fail(); return; }
    for (size_t k = 0; k < nChunks; k++) {
      if (k >= 3 && !P.wait(P.computed, k - 2)) return;
      const size_t f0 = k * chunkFrames, f1 = std::min(nFrames, f0 + chunkFrames);
      const size_t len = ends[f1 - 1] - starts[f0];
      if (len && hipMemcpy(dIn + (k % 3) * inSlot, hSpan + starts[f0], len, hipMemcpyHostToDevice) != hipSuccess) { P.fail(); return; }
      P.bump(P.uploaded);
    }
  });
  std::thread down([&] {
    if (hipSetDevice(dev) != hipSuccess) { P.fail(); return; }
    for (size_t k = 0; k < nChunks; k++) {
      if (!P.wait(P.computed, k + 1)) return;
      const uint64_t o = (uint64_t)k * outSlot;
      const size_t len = o >= total ? 0 : (size_t)std::min<uint64_t>(outSlot, total - o);
      if (len && hipMemcpy(hOut + o, dOut + (k % 2) * outSlot, len, hipMemcpyDeviceToHost) != hipSuccess) { P.fail(); return; }
      P.bump(P.downloaded);
    }
  });
  Status st = ok();
  for (size_t k = 0; k < nChunks; k++) {
    if (!P.wait(P.uploaded, k + 1) || (k >= 2 && !P.wait(P.downloaded, k - 1))) { st = zerr(1); break; }
    const size_t f0 = k * chunkFrames, f1 = std::min(nFrames, f0 + chunkFrames);
    st = decode_jobs(dIn + (k % 3) * inSlot, ends[f1 - 1] - starts[f0], frameOff_.as<uint64_t>() + 2 * f0, dOut + (k % 2) * outSlot,
                     outOff_.as<uint64_t>() + f0, expect_.as<uint32_t>() + f0, (uint32_t)(f1 - f0), frameSize, 2, 0);
    if (st.zra) { P.fail(); break; }
    P.bump(P.computed);
  }
  up.join(); down.join();
  if (st.zra == kZStdError && st.zstd != 1 && st.zstd != 64) { *fallBack = true; return ok(); }   // a frame failed: the one-piece path reports it
  if (!st.zra && P.failed) st = zerr(1);
  if (st.zra) (void)hipDeviceSynchronize();
  return st;
}

}  // namespace zra_eng
