// zratool_amd — command-line counterpart of the reference's programs/zratool.cpp: same modes, same argv, same output names
//   zratool_amd c|imc|d|imd|b {file} ...     (zratool.cpp:98-286; the positions are listed above main)
// written against include/zra.hpp only, so it doubles as a source-compatibility check of the C++ API.
//   c   : streaming compress  (Compressor, 10 MB chunks rounded to the frame size, header written last at offset 0)
//   d   : streaming decompress (FullDecompressor)
//   imc : in-memory CompressBuffer          imd : in-memory DecompressBuffer
//   b   : benchmark of all four + one random-access query with a memcmp check
#include <zra.hpp>
#include <zra.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

namespace {
using Clock = std::chrono::steady_clock;
double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }

zra::Buffer read_file(const char* path) {
  std::ifstream f(path, std::ios::binary | std::ios::ate);
  if (!f) { std::perror(path); std::exit(2); }
  zra::Buffer b((size_t)f.tellg());
  f.seekg(0);
  f.read(reinterpret_cast<char*>(b.data()), (std::streamsize)b.size());
  return b;
}
void write_file(const char* path, const zra::u8* p, size_t n) {
  std::ofstream f(path, std::ios::binary);
  f.write(reinterpret_cast<const char*>(p), (std::streamsize)n);
}

size_t stream_compress(const char* in, const char* out, zra::i8 level, zra::u32 frameSize, size_t bufferSize) {
  std::ifstream fi(in, std::ios::binary | std::ios::ate);
  if (!fi) { std::perror(in); std::exit(2); }
  const size_t size = (size_t)fi.tellg();
  fi.seekg(0);
  std::ofstream fo(out, std::ios::binary);
  zra::Compressor comp(size, level, frameSize);
  const size_t chunk = bufferSize - (bufferSize % frameSize) + frameSize;      // zratool.cpp:131
  zra::Buffer ibuf(chunk), obuf;
  fo.seekp((std::streamoff)comp.GetHeaderSize());
  size_t done = 0, body = 0;
  while (done < size) {
    const size_t n = std::min(chunk, size - done);
    fi.read(reinterpret_cast<char*>(ibuf.data()), (std::streamsize)n);
    comp.Compress(zra::BufferView(ibuf.data(), n), obuf);
    fo.write(reinterpret_cast<const char*>(obuf.data()), (std::streamsize)obuf.size());
    body += obuf.size();
    done += n;
  }
  if (size == 0) comp.Compress(zra::BufferView(ibuf.data(), 0), obuf);
  const zra::Buffer& h = comp.GetHeader();
  fo.seekp(0);
  fo.write(reinterpret_cast<const char*>(h.data()), (std::streamsize)h.size());
  return body + h.size();
}

size_t stream_decompress(const char* in, const char* out, size_t bufferSize) {
  std::ifstream fi(in, std::ios::binary);
  if (!fi) { std::perror(in); std::exit(2); }
  std::ofstream fo(out, std::ios::binary);
  zra::FullDecompressor dec([&fi](size_t off, size_t n, void* buf) {
    fi.seekg((std::streamoff)off);
    fi.read(static_cast<char*>(buf), (std::streamsize)n);
  });
  zra::Buffer obuf(bufferSize);                                                // zratool.cpp:195
  size_t total = 0;
  for (;;) {
    const size_t n = dec.Decompress(obuf);
    if (!n) break;
    fo.write(reinterpret_cast<const char*>(obuf.data()), (std::streamsize)n);
    total += n;
  }
  return total;
}

// output name of the decompress modes: the ".zra" suffix removed (zratool.cpp:90-95; a name without it is used as it is)
std::string remove_extension(std::string name) {
  const auto pos = name.find_last_of('.');
  if (pos != std::string::npos && name.substr(pos) == ".zra") return name.substr(0, pos);
  return name;
}
}  // namespace

// argv of the reference tool, position by position (zratool.cpp:98-125,213-221):
//   c   {file} {level = 0} {frameSize = 16384} {stream buffer MB = 10}      -> {file}.zra
//   imc {file} {level = 0} {frameSize = 16384}                               -> {file}.zra
//   d   {file} {stream buffer MB = 10}                                       -> {file} without ".zra"
//   imd {file}                                                               -> {file} without ".zra"
//   b   {file} {level} {frameSize} {stream buffer MB} {offset = 0x1000} {size = 0x10000}
int main(int argc, char** argv) {
  if (argc < 3) {
    std::printf("%s {mode} {file} ...\n"
                "c {file} {compression level = 3} {frame size = 16384} {stream buffer size = 10MB} - Streaming Compression\n"
                "imc  {file} {compression level = 3} {frame size = 16384} - In-memory Compression\n"
                "d {file} {stream buffer size = 10MB} - Streaming Decompression\n"
                "imd  {file} - In-memory Decompression\n"
                "b  {file} {compression level = 3} {frame size = 16384} {stream buffer size = 10MB} {offset = 0x1000} {size = 0x10000} - Benchmark (Memory Intensive)\n",
                argv[0]);
    return 0;
  }
  const std::string mode = argv[1];
  const bool comp = mode == "c" || mode == "imc" || mode == "b";
  const zra::i8 level = comp && argc > 3 ? (zra::i8)std::atoi(argv[3]) : 0;
  const zra::u32 frameSize = comp && argc > 4 ? (zra::u32)std::strtoul(argv[4], nullptr, 10) : 16384;
  const int bufArg = mode == "d" ? 3 : 5;
  const size_t bufferSize = (mode == "c" || mode == "d" || mode == "b") && argc > bufArg ? (size_t)std::atoi(argv[bufArg]) * 1'000'000 : 10'000'000;
  std::string fileName; size_t fileSize = 0;
  try {
    if (mode == "c") {
      fileName = std::string(argv[2]) + ".zra";
      fileSize = stream_compress(argv[2], fileName.c_str(), level, frameSize, bufferSize);
    } else if (mode == "d") {
      fileName = remove_extension(argv[2]);
      fileSize = stream_decompress(argv[2], fileName.c_str(), bufferSize);
    } else if (mode == "imc") {
      zra::Buffer in = read_file(argv[2]);
      zra::Buffer out = zra::CompressBuffer(in, level, frameSize);
      fileName = std::string(argv[2]) + ".zra"; fileSize = out.size();
      write_file(fileName.c_str(), out.data(), out.size());
    } else if (mode == "imd") {
      zra::Buffer in = read_file(argv[2]);
      zra::Buffer out = zra::DecompressBuffer(in);
      fileName = remove_extension(argv[2]); fileSize = out.size();
      write_file(fileName.c_str(), out.data(), out.size());
    } else if (mode == "b") {
      zra::Buffer in = read_file(argv[2]);
      const std::string arcName = std::string(argv[2]) + ".bench.zra", backName = std::string(argv[2]) + ".bench.out";
      auto t = Clock::now();
      zra::Buffer arc = zra::CompressBuffer(in, level, frameSize);
      double m = ms_since(t);
      std::printf("in-memory compress   : %8.1f ms  %8.1f MB/s  (%zu -> %zu)\n", m, in.size() / 1e3 / m, in.size(), arc.size());
      t = Clock::now();
      zra::Buffer back = zra::DecompressBuffer(arc);
      m = ms_since(t);
      std::printf("in-memory decompress : %8.1f ms  %8.1f MB/s  %s\n", m, in.size() / 1e3 / m, back == in ? "ok" : "MISMATCH");
      t = Clock::now();
      size_t n = stream_compress(argv[2], arcName.c_str(), level, frameSize, bufferSize);
      m = ms_since(t);
      std::printf("streaming compress   : %8.1f ms  %8.1f MB/s  (%zu bytes)\n", m, in.size() / 1e3 / m, n);
      t = Clock::now();
      n = stream_decompress(arcName.c_str(), backName.c_str(), bufferSize);
      m = ms_since(t);
      std::printf("streaming decompress : %8.1f ms  %8.1f MB/s  (%zu bytes)\n", m, in.size() / 1e3 / m, n);
      {
        // steady state of the drop-in C ABI (caller-owned buffers, scratch already allocated): what a long-running host sees
        std::vector<zra::u8> obuf(ZraGetCompressedOutputBufferSize(in.size(), frameSize)), rbuf(in.size());
        for (int rep = 0; rep < 3; rep++) {
          size_t osz = 0;
          t = Clock::now();
          ZraStatus st = ZraCompressBuffer(in.data(), in.size(), obuf.data(), &osz, level, frameSize, true, nullptr, 0);
          const double mc = ms_since(t);
          t = Clock::now();
          ZraStatus sd = ZraDecompressBuffer(obuf.data(), osz, rbuf.data());
          const double md = ms_since(t);
          std::printf("C ABI rep %d: compress %8.1f ms %8.1f MB/s (status %d)   decompress %8.1f ms %8.1f MB/s (status %d, %s)\n", rep, mc,
                      in.size() / 1e3 / mc, (int)st.zra, md, in.size() / 1e3 / md, (int)sd.zra, std::memcmp(rbuf.data(), in.data(), in.size()) == 0 ? "ok" : "MISMATCH");
        }
      }
      {
        // the reference's random-access probe (zratool.cpp:264-279): offset / size from argv, an archive at the DEFAULT frame size
        const size_t off = argc > 6 ? (size_t)std::atoll(argv[6]) : 0x1000, want = argc > 7 ? (size_t)std::atoll(argv[7]) : 0x10000;
        if (in.size() > off + 1) {
          const size_t len = std::min<size_t>(in.size() - off - 1, want);
          zra::Buffer arcDefault = zra::CompressBuffer(in, level);
          t = Clock::now();
          zra::Buffer ra = zra::DecompressRA(arcDefault, off, len);
          m = ms_since(t);
          std::printf("random access %zu B @ %zu (in-memory) : %8.3f ms  %s\n", len, off, m, std::memcmp(ra.data(), in.data() + off, len) == 0 ? "ok" : "MISMATCH");
          zra::Decompressor dec([&arcDefault](size_t o, size_t sz, void* out) { std::memcpy(out, arcDefault.data() + o, sz); });
          t = Clock::now();
          zra::Buffer rs = dec.Decompress(off, len);
          m = ms_since(t);
          std::printf("random access %zu B @ %zu (streaming) : %8.3f ms  %s\n", len, off, m, std::memcmp(rs.data(), in.data() + off, len) == 0 ? "ok" : "MISMATCH");
        }
      }
      zra::Buffer streamed = read_file(arcName.c_str());
      std::printf("streaming archive %s in-memory archive\n", streamed == arc ? "==" : "!=");
      std::remove(arcName.c_str()); std::remove(backName.c_str());
    } else {
      char* again[1] = {argv[0]};
      return main(1, again);
    }
  } catch (const zra::Exception& e) {
    std::fprintf(stderr, "zra error: %s\n", e.what());
    return 3;
  }
  if (fileSize || !fileName.empty()) std::printf("Output Size (%s): %zu bytes\n", fileName.c_str(), fileSize);
  return 0;
}
