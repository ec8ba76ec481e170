// zratool_amd — command-line counterpart of the reference's programs/zratool.cpp (modes and argv order kept:
//   zratool_amd c|imc|d|imd|b <input> <output> [level=0] [frameSize=16384]     zratool.cpp:98-286)
// written against include/zra.hpp only, so it doubles as a source-compatibility check of the C++ API.
//   c   : streaming compress  (Compressor, 10 MB chunks rounded to the frame size, header written last at offset 0)
//   d   : streaming decompress (FullDecompressor)
//   imc : in-memory CompressBuffer          imd : in-memory DecompressBuffer
//   b   : benchmark of all four + one random-access query with a memcmp check
#include <zra.hpp>
#include <zra.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
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

size_t stream_compress(const char* in, const char* out, zra::i8 level, zra::u32 frameSize) {
  std::ifstream fi(in, std::ios::binary | std::ios::ate);
  if (!fi) { std::perror(in); std::exit(2); }
  const size_t size = (size_t)fi.tellg();
  fi.seekg(0);
  std::ofstream fo(out, std::ios::binary);
  zra::Compressor comp(size, level, frameSize);
  const size_t chunk = ((10'000'000 + frameSize - 1) / frameSize) * frameSize;
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

size_t stream_decompress(const char* in, const char* out) {
  std::ifstream fi(in, std::ios::binary);
  if (!fi) { std::perror(in); std::exit(2); }
  std::ofstream fo(out, std::ios::binary);
  zra::FullDecompressor dec([&fi](size_t off, size_t n, void* buf) {
    fi.seekg((std::streamoff)off);
    fi.read(static_cast<char*>(buf), (std::streamsize)n);
  });
  const size_t fs = dec.header.frameSize ? dec.header.frameSize : 1;
  zra::Buffer obuf(((10'000'000 + fs - 1) / fs) * fs);
  size_t total = 0;
  for (;;) {
    const size_t n = dec.Decompress(obuf);
    if (!n) break;
    fo.write(reinterpret_cast<const char*>(obuf.data()), (std::streamsize)n);
    total += n;
  }
  return total;
}
}  // namespace

int main(int argc, char** argv) {
  if (argc < 4) {
    std::fprintf(stderr, "usage: %s c|imc|d|imd|b <input> <output> [level=0] [frameSize=16384]\n", argv[0]);
    return 1;
  }
  const std::string mode = argv[1];
  const zra::i8 level = argc > 4 ? (zra::i8)std::atoi(argv[4]) : 0;
  const zra::u32 frameSize = argc > 5 ? (zra::u32)std::strtoul(argv[5], nullptr, 10) : 16384;
  try {
    if (mode == "c") {
      std::printf("compressed: %zu bytes\n", stream_compress(argv[2], argv[3], level, frameSize));
    } else if (mode == "d") {
      std::printf("decompressed: %zu bytes\n", stream_decompress(argv[2], argv[3]));
    } else if (mode == "imc") {
      zra::Buffer in = read_file(argv[2]);
      zra::Buffer out = zra::CompressBuffer(in, level, frameSize);
      write_file(argv[3], out.data(), out.size());
      std::printf("compressed: %zu -> %zu bytes\n", in.size(), out.size());
    } else if (mode == "imd") {
      zra::Buffer in = read_file(argv[2]);
      zra::Buffer out = zra::DecompressBuffer(in);
      write_file(argv[3], out.data(), out.size());
      std::printf("decompressed: %zu -> %zu bytes\n", in.size(), out.size());
    } else if (mode == "b") {
      zra::Buffer in = read_file(argv[2]);
      auto t = Clock::now();
      zra::Buffer arc = zra::CompressBuffer(in, level, frameSize);
      double m = ms_since(t);
      std::printf("in-memory compress   : %8.1f ms  %8.1f MB/s  (%zu -> %zu)\n", m, in.size() / 1e3 / m, in.size(), arc.size());
      t = Clock::now();
      zra::Buffer back = zra::DecompressBuffer(arc);
      m = ms_since(t);
      std::printf("in-memory decompress : %8.1f ms  %8.1f MB/s  %s\n", m, in.size() / 1e3 / m, back == in ? "ok" : "MISMATCH");
      t = Clock::now();
      size_t n = stream_compress(argv[2], argv[3], level, frameSize);
      m = ms_since(t);
      std::printf("streaming compress   : %8.1f ms  %8.1f MB/s  (%zu bytes)\n", m, in.size() / 1e3 / m, n);
      std::string tmp = std::string(argv[3]) + ".out";
      t = Clock::now();
      n = stream_decompress(argv[3], tmp.c_str());
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
      if (in.size() > 4096) {
        const size_t off = in.size() / 3, len = std::min<size_t>(in.size() - off - 1, 1 << 20);
        t = Clock::now();
        zra::Buffer ra = zra::DecompressRA(arc, off, len);
        m = ms_since(t);
        std::printf("random access %zu B    : %8.3f ms  %s\n", len, m, std::memcmp(ra.data(), in.data() + off, len) == 0 ? "ok" : "MISMATCH");
      }
      zra::Buffer streamed = read_file(argv[3]);
      std::printf("streaming archive %s in-memory archive\n", streamed == arc ? "==" : "!=");
    } else {
      std::fprintf(stderr, "unknown mode %s\n", mode.c_str());
      return 1;
    }
  } catch (const zra::Exception& e) {
    std::fprintf(stderr, "zra error: %s\n", e.what());
    return 3;
  }
  return 0;
}
