// zra_amd — ZRA container layer on the host: 38-byte fixed header, 40-bit seek-table entries, CRC-32.
// Written from the format facts of the reference (zra.cpp:88-139, README.md:8-19); bit-exact with it.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

namespace zra_fmt {

constexpr uint32_t kSkippableMagic = 0x184D2A50u;  // zra.cpp:112 frameId
constexpr uint32_t kZraMagic = 0x3041525Au;        // "ZRA0", zra.cpp:114
constexpr uint16_t kVersion = 1;                   // zra.cpp:84
constexpr size_t kFixedSize = 38;                  // sizeof(FixedHeader), packed (zra.cpp:111-126)
constexpr size_t kEntrySize = 5;                   // 40-bit LE offset (zra.cpp:96-107)
constexpr uint64_t kMaxCompressedSize = 1ULL << 40;  // zra.cpp:109

inline uint32_t rd32(const uint8_t* p) { uint32_t v; std::memcpy(&v, p, 4); return v; }
inline uint16_t rd16(const uint8_t* p) { uint16_t v; std::memcpy(&v, p, 2); return v; }
inline uint64_t rd64(const uint8_t* p) { uint64_t v; std::memcpy(&v, p, 8); return v; }
inline void wr32(uint8_t* p, uint32_t v) { std::memcpy(p, &v, 4); }
inline void wr16(uint8_t* p, uint16_t v) { std::memcpy(p, &v, 2); }
inline void wr64(uint8_t* p, uint64_t v) { std::memcpy(p, &v, 8); }

inline uint64_t entry_get(const uint8_t* p) { return (uint64_t)rd32(p) | ((uint64_t)p[4] << 32); }
inline void entry_put(uint8_t* p, uint64_t v) { wr32(p, (uint32_t)v); p[4] = (uint8_t)(v >> 32); }

// ZSTD_compressBound (zra.cpp:191,196,316): fixes the per-frame output slot stride for parallel writers
inline size_t compress_bound(size_t n) { return n + (n >> 8) + (n < (128u << 10) ? (((128u << 10) - n) >> 11) : 0); }

// number of seek-table entries = frames + 1 end sentinel (zra.cpp:190,195,304)
inline uint32_t table_size(size_t inputSize, uint32_t frameSize) {
  return (uint32_t)(inputSize / frameSize) + ((inputSize % frameSize) ? 2u : 1u);
}

// CRC-32 (poly 0x04C11DB7 reflected, init/xorout 0xFFFFFFFF) == CRCpp CRC_32() == zlib; slice-by-8 on the host.
uint32_t crc32(uint32_t crc, const void* data, size_t n);

// fixed header fields at their packed offsets (zra.cpp:111-126)
inline void write_fixed(uint8_t* h, uint64_t origSize, uint32_t tableSize, uint32_t frameSize, uint32_t metaSize) {
  wr32(h + 0, kSkippableMagic);
  wr32(h + 4, (uint32_t)(kFixedSize + metaSize + (size_t)tableSize * kEntrySize - 8));  // headerSize excludes frameId+itself
  wr32(h + 8, kZraMagic);
  wr16(h + 12, kVersion);
  wr32(h + 14, 0);  // hash, filled last
  wr64(h + 18, origSize);
  wr32(h + 26, tableSize);
  wr32(h + 30, frameSize);
  wr32(h + 34, metaSize);
}
// CalculateHash (zra.cpp:128-133): CRC over [0,14) || [18,38) || the bytes that follow the fixed header
inline uint32_t header_hash(const uint8_t* fixed, const uint8_t* rest) {
  uint32_t c = crc32(0, fixed, 14);
  c = crc32(c, fixed + 18, 20);
  return crc32(c, rest, rd32(fixed + 4) - kFixedSize + 8);
}

}  // namespace zra_fmt
