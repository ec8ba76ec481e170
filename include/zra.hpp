// zra_amd — C++ API of the MI355X-native ZRA engine, source-compatible with the reference's zra.hpp
// (same namespace, names, argument order, defaults and exception type), so `#include <zra.hpp>` users
// recompile unchanged. Class internals are ours (the reference exposes them as private members only).
// Reference interface replaced: include/zra.hpp:21-324; implementation source/zra.cpp:18-437.
#pragma once

#include <cstddef>
#include <cstdint>
#include <exception>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#ifndef ZRA_EXPORT
#define ZRA_EXPORT __attribute__((visibility("default")))
#endif

namespace zra {
  using u64 = std::uint64_t;
  using u32 = std::uint32_t;
  using u16 = std::uint16_t;
  using u8 = std::uint8_t;
  using i64 = std::int64_t;
  using i32 = std::int32_t;
  using i16 = std::int16_t;
  using i8 = std::int8_t;

  using Buffer = std::vector<u8>;

  /// Non-owning (pointer, length) pair; implicitly built from a Buffer. (zra.hpp:37-53)
  struct BufferView {
    u8* data{nullptr};
    std::size_t size{0};
    BufferView() = default;
    constexpr BufferView(void* data, std::size_t size) : data(static_cast<u8*>(data)), size(size) {}
    BufferView(const Buffer& buffer) : data(const_cast<u8*>(buffer.data())), size(buffer.size()) {}
  };

  /// Same order and values as ZraStatusCode in zra.h (zra.hpp:58-68).
  enum class StatusCode {
    Success, ZStdError, ZraVersionLow, HeaderInvalid, HeaderIncomplete, OutOfBoundsAccess, OutputBufferTooSmall,
    CompressedSizeTooLarge, InputFrameSizeMismatch,
  };

  /// Thrown by every failing call (zra.hpp:70-86, zra.cpp:46-82).
  struct ZRA_EXPORT Exception : std::exception {
    StatusCode code;
    int zstdCode;
    Exception(StatusCode code, i32 zstdCode = {});
    static const char* GetExceptionString(StatusCode code);
    const char* what() const noexcept override;
  };

  ZRA_EXPORT u16 GetVersion();   // zra.hpp:91 -> 1

  /// Parsed view of an archive header (zra.hpp:96-131, zra.cpp:141-187).
  class ZRA_EXPORT Header {
    std::function<void(std::size_t, std::size_t, void*)> readFunction;
   public:
    u16 version;
    u32 size;              ///< whole header incl. metadata and seek table
    u64 uncompressedSize;
    u32 frameSize;
    u32 metaOffset;
    u32 metaSize;
    u32 seekTableOffset;
    u32 seekTableSize;     ///< bytes (5 per entry)

    Header(const std::function<void(std::size_t offset, std::size_t size, void* buffer)>& readFunction);
    Header(const BufferView& buffer);
    void GetMetadata(const BufferView& buffer) const;
    Buffer GetMetadata() const;
    Buffer GetSeekTable() const;
  };

  ZRA_EXPORT std::size_t GetOutputBufferSize(std::size_t inputSize, u32 frameSize, u32 metaSize = 0);   // zra.hpp:139

  // In-memory calls (zra.hpp:151-194): host buffers in, host buffers out; the per-frame codec work runs on the GPU.
  ZRA_EXPORT std::size_t CompressBuffer(const BufferView& input, const BufferView& output, i8 compressionLevel = 0, u32 frameSize = 16384,
                                        bool checksum = true, const BufferView& meta = {});
  ZRA_EXPORT Buffer CompressBuffer(const BufferView& buffer, i8 compressionLevel = 0, u32 frameSize = 16384, bool checksum = true,
                                   const BufferView& meta = {});
  ZRA_EXPORT void DecompressBuffer(const BufferView& input, const BufferView& output);
  ZRA_EXPORT Buffer DecompressBuffer(const BufferView& buffer);
  ZRA_EXPORT void DecompressRA(const BufferView& input, const BufferView& output, std::size_t offset, std::size_t size);
  ZRA_EXPORT Buffer DecompressRA(const BufferView& buffer, std::size_t offset, std::size_t size);

  /// Streaming compressor (zra.hpp:202-254, zra.cpp:304-365): chunks must be multiples of frameSize except the last.
  class ZRA_EXPORT Compressor {
    struct Impl;
    std::shared_ptr<Impl> impl;
   public:
    Compressor(std::size_t size, i8 compressionLevel = 0, u32 frameSize = 16384, bool checksum = true, const BufferView& meta = {});
    std::size_t GetOutputBufferSize(std::size_t inputSize) const;
    std::size_t Compress(const BufferView& input, const BufferView& output);
    void Compress(const BufferView& input, Buffer& output);
    const Buffer& GetHeader();
    std::size_t GetHeaderSize();
  };

  /// Streaming random-access decompressor (zra.hpp:261-298, zra.cpp:367-424).
  class ZRA_EXPORT Decompressor {
    std::function<void(std::size_t, std::size_t, void*)> readFunction;
   public:
    Header header;
   private:
    Buffer seekTable;
    Buffer cache;
    std::size_t maxCacheSize;
   public:
    Decompressor(const std::function<void(std::size_t offset, std::size_t size, void* buffer)>& readFunction,
                 std::size_t maxCacheSize = 1024 * 1024 * 20);
    void Decompress(std::size_t offset, std::size_t size, const BufferView& output);
    void Decompress(std::size_t offset, std::size_t size, Buffer& output);
    Buffer Decompress(std::size_t offset, std::size_t size);
  };

  /// Sequential whole-archive streaming decompressor (zra.hpp:303-324, zra.cpp:426-436).
  class ZRA_EXPORT FullDecompressor {
    std::function<void(std::size_t, std::size_t, void*)> readFunction;
   public:
    Header header;
   private:
    Buffer seekTable;
    Buffer cache;
    std::size_t entryIndex{0};
    struct Ahead;                        ///< frames decoded ahead of the caller (one large device batch serves many calls)
    std::shared_ptr<Ahead> ahead;
   public:
    FullDecompressor(const std::function<void(std::size_t offset, std::size_t size, void* buffer)>& readFunction);
    std::size_t Decompress(const BufferView& output);   ///< returns 0 once everything has been produced
  };
}  // namespace zra
