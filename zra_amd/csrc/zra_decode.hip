// zra_amd — frame DECODE kernel for gfx950 (MI355X).
//
// Replaces the reference's per-frame ZSTD_decompressDCtx work (zra.cpp:249,280,289,293,397,406,410,435).
// One independent zstd frame per workgroup; the workgroup is ONE 64-lane wave and keeps only ~17 KiB of LDS (Huffman +
// FSE decode tables, a 64-sequence batch), so 8-9 frames are in flight per CU (persistent grid, atomic frame queue):
// decoding is a chain of dependent memory round trips, and frames in flight are what hides them.
// Per compressed block:  lane 0 parses headers; all lanes build the tables; 4 lanes decode the 4 Huffman streams;
// then batches of 64 sequences: lane 0 walks the FSE bit chain (inherently serial), a wave prefix-scan turns
// (litLength, matchLength) into output positions, every lane copies ITS sequence's literals, and match copies run in
// dependency rounds: a lane is ready once its source range ends before the first unfinished match destination, so
// independent matches of a batch execute in the same round trip. The output window is the destination buffer (HBM/L2).
// Format per RFC 8878 / SURVEY.md Appendix A.1-A.3.
#include "zra_dev.h"
#include "zra_kernels.h"

using namespace zra_dev;

namespace {

constexpr int DEC_THREADS = 64;
constexpr int BATCH = 64;           // sequences decoded + executed per round (one per lane)
constexpr u32 BLOCK_MAX = 128u << 10;

__constant__ u32 c_ll_base[36] = {0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,18,20,22,24,28,32,40,48,64,128,256,512,1024,2048,4096,8192,16384,32768,65536};
__constant__ u8 c_ll_bits[36] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16};
__constant__ u32 c_ml_base[53] = {3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,
                                  35,37,39,41,43,47,51,59,67,83,99,131,259,515,1027,2051,4099,8195,16387,32771,65539};
__constant__ u8 c_ml_bits[53] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,
                                 1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16};
__constant__ short c_ll_defnorm[36] = {4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1};
__constant__ short c_ml_defnorm[53] = {1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,
                                       1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1};
__constant__ short c_of_defnorm[29] = {1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1};

// FSE decode cell of the Huffman-weight table (8 bytes): symbol | nbBits<<40 | nextBase<<48
__device__ __forceinline__ u64 mk_seqsym(u32 baseValue, u32 addBits, u32 nbBits, u32 nextBase) {
  return (u64)baseValue | ((u64)addBits << 32) | ((u64)nbBits << 40) | ((u64)nextBase << 48);
}
// decode-table cell for LL / ML / OF (4 bytes): symbol | extraBits<<8 | nbBits<<16 | nextBase<<20. Everything that positions
// the bit reader (extra bits, state bits) is in the cell; the base value is looked up by symbol off that chain. 4-byte cells
// (2 KiB per 512-cell table instead of 4) are what lets 20 frames share a CU's LDS: the decoder is latency-bound and its
// throughput is proportional to the frames in flight (profiles/r01_mf_occupancy_sweep.log, decode sweep).
__device__ __forceinline__ u32 mk_cell(u32 sym, u32 addBits, u32 nbBits, u32 nextBase) {
  return sym | (addBits << 8) | (nbBits << 16) | (nextBase << 20);
}

struct __attribute__((aligned(16))) DecShared {
  // The Huffman literal table is only live while the literal streams are decoded, the LL and ML tables only while sequences are
  // decoded: they share 4 KiB. What is needed to rebuild them for a later block of the same frame (treeless literals, repeat-mode
  // tables) is kept below (weights / llNorm / mlNorm).
  union {
    struct { u32 llT[512]; u32 mlT[512]; };
    u16 huf[2048];          // sym | nbBits<<8 ; doubles as scratch while a tree description is parsed
  };
  u32 ofT[256];
  u32 llBase[36], mlBase[53];   // base values by symbol (copied from constant memory once per workgroup)
  short llNorm[36];         // normalised counts of the last non-RLE LL / ML table (rebuild on repeat mode)
  short mlNorm[53];
  u32 llSaveLog, llSaveMax, llSaveRle;   // xxSaveRle: 0 = FSE table described by xxNorm, else 1 + RLE symbol
  u32 mlSaveLog, mlSaveMax, mlSaveRle;
  union {
    struct { short norm[256]; u8 spread[512]; };      // scratch while a table is described / built
    struct { u32 seqLL[BATCH], seqML[BATCH], seqOF[BATCH]; };   // one batch of decoded sequences
  };
  u8 weights[256];
  u32 rankStart[16];
  // control words (written by lane 0, read by the wave after a wave sync)
  u32 err;
  u32 frame;
  u32 blkType, blkSize, blkLast, blkPos;
  u32 litType, litRegen, litComp, litHdr, litStreams, litRle;
  u32 hufValid, hufMaxBits, hufNSym;
  u32 nbSeq, seqPos, seqModes;
  u32 llLog, mlLog, ofLog, llValid, mlValid, ofValid;
  u32 rep[3];
  u32 streamOff[4], streamLen[4];
  u32 tl, ms, used;
  u32 produced;             // bytes produced in this frame so far
  u32 seqHdrErr;            // error of the sequences header, raised after the literal stage
  u32 batchValid;           // sequences of the current batch decoded before the bitstream went bad (== batch size when it did not)
  u32 fcsLo, fcsHi, fcsHave;   // Frame_Content_Size when the header declares one (checked at the frame end, before the checksum)
};

// wave-level sync: LDS and global traffic of the wave is complete and visible to its other lanes
__device__ __forceinline__ void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// ---------------------------------------------------------------------------------------------
// FSE table description (A.3) -> norm[] ; single thread. returns bytes consumed, 0 on corruption
__device__ u32 read_ncount(short* norm, u32* maxSymIO, u32* tableLogOut, const u8* src, u32 n, u32 maxAL, const u8* lim) {
  if (n < 1) return 0;
  u32 bitpos = 0, nbits = n * 8;
  auto peekf = [&](u32 k) -> u32 {
    u32 byte = bitpos >> 3;
    u64 v = byte < n ? ld64_safe(src + byte, src + n < lim ? src + n : lim) : 0;
    return (u32)((v >> (bitpos & 7)) & ((1u << k) - 1));
  };
  u32 AL = peekf(4) + 5; bitpos += 4;
  if (AL > maxAL) return 0;
  i32 remaining = (1 << AL) + 1, thr = 1 << AL, nb = (i32)AL + 1;
  u32 sym = 0, maxSym = *maxSymIO;
  bool prev0 = false;
  for (u32 s = 0; s <= maxSym; s++) norm[s] = 0;
  while (remaining > 1 && sym <= maxSym) {
    if (prev0) {
      for (;;) {
        u32 f = peekf(2); bitpos += 2;
        sym += f;
        if (f != 3) break;
        if (bitpos > nbits) return 0;
      }
      if (sym > maxSym) return 0;
    }
    i32 max = (2 * thr - 1) - remaining, v;
    u32 low = peekf(nb - 1);
    if ((i32)low < max) { v = (i32)low; bitpos += nb - 1; }
    else { v = (i32)peekf(nb); if (v >= thr) v -= max; bitpos += nb; }
    v--;
    remaining -= v < 0 ? -v : v;
    norm[sym++] = (short)v;
    prev0 = (v == 0);
    if (remaining < 1) return 0;
    while (remaining < thr) { nb--; thr >>= 1; }
    if (bitpos > nbits) return 0;
  }
  if (remaining != 1 || bitpos > nbits) return 0;
  *maxSymIO = sym - 1;
  *tableLogOut = AL;
  return (bitpos + 7) >> 3;
}

// Build an FSE decode table from norm[] (single wave; lanes cooperate on the final fill).
// kind: 0 = LL, 1 = ML, 2 = OF
__device__ void build_fse_dtable(u32* table, const short* norm, u32 maxSym, u32 tableLog, int kind, u8* spread, int lane) {
  u32 size = 1u << tableLog, mask = size - 1;
  if (lane == 0) {
    u32 high = size - 1, step = (size >> 1) + (size >> 3) + 3, pos = 0;
    for (u32 s = 0; s <= maxSym; s++) if (norm[s] == -1) spread[high--] = (u8)s;
    for (u32 s = 0; s <= maxSym; s++) {
      for (int i = 0; i < norm[s]; i++) {
        spread[pos] = (u8)s;
        pos = (pos + step) & mask;
        while (pos > high) pos = (pos + step) & mask;
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  // cell u gets x = next[sym]++ in ascending u. One lane per symbol walks all cells (broadcast LDS reads).
  if ((u32)lane <= maxSym && norm[lane] != 0) {
    const u32 s = (u32)lane;
    u32 x = norm[s] == -1 ? 1u : (u32)norm[s];
    for (u32 u = 0; u < size; u++) {
      if (spread[u] != s) continue;
      u32 nbBits = tableLog - hb32(x);
      u32 nextBase = (x << nbBits) - size;
      table[u] = mk_cell(s, kind == 0 ? c_ll_bits[s] : kind == 1 ? c_ml_bits[s] : s, nbBits, nextBase);
      x++;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// cooperative byte copy global->global by the calling group of `nthreads` threads (rank `t`)
__device__ __forceinline__ void copy_bytes(u8* dst, const u8* src, u32 n, int t, int nthreads) {
  // 16-byte body when both are 16B-aligned relative to each other is not guaranteed; use 8-byte unaligned moves
  u32 n8 = n >> 3;
  for (u32 i = t; i < n8; i += nthreads) st64(dst + 8 * i, ld64(src + 8 * i));
  for (u32 i = (n8 << 3) + t; i < n; i += nthreads) dst[i] = src[i];
}
__device__ __forceinline__ u32 bcast_u32(u32 v, u32 l) { return (u32)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ void fill_bytes(u8* dst, u8 v, u32 n, int t, int nthreads) {
  u64 vv = 0x0101010101010101ull * v;
  u32 n8 = n >> 3;
  for (u32 i = t; i < n8; i += nthreads) st64(dst + 8 * i, vv);
  for (u32 i = (n8 << 3) + t; i < n; i += nthreads) dst[i] = v;
}

// ---------------------------------------------------------------------------------------------
// literals section header + Huffman tree description; thread 0 only. Sets S.lit* / S.huf* / S.err.
__device__ void parse_literals_header(DecShared& S, const u8* src, u32 n, const u8* lim) {
  if (n < 1) { S.err = ZE_CORRUPTION; return; }
  u32 b0 = src[0], type = b0 & 3, sf = (b0 >> 2) & 3;
  S.litType = type;
  if (type < 2) {
    u32 size, lh;
    if (sf == 0 || sf == 2) { size = b0 >> 3; lh = 1; }
    else if (sf == 1) { if (n < 2) { S.err = ZE_CORRUPTION; return; } size = ld16(src) >> 4; lh = 2; }
    else { if (n < 3) { S.err = ZE_CORRUPTION; return; } size = ld24(src) >> 4; lh = 3; }
    if (size > BLOCK_MAX) { S.err = ZE_CORRUPTION; return; }
    u32 payload = type == 0 ? size : 1;
    if (lh + payload > n) { S.err = ZE_CORRUPTION; return; }
    S.litRegen = size; S.litHdr = lh; S.litComp = payload;
    if (type == 1) S.litRle = src[lh];
    return;
  }
  u32 need = sf < 2 ? 3 : sf == 2 ? 4 : 5;
  if (n < need) { S.err = ZE_CORRUPTION; return; }
  u32 regen, comp, lh, streams;
  if (sf < 2) { u32 v = ld24(src); regen = (v >> 4) & 0x3FF; comp = v >> 14; lh = 3; streams = sf == 0 ? 1 : 4; }
  else if (sf == 2) { u32 v = ld32(src); regen = (v >> 4) & 0x3FFF; comp = v >> 18; lh = 4; streams = 4; }
  else { u64 v = (u64)ld32(src) | ((u64)src[4] << 32); regen = (u32)(v >> 4) & 0x3FFFF; comp = (u32)(v >> 22); lh = 5; streams = 4; }
  if (regen > BLOCK_MAX || lh + comp > n) { S.err = ZE_CORRUPTION; return; }
  S.litRegen = regen; S.litComp = comp; S.litHdr = lh; S.litStreams = streams;
  const u8* p = src + lh; u32 rem = comp;
  if (type == 2) {
    // ---- tree description -> S.weights[0..nSym)
    if (rem < 1) { S.err = ZE_CORRUPTION; return; }
    u32 hbyte = p[0], nw = 0, used;
    if (hbyte >= 128) {
      nw = hbyte - 127; used = 1 + (nw + 1) / 2;
      if (used > rem) { S.err = ZE_CORRUPTION; return; }
      for (u32 i = 0; i < nw; i += 2) { u32 b = p[1 + i / 2]; S.weights[i] = (u8)(b >> 4); S.weights[i + 1] = (u8)(b & 15); }
    } else {
      used = 1 + hbyte;
      if (used > rem || hbyte < 1) { S.err = ZE_CORRUPTION; return; }
      u32 maxSym = 255, tl;
      u32 h = read_ncount(S.norm, &maxSym, &tl, p + 1, hbyte, 6, lim);
      if (!h) { S.err = ZE_CORRUPTION; return; }
      // small serial FSE decode of the weights (<= 255 symbols); table (<= 64 cells) in the not-yet-filled Huffman table
      u64* const wt = (u64*)S.huf;                 // 64 cells max (accuracy <= 6)
      u16* const next = S.huf + 1024;
      {
        u32 size = 1u << tl, mask = size - 1, high = size - 1, step = (size >> 1) + (size >> 3) + 3, pos = 0;
        for (u32 s = 0; s <= maxSym; s++) { next[s] = S.norm[s] == -1 ? 1 : (u16)S.norm[s]; if (S.norm[s] == -1) S.spread[high--] = (u8)s; }
        for (u32 s = 0; s <= maxSym; s++)
          for (int i = 0; i < S.norm[s]; i++) { S.spread[pos] = (u8)s; pos = (pos + step) & mask; while (pos > high) pos = (pos + step) & mask; }
        if (pos != 0) { S.err = ZE_CORRUPTION; return; }
        for (u32 u = 0; u < size; u++) {
          u32 s = S.spread[u], x = next[s]++;
          u32 nbBits = tl - hb32(x);
          wt[u] = mk_seqsym(s, 0, nbBits, (x << nbBits) - size);
        }
      }
      BitR br;
      if (br.init(p + 1 + h, hbyte - h, lim)) { S.err = ZE_CORRUPTION; return; }
      u32 s1 = br.read((int)tl), s2 = br.read((int)tl);
      for (;;) {
        if (nw >= 254) { S.err = ZE_CORRUPTION; return; }
        u64 e1 = wt[s1];
        S.weights[nw++] = (u8)e1;
        s1 = (u32)(e1 >> 48) + br.read((int)((e1 >> 40) & 0xFF));
        if (br.pos < 0) { S.weights[nw++] = (u8)wt[s2]; break; }
        if (nw >= 254) { S.err = ZE_CORRUPTION; return; }
        u64 e2 = wt[s2];
        S.weights[nw++] = (u8)e2;
        s2 = (u32)(e2 >> 48) + br.read((int)((e2 >> 40) & 0xFF));
        if (br.pos < 0) { S.weights[nw++] = (u8)wt[s1]; break; }
      }
    }
    u32 total = 0;
    for (u32 i = 0; i < nw; i++) { u32 w = S.weights[i]; if (w > 11) { S.err = ZE_CORRUPTION; return; } total += (1u << w) >> 1; }
    if (total == 0) { S.err = ZE_CORRUPTION; return; }
    u32 maxBits = hb32(total) + 1;
    if (maxBits > 11) { S.err = ZE_CORRUPTION; return; }
    u32 rest = (1u << maxBits) - total;
    if (rest == 0 || (rest & (rest - 1))) { S.err = ZE_CORRUPTION; return; }
    S.weights[nw] = (u8)(hb32(rest) + 1);
    S.hufNSym = nw + 1; S.hufMaxBits = maxBits;
    // start cell of each weight class (cells ordered by weight ascending)
    // counts per weight, then turned in place into start cells (kept in LDS: a dynamically indexed local array would live in scratch)
    for (u32 w = 0; w <= 12; w++) S.rankStart[w] = 0;
    for (u32 i = 0; i <= nw; i++) S.rankStart[S.weights[i]]++;
    u32 acc = 0;
    for (u32 w = 1; w <= maxBits; w++) { const u32 c = S.rankStart[w]; S.rankStart[w] = acc; acc += c << (w - 1); }
    if (acc != (1u << maxBits)) { S.err = ZE_CORRUPTION; return; }
    S.hufValid = 2;   // 2 = new table to be filled by the workgroup
    p += used; rem -= used;
  } else {
    if (!S.hufValid) { S.err = ZE_DICT_CORRUPTED; return; }
  }
  // stream layout
  u32 base = (u32)(p - src);
  if (streams == 1) { S.streamOff[0] = base; S.streamLen[0] = rem; }
  else {
    if (rem < 10) { S.err = ZE_CORRUPTION; return; }
    u32 s1 = ld16(p), s2 = ld16(p + 2), s3 = ld16(p + 4);
    if (6 + s1 + s2 + s3 > rem) { S.err = ZE_CORRUPTION; return; }
    u32 seg = (regen + 3) / 4;
    if (seg * 3 > regen) { S.err = ZE_CORRUPTION; return; }
    S.streamOff[0] = base + 6; S.streamLen[0] = s1;
    S.streamOff[1] = base + 6 + s1; S.streamLen[1] = s2;
    S.streamOff[2] = base + 6 + s1 + s2; S.streamLen[2] = s3;
    S.streamOff[3] = base + 6 + s1 + s2 + s3; S.streamLen[3] = rem - 6 - s1 - s2 - s3;
  }
}

// sequences section header (nbSeq, modes, table descriptions -> S.norm per table is consumed immediately
// by the workgroup, so this only parses nbSeq + modes; tables are parsed one at a time). thread 0.
__device__ void parse_seq_header(DecShared& S, const u8* p, u32 rem) {
  if (rem < 1) { S.err = ZE_SRCSIZE_WRONG; return; }
  u32 nb = p[0], used;
  if (nb == 0) { used = 1; if (rem != 1) { S.err = ZE_CORRUPTION; return; } }
  else if (nb < 128) used = 1;
  else if (nb < 255) { if (rem < 2) { S.err = ZE_SRCSIZE_WRONG; return; } nb = ((nb - 128) << 8) + p[1]; used = 2; }
  else { if (rem < 3) { S.err = ZE_SRCSIZE_WRONG; return; } nb = (u32)p[1] + ((u32)p[2] << 8) + 0x7F00; used = 3; }
  S.nbSeq = nb;
  if (nb) {
    if (rem < used + 1) { S.err = ZE_SRCSIZE_WRONG; return; }
    S.seqModes = p[used]; used++;
    if (S.seqModes & 3) { S.err = ZE_CORRUPTION; return; }
  }
  S.seqPos += used;
}

// one LL/ML/OF table: thread 0 parses (fills S.norm + logs), then wave 0 builds. returns via S fields.
// kind 0 LL, 1 ML, 2 OF
__device__ void seq_table_parse(DecShared& S, int kind, u32 mode, const u8* p, u32 rem, u32* tlOut, u32* msOut, u32* usedOut, const u8* lim) {
  const u32 maxSymK = kind == 0 ? 35 : kind == 1 ? 52 : 31;
  const u32 maxALK = kind == 2 ? 8 : 9;
  *usedOut = 0;
  if (mode == 0) {
    u32 ms = kind == 0 ? 35 : kind == 1 ? 52 : 28;
    for (u32 s = 0; s <= ms; s++) S.norm[s] = kind == 0 ? c_ll_defnorm[s] : kind == 1 ? c_ml_defnorm[s] : c_of_defnorm[s];
    *tlOut = kind == 2 ? 5 : 6; *msOut = ms;
  } else if (mode == 1) {
    if (rem < 1 || p[0] > maxSymK) { S.err = ZE_CORRUPTION; return; }
    *tlOut = 0; *msOut = p[0]; *usedOut = 1;
  } else if (mode == 2) {
    u32 ms = maxSymK, tl;
    u32 h = read_ncount(S.norm, &ms, &tl, p, rem, maxALK, lim);
    if (!h) { S.err = ZE_CORRUPTION; return; }
    *tlOut = tl; *msOut = ms; *usedOut = h;
  }
}

}  // namespace

// =================================================================================================
extern "C" __global__ void __launch_bounds__(DEC_THREADS, 6)   // 6 waves/SIMD; LDS (7 KiB per frame) admits 22 frames per CU
zra_decode_frames_kernel(ZraDecodeArgs a) {
  __shared__ DecShared S;
  const int lane = threadIdx.x;
  u8* const litScratch = a.litScratch + (size_t)blockIdx.x * ZRA_LIT_STRIDE;
  if (lane < 36) S.llBase[lane] = c_ll_base[lane];
  if (lane < 53) S.mlBase[lane] = c_ml_base[lane];

  for (;;) {
    if (lane == 0) S.frame = atomicAdd(a.queue, 1u);
    wsync();
    const u32 f = S.frame;
    wsync();
    if (f >= a.nFrames) return;

    const u64 so = a.frameOff[(size_t)f * a.offStride], se = a.frameOff[(size_t)f * a.offStride + 1];
    const u8* const src = a.body + so;
    const u32 srcSize = (u32)(se - so);
    const u8* const lim = a.body + a.bodySize;
    u8* const dst = a.out + a.outOff[f];
    const u32 dstCap = a.outCap[f];

    if (lane == 0) {
      S.err = 0; S.produced = 0; S.hufValid = 0; S.llValid = S.mlValid = S.ofValid = 0;
      S.rep[0] = 1; S.rep[1] = 4; S.rep[2] = 8; S.blkLast = 0;
      // ---- frame header (A.1)
      u32 hs = 0, checksum = 0;
      if (se < so || se > a.bodySize || srcSize < 5) S.err = ZE_SRCSIZE_WRONG;
      else if (ld32(src) != 0xFD2FB528u) S.err = ZE_PREFIX_UNKNOWN;
      else {
        u32 fhd = src[4], did = fhd & 3, ss = (fhd >> 5) & 1, fcs = fhd >> 6;
        u32 didSize = did == 3 ? 4 : did;
        u32 fcsSize = fcs == 0 ? ss : fcs == 1 ? 2 : fcs == 2 ? 4 : 8;
        hs = 5 + !ss + didSize + fcsSize;
        if (srcSize < hs) S.err = ZE_SRCSIZE_WRONG;
        else if (fhd & 8) S.err = ZE_FRAMEPARAM_UNSUPPORTED;
        else if (!ss) {
          u32 b = src[5], wl = 10 + (b >> 3);
          if (wl > 31) S.err = ZE_FRAMEPARAM_UNSUPPORTED;
          else if (((1ull << wl) + ((1ull << wl) >> 3) * (b & 7)) > (1ull << 27) + 1) S.err = ZE_WINDOW_TOO_LARGE;
        }
        checksum = (fhd >> 2) & 1;
        S.fcsHave = 0;
        if (!S.err) {
          // a dictionary id cannot be honoured (the reference never loads one): dictionary_wrong, as ZSTD_decompressFrame reports it;
          // the declared content size (1/2/4/8 bytes, the 2-byte form biased by 256) must equal what the frame regenerates
          const u8* q = src + 5 + !ss;
          const u32 dict = did == 0 ? 0u : did == 1 ? (u32)q[0] : did == 2 ? (u32)ld16(q) : ld32(q);
          if (dict) S.err = ZE_DICT_WRONG;
          q += didSize;
          if (fcsSize) {
            const u64 v = fcsSize == 1 ? (u64)q[0] : fcsSize == 2 ? (u64)ld16(q) + 256 : fcsSize == 4 ? (u64)ld32(q) : ld64(q);
            S.fcsLo = (u32)v; S.fcsHi = (u32)(v >> 32); S.fcsHave = 1;
          }
        }
      }
      S.blkPos = hs;
      a.frameMeta[2 * (size_t)f] = checksum;   // [2f] = has checksum, [2f+1] = stored checksum (set at frame end)
    }
    wsync();

    // ------------------------------------------------------------------ block loop
    for (;;) {
      const bool done = S.err || S.blkLast;    // sampled by every lane before lane 0 may overwrite it
      wsync();
      if (done) break;
      if (lane == 0) {
        u32 pos = S.blkPos;
        if (srcSize - pos < 3) S.err = ZE_SRCSIZE_WRONG;
        else {
          u32 bh = ld24(src + pos);
          S.blkLast = bh & 1; S.blkType = (bh >> 1) & 3; S.blkSize = bh >> 3;
          pos += 3;
          u32 payload = S.blkType == 1 ? 1 : S.blkSize;
          if (S.blkType == 3) S.err = ZE_CORRUPTION;
          else if (payload > srcSize - pos) S.err = ZE_SRCSIZE_WRONG;
          else if (S.blkType == 2 && S.blkSize >= BLOCK_MAX) S.err = ZE_SRCSIZE_WRONG;   // the constant, not the window-derived maximum (as the dependency's one-shot decoder)
          else if (S.blkType != 2 && S.blkSize > dstCap - S.produced) S.err = ZE_DSTSIZE_TOOSMALL;
          S.blkPos = pos;
        }
      }
      wsync();
      if (S.err) break;
      const u32 btype = S.blkType, bsize = S.blkSize, bpos = S.blkPos, produced0 = S.produced;
      u8* const out = dst + produced0;           // this block's output start
      const u32 outCap = dstCap - produced0;

      if (btype == 0 || btype == 1) {
        if (btype == 0) copy_bytes(out, src + bpos, bsize, lane, DEC_THREADS);
        else fill_bytes(out, src[bpos], bsize, lane, DEC_THREADS);
        wsync();
        if (lane == 0) { S.produced = produced0 + bsize; S.blkPos = bpos + (btype == 0 ? bsize : 1); }
        wsync();
        continue;
      }

      // ------------------------------------------------------------ compressed block
      const u8* const blk = src + bpos;
      if (lane == 0) {
        S.litStreams = 1; S.litRle = 0;
        parse_literals_header(S, blk, bsize, lim);
        S.seqHdrErr = 0;
        if (!S.err) {
          // the sequences header is parsed now (its fields steer the literal stage's scratch), but an error in it is only raised
          // after the literals have been decoded: the reference decodes the literals section first (ZSTD_decodeLiteralsBlock)
          S.seqPos = S.litHdr + S.litComp;
          parse_seq_header(S, blk + S.seqPos, bsize - S.seqPos);
          S.seqHdrErr = S.err; S.err = 0;
          if (S.seqHdrErr) S.nbSeq = 0;
        }
      }
      wsync();
      if (S.err) break;

      // ---- Huffman decode table fill: cells ordered by weight, then symbol
      if (S.litType >= 2) {                              // new table, or treeless: rebuilt from the kept weights (LDS shared with llT)
        const u32 nSym = S.hufNSym, maxBits = S.hufMaxBits;
        for (u32 sy = lane; sy < nSym; sy += DEC_THREADS) {
          u32 w = S.weights[sy];
          if (w == 0) continue;
          u32 before = 0;
          for (u32 t = 0; t < sy; t++) before += (S.weights[t] == w);
          u32 len = 1u << (w - 1), start = S.rankStart[w] + before * len;
          u16 e = (u16)(sy | ((maxBits + 1 - w) << 8));
          for (u32 c = 0; c < len; c++) S.huf[start + c] = e;
        }
        wsync();
        if (lane == 0) S.hufValid = 1;
        wsync();
      }
      const u32 nbSeq = S.nbSeq;
      // ---- literals: raw -> point into the source; RLE -> fill scratch; Huffman -> up to 4 lanes, one per stream
      const u32 litType = S.litType, regen = S.litRegen;
      const u8* lit = litScratch;
      if (litType == 0) lit = blk + S.litHdr;
      else if (litType == 1) fill_bytes(litScratch, (u8)S.litRle, regen, lane, DEC_THREADS);
      else {
        const u32 nStreams = S.litStreams;
        if ((u32)lane < nStreams && !(a.debugSkip & 4)) {
          const u32 seg = nStreams == 1 ? regen : (regen + 3) / 4;
          const u32 myLen = nStreams == 1 ? regen : (lane < 3 ? seg : regen - 3 * seg);
          u8* o = litScratch + (size_t)lane * seg;
          BitR hb;
          bool bad = hb.init(blk + S.streamOff[lane], S.streamLen[lane], lim) != 0;
          if (!bad) {
            const int mb = (int)S.hufMaxBits;
            u32 i = 0;
            for (; i + 4 <= myLen; i += 4) {        // 4 symbols per window reload (4*11 = 44 <= 56 guaranteed bits)
              hb.ensure(4 * mb);
              u32 packed = 0;
#pragma unroll
              for (int k = 0; k < 4; k++) {
                u32 e = S.huf[hb.peek(mb)];
                packed |= (e & 0xFF) << (8 * k);
                hb.skip((int)(e >> 8));
              }
              st32(o + i, packed);
            }
            for (; i < myLen; i++) { hb.ensure(mb); u32 e = S.huf[hb.peek(mb)]; o[i] = (u8)e; hb.skip((int)(e >> 8)); }
            if (hb.pos != 0) bad = true;
          }
          if (bad) S.err = ZE_CORRUPTION;
        }
      }
      wsync();
      if (S.err) break;
      if (S.seqHdrErr) { wsync(); if (lane == 0) S.err = S.seqHdrErr; wsync(); break; }

      // ---- sequence decode tables: lane 0 parses each description, the wave builds it
      if (nbSeq) {
        bool bad = false;
        for (int kind = 0; kind < 3 && !bad; kind++) {
          const int k = kind == 0 ? 0 : kind == 1 ? 2 : 1;          // wire order is LL, OF, ML
          const u32 mode = k == 0 ? (S.seqModes >> 6) : k == 2 ? ((S.seqModes >> 4) & 3) : ((S.seqModes >> 2) & 3);
          if (lane == 0) {
            u32 tl = 0, ms = 0, used = 0;
            seq_table_parse(S, k, mode, blk + S.seqPos, bsize - S.seqPos, &tl, &ms, &used, lim);
            S.seqPos += used; S.tl = tl; S.ms = ms;
            if (mode == 3) { u32 v = k == 0 ? S.llValid : k == 1 ? S.mlValid : S.ofValid; if (!v) S.err = ZE_CORRUPTION; }
          }
          wsync();
          if (S.err) { bad = true; break; }
          u32* table = k == 0 ? S.llT : k == 1 ? S.mlT : S.ofT;
          const u32 tl = S.tl, ms = S.ms;
          if (mode == 1) {
            if (lane == 0) table[0] = mk_cell(ms, k == 0 ? c_ll_bits[ms] : k == 1 ? c_ml_bits[ms] : ms, 0, 0);
            if (k == 0 && lane == 0) S.llSaveRle = 1 + ms;
            if (k == 1 && lane == 0) S.mlSaveRle = 1 + ms;
          } else if (mode != 3) {
            build_fse_dtable(table, S.norm, ms, tl, k, S.spread, lane);
            // remember how to rebuild LL / ML (their LDS is reused by the next block's Huffman table)
            if (k == 0) {
              if ((u32)lane <= ms) S.llNorm[lane] = S.norm[lane];
              if (lane == 0) { S.llSaveLog = tl; S.llSaveMax = ms; S.llSaveRle = 0; }
            } else if (k == 1) {
              if ((u32)lane <= ms) S.mlNorm[lane] = S.norm[lane];
              if (lane == 0) { S.mlSaveLog = tl; S.mlSaveMax = ms; S.mlSaveRle = 0; }
            }
          } else if (k == 0) {                             // repeat mode: the literal decode of this block overwrote the table
            if (S.llSaveRle) { if (lane == 0) { const u32 sy = S.llSaveRle - 1; table[0] = mk_cell(sy, c_ll_bits[sy], 0, 0); } }
            else build_fse_dtable(table, S.llNorm, S.llSaveMax, S.llSaveLog, 0, S.spread, lane);
          } else if (k == 1) {
            if (S.mlSaveRle) { if (lane == 0) { const u32 sy = S.mlSaveRle - 1; table[0] = mk_cell(sy, c_ml_bits[sy], 0, 0); } }
            else build_fse_dtable(table, S.mlNorm, S.mlSaveMax, S.mlSaveLog, 1, S.spread, lane);
          }
          if (lane == 0 && mode != 3) {
            if (k == 0) { S.llLog = tl; S.llValid = 1; } else if (k == 1) { S.mlLog = tl; S.mlValid = 1; } else { S.ofLog = tl; S.ofValid = 1; }
          }
          wsync();
        }
        if (bad) break;
      }

      // ---- sequences: batches of 64 (lane 0 decodes, the wave executes)
      BitR br; br.base = blk; br.lim = lim; br.pos = 0; br.wlo = 0; br.w = 0;
      u32 sLL = 0, sML = 0, sOF = 0;
      u32 rep0 = S.rep[0], rep1 = S.rep[1], rep2 = S.rep[2];
      if (lane == 0 && nbSeq) {
        if (br.init(blk + S.seqPos, bsize - S.seqPos, lim)) S.err = ZE_CORRUPTION;
        else {
          sLL = br.read((int)S.llLog); sOF = br.read((int)S.ofLog); sML = br.read((int)S.mlLog);
          if (br.pos < 0) S.err = ZE_CORRUPTION;
        }
      }
      wsync();
      if (S.err) break;
      u32 outBase = 0, litBase = 0;              // running positions (wave-uniform)
      bool fail = false;
      for (u32 first = 0; first < nbSeq; first += BATCH) {
        const u32 cntAll = min((u32)BATCH, nbSeq - first);
        // -------- stage A: lane 0 — FSE sequence decode of this batch
        if (lane == 0) {
          // A malformed sequence stops the decode, but the sequences before it are still executed first: the reference decodes and
          // executes one sequence at a time, so an execution error of an earlier sequence wins over the decode error of a later one
          u32 bad = 0, valid = cntAll;
          for (u32 i = 0; i < cntAll; i++) {
            const u32 eL = S.llT[sLL], eM = S.mlT[sML], eO = S.ofT[sOF];
            const u32 ofBits = (eO >> 8) & 0xFF, mlBits = (eM >> 8) & 0xFF, llBits = (eL >> 8) & 0xFF;
            if (ofBits > 31) { bad = 1; valid = i; break; }
            const u32 baseL = S.llBase[eL & 0xFF], baseM = S.mlBase[eM & 0xFF];   // by symbol; not on the bit-position chain
            // offset, match-length and literal-length extra bits in ONE extraction when they fit the 57 bits a reload guarantees
            // (always, for windows <= 128 KiB); bitstream order: OF, ML, LL = topmost ... lowest
            const u32 lm = mlBits + llBits, t1 = ofBits + lm;
            u32 offVal, both;
            if (t1 <= 56) {
              br.ensure((int)t1);
              const u64 x = (br.w >> (br.pos - (i32)t1 - br.wlo)) & ((1ull << t1) - 1);
              br.pos -= (i32)t1;
              offVal = (1u << ofBits) + (u32)(x >> lm);
              both = (u32)(x & ((1ull << lm) - 1));
            } else {
              offVal = (1u << ofBits) + br.read((int)ofBits);
              both = br.read((int)lm);
            }
            const u32 ml = baseM + (both >> llBits), ll = baseL + (both & ((1u << llBits) - 1));
            if (first + i + 1 < nbSeq) {
              const int nL = (int)((eL >> 16) & 0xF), nM = (int)((eM >> 16) & 0xF), nO = (int)((eO >> 16) & 0xF);
              // LL, ML, OF state bits in one extraction (<= 26 bits): LL is read first = topmost
              const u32 st = br.read(nL + nM + nO);
              sLL = (eL >> 20) + (st >> (nM + nO));
              sML = (eM >> 20) + ((st >> nO) & ((1u << nM) - 1));
              sOF = (eO >> 20) + (st & ((1u << nO) - 1));
            }
            if (br.pos < 0) { bad = 1; valid = i; break; }
            u32 off;                                   // repcode resolution (A.3)
            if (offVal > 3) { off = offVal - 3; rep2 = rep1; rep1 = rep0; rep0 = off; }
            else {
              u32 idx = offVal + (ll == 0);
              if (idx == 1) off = rep0;
              else {
                off = idx == 2 ? rep1 : idx == 3 ? rep2 : rep0 - 1;
                off += !off;                               // zstd 1.4.9 forces a zero offset to 1 (no error)
                if (idx != 2) rep2 = rep1;
                rep1 = rep0; rep0 = off;
              }
            }
            S.seqLL[i] = ll; S.seqML[i] = ml; S.seqOF[i] = off;
          }
          if (!bad && first + cntAll == nbSeq && br.pos != 0) bad = 1;     // checked after the last sequence has been executed
          S.batchValid = bad ? (valid | 0x80000000u) : valid;
        }
        wsync();
        const u32 bv = S.batchValid;
        const bool chainBad = bv >> 31;
        const u32 cnt = bv & 0x7FFFFFFFu;
        // -------- stage B: positions by wave prefix scan; every lane copies its sequence's literals
        const bool act = (u32)lane < cnt;
        const u32 ll = act ? S.seqLL[lane] : 0, ml = act ? S.seqML[lane] : 0, off = act ? S.seqOF[lane] : 1;
        const u32 tot = ll + ml;
        const u32 incT = wave_incl_scan(tot), incL = wave_incl_scan(ll);
        const u32 oStart = outBase + incT - tot, lStart = litBase + incL - ll;
        const u32 mdst = oStart + ll;
        u32 e = 0;
        if (act) {
          // ZSTD_execSequenceEnd order: destination room first, then the literal buffer, then the offset
          if (oStart > outCap || tot > outCap - oStart) e = ZE_DSTSIZE_TOOSMALL;
          else if (lStart > regen || ll > regen - lStart) e = ZE_CORRUPTION;
          else if (off > produced0 + mdst) e = ZE_CORRUPTION;
        }
        const u64 em = __ballot(e != 0);
        if (em) { if ((u32)lane == (u32)__builtin_ctzll(em)) S.err = e; wsync(); fail = true; break; }
        {
          const u8* lp = lit + lStart; u8* op = out + oStart;
          const bool longLit = ll > 32;
          if (!longLit && !(a.debugSkip & 2)) for (u32 b = 0; b < ll; b++) op[b] = lp[b];
          u64 lm = __ballot(longLit);
          while (lm) {                               // long literal runs: the whole wave copies, coalesced
            const u32 j = (u32)__builtin_ctzll(lm); lm &= lm - 1;
            const u32 jl = bcast_u32(ll, j), jo = bcast_u32(oStart, j), js = bcast_u32(lStart, j);
            copy_bytes(out + jo, lit + js, jl, lane, DEC_THREADS);
          }
        }
        wsync();
        // -------- stage C: match copies in dependency rounds
        {
          const u32 msrc = mdst - off;
          const u32 msrcEnd = min(msrc + ml, mdst);
          u64 pending = (a.debugSkip & 1) ? 0ull : __ballot(act);
          while (pending) {
            const u32 fnd = (u32)__builtin_ctzll(pending);
            const u32 frontier = bcast_u32(mdst, fnd);
            const bool mine = (pending >> lane) & 1;
            const bool ready = mine && (msrcEnd <= frontier || (u32)lane == fnd);
            const bool longM = ready && ml > 64;
            if (ready && !longM) {
              u8* dp = out + mdst; const u8* sp = dp - off;
              if (off >= ml) {                           // no overlap: 8-byte moves + byte tail
                u32 k = 0;
                for (; k + 8 <= ml; k += 8) st64(dp + k, ld64(sp + k));
                for (; k < ml; k++) dp[k] = sp[k];
              } else {                                   // overlapping match = period `off`: read only bytes in front of the destination
                u32 j = 0;
                for (u32 k = 0; k < ml; k++) { dp[k] = sp[j]; j = j + 1 == off ? 0 : j + 1; }
              }
            }
            u64 lmk = __ballot(longM);
            while (lmk) {                            // long matches: the whole wave copies (period-safe modular source)
              const u32 j = (u32)__builtin_ctzll(lmk); lmk &= lmk - 1;
              const u32 jml = bcast_u32(ml, j), jd = bcast_u32(mdst, j), jof = bcast_u32(off, j);
              u8* dp = out + jd; const u8* sp = dp - jof;
              if (jof >= jml) { for (u32 k = lane; k < jml; k += WAVE) dp[k] = sp[k]; }
              else { for (u32 k = lane; k < jml; k += WAVE) dp[k] = sp[k % jof]; }
            }
            pending &= ~__ballot(ready);
            wsync();
          }
        }
        outBase += bcast_u32(incT, 63); litBase += bcast_u32(incL, 63);
        if (chainBad) { if (lane == 0) S.err = ZE_CORRUPTION; wsync(); fail = true; break; }
      }
      if (fail || S.err) break;

      // ---- block tail: remaining literals
      if (litBase > regen) { if (lane == 0) S.err = ZE_CORRUPTION; }
      else if (outBase > outCap || regen - litBase > outCap - outBase) { if (lane == 0) S.err = ZE_DSTSIZE_TOOSMALL; }
      else copy_bytes(out + outBase, lit + litBase, regen - litBase, lane, DEC_THREADS);
      wsync();
      if (lane == 0 && !S.err) {
        S.produced = produced0 + outBase + (regen - litBase);
        S.blkPos = bpos + bsize;
        S.rep[0] = rep0; S.rep[1] = rep1; S.rep[2] = rep2;
      }
      wsync();
    }

    // ------------------------------------------------------------------ frame end
    if (lane == 0) {
      u32 err = S.err;
      if (!err) {
        u32 pos = S.blkPos;
        if (S.fcsHave && (S.fcsHi != 0 || S.fcsLo != S.produced)) err = ZE_CORRUPTION;   // declared size first, then the checksum
        if (!err && a.frameMeta[2 * (size_t)f]) {
          if (srcSize - pos < 4) err = ZE_CHECKSUM_WRONG;
          else { a.frameMeta[2 * (size_t)f + 1] = ld32(src + pos); pos += 4; }
        }
        if (!err && pos != srcSize) err = ZE_SRCSIZE_WRONG;          // seek table and frame walk disagree
      }
      a.status[f] = err;
      a.produced[f] = S.produced;
    }
    wsync();
  }
}
