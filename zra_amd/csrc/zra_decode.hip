// zra_amd — frame DECODE kernels for gfx950 (MI355X).
//
// Replaces the reference's per-frame ZSTD_decompressDCtx work (zra.cpp:249,280,289,293,397,406,410,435).
// A zstd frame is a chain of serial sections (header parses, the FSE sequence bit chain) around data-parallel ones (table fills,
// literal and match copies). One wave per frame left 63 lanes idle in the serial sections, and the sequence chain alone was more
// than half of all instructions. So the work of a frame is split by its SHAPE into four kernels that run once per round
// (a round = one compressed block of every unfinished frame; 64 KiB ZRA frames take exactly one):
//
//   zra_dec_parse_kernel   wave per frame. Frame / block / literal / sequence headers and the bit-serial table descriptions (lane 0);
//                          the Huffman weights (FSE table in a register, one cell per lane), the tree's rank arithmetic and the
//                          three FSE decode tables wave-wide (built in LDS, stored as 4-byte cells to the frame's table scratch in
//                          HBM). Raw and RLE blocks and the frame end are handled on the way; a compressed block is handed on.
//   zra_dec_huf_kernel     the whole wave on one frame's literal streams: 16 self-synchronising runs per stream decoded side by side
//                          from guessed starts, restarted from the predecessor's hand-over point until nothing moves, placed by a
//                          prefix sum of the counts (huf_decode_wave); damaged or unusual streams go to the serial X1 / X2 decoders
//                          of libzstd, one lane per stream. Literals to a bump-allocated scratch, sixteen at a time.
//   zra_dec_chain_kernel   LANE per frame: the FSE sequence chains of 64 frames advance together in the 64 lanes of a wave
//   zra_dec_chain_lds_kernel  (tables read from L2/HBM — or, in the second kernel, one workgroup per CU beside the first, from 31
//                          LDS slots — one dependent round trip per sequence). Every check of the reference's sequence loop lives
//                          here, in its order, so the execute kernel moves bytes without looking at them. Sequences go to a
//                          bump-allocated scratch.
//   zra_dec_exec_kernel    wave per frame: 64 sequences per step — scan of the lengths, the step assembled in an LDS window
//                          (literal runs, then match copies in dependency rounds: a match waits for the sequences its source
//                          touches), 16-byte stores, one drain; then block commit, frame end, and (random access) the query slices.
//   zra_ra_small_kernel    small random-access batches: one workgroup of three waves takes a frame through all of the above without
//                          leaving the kernel (six-lane sequence chain out of LDS, literals on a second wave, a third wave that
//                          validates and executes behind the chain); anything unusual is handed back to the four kernels.
//
// Statuses are results: the control flow restates libzstd 1.4.9's (oracle/zo_decode.c is the CPU twin, pinned against the library
// on 44,000 damaged archives): its BIT_DStream reader incl. what an over-read returns, both of its Huffman decoders, both of its
// sequence loops, its check order. Format per RFC 8878 / SURVEY.md Appendix A.1-A.3.
#include "zra_dev.h"
#include "zra_kernels.h"

using namespace zra_dev;

namespace {

constexpr int DEC_THREADS = 64;
constexpr int BATCH = 64;           // sequences executed per step (one per lane)
#ifndef ZRA_XWIN
#define ZRA_XWIN 4032
#endif
constexpr u32 XPRE = 64, XWIN = ZRA_XWIN; // execute stage: bytes in front of a step kept in LDS / largest step put together in LDS
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
// HUF_selectDecoder of zstd 1.4.9 (algoTime[Q][single|double]): {tableTime, decode256Time}
__constant__ u16 c_huf_t0[16][2] = {{0,0},{0,0},{38,130},{448,128},{556,128},{714,128},{883,128},{897,128},{926,128},{947,128},{1107,128},{1177,128},{1242,128},{1349,128},{1455,128},{722,128}};
__constant__ u16 c_huf_t1[16][2] = {{1,1},{1,1},{1313,74},{1353,74},{1353,74},{1418,74},{1437,74},{1515,75},{1613,75},{1729,77},{2083,81},{2379,87},{2415,93},{2644,106},{2422,124},{1891,145}};

// FSE decode cell of the Huffman-weight table (8 bytes): symbol | nbBits<<40 | nextBase<<48
__device__ __forceinline__ u64 mk_seqsym(u32 baseValue, u32 addBits, u32 nbBits, u32 nextBase) {
  return (u64)baseValue | ((u64)addBits << 32) | ((u64)nbBits << 40) | ((u64)nextBase << 48);
}
// low word of a decode-table cell for LL / ML / OF: symbol | extraBits<<8 | stateBits<<16 | nextBase<<20
__device__ __forceinline__ u32 mk_cell(u32 sym, u32 addBits, u32 nbBits, u32 nextBase) {
  return sym | (addBits << 8) | (nbBits << 16) | (nextBase << 20);
}

#ifdef ZRA_SMALL_PROFILE
// bring-up (never in the shipped library): s_memtime sums per stage of zra_ra_small_kernel: parse, Huffman (wave 1), chain (wave 0), execute, jobs
__device__ unsigned long long zra_small_prof[64];
extern "C" __attribute__((visibility("default"))) void ZraHipDebugReadSmallProfile(unsigned long long* out16, int reset) {
  (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(zra_small_prof), sizeof(unsigned long long) * 64);
  if (reset) { unsigned long long z[64] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(zra_small_prof), z, sizeof(z)); }
}
#define SPROF(k) { __builtin_amdgcn_s_waitcnt(0); const u64 n_ = __builtin_amdgcn_s_memtime(); if (lane == 0) atomicAdd(&zra_small_prof[k], n_ - spt_); spt_ = n_; }
#define SPROF_T0 u64 spt_ = __builtin_amdgcn_s_memtime();
#define SPROF_RESET spt_ = __builtin_amdgcn_s_memtime();
#else
#define SPROF(k)
#define SPROF_T0
#define SPROF_RESET
#endif
#ifdef ZRA_SMALL_PROFILE
#define PPROF_T0 u64 ppt_ = __builtin_amdgcn_s_memtime();
#define PPROF(k) if (FUSED) { __builtin_amdgcn_s_waitcnt(0); const u64 n_ = __builtin_amdgcn_s_memtime(); if (lane == 0) atomicAdd(&zra_small_prof[k], n_ - ppt_); ppt_ = n_; }
#define LPROF_T0 u64 lpt_ = __builtin_amdgcn_s_memtime();
#define LPROF(k) { __builtin_amdgcn_s_waitcnt(0); const u64 n_ = __builtin_amdgcn_s_memtime(); atomicAdd(&zra_small_prof[k], n_ - lpt_); lpt_ = n_; }
#else
#define PPROF_T0
#define PPROF(k)
#define LPROF_T0
#define LPROF(k)
#endif
constexpr u32 STAGE_BYTES = 512;     // header bytes copied to LDS per window (frame + block + literals header + tree description: <= 160)
struct __attribute__((aligned(16))) ParseShared {
  // lane 0 parses headers byte by byte: from HBM that is a dependent round trip per byte (~1-2 us each under load), so the wave
  // first copies the header region into LDS with one coalesced load (w0: from the block header on; w1: the sequences header)
  u8 w0[STAGE_BYTES + 16];
  u8 w1[STAGE_BYTES + 16];
  union {
    u64 wt[64];             // FSE decode table of the Huffman weights (accuracy <= 6)
    u32 stage[1024];        // one FSE table under construction (LL / ML: 512 cells x 2 words; OF: 256 x 1)
  };
  u16 wnext[256];           // scratch of the weight-table build
  short norm[256];          // scratch while a table is described
  u8 spread[512];
  u8 weights[256];
  u16 hufStart[256];        // first decode-table cell of every symbol (cells ordered by weight, then symbol)
  u32 rankStart[16];
  // control words (written by lane 0, read by the wave after a wave sync)
  u32 err, job;
  u32 blkType, blkSize, blkLast, blkPos, hdrPos, frameEnd, winPos, winLen;
  u32 litType, litRegen, litComp, litHdr, litStreams, litRle;
  u32 hufValid, hufMaxBits, hufNSym, hufX2;
  u32 hufNw, hufUsed, hufPhase;   // a tree description read by lane 0 (weights[0..hufNw)), waiting for the wave-wide part (phase 1);
  u32 hufTl, hufMaxSym, hufNcBytes, hufHbyte;   // ... or (phase 2) an FSE-coded one whose table description lane 0 has read into norm[]
  u32 nbSeq, seqPos, seqModes, seqTables;
  u32 llLog, mlLog, ofLog, llValid, mlValid, ofValid, ofShare;
  u32 rep[3];
  u32 streamOff[4], streamLen[4];
  u32 tl, ms, used;
  u32 produced, lateErr;
  u32 fcsLo, fcsHi, fcsHave, hasChecksum, bigWindow;
  u32 litKind, litArg, alloc;
  u64 litBase, seqBase;
};

// wave-level sync: LDS and global traffic of the wave is complete and visible to its other lanes
__device__ __forceinline__ void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ u32 bcast_u32(u32 v, u32 l) { return (u32)__builtin_amdgcn_readlane((int)v, (int)l); }
// the frame a job belongs to: jobs are frames — or, in the block-parallel pass (a.bpf > 1), blocks: job = frame * bpf + ordinal
__device__ __forceinline__ size_t frame_of(const ZraDecodeArgs& a, u32 j) { return a.bpf > 1 ? (size_t)(j / a.bpf) : (size_t)j; }
__device__ __forceinline__ u32 rfl(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
// lane l of `old` := val (val and l wave-uniform): v_writelane_b32 with the lane select in M0
__device__ __forceinline__ u32 wrlane_d(u32 old, u32 val, u32 l) {
  asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(old) : "s"(val), "s"(l) : "m0");
  return old;
}

// ---------------------------------------------------------------------------------------------
// FSE table description -> normalised counts; bytes consumed, 0 = corruption. One lane.
// Restates FSE_readNCount of libzstd 1.4.9 (lib/common/entropy_common.c) including its behaviour near the END of its input (oracle/
// zo_entropy.c: zo_fse_read_ncount has the long version): 32 bits are read at a byte position that never passes end - 4; a read position
// behind that is clamped there with the bit offset taken modulo 32, so a description that runs past the end re-reads earlier bits instead
// of failing, and only the position behind the last symbol is checked. Inputs shorter than 8 bytes are read from a zero-padded copy.
// (Round 5: the reader written from the format description — zeros past the end, every overrun an error — called a damaged frame corrupt
// that libzstd decodes: soak seed 91417.)
__device__ __forceinline__ u32 read_ncount(short* norm, u32* maxSymIO, u32* tableLogOut, const u8* src, u32 n, u32 maxAL, const u8* lim) {
  if (n < 1) return 0;
  const bool small = n < 8;
  u64 padv = 0;
  if (small) for (u32 i = 0; i < n; i++) padv |= (u64)src[i] << (8 * i);
  const i32 iend = small ? 8 : (i32)n;                   // positions are byte offsets from src
  auto rd32 = [&](i32 off) -> u32 { return small ? (u32)(padv >> (8 * off)) : ld32(src + off); };
  i32 ip = 0;
  const u32 maxSV1 = *maxSymIO + 1;
  u32 bitStream = rd32(0);
  i32 nbBits = (i32)(bitStream & 0xF) + 5, remaining, threshold, bitCount = 4;
  bool previous0 = false;
  u32 charnum = 0;
  if (nbBits > 15) return 0;
  bitStream >>= 4;
  const u32 AL = (u32)nbBits;
  remaining = (1 << nbBits) + 1; threshold = 1 << nbBits; nbBits++;
  for (u32 s = 0; s < maxSV1; s++) norm[s] = 0;
  auto advance = [&]() {
    if (ip <= iend - 7 || ip + (bitCount >> 3) <= iend - 4) { ip += bitCount >> 3; bitCount &= 7; }
    else { bitCount -= 8 * (iend - 4 - ip); bitCount &= 31; ip = iend - 4; }
    bitStream = rd32(ip) >> bitCount;
  };
  for (;;) {
    if (previous0) {
      i32 repeats = (i32)__builtin_ctz(~bitStream | 0x80000000u) >> 1;
      while (repeats >= 12) {
        charnum += 3 * 12;
        if (ip <= iend - 7) ip += 3;
        else { bitCount -= 8 * (iend - 7 - ip); bitCount &= 31; ip = iend - 4; }
        bitStream = rd32(ip) >> bitCount;
        repeats = (i32)__builtin_ctz(~bitStream | 0x80000000u) >> 1;
      }
      charnum += 3 * (u32)repeats;
      bitStream >>= 2 * repeats; bitCount += 2 * repeats;
      charnum += bitStream & 3; bitCount += 2;
      if (charnum >= maxSV1) break;
      advance();
    }
    const i32 mx = (2 * threshold - 1) - remaining;
    i32 count;
    if ((bitStream & (u32)(threshold - 1)) < (u32)mx) { count = (i32)(bitStream & (u32)(threshold - 1)); bitCount += nbBits - 1; }
    else { count = (i32)(bitStream & (u32)(2 * threshold - 1)); if (count >= threshold) count -= mx; bitCount += nbBits; }
    count--;
    if (count >= 0) remaining -= count; else remaining += count;
    norm[charnum++] = (short)count;
    previous0 = count == 0;
    if (remaining < threshold) {
      if (remaining <= 1) break;
      nbBits = (i32)hb32((u32)remaining) + 1;
      threshold = 1 << (nbBits - 1);
    }
    if (charnum >= maxSV1) break;
    advance();
  }
  if (remaining != 1 || charnum > maxSV1 || bitCount > 32 || AL > maxAL) return 0;
  *maxSymIO = charnum - 1;
  *tableLogOut = AL;
  const u32 h = (u32)ip + (u32)((bitCount + 7) >> 3);
  return (small && h > n) ? 0u : h;
}

// wave-wide inclusive scans on the DPP network (no LDS traffic): rows of 16 by row_shr 1/2/4/8, then the row totals by row_bcast 15 / 31
__device__ __forceinline__ u32 dpp_scan_add(u32 v) {
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
  return v;
}
__device__ __forceinline__ u32 dpp_scan_max(u32 v) {
  v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true));
  v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true));
  v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true));
  v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true));
  v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false));
  v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false));
  return v;
}

// Build an FSE decode table from norm[] into the LDS staging area — FSE_buildDTable / ZSTD_buildFSETable of the dependency, every step
// of it wave-wide (it was one lane's 512-step walk, twice: two thirds of the parse stage):
//  * spread: the serial walk visits pos_i = i * step mod size, i = 0, 1, ..., and hands the positions <= highThreshold, in that order, to
//    the symbols in ascending order, norm[s] each. So the r-th valid visit belongs to the symbol whose run of ranks holds r: ranks from a
//    ballot prefix over 64 visits at a time, rank -> symbol from a running maximum over "symbol s starts at rank cum[s]" marks.
//  * cells: cell u takes x = next[s]++ in ascending u. 64 cells at a time: the lanes of one symbol find each other through a 64-bit
//    mask per symbol in LDS (atomic OR), their order inside the chunk is a popcount below the lane, the run continues in next[s].
// kind: 0 = LL, 1 = ML (two words per cell: packed fields, base value), 2 = OF (one word per cell).
// tmp: 1 KiB of scratch that may alias `stage`; masks: 64 x u64; next: 64 x u16. maxSym <= 52.
__device__ __forceinline__ void build_fse_dtable(u32* stage, const short* norm, u32 maxSym, u32 tableLog, int kind, u8* spread, u64* masks, u16* next, int lane) {
  const u32 size = 1u << tableLog, mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
  const u32 s = (u32)lane;
  const u64 lt = (1ull << lane) - 1ull;
  const int nv = s <= maxSym ? (int)norm[s] : 0;
  const bool low = nv == -1;
  const u32 cnt = nv > 0 ? (u32)nv : 0u;
  const u64 lowM = __ballot(low);
  const u32 high = size - 1 - (u32)__popcll(lowM);
  u8* const M = (u8*)stage;            // marks, then the symbol of every rank
  for (u32 i = (u32)lane; i < (size >> 2); i += DEC_THREADS) ((u32*)M)[i] = 0;
  masks[lane] = 0;
  next[lane] = (u16)(low ? 1u : cnt);
  if (low) spread[size - 1 - (u32)__popcll(lowM & lt)] = (u8)s;
  const u32 cum = dpp_scan_add(cnt) - cnt;
  wsync();
  if (cnt) M[cum] = (u8)s;
  wsync();
  u32 carry = 0;
  for (u32 c = 0; c < size; c += DEC_THREADS) {
    const u32 r = c + (u32)lane;
    u32 v = r < size ? M[r] : 0u;
    v = max(dpp_scan_max(v), carry);
    carry = bcast_u32(v, 63);
    if (r < size) M[r] = (u8)v;
  }
  wsync();
  u32 rank0 = 0;
  for (u32 c = 0; c < size; c += DEC_THREADS) {
    const u32 i = c + (u32)lane;
    const u32 p = (i * step) & mask;
    const bool ok = i < size && p <= high;
    const u64 bm = __ballot(ok);
    if (ok) spread[p] = M[rank0 + (u32)__popcll(bm & lt)];
    rank0 += (u32)__popcll(bm);
  }
  wsync();
  // what a symbol's cells share: extra bits and base value, held by lane = symbol
  const u32 addS = s <= maxSym ? (kind == 0 ? c_ll_bits[s] : kind == 1 ? c_ml_bits[s] : s) : 0u;
  const u32 baseS = s <= maxSym ? (kind == 0 ? c_ll_base[s] : kind == 1 ? c_ml_base[s] : 0u) : 0u;
  for (u32 c = 0; c < size; c += DEC_THREADS) {
    const u32 u = c + (u32)lane;
    const bool on = u < size;
    const u32 sy = on ? (u32)spread[u] : 0u;
    if (on) atomicOr((unsigned long long*)&masks[sy], 1ull << lane);
    wsync();
    const u64 m = on ? masks[sy] : 0ull;
    const u32 x0 = next[sy];
    wsync();
    const u32 before = (u32)__popcll(m & lt);
    if (on && before == 0) { next[sy] = (u16)(x0 + (u32)__popcll(m)); masks[sy] = 0; }
    const u32 add = (u32)__shfl((int)addS, (int)sy, 64), base = (u32)__shfl((int)baseS, (int)sy, 64);
    if (on) {
      const u32 x = x0 + before;
      const u32 nbBits = tableLog - hb32(x);
      const u32 nextBase = (x << nbBits) - size;
      if (kind == 2) stage[u] = mk_cell(sy, add, nbBits, nextBase);
      else { stage[2 * u] = mk_cell(sy, add, nbBits, nextBase); stage[2 * u + 1] = base; }
    }
    wsync();
  }
}

// ---------------------------------------------------------------------------------------------
// cooperative byte copy global->global by the calling group of `nthreads` threads (rank `t`)
__device__ __forceinline__ void copy_bytes(u8* dst, const u8* src, u32 n, int t, int nthreads) {
  u32 n8 = n >> 3;
  for (u32 i = t; i < n8; i += nthreads) st64(dst + 8 * i, ld64(src + 8 * i));
  for (u32 i = (n8 << 3) + t; i < n; i += nthreads) dst[i] = src[i];
}
__device__ __forceinline__ void fill_bytes(u8* dst, u8 v, u32 n, int t, int nthreads) {
  u64 vv = 0x0101010101010101ull * v;
  u32 n8 = n >> 3;
  for (u32 i = t; i < n8; i += nthreads) st64(dst + 8 * i, vv);
  for (u32 i = (n8 << 3) + t; i < n; i += nthreads) dst[i] = v;
}

// Up to 64 bytes, one lane. Two rules shape this: (1) every load is issued before the first store — the compiler cannot do that
// across a load/store loop (the ranges might alias), and a loop of dependent HBM round trips is what it turns into; (2) as few
// vector memory INSTRUCTIONS as possible — each one occupies the CU's address unit for the whole wave whatever its width, and the
// execute kernel is bound by exactly that. So: 8-byte words, the last one placed to END at byte n (it overlaps its neighbour
// instead of leaving a tail of byte accesses); below 8 bytes the same with 4-, 2-, 1-byte accesses: never more than 2 + 2.
typedef u16 __attribute__((aligned(1))) u16_a1;
__device__ __forceinline__ void copy_le64(u8* dp, const u8* sp, u32 n) {
  if (n >= 8) {
    u64 w[7];
    const u32 lastAt = n - 8;
#pragma unroll
    for (int c = 0; c < 7; c++) { w[c] = 0; if (8u * c < lastAt) w[c] = ld64(sp + 8 * c); }   // words at 0, 8, ... below the last one
    const u64 wl = ld64(sp + lastAt);                                                          // the last word ends at byte n
#pragma unroll
    for (int c = 0; c < 7; c++) if (8u * c < lastAt) st64(dp + 8 * c, w[c]);
    st64(dp + lastAt, wl);
  } else if (n >= 4) {
    const u32 a0 = ld32(sp), a1 = ld32(sp + n - 4);
    st32(dp, a0); st32(dp + n - 4, a1);
  } else if (n >= 2) {
    const u32 a0 = *(const u16_a1*)sp, a1 = sp[n - 1];
    *(u16_a1*)dp = (u16)a0; dp[n - 1] = (u8)a1;
  } else if (n == 1) dp[0] = sp[0];
}
// an overlapping match of up to 64 bytes at distance `off` < n: the first period (final bytes in front of the destination) is read
// once — into registers (period < 8: the repeated pattern is built with shifts) or into the lane's 64-byte LDS slot — and the
// output is produced from there: no load ever waits for a store of this copy
__device__ __forceinline__ void copy_periodic_le64(u8* dp, const u8* sp, u32 n, u32 off, u8* slot) {
  if (off >= 8) {
    copy_le64(slot, sp, off);                                  // (HBM -> LDS: loads first, then LDS stores)
    u32 p = 0;
    for (u32 k = 0; k < n; k++) { dp[k] = slot[p]; p = p + 1 == off ? 0 : p + 1; }
    return;
  }
  u64 raw = 0;
#pragma unroll
  for (int b = 0; b < 7; b++) if ((u32)b < off) raw |= (u64)sp[b] << (8 * b);
  u64 e0 = 0, e1 = 0;                                        // 16 bytes of the repeated pattern
  for (u32 b = 0; b < 8; b += off) e0 |= raw << (8 * b);
  { const u32 ph = 8 % off; const u64 rot = ph ? ((raw >> (8 * ph)) | (raw << (8 * (off - ph)))) : raw;
    const u64 rp = rot & ((1ull << (8 * off)) - 1);
    for (u32 b = 0; b < 8; b += off) e1 |= rp << (8 * b); }
  u32 k = 0, ph = 0;
  for (; k + 8 <= n; k += 8) {
    const u64 v = ph ? ((e0 >> (8 * ph)) | (e1 << (64 - 8 * ph))) : e0;
    st64(dp + k, v);
    ph += 8 % off; if (ph >= off) ph -= off;
  }
  if (k < n) { const u64 v = ph ? ((e0 >> (8 * ph)) | (e1 << (64 - 8 * ph))) : e0; for (u32 b = 0; k + b < n; b++) dp[k + b] = (u8)(v >> (8 * b)); }
}

// ---------------------------------------------------------------------------------------------
// literals section header + Huffman tree description; thread 0 only. Sets S.lit* / S.huf* / S.err.
// (Forced inline like every helper that is handed `win + x` pointers: those are LDS window addresses minus the window's frame position,
// 32-bit arithmetic that only comes out right while it stays in the LDS address space — a real call takes them as flat pointers.)
// Check order of ZSTD_decodeLiteralsBlock (zstd_decompress_block.c of 1.4.9).
__device__ __forceinline__ void parse_literals_header(ParseShared& S, const u8* src, u32 n, const u8* lim) {
  S.hufPhase = 0;
  LPROF_T0
  if (n < 3) { S.err = ZE_CORRUPTION; return; }                       // MIN_CBLOCK_SIZE
  u32 b0 = src[0], type = b0 & 3, sf = (b0 >> 2) & 3;
  S.litType = type;
  if (type < 2) {
    u32 size, lh;
    if (sf == 0 || sf == 2) { size = b0 >> 3; lh = 1; }
    else if (sf == 1) { size = ld16(src) >> 4; lh = 2; }
    else { size = ld24(src) >> 4; lh = 3; }
    if (size > BLOCK_MAX) { S.err = ZE_CORRUPTION; return; }
    u32 payload = type == 0 ? size : 1;
    if (lh + payload > n) { S.err = ZE_CORRUPTION; return; }
    S.litRegen = size; S.litHdr = lh; S.litComp = payload;
    if (type == 1) S.litRle = src[lh];
    return;
  }
  if (type == 3 && !S.hufValid) { S.err = ZE_DICT_CORRUPTED; return; }   // set_repeat without a table, before any size check
  if (n < 5) { S.err = ZE_CORRUPTION; return; }                          // "we need up to 5 for case 3", whatever the size format
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
      LPROF(20)
      u32 h = read_ncount(S.norm, &maxSym, &tl, p + 1, hbyte, 6, lim);
      if (!h) { S.err = ZE_CORRUPTION; return; }
      LPROF(21)
      // the FSE-coded weights themselves: huf_weights_decode, by the wave
      S.hufTl = tl; S.hufMaxSym = maxSym; S.hufNcBytes = h; S.hufHbyte = hbyte; S.hufUsed = used; S.hufPhase = 2;
      return;
    }
    S.hufNw = nw; S.hufUsed = used; S.hufPhase = 1;
  }
}

// FSE-coded Huffman weights (FSE_decompress_wksp inside HUF_readStats): two interleaved states over a table of <= 64 cells. The table
// is built wave-wide (build_fse_dtable, one cell per lane) and stays in a REGISTER, cell u in lane u: the decode — serial by nature,
// one lane's worth of work run by all lanes in step — fetches a cell with v_readlane instead of an LDS round trip per weight (it was
// 71 k of the parse stage's 190 k cycles per frame, the serial table build 15 k more). Descriptions that name symbols beyond 63
// (no valid tree has them) take the serial build and LDS cells.
__device__ __forceinline__ void huf_weights_decode(ParseShared& S, const u8* src, const u8* lim, const int lane) {
  const u32 tl = S.hufTl, maxSym = S.hufMaxSym, h = S.hufNcBytes, hbyte = S.hufHbyte;
  const u8* const p = src + S.litHdr;
  const u32 size = 1u << tl;
  const bool inRegs = maxSym < 64;
  u32 cellReg = 0;
  if (inRegs) {
    build_fse_dtable(S.stage, S.norm, maxSym, tl, 2, S.spread, (u64*)S.w1, S.wnext, lane);
    wsync();
    cellReg = (u32)lane < size ? S.stage[lane] : 0u;
  } else {
    if (lane == 0) {
      u64* const wt = S.wt; u16* const next = S.wnext;
      u32 mask = size - 1, high = size - 1, step = (size >> 1) + (size >> 3) + 3, pos = 0;
      for (u32 s = 0; s <= maxSym; s++) { next[s] = S.norm[s] == -1 ? 1 : (u16)S.norm[s]; if (S.norm[s] == -1) S.spread[high--] = (u8)s; }
      for (u32 s = 0; s <= maxSym; s++)
        for (int i = 0; i < S.norm[s]; i++) { S.spread[pos] = (u8)s; pos = (pos + step) & mask; while (pos > high) pos = (pos + step) & mask; }
      for (u32 u = 0; u < size; u++) {
        u32 s = S.spread[u], x = next[s]++;
        u32 nbBits = tl - hb32(x);
        wt[u] = mk_seqsym(s, 0, nbBits, (x << nbBits) - size);
      }
    }
    wsync();
  }
  // cell of a state: {symbol, state bits, base of the next state}
  auto cell = [&](u32 st, u32& sym, u32& nb, u32& nextBase) {
    if (inRegs) { const u32 e = bcast_u32(cellReg, st); sym = e & 0xFF; nb = (e >> 16) & 0xF; nextBase = e >> 20; }
    else { const u64 e = S.wt[st]; sym = (u32)e & 0xFF; nb = (u32)(e >> 40) & 0xFF; nextBase = (u32)(e >> 48); }
  };
  bool bad = false;
  u32 nw = 0;
  BitR br;
  if (br.init(p + 1 + h, hbyte - h, lim)) bad = true;
  else {
    u32 s1 = br.read((int)tl), s2 = br.read((int)tl);
    for (;;) {
      u32 sym, nb, nx;
      if (nw >= 254) { bad = true; break; }
      cell(s1, sym, nb, nx);
      if (lane == 0) S.weights[nw] = (u8)sym;
      nw++;
      s1 = nx + br.read((int)nb);
      if (br.pos < 0) { cell(s2, sym, nb, nx); if (lane == 0) S.weights[nw] = (u8)sym; nw++; break; }
      if (nw >= 254) { bad = true; break; }
      cell(s2, sym, nb, nx);
      if (lane == 0) S.weights[nw] = (u8)sym;
      nw++;
      s2 = nx + br.read((int)nb);
      if (br.pos < 0) { cell(s1, sym, nb, nx); if (lane == 0) S.weights[nw] = (u8)sym; nw++; break; }
    }
  }
  if (lane == 0) { if (bad) S.err = ZE_CORRUPTION; else { S.hufNw = nw; S.hufPhase = 1; } }
}

// The rest of HUF_readStats / HUF_readDTableX1 for a tree description lane 0 has read (S.weights[0..hufNw)), by the whole wave: the
// implied last weight, the per-weight counts and the first decode-table cell of every symbol (cells ordered by weight, then by symbol).
// Every rejection here is corruption_detected, so their order does not matter.
__device__ __forceinline__ void huf_tree_finish(ParseShared& S, const int lane) {
  const u32 nw = S.hufNw;
  const u64 lt = (1ull << lane) - 1ull;
  u32 wv[4];
  u32 part = 0; bool over = false;
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const u32 i = 64u * c + (u32)lane;
    wv[c] = i < nw ? (u32)S.weights[i] : 0u;
    over |= wv[c] > 11;
    part += (1u << wv[c]) >> 1;
  }
  const u32 total = bcast_u32(dpp_scan_add(part), 63);
  bool bad = __ballot(over) != 0 || total == 0;
  const u32 maxBits = bad ? 1u : hb32(total) + 1;
  bad |= maxBits > 12;                                                // HUF_TABLELOG_MAX of libzstd (the format says 11)
  const u32 rest = (1u << (maxBits & 31)) - total;
  bad |= rest == 0 || (rest & (rest - 1));
  if (bad) { if (lane == 0) S.err = ZE_CORRUPTION; return; }
  const u32 lastW = hb32(rest) + 1;
#pragma unroll
  for (int c = 0; c < 4; c++) if (64u * c + (u32)lane == nw) wv[c] = lastW;       // (nw <= 255: the implied symbol has a slot)
  // counts per weight -> first cell of every weight
  u32 start[13];
#pragma unroll
  for (int w = 1; w <= 12; w++) {
    u32 n = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) n += (u32)__popcll(__ballot(wv[c] == (u32)w));
    start[w] = n;
  }
  bad = start[1] < 2;                                                 // HUF_readStats: "at least 2 elts of rank 1"
  u32 acc = 0;
#pragma unroll
  for (int w = 1; w <= 12; w++) { const u32 n = start[w]; start[w] = acc; acc += n << (w - 1); }
  bad |= acc != (1u << maxBits);
  bad |= S.hufUsed >= S.litComp;                                      // "hSize >= cSrcSize"
  if (bad) { if (lane == 0) S.err = ZE_CORRUPTION; return; }
#pragma unroll
  for (int c = 0; c < 4; c++) {
    u32 hs = 0;
#pragma unroll
    for (int w = 1; w <= 12; w++) {
      const u64 m = __ballot(wv[c] == (u32)w);
      if (wv[c] == (u32)w) hs = start[w] + ((u32)__popcll(m & lt) << (w - 1));
      start[w] += (u32)__popcll(m) << (w - 1);
    }
    const u32 i = 64u * c + (u32)lane;
    if (i <= nw) S.hufStart[i] = (u16)hs;
  }
  if (lane == 0) {
    S.weights[nw] = (u8)lastW;
    S.hufNSym = nw + 1; S.hufMaxBits = maxBits;
    S.hufValid = 2;   // 2 = a new tree (its description goes to the frame record)
    // which of libzstd's two decoders reads this table (they accept different DAMAGED streams): one stream -> single-symbol;
    // four streams -> HUF_selectDecoder(regenerated size, compressed size incl. the tree); treeless blocks keep the table's kind
    const u32 regen = S.litRegen, comp = S.litComp;
    if (S.litStreams == 4) {
      const u32 Q = comp >= regen ? 15u : (comp * 16u / regen), D256 = regen >> 8;
      const u32 d0 = c_huf_t0[Q][0] + c_huf_t0[Q][1] * D256;
      u32 d1 = c_huf_t1[Q][0] + c_huf_t1[Q][1] * D256;
      d1 += d1 >> 3;
      S.hufX2 = d1 < d0;
    } else S.hufX2 = 0;
  }
}

// where the literal streams of a Huffman-coded section lie (lane 0, after the tree)
__device__ __forceinline__ void literal_streams_layout(ParseShared& S, const u8* src) {
  const u32 type = S.litType, regen = S.litRegen, streams = S.litStreams;
  const u32 used = type == 2 ? S.hufUsed : 0u;
  const u8* const p = src + S.litHdr + used;
  const u32 rem = S.litComp - used;
  // stream layout
  u32 base = (u32)(p - src);
  if (streams == 1) { S.streamOff[0] = base; S.streamLen[0] = rem; }
  else {
    if (regen == 0 && type == 2) { S.err = ZE_CORRUPTION; return; }    // HUF_decompress4X_hufOnly_wksp: dstSize == 0
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

// sequences section header (nbSeq + modes byte; ZSTD_decodeSeqHeaders). thread 0.
__device__ __forceinline__ void parse_seq_header(ParseShared& S, const u8* p, u32 rem) {
  if (rem < 1) { S.err = ZE_SRCSIZE_WRONG; return; }
  u32 nb = p[0], used;
  if (nb == 0) { used = 1; if (rem != 1) { S.err = ZE_SRCSIZE_WRONG; return; } }
  else if (nb < 128) used = 1;
  else if (nb < 255) { if (rem < 2) { S.err = ZE_SRCSIZE_WRONG; return; } nb = ((nb - 128) << 8) + p[1]; used = 2; }
  else { if (rem < 3) { S.err = ZE_SRCSIZE_WRONG; return; } nb = (u32)p[1] + ((u32)p[2] << 8) + 0x7F00; used = 3; }
  S.nbSeq = nb;
  // ZSTD_decodeSeqHeaders leaves early only for a FIRST BYTE of zero: a count of zero in the two-byte form (0x80 0x00; no encoder writes
  // it — round-6 soak on damaged archives, seeds 145238 / 146031) still goes through the table descriptions (their errors count, their
  // tables stay for later blocks' repeat modes), then no sequence is decoded and the bitstream is never opened
  S.seqTables = p[0] != 0;
  if (S.seqTables) {
    if (rem < used + 1) { S.err = ZE_SRCSIZE_WRONG; S.seqTables = 0; return; }
    S.seqModes = p[used]; used++;          // the two reserved bits are not looked at by libzstd 1.4.9
  }
  S.seqPos += used;
}

// one LL/ML/OF table: thread 0 parses (fills S.norm + logs), then the wave builds.  kind 0 LL, 1 ML, 2 OF
__device__ __forceinline__ void seq_table_parse(ParseShared& S, int kind, u32 mode, const u8* p, u32 rem, u32* tlOut, u32* msOut, u32* usedOut, const u8* lim) {
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

// ---------------------------------------------------------------------------------------------
// frame end (shared by the parse and execute kernels): frame-level checks in the order of ZSTD_decompressFrame, the per-frame
// result words, and — random access — the query slices of this frame
// (lane / nthreads: rank and size of the calling group)
__device__ __forceinline__ void frame_finish(const ZraDecodeArgs& a, u32 j, const u8* src, u32 srcSize, u32 err, u32 produced, u32 endPos, bool truncated,
                             u32 fcsHave, u32 fcsLo, u32 fcsHi, u32 hasChecksum, int lane, int nthreads = DEC_THREADS) {
  if (lane == 0) {
    u32 ck = 0;
    if (!err && !truncated) {
      u32 pos = endPos;
      if (fcsHave && (fcsHi != 0 || fcsLo != produced)) err = ZE_CORRUPTION;   // declared size first, then the checksum
      if (!err && hasChecksum) {
        if (srcSize - pos < 4) err = ZE_CHECKSUM_WRONG;
        else { ck = ld32(src + pos); pos += 4; }
      }
      if (!err && pos != srcSize) err = ZE_SRCSIZE_WRONG;          // seek table and frame walk disagree
    }
    a.frameMeta[2 * (size_t)j] = truncated ? 2u : hasChecksum;
    a.frameMeta[2 * (size_t)j + 1] = ck;
    a.status[j] = err;
    a.produced[j] = produced;
    a.frames[j].done = 1;
  }
  if (a.pieces) {
    // every thread of the group recomputes `err` (lane 0 only looked at frame-end conditions that cannot hold for a stopped-early frame)
    if (!err && !truncated) {
      u32 pos = endPos;
      if (fcsHave && (fcsHi != 0 || fcsLo != produced)) err = ZE_CORRUPTION;
      if (!err && hasChecksum) { if (srcSize - pos < 4) err = ZE_CHECKSUM_WRONG; else pos += 4; }
      if (!err && pos != srcSize) err = ZE_SRCSIZE_WRONG;
    }
    if (!err) {
      const u32 p0 = a.pieceBase[j], p1 = a.pieceBase[j + 1];
      for (u32 p = p0; p < p1; p++) {
        const ZraRaPiece q = a.pieces[p];
        // A frame that regenerated fewer bytes than a slice asks for, without an error (only a damaged frame can): what there is, then
        // zeros — deterministic, whatever the caller's buffer held (round 5: such a slice used to be left out). What the REFERENCE
        // returns there is its reused frame buffer's earlier content (zra.cpp:272-295: the first frame's bytes, or what libzstd's wide
        // copies scribbled) — not something a batch of independent queries can reproduce; the host-pointer call does not come here.
        const u32 have = q.srcOff >= produced ? 0u : min(q.len, produced - q.srcOff);
        if (have) copy_bytes(a.raOut + q.dstOff, a.out + a.outOff[j] + q.srcOff, have, lane, nthreads);
        if (have < q.len) fill_bytes(a.raOut + q.dstOff + have, 0, q.len - have, lane, nthreads);
      }
    }
  }
}

}  // namespace

// =================================================================================================
// stage 1: parse — one job (frame j, its next block) by one wave. FUSED (the one-launch path for small random-access batches,
// zra_ra_small_kernel below): nothing is appended to the stage lists, the sequence tables are also left in LDS (ldsT), and the
// outcome comes back as a code: 0 = the frame is finished (frame_finish has run), 1 = a compressed block was handed on,
// 2 = no scratch for it in this round.
namespace {
// ALL (round 6, the block-parallel pass for frames of several blocks, zra_dec_parse_all_kernel): the wave walks ALL blocks of the frame in
// one go instead of one per round. Every compressed block becomes a job of its own for the Huffman and chain stages (record and tables
// in a.blkRecs / a.blkTables at frame * bpf + ordinal), on two assumptions that the execute stage verifies and that hold for every frame
// zstd itself writes: a compressed block that is not the frame's last regenerates ZRA_FMB_BLOCK bytes (so the next block's place in the
// output is known), and nothing is wrong with the frame. A frame that breaks either — or needs more block jobs than bpf, the long-offset
// mode, more scratch than there is — goes on the bail list (a.nextActive) and takes the classic rounds afterwards, where every status of
// the reference is reproduced. A block's initial repeat offsets are markers (ZRA_REP_MARK) from the second compressed block on.
template <bool FUSED, bool ALL = false>
__device__ __forceinline__ u32 parse_job(const ZraDecodeArgs& a, const u32 j, ParseShared& S, const int lane, u32* ldsT) {
  u32 outcome = 0;
  {
    const u64 so = a.frameOff[(size_t)j * a.offStride], se = a.frameOff[(size_t)j * a.offStride + 1];
    const u8* const src = a.body + so;
    const bool spanOk = se >= so && se <= a.bodySize;
    const u32 srcSize = spanOk ? (u32)(se - so) : 0u;
    u8* const dst = a.out + a.outOff[j];
    const u32 dstCap = a.outCap[j];
    const u32 limit = a.limit ? a.limit[j] : 0xFFFFFFFFu;
    ZraDecFrame* const F = &a.frames[j];
    u32 nBlk = 0; bool bail = false;                     // ALL: compressed blocks handed on so far; the frame leaves the pass
    const u32* prevT[3] = {nullptr, nullptr, nullptr};  // ALL: the block tables that hold the current LL / ML / OF table (repeat mode copies from there)

    // copy of [pos, pos + STAGE_BYTES) of the frame (clipped to its end) into an LDS window: one coalesced load for the wave
    auto stage = [&](u8* w, u32 pos) -> u32 {
      const u32 avail = pos < srcSize ? min(srcSize - pos, STAGE_BYTES) : 0u;
      const u32 at = 8u * (u32)lane;
      u64 v = 0;
      if (at + 8 <= avail) v = ld64(src + pos + at);
      else for (u32 k = 0; at + k < avail && k < 8; k++) v |= (u64)src[pos + at + k] << (8 * k);
      *(u64*)(w + at) = v;
      if (lane < 2) *(u64*)(w + STAGE_BYTES + 8 * lane) = 0;
      return avail;
    };
    PPROF_T0
    const u32 startPos = a.round == 0 ? 0u : F->blkPos;
    const u32 staged0 = stage(S.w0, startPos);

    if (lane == 0) {
      S.err = 0; S.frameEnd = 0; S.winPos = startPos; S.winLen = staged0;
      if (a.round == 0) {
        S.produced = 0; S.hufValid = 0; S.llValid = S.mlValid = S.ofValid = 0; S.ofShare = 0;
        S.rep[0] = 1; S.rep[1] = 4; S.rep[2] = 8;
      }
    }
    wsync();
    if (lane == 0) {
      if (a.round == 0) {
        // ---- frame header (A.1), check order of ZSTD_decompressFrame + ZSTD_getFrameHeader_advanced: sizes before the magic number
        const u8* const hsrc = S.w0;          // bytes [0, STAGE_BYTES) of the frame
        u32 hs = 0;
        S.fcsHave = 0; S.hasChecksum = 0; S.bigWindow = 0;
        if (!spanOk || srcSize < 9) S.err = ZE_SRCSIZE_WRONG;
        else {
          u32 fhd = hsrc[4], did = fhd & 3, ss = (fhd >> 5) & 1, fcs = fhd >> 6;
          u32 didSize = did == 3 ? 4 : did;
          u32 fcsSize = fcs == 0 ? ss : fcs == 1 ? 2 : fcs == 2 ? 4 : 8;
          hs = 5 + !ss + didSize + fcsSize;
          if (srcSize < hs + 3) S.err = ZE_SRCSIZE_WRONG;
          else if (ld32(hsrc) != 0xFD2FB528u) S.err = ZE_PREFIX_UNKNOWN;
          else if (fhd & 8) S.err = ZE_FRAMEPARAM_UNSUPPORTED;
          else {
            u64 window = 0;
            if (!ss) {
              u32 b = hsrc[5], wl = 10 + (b >> 3);
              if (wl > 31) S.err = ZE_WINDOW_TOO_LARGE;                  // windowLog > ZSTD_WINDOWLOG_MAX; the one-shot decoder has no other limit
              else window = (1ull << wl) + ((1ull << wl) >> 3) * (b & 7);
            }
            S.hasChecksum = (fhd >> 2) & 1;
            if (!S.err) {
              // a dictionary id cannot be honoured (the reference never loads one): dictionary_wrong, as ZSTD_decompressFrame reports it;
              // the declared content size (1/2/4/8 bytes, the 2-byte form biased by 256) must equal what the frame regenerates
              const u8* q = hsrc + 5 + !ss;
              const u32 dict = did == 0 ? 0u : did == 1 ? (u32)q[0] : did == 2 ? (u32)ld16(q) : ld32(q);
              if (dict) S.err = ZE_DICT_WRONG;
              q += didSize;
              if (fcsSize) {
                const u64 v = fcsSize == 1 ? (u64)q[0] : fcsSize == 2 ? (u64)ld16(q) + 256 : fcsSize == 4 ? (u64)ld32(q) : ld64(q);
                S.fcsLo = (u32)v; S.fcsHi = (u32)(v >> 32); S.fcsHave = 1;
                if (ss) window = v;
              }
              S.bigWindow = window > (1ull << 24);                     // selects libzstd's long-offset sequence loop (chain kernel)
            }
          }
        }
        S.blkPos = hs;
      } else {
        S.produced = F->produced; S.blkPos = F->blkPos;
        S.hufValid = F->hufValid; S.hufMaxBits = F->hufMaxBits; S.hufNSym = F->hufNSym; S.hufX2 = F->hufX2;
        S.llValid = F->llValid; S.mlValid = F->mlValid; S.ofValid = F->ofValid; S.llLog = F->llLog; S.mlLog = F->mlLog; S.ofLog = F->ofLog;
        S.ofShare = F->ofShare;
        S.rep[0] = F->rep[0]; S.rep[1] = F->rep[1]; S.rep[2] = F->rep[2];
        S.fcsLo = F->fcsLo; S.fcsHi = F->fcsHi; S.fcsHave = F->fcsHave; S.hasChecksum = F->hasChecksum; S.bigWindow = F->bigWindow;
      }
    }
    wsync();

    // what a frame carries from round to round (lane 0)
    auto save_persistent_to = [&](ZraDecFrame* F, u32 blkPos, u32 produced, u32 hv, u32 hmb, u32 hns, u32 hx2) {
      F->blkPos = blkPos; F->produced = produced; F->done = 0;
      F->hufValid = hv ? 1u : 0u; F->hufMaxBits = hmb; F->hufNSym = hns; F->hufX2 = hx2;
      F->llValid = S.llValid; F->mlValid = S.mlValid; F->ofValid = S.ofValid; F->llLog = S.llLog; F->mlLog = S.mlLog; F->ofLog = S.ofLog;
      F->ofShare = S.ofShare;
      F->rep[0] = S.rep[0]; F->rep[1] = S.rep[1]; F->rep[2] = S.rep[2];
      F->fcsLo = S.fcsLo; F->fcsHi = S.fcsHi; F->fcsHave = S.fcsHave; F->hasChecksum = S.hasChecksum; F->bigWindow = S.bigWindow;
    };
    auto save_persistent = [&](u32 blkPos, u32 produced, u32 hv, u32 hmb, u32 hns, u32 hx2) { save_persistent_to(F, blkPos, produced, hv, hmb, hns, hx2); };
    if (ALL && S.bigWindow) bail = true;                 // (the long-offset sequence loop: the classic rounds)

    // ------------------------------------------------------------------ block loop: until a compressed block is handed on,
    //                                                                    the frame ends, or an error stops it
    bool handed = false, truncated = false;
    for (;;) {
      const bool stop = S.err || S.frameEnd;    // sampled by every lane before lane 0 may overwrite it
      wsync();
      if (stop || (ALL && bail)) break;
      // (ALL: the positions are assumed ones; one beyond the frame's room means the frame is not what the pass takes — and every
      //  "room left" below is a subtraction from dstCap)
      if (ALL && S.produced > dstCap) { bail = true; break; }
      if (S.produced >= limit) { truncated = true; break; }        // random access: every byte a query needs exists
      // the header window must hold this block's header, literals header and tree description (<= 160 bytes) or reach the frame's end
      if (S.blkPos < S.winPos || S.blkPos + 160 > S.winPos + S.winLen) {
        const u32 at = S.blkPos;
        wsync();
        const u32 got = stage(S.w0, at);
        if (lane == 0) { S.winPos = at; S.winLen = got; }
        wsync();
      }
      const u8* const win = S.w0 - S.winPos;     // win + x == LDS copy of frame byte x, for x inside the window
      if (lane == 0) {
        u32 pos = S.blkPos;
        S.hdrPos = pos;
        if (srcSize - pos < 3) S.err = ZE_SRCSIZE_WRONG;
        else {
          u32 bh = ld24(win + pos);
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

      if (btype == 0 || btype == 1) {
        u8* const out = dst + produced0;
        if (btype == 0) copy_bytes(out, src + bpos, bsize, lane, DEC_THREADS);
        else fill_bytes(out, src[bpos], bsize, lane, DEC_THREADS);
        wsync();
        if (lane == 0) { S.produced = produced0 + bsize; S.blkPos = bpos + (btype == 0 ? bsize : 1); S.frameEnd = S.blkLast; }
        wsync();
        continue;
      }

      // ------------------------------------------------------------ compressed block
      PPROF(8)
      const u32 hv0 = S.hufValid, hmb0 = S.hufMaxBits, hns0 = S.hufNSym, hx20 = S.hufX2;     // the kept tree as of before this block
      if (lane == 0) {
        S.litStreams = 1; S.litRle = 0; S.alloc = 1; S.lateErr = 0; S.nbSeq = 0;
        parse_literals_header(S, win + bpos, bsize, S.w0 + STAGE_BYTES + 8);
      }
      wsync();
      if (!S.err && S.hufPhase == 2) { huf_weights_decode(S, win + bpos, S.w0 + STAGE_BYTES + 8, lane); wsync(); }
      if (!S.err && S.hufPhase == 1) huf_tree_finish(S, lane);
      wsync();
      if (lane == 0) {
        if (!S.err && S.litType >= 2) literal_streams_layout(S, win + bpos);
        if (!S.err) {
          // literal scratch of this round (Huffman-coded literals only: raw ones are read in place, RLE ones are a byte)
          if (S.litType >= 2) {
            const u64 need = ((u64)S.litRegen + 15) & ~15ull;
            const u64 at = atomicAdd((unsigned long long*)&a.counters[ZRA_DC_LITCUR], (unsigned long long)need);
            if (at + need > a.litCap) S.alloc = 0;
            S.litBase = at; S.litKind = 2;
          } else if (S.litType == 1) { S.litKind = 1; S.litArg = S.litRle; }
          else { S.litKind = 0; S.litArg = bpos + S.litHdr; }
          S.seqPos = S.litHdr + S.litComp;
        }
      }
      wsync();
      PPROF(9)
      if (S.err) break;
      // ---- sequences header + table descriptions, from their own LDS window. Whatever is wrong from here on is raised only after
      //      the literals have been decoded (the reference decodes the literals section first): it travels as lateErr
      {
        const u32 at = bpos + S.seqPos;
        const u32 got = stage(S.w1, at);
        (void)got;
        wsync();
      }
      const u8* const sw = S.w1 - S.seqPos;        // sw + x == LDS copy of block byte x, for x from the sequences header on
      const u8* const swLim = S.w1 + STAGE_BYTES + 8;
      if (lane == 0) {
        parse_seq_header(S, sw + S.seqPos, bsize - S.seqPos);
        if (S.err) { S.lateErr = S.err; S.err = 0; S.nbSeq = 0; S.seqTables = 0; }
        if (S.alloc && S.nbSeq) {
          // multiples of four entries per frame: the chain kernel writes its sequences four at a time (aligned 32 bytes)
          const u64 take = ((u64)S.nbSeq + 3) & ~3ull;
          const u64 at = atomicAdd((unsigned long long*)&a.counters[ZRA_DC_SEQCUR], (unsigned long long)take);
          if (at + take > a.seqCap) S.alloc = 0;
          S.seqBase = at;
        }
      }
      wsync();
      if (ALL && (!S.alloc || nBlk >= a.bpf)) { bail = true; break; }      // (no scratch for the whole frame at once / more blocks than jobs)
      if (!S.alloc) {
        // this round's scratch is full: the frame keeps its state as of this block's header and takes the next round
        if (lane == 0) {
          save_persistent(S.hdrPos, produced0, hv0, hmb0, hns0, hx20);
          if (!FUSED) a.nextActive[atomicAdd(&a.counters[ZRA_DC_NNEXT], 1u)] = j;
        }
        handed = true; outcome = 2;          // (nothing pending, but the frame is not finished either)
        wsync();
        break;
      }

      PPROF(10)
      // ---- sequence decode tables: lane 0 parses each description, the wave builds it in LDS and stores it to the table scratch
      const u32 nbSeq = S.nbSeq, regen = S.litRegen;
      const size_t jb = ALL ? (size_t)j * a.bpf + nBlk : (size_t)j;
      u32* const T = ALL ? a.blkTables + jb * ZRA_DEC_TBL_WORDS : a.tables + (size_t)j * ZRA_DEC_TBL_WORDS;
      if (S.seqTables) {                                   // (uniform: written by lane 0 before the wave sync above)
        for (int kind = 0; kind < 3; kind++) {
          const int k = kind == 0 ? 0 : kind == 1 ? 2 : 1;          // wire order is LL, OF, ML
          const u32 mode = k == 0 ? (S.seqModes >> 6) : k == 2 ? ((S.seqModes >> 4) & 3) : ((S.seqModes >> 2) & 3);
          if (lane == 0) {
            u32 tl = 0, ms = 0, used = 0;
            seq_table_parse(S, k, mode, sw + S.seqPos, bsize - S.seqPos, &tl, &ms, &used, swLim);
            S.seqPos += used; S.tl = tl; S.ms = ms;
            if (mode == 3) { u32 v = k == 0 ? S.llValid : k == 1 ? S.mlValid : S.ofValid; if (!v) S.err = ZE_CORRUPTION; }
            if (S.err) { S.lateErr = S.err; S.err = 0; }
          }
          wsync();
          PPROF(11)
          if (S.lateErr) break;
          const u32 tl = S.tl, ms = S.ms;
          u32* const G = T + (k == 0 ? ZRA_DEC_TBL_LL : k == 1 ? ZRA_DEC_TBL_ML : ZRA_DEC_TBL_OF);
          if (mode == 1) {
            if (lane == 0) {
              if (k == 2) { G[0] = mk_cell(ms, ms, 0, 0); S.ofShare = (ms > 22) ? 256u : 0u; }
              else G[0] = mk_cell(ms, k == 0 ? c_ll_bits[ms] : c_ml_bits[ms], 0, 0);
              if (FUSED) { u32* const GL = ldsT + (k == 0 ? ZRA_LDS_TBL_LL : k == 1 ? ZRA_LDS_TBL_ML : ZRA_LDS_TBL_OF); GL[0] = G[0]; if (k != 2) GL[1] = k == 0 ? c_ll_base[ms] : c_ml_base[ms]; }
            }
          } else if (mode != 3) {
            build_fse_dtable(S.stage, S.norm, ms, tl, k, S.spread, (u64*)S.w0, S.wnext, lane);
            wsync();
            const u32 words = (k == 2 ? 1u : 2u) << tl;
            if (!FUSED) for (u32 i = lane; i < (1u << tl); i += DEC_THREADS) G[i] = S.stage[k == 2 ? i : 2 * i];      // (the one-launch kernel never reads the global copy)
            if (FUSED) { u32* const GL = ldsT + (k == 0 ? ZRA_LDS_TBL_LL : k == 1 ? ZRA_LDS_TBL_ML : ZRA_LDS_TBL_OF); for (u32 i = lane; i < words; i += DEC_THREADS) GL[i] = S.stage[i]; }
            if (k == 2) {
              // share of long offset codes (ZSTD_getLongOffsetsShare): cells whose code needs more than 22 extra bits, scaled to 8 bits
              u32 cnt = 0;
              for (u32 i = lane; i < (1u << tl); i += DEC_THREADS) cnt += (S.stage[i] & 0xFF) > 22;
              cnt = wave_sum(cnt);
              if (lane == 0) S.ofShare = cnt << (8 - tl);
            }
            wsync();
          }
          if (lane == 0 && mode != 3) {
            if (k == 0) { S.llLog = tl; S.llValid = 1; } else if (k == 1) { S.mlLog = tl; S.mlValid = 1; } else { S.ofLog = tl; S.ofValid = 1; }
          }
          if (ALL) {
            // a repeated table lives in the tables of the block that built it: into this block's own
            if (mode == 3 && prevT[k] && prevT[k] != G) {
              const u32 lg = k == 0 ? S.llLog : k == 1 ? S.mlLog : S.ofLog;
              for (u32 i = lane; i < (1u << lg); i += DEC_THREADS) G[i] = prevT[k][i];
            }
            prevT[k] = G;
          }
          wsync();
          PPROF(12)
        }
      }

      // ---- hand the block on: persistent state + the block record; Huffman-coded literals go through the Huffman kernel first
      if (ALL) {
        ZraDecFrame* const D = &a.blkRecs[jb];
        if (lane == 0) {
          save_persistent_to(D, S.hdrPos, produced0, S.hufValid, S.hufMaxBits, S.hufNSym, S.hufX2);
          if (nBlk) { D->rep[0] = ZRA_REP_MARK(0); D->rep[1] = ZRA_REP_MARK(1); D->rep[2] = ZRA_REP_MARK(2); }
          D->bpos = bpos; D->bsize = bsize; D->blast = S.blkLast;
          D->litKind = S.litKind; D->litRegen = regen; D->litArg = S.litArg; D->litBase = S.litBase;
          D->litStreams = S.litStreams;
          for (int k = 0; k < 4; k++) { D->streamOff[k] = S.streamOff[k]; D->streamLen[k] = S.streamLen[k]; }
          D->hufErr = 0; D->lateErr = S.lateErr;
          D->nbSeq = S.lateErr ? 0u : nbSeq; D->seqPos = S.seqPos; D->seqBase = S.seqBase;
          D->longMode = 0;
          D->chainErr = 0; D->nSeqValid = 0; D->seqOut = 0; D->seqLit = 0; D->truncated = 0;
          a.pending[atomicAdd(&a.counters[ZRA_DC_NPENDING], 1u)] = (u32)jb;
          if (S.litKind == 2) a.hufJobs[atomicAdd(&a.counters[ZRA_DC_NHUF], 1u)] = (u32)jb;
          // where the next block goes: this one regenerates a whole block unless it is the frame's last (the execute stage checks)
          S.produced = produced0 + (S.blkLast ? 0u : ZRA_FMB_BLOCK); S.blkPos = bpos + bsize; S.frameEnd = S.blkLast;
          // (the table builds used the header window as scratch — build_fse_dtable's masks — and a highly compressible block's successor
          //  starts inside the same 512 bytes: stage again. Found by the round's soak, seed 150028: periodic data, blocks of ~50 bytes)
          S.winLen = 0;
        }
        if (S.litKind == 2) {                             // the tree in use (new, or kept from an earlier block of this walk) travels with the block
          for (u32 i = lane; i < 256; i += DEC_THREADS) { D->weights[i] = S.weights[i]; D->hufStart[i] = S.hufStart[i]; }
        }
        if (S.lateErr) bail = true;                       // (a status of the reference: the classic rounds report it)
        nBlk++;
        wsync();
        continue;
      }
      if (lane == 0) {
        save_persistent(S.hdrPos, produced0, S.hufValid, S.hufMaxBits, S.hufNSym, S.hufX2);
        F->bpos = bpos; F->bsize = bsize; F->blast = S.blkLast;
        F->litKind = S.litKind; F->litRegen = regen; F->litArg = S.litArg; F->litBase = S.litBase;
        F->litStreams = S.litStreams;
        for (int k = 0; k < 4; k++) { F->streamOff[k] = S.streamOff[k]; F->streamLen[k] = S.streamLen[k]; }
        F->hufErr = 0; F->lateErr = S.lateErr;
        F->nbSeq = S.lateErr ? 0u : nbSeq; F->seqPos = S.seqPos; F->seqBase = S.seqBase;
        F->longMode = (S.bigWindow && nbSeq > 4 && S.ofShare >= 7) ? 1u : 0u;
        if (!FUSED) {
          a.pending[atomicAdd(&a.counters[ZRA_DC_NPENDING], 1u)] = j;
          if (S.litKind == 2) a.hufJobs[atomicAdd(&a.counters[ZRA_DC_NHUF], 1u)] = j;
        }
      }
      if (S.litType == 2) {                               // a new tree: its description stays with the frame (treeless blocks reuse it)
        for (u32 i = lane; i < 256; i += DEC_THREADS) { F->weights[i] = S.weights[i]; F->hufStart[i] = S.hufStart[i]; }
      }
      handed = true; outcome = 1;
      wsync();
      PPROF(13)
      break;
    }
    if (ALL) {
      if (bail || (S.err && nBlk)) {
        if (lane == 0) a.nextActive[atomicAdd(&a.counters[ZRA_DC_NNEXT], 1u)] = j;
      } else if (nBlk == 0) {
        frame_finish(a, j, src, srcSize, S.err, S.produced, S.blkPos, truncated, S.fcsHave, S.fcsLo, S.fcsHi, S.hasChecksum, lane);
      } else if (lane == 0) {
        save_persistent(S.blkPos, S.produced, S.hufValid, S.hufMaxBits, S.hufNSym, S.hufX2);
        F->nBlk = nBlk; F->endPos = S.blkPos; F->parseTrunc = truncated ? 1u : 0u;
        a.execList[atomicAdd(&a.counters[ZRA_DC_NEXEC], 1u)] = j;
      }
    } else if (!handed) {
      frame_finish(a, j, src, srcSize, S.err, S.produced, S.blkPos, truncated, S.fcsHave, S.fcsLo, S.fcsHi, S.hasChecksum, lane);
    }
    wsync();
  }
  return outcome;
}
}  // namespace

// waves per SIMD the parse kernel is compiled for. The stage is one lane's serial work per frame (header fields, weight decode, table
// spreads): what it needs is frames in flight. A/B on one box, 8 GiB decode, parse stage: 2 waves per SIMD (170 VGPRs, the compiler's
// own choice) 22.5 ms, 3 (162) 15.2, 4 (128, 9 spilled) 11.9, 5 (96, 40 spilled, 164 B scratch) 10.7, 6 (80, 55 spilled) 10.9
#ifndef ZRA_PARSE_WAVES
#define ZRA_PARSE_WAVES 5
#endif
extern "C" __global__ void __launch_bounds__(DEC_THREADS, ZRA_PARSE_WAVES)
zra_dec_parse_kernel(ZraDecodeArgs a) {
  __shared__ ParseShared S;
  const int lane = threadIdx.x;
  // a round enqueued behind the previous one without a host synchronisation learns its job count on the device
  const u32 nActive = a.nActivePtr ? __hip_atomic_load(a.nActivePtr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : a.nActive;
  for (;;) {
    if (lane == 0) S.job = atomicAdd(&a.counters[ZRA_DC_QPARSE], 1u);
    wsync();
    const u32 qi = S.job;
    wsync();
    if (qi >= nActive) return;
    (void)parse_job<false>(a, a.active ? a.active[qi] : qi, S, lane, nullptr);
  }
}

// the block-parallel pass's parse: every block of every frame in one launch (parse_job<.., ALL>)
extern "C" __global__ void __launch_bounds__(DEC_THREADS, ZRA_PARSE_WAVES)
zra_dec_parse_all_kernel(ZraDecodeArgs a) {
  __shared__ ParseShared S;
  const int lane = threadIdx.x;
  for (;;) {
    if (lane == 0) S.job = atomicAdd(&a.counters[ZRA_DC_QPARSE], 1u);
    wsync();
    const u32 qi = S.job;
    wsync();
    if (qi >= a.nActive) return;
    (void)parse_job<false, true>(a, a.active ? a.active[qi] : qi, S, lane, nullptr);
  }
}

// =================================================================================================
// stage 1b: Huffman literal streams, lane = stream.
// A stream is a serial chain (table lookup -> code length -> next bit position); the four streams of a frame gave a wave four busy
// lanes. Here a wave takes 16 frames: their 16 decode tables are built in LDS (64 KiB) and the 64 streams advance together.
namespace {
constexpr int HUF_FRAMES = ZRA_HUF_FRAMES;
struct __attribute__((aligned(16))) HufShared {
  u16 tab[HUF_FRAMES][2048];   // sym | nbBits<<8 (bit 15: a pair of 12-bit codes, resolved through w1[])
  u8 w1[HUF_FRAMES][256];      // weight-1 symbols in order (only tables of depth 12 need them)
  u32 base;
};
// Huffman symbol lookup on 12 bits v (left-aligned code bits): symbol | nbBits<<8
__device__ __forceinline__ u32 huf_lookup12(const u16* tab, const u8* w1, u32 v12, u32 mb) {
  if (mb < 12) return tab[v12 >> (12 - mb)] & 0x7FFFu;
  const u32 e = tab[v12 >> 1];
  if (e & 0x8000u) return (u32)w1[v12] | (12u << 8);
  return e;
}
// bits [pos-12, pos) of a backward stream, zeros below bit 0 (BIT_lookBitsFast while bits remain)
__device__ __forceinline__ u32 peek12(const u8* base, const u8* lim, i32 pos) {
  const i32 lo = pos - 12;
  if (lo >= 0) { const u64 w = ld64_safe(base + (lo >> 3), lim); return (u32)(w >> (lo & 7)) & 0xFFFu; }
  if (pos <= 0) return 0;
  const u64 w = ld64_safe(base, lim);
  return (u32)((w << (u32)(-lo)) & 0xFFFu);
}
// libzstd's double-symbol decoder (HUF_decodeStreamX2) on ONE stream that the single-symbol rules rejected: a 12-bit lookup yields
// one symbol or a PAIR (when both codes fit in 12 bits); if the walk ends one output position short, HUF_decodeLastSymbolX2 takes
// the first symbol of the entry under the cursor and, for a pair entry, skips the bits of both codes clamped to the end of the
// stream (nothing when no bit is left). Returns true when that decoder accepts the stream; `o` gets the symbols it writes.
__device__ __forceinline__ bool huf_stream_x2(const u16* tab, const u8* w1, u32 mb, const u8* base, u32 n, const u8* lim, u8* o, u32 regen) {
  if (n == 0) return false;
  const u32 last = base[n - 1];
  if (last == 0) return false;
  i32 pos = (i32)(n - 1) * 8 + (i32)hb32(last);
  u32 i = 0;
  while (i + 2 <= regen) {
    if (pos <= 0) return false;
    const u32 e1 = huf_lookup12(tab, w1, peek12(base, lim, pos), mb), a = e1 >> 8;
    const u32 e2 = huf_lookup12(tab, w1, peek12(base, lim, pos - (i32)a), mb), b = e2 >> 8;
    o[i] = (u8)e1;
    if (a + b <= 12) { o[i + 1] = (u8)e2; pos -= (i32)(a + b); i += 2; }
    else { pos -= (i32)a; i += 1; }
  }
  if (i < regen) {
    if (pos < 0) return false;
    u32 v1;
    if (pos > 0) v1 = peek12(base, lim, pos);
    else {                                         // no bit left: the container's TOP 12 bits come back (shift by 64 & 63 = 0)
      u64 c = 0; for (u32 k = 0; k < 8 && k < n; k++) c |= (u64)base[k] << (8 * k);
      v1 = (u32)(c >> 52);
    }
    const u32 e1 = huf_lookup12(tab, w1, v1, mb), a = e1 >> 8;
    const u32 v2 = pos > 0 ? peek12(base, lim, pos - (i32)a) : ((v1 << a) & 0xFFFu);
    const u32 e2 = huf_lookup12(tab, w1, v2, mb), b = e2 >> 8;
    o[i] = (u8)e1;
    if (a + b <= 12) { if (pos > 0) { pos -= (i32)(a + b); if (pos < 0) pos = 0; } }
    else pos -= (i32)a;
  }
  return pos == 0;
}
}  // namespace

namespace {
// decode table of one frame's kept Huffman description, by the whole wave: lane l owns symbols 4l .. 4l+3
__device__ __forceinline__ void huf_build_table(const ZraDecFrame* const F, u16* const tab, u8* const w1, const int lane) {
  const u32 maxBits = F->hufMaxBits, nSym = F->hufNSym;
  const u32 sh = maxBits == 12 ? 1u : 0u;          // depth 12: cells are indexed by the top 11 bits, the 12-bit codes come in pairs
  const u32 w4 = *(const u32*)(F->weights + 4 * lane);
  const u64 st4 = *(const u64*)(F->hufStart + 4 * lane);
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const u32 sy = 4 * (u32)lane + k;
    const u32 w = sy < nSym ? (w4 >> (8 * k)) & 0xFF : 0u;
    u32 start = (u32)(st4 >> (16 * k)) & 0xFFFFu, len = w ? 1u << (w - 1) : 0u;
    const u16 e = (u16)(sy | ((maxBits + 1 - w) << 8));
    if (sh && w == 1) { w1[start] = (u8)sy; if (!(start & 1)) tab[start >> 1] = (u16)(0x8000u | (12u << 8)); len = 0; }
    len >>= sh; start >>= sh;
    // short runs by their own lane, long ones by the whole wave
    const bool big = len >= 32;
    if (!big) for (u32 c = 0; c < len; c++) tab[start + c] = e;
    u64 bm = __ballot(big);
    while (bm) {
      const u32 l2 = (u32)__builtin_ctzll(bm); bm &= bm - 1;
      const u32 bs = bcast_u32(start, l2), bl = bcast_u32(len, l2), be = bcast_u32((u32)e, l2);
      for (u32 c = lane; c < bl; c += DEC_THREADS) tab[bs + c] = (u16)be;
    }
  }
}
// `count` symbols of a stream through the single-symbol table (maxBits < 12), the reader standing on the first of them:
// 16 symbols per store (write requests are the expensive ones), 4 symbols per window reload (4*11 = 44 <= 56 guaranteed bits)
template <typename Reader>
__device__ __forceinline__ void huf_x1_emit(Reader& hb, const u16* const tab, const int mb, u8* const o, const u32 count) {
  u32 i = 0;
  for (; i + 16 <= count; i += 16) {
    u32 pk[4];
#pragma unroll
    for (int g = 0; g < 4; g++) {
      hb.ensure(4 * mb);
      u32 packed = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        u32 e = tab[hb.peek(mb)];
        packed |= (e & 0xFF) << (8 * k);
        hb.skip((int)(e >> 8));
      }
      pk[g] = packed;
    }
    st128(o + i, pk[0], pk[1], pk[2], pk[3]);
  }
  for (; i + 4 <= count; i += 4) {
    hb.ensure(4 * mb);
    u32 packed = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      u32 e = tab[hb.peek(mb)];
      packed |= (e & 0xFF) << (8 * k);
      hb.skip((int)(e >> 8));
    }
    st32(o + i, packed);
  }
  for (; i < count; i++) { hb.ensure(mb); u32 e = tab[hb.peek(mb)]; o[i] = (u8)e; hb.skip((int)(e >> 8)); }
}

// one literal stream of frame job j by ONE lane (strm < F->litStreams); returns false when libzstd would reject the stream.
// stopAt: symbols of this stream that are needed (random access that stops early: the rest is not decoded; only without X2 fall-back)
__device__ __forceinline__ bool huf_decode_stream(const ZraDecodeArgs& a, ZraDecFrame* const F, const u32 j, const u16* const tab, const u8* const w1,
                                                  const u32 strm, const u8* const lim) {
  const u32 nStreams = F->litStreams, regen = F->litRegen;
  const u32 seg = nStreams == 1 ? regen : (regen + 3) / 4;
  const u32 myLen = nStreams == 1 ? regen : (strm < 3 ? seg : regen - 3 * seg);
  u8* o = a.lits + F->litBase + (size_t)strm * seg;
  const u8* const blk = a.body + a.frameOff[frame_of(a, j) * a.offStride] + F->bpos;
  const u8* const sb = blk + F->streamOff[strm]; const u32 sl = F->streamLen[strm];
  const int mb = (int)F->hufMaxBits;
#ifndef ZRA_HUF_READER_AHEAD
#define ZRA_HUF_READER_AHEAD 1
#endif
#if ZRA_HUF_READER_AHEAD
  BitRS hb;
#else
  BitR hb;
#endif
  bool bad = hb.init(sb, sl, lim) != 0;
  if (!bad) {
    u32 i = 0;
    if (mb < 12) {
      huf_x1_emit(hb, tab, mb, o, myLen);
      i = myLen;
    } else {
      for (; i < myLen; i++) { hb.ensure(12); const u32 e = huf_lookup12(tab, w1, hb.peek(12), 12); o[i] = (u8)e; hb.skip((int)(e >> 8)); }
    }
    if (hb.pos != 0) bad = true;
  }
  // a stream the single-symbol rules reject may still pass libzstd's double-symbol decoder, if that is the one it would use
  if (bad && F->hufX2) bad = !huf_stream_x2(tab, w1, (u32)mb, sb, sl, lim, o, myLen);
  return !bad;
}

// The literal streams of one frame by the WHOLE wave. A Huffman stream can only be decoded from its end, one code after the other — but
// a decoder that starts in the middle of a stream falls into step with the real code boundaries after a few codes (self-synchronisation).
// So every stream is cut into 16 runs of bits (a lone stream: 64), every lane decodes its run from a guessed start, counts the codes
// and notes where the first code of the NEXT run starts; the lanes then restart from their predecessor's hand-over point until nobody's
// start moves — lane 0 of a stream starts at the true end mark, so the fixed point is the real decode — and a last pass writes the
// symbols at the prefix sums of the counts. Returns false unless every stream regenerates exactly its share and ends on bit 0: the
// caller then runs the serial decoders, which decide what a damaged stream returns (what was written here is overwritten or unused).
__device__ __forceinline__ bool huf_decode_wave(const ZraDecodeArgs& a, const ZraDecFrame* const F, const u32 j, const u16* const tab, const int lane, const u8* const lim) {
  const u32 nStreams = F->litStreams, regen = F->litRegen;
  const int mb = (int)F->hufMaxBits;
  if (mb >= 12 || regen < 64) return false;
  const u32 L = nStreams == 1 ? 64u : 16u;
  const u32 strm = nStreams == 1 ? 0u : (u32)lane >> 4, t = (u32)lane & (L - 1);
  const u32 seg = nStreams == 1 ? regen : (regen + 3) / 4;
  const u32 myLen = nStreams == 1 ? regen : (strm < 3 ? seg : regen - 3 * seg);
  const u8* const blk = a.body + a.frameOff[frame_of(a, j) * a.offStride] + F->bpos;
  const u8* const sb = blk + F->streamOff[strm]; const u32 sl = F->streamLen[strm];
  const u32 last = sl ? (u32)sb[sl - 1] : 0u;
  if (__ballot(last == 0)) return false;
  const u32 P = (sl - 1) * 8 + hb32(last);                            // unread bits under the end mark
  const u32 C = max((P + L - 1) / L, 128u);                           // bits per run
  const bool active = t * C < P;
  const i32 lower = (t + 1) * C >= P ? 0 : (i32)(P - (t + 1) * C);    // codes that START above this bit are the lane's
  i32 start = (i32)(P - t * C), end = 0;
  u32 n = 0;
  bool dirty = active;
  BitRS hb;
  i32 prevStart = start;
  for (int it = 0; it < 6 && __ballot(dirty); it++) {
    if (dirty) {
      // a restarted lane first walks its new path and its previous one side by side, a code at a time on whichever is behind: the
      // two fall into step after a few codes, and from there on count and hand-over point are the previous pass's — the run is not
      // decoded again. (No meeting point within 96 codes, or the first pass: the whole run.)
      bool merged = false;
      if (it) {
        BitRS ho;
        hb.init_at(sb, lim, start); ho.init_at(sb, lim, prevStart);
        u32 ka = 0, kb = 0;
        for (int k = 0; k < 96; k++) {
          if (hb.pos == ho.pos) { merged = true; break; }
          if (hb.pos <= lower || ho.pos <= lower) break;
          if (hb.pos > ho.pos) { hb.ensure(mb); const u32 e = tab[hb.peek(mb)]; hb.skip((int)(e >> 8)); ka++; }
          else { ho.ensure(mb); const u32 e = tab[ho.peek(mb)]; ho.skip((int)(e >> 8)); kb++; }
        }
        if (merged) n = ka + (n - kb);
      }
      if (!merged) {
        hb.init_at(sb, lim, start); n = 0;
        while (hb.pos > lower) {
          hb.ensure(4 * mb);
#pragma unroll
          for (int k = 0; k < 4; k++) if (hb.pos > lower) { const u32 e = tab[hb.peek(mb)]; hb.skip((int)(e >> 8)); n++; }
        }
        end = hb.pos;
      }
    }
    const i32 handed = (i32)__shfl_up(end, 1, 64);                   // (every lane takes part: a lane that sits out returns 0 to its reader)
    const i32 ns = t == 0 ? (i32)P : handed;
    prevStart = start;
    dirty = active && ns != start;
    start = ns;
#ifdef ZRA_SMALL_PROFILE
    { const u64 dm = __ballot(dirty); if (lane == 0) atomicAdd(&zra_small_prof[16 + it], (u64)__popcll(dm)); }
#endif
  }
#ifdef ZRA_SMALL_PROFILE
  if (lane == 0) atomicAdd(&zra_small_prof[6], 1ull);
  if (__ballot(dirty) && lane == 0) atomicAdd(&zra_small_prof[7], 1ull);
#endif
  if (__ballot(dirty)) return false;
  u32 incl = active ? n : 0u;
  if (L == 64) incl = dpp_scan_add(incl);
  else {
    incl += (u32)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xF, 0xF, true);
    incl += (u32)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xF, 0xF, true);
    incl += (u32)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xF, 0xF, true);
    incl += (u32)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xF, 0xF, true);
  }
  const u32 total = (u32)__shfl((int)incl, (int)(((u32)lane & ~(L - 1)) + L - 1), 64);
  if (__ballot(total != myLen || (active && lower == 0 && end != 0))) return false;
  if (active && n) {
    hb.init_at(sb, lim, start);
    huf_x1_emit(hb, tab, mb, a.lits + F->litBase + (size_t)strm * seg + (incl - n), n);
  }
  return true;
}
}  // namespace

extern "C" __global__ void __launch_bounds__(DEC_THREADS)
zra_dec_huf_kernel(ZraDecodeArgs a) {
  __shared__ HufShared S;
  const int lane = threadIdx.x;
  const u32 nJobs = a.counters[ZRA_DC_NHUF];
  const u8* const lim = a.body + a.bodySize;
  for (;;) {
    wsync();
    if (lane == 0) S.base = atomicAdd(&a.counters[ZRA_DC_QHUF], (u32)HUF_FRAMES);
    wsync();
    const u32 base = S.base;
    if (base >= nJobs) return;
    const u32 nHere = min((u32)HUF_FRAMES, nJobs - base);
    // ---- the tables of this batch, one frame after the other, all lanes on each
    for (u32 s = 0; s < nHere; s++) huf_build_table(&a.frames[a.hufJobs[base + s]], S.tab[s], S.w1[s], lane);
    wsync();
    // ---- all lanes on one frame after the other; what that leaves (damaged or unusual streams): lane -> (frame slot, stream)
    u32 doneMask = 0;
    for (u32 s = 0; s < nHere; s++) {
      const u32 j = a.hufJobs[base + s];
      if (huf_decode_wave(a, &a.frames[j], j, S.tab[s], lane, lim)) doneMask |= 1u << s;
    }
    const u32 slot = (u32)lane >> 2, strm = (u32)lane & 3;
    if (slot < nHere && !((doneMask >> slot) & 1u)) {
      const u32 j = a.hufJobs[base + slot];
      ZraDecFrame* const F = &a.frames[j];
      if (strm < F->litStreams && !huf_decode_stream(a, F, j, S.tab[slot], S.w1[slot], strm, lim)) F->hufErr = 1;
    }
  }
}

// =================================================================================================
// stage 2: the FSE sequence chains, lane = frame
namespace {

// libzstd's BIT_DStream_t (bitstream.h of 1.4.9), restated field for field: a 64-bit container refilled from the END of the stream
// towards its start. Exact, because an over-read stream is not an error inside the sequence loop of 1.4.9: the wrapped container
// bits it then reads decide what the frame finally returns (oracle/zo_decode.c: zds).
// RING (round 5, the LDS-table chain kernel): the stream's bytes reach the container through a per-lane ring of CHAIN_RING bytes in LDS
// that is filled AHEAD of the read position — the stream is read strictly from its end towards its start, so the next 16 bytes below
// the ring's lowest offset are always the right ones to fetch — instead of through one dependent global load per sequence. ring[s & (R-1)]
// = byte s of the stream for s in [rLo, rLo + R) (the first 8 bytes are mirrored behind the ring so that a container that wraps reads
// straight on). Every container read is ONE ds_read without a branch around it; once per step ring_guard looks whether the prefetch fell
// behind (more than 4 bytes consumed per step for a while) and then refills the ring at the read position, so nothing depends on timing.
constexpr u32 CHAIN_RING = 128, CHAIN_RING_WORDS = (CHAIN_RING + 16) / 4;
static_assert(CHAIN_RING_WORDS == ZRA_CHAIN_RING_WORDS, "ring size: zra_kernels.h");
__device__ __attribute__((noinline)) uint4 chain_ld128_tail(const u8* p, const u8* lim) {   // 16 bytes at p, zeros beyond the readable body (the archive's last bytes)
  u32 w[4] = {0, 0, 0, 0};
  for (u32 k = 0; k < 16; k++) if (p + k < lim) w[k >> 2] |= (u32)p[k] << (8 * (k & 3));
  return make_uint4(w[0], w[1], w[2], w[3]);
}
template <bool RING>
struct ZdsT {
  u64 c; u32 bc; u32 ptr;      // ptr: byte offset of the container inside the stream (libzstd: ptr - start)
  const u8* base;
  // RING only: the lane's ring, the lowest stream offset it holds, the readable end of the body, the piece on its way from memory
  u8* ring; u32 rLo; const u8* lim; uint4 pf; u32 pfState, pfAge;
  enum : int { UNFINISHED = 0, ENDOFBUFFER = 1, COMPLETED = 2, OVERFLOW = 3 };
  __device__ __forceinline__ uint4 ld16(const u8* p) const {
    if (p + 16 <= lim) { const u128_u q = *(const u128_u*)p; return make_uint4(q.a, q.b, q.c, q.d); }
    return chain_ld128_tail(p, lim);
  }
  __device__ __forceinline__ void ring_write(u32 off, uint4 q) {   // stream bytes [off, off + 16), off a multiple of 16
    const u32 o = off & (CHAIN_RING - 1);
    lds_st128(ring + o, q);
    if (o == 0) lds_st64(ring + CHAIN_RING, (u64)q.x | ((u64)q.y << 32));
  }
  // the ring around position p (the start of a stream, or the prefetch fell behind): five pieces from 48 bytes below p's piece upwards
  __device__ __forceinline__ void ring_resync(u32 p) {
    const u32 pc = p & ~15u, lo = pc >= 48 ? pc - 48 : 0u;
    uint4 q[5];
#pragma unroll
    for (u32 k = 0; k < 5; k++) q[k] = ld16(base + lo + 16 * k);
#pragma unroll
    for (u32 k = 0; k < 5; k++) ring_write(lo + 16 * k, q[k]);
    rLo = lo; pfState = 0;
  }
  // once per decoded sequence: a piece is asked for when fewer than 64 bytes lie below the read position, and written two steps later
  // (it has arrived by then) once the bytes it overwrites — the ring's top 16 — are behind the container
  __device__ __forceinline__ void ring_advance() {
    if (pfState) {
      if (++pfAge >= 2 && ptr + 8 + 16 <= rLo + CHAIN_RING) { ring_write(rLo - 16, pf); rLo -= 16; pfState = 0; }
    } else if (rLo >= 16 && ptr < rLo + 64) { pf = ld16(base + rLo - 16); pfState = 1; pfAge = 0; }
  }
  // (RING: no test here — ring_guard, once per step of the caller, keeps 16 bytes of margin below the container, more than a step consumes)
  __device__ __forceinline__ u64 fetch(u32 p) {
    if (RING) return *(const zra_lds_u64u_t*)(ring + (p & (CHAIN_RING - 1)));
    return ld64(base + p);
  }
  __device__ __forceinline__ void ring_guard() { if (rLo && ptr < rLo + 16) ring_resync(ptr); }
  __device__ __forceinline__ bool init(const u8* b, u32 n) {
    base = b;
    if (n < 1) return false;
    if (RING) ring_resync(n >= 8 ? n - 8 : 0u);            // the top of the stream into the ring
    if (n >= 8) { ptr = n - 8; c = fetch(ptr); }
    else { ptr = 0; c = 0; for (u32 i = 0; i < n; i++) c |= (u64)b[i] << (8 * i); }
    const u32 last = b[n - 1];
    if (last == 0) return false;
    bc = 8 - hb32(last);
    if (n < 8) bc += (8 - n) * 8;
    return true;
  }
  __device__ __forceinline__ u32 read(u32 nb) {           // BIT_readBits: BIT_getMiddleBits(container, 64 - bc - nb, nb)
    const u32 start = 64u - bc - nb;
    const u32 v = (u32)(c >> (start & 63)) & ((1u << nb) - 1u);      // nb <= 9 here
    bc += nb; return v;
  }
  __device__ __forceinline__ u32 read_fast(u32 nb) {      // BIT_readBitsFast: nb >= 1
    const u32 v = (u32)((c << (bc & 63)) >> ((64 - nb) & 63));
    bc += nb; return v;
  }
  __device__ __forceinline__ int reload() {
    if (bc > 64) return OVERFLOW;
    if (ptr >= 8) { ptr -= bc >> 3; bc &= 7; c = fetch(ptr); return UNFINISHED; }
    if (ptr == 0) return bc < 64 ? ENDOFBUFFER : COMPLETED;
    u32 nbBytes = bc >> 3; int r = UNFINISHED;
    if (ptr < nbBytes) { nbBytes = ptr; r = ENDOFBUFFER; }
    ptr -= nbBytes; bc -= nbBytes * 8; c = fetch(ptr);
    return r;
  }
  // the same transition without its status and without control flow around the load (so that the load can be in flight together with
  // the table cells of the next sequence): all three cases above move min(bc / 8, ptr) bytes, and for streams of 8 bytes and more
  // the container always equals the 8 bytes at ptr, so an unconditional load is exact. `pad`: 8 readable bytes for shorter streams.
  __device__ __forceinline__ void reload_quiet(bool wide, const u8* pad) {
    const u32 nb = bc <= 64 ? min(bc >> 3, ptr) : 0u;
    ptr -= nb; bc -= nb * 8;
    const u64 v = RING ? fetch(ptr) : ld64(wide ? base + ptr : pad);      // (RING, a short stream: reads what lies in the ring; nb is 0)
    c = nb ? v : c;
  }
};
typedef ZdsT<false> Zds;

}  // namespace

namespace {
// LDSTAB: the tables of the wave's frames live in LDS (ZRA_CHAIN_LDS_FRAMES slots of 5 KiB: lanes beyond them take no frames); a
// frame's table is copied there by the whole wave when a lane takes the frame. The step itself is the same code.
template <bool LDSTAB>
__device__ __forceinline__ void chain_body(const ZraDecodeArgs& a, u32* const ldsTabs, u32* const baseLL, u32* const baseML, u32* const ldsRings = nullptr) {
  const int lane = threadIdx.x;
  const u32 nPend = a.counters[ZRA_DC_NPENDING];
  // per-lane job state
  bool have = false, drained = false;
  ZraDecFrame* F = nullptr;
  const u32* T = nullptr;
  u64* sq = nullptr;
  ZdsT<LDSTAB> br; br.c = 0; br.bc = 0; br.ptr = 0; br.base = nullptr; br.ring = nullptr; br.rLo = 0;      // (rLo == 0: nothing to guard or prefetch — no stream yet, or the ring holds the stream's start)
  br.lim = a.body + a.bodySize; br.pf = make_uint4(0, 0, 0, 0); br.pfState = 0; br.pfAge = 0;
  if (LDSTAB) br.ring = (u8*)(ldsRings + (u32)lane * CHAIN_RING_WORDS);
  u32 sLL = 0, sOF = 0, sML = 0, rep0 = 1, rep1 = 4, rep2 = 8;
  u32 i = 0, nbSeq = 0, outPos = 0, litPos = 0, outCap = 0, regen = 0, produced0 = 0, limit = 0;
  u32 longMode = 0, err = 0, jErr = 0xFFFFFFFFu, valid = 0, validOut = 0, validLit = 0, truncated = 0;
  // decoded sequences leave four at a time: a random write costs the memory system about two random reads whatever its width, so 32 bytes
  // per request instead of 8
  u64 q0 = 0, q1 = 0, q2 = 0;
  // the table cells of the sequence about to be decoded: requested one step ahead, together with the container reload
  u32 eL = 0, eM = 0, eO = 0;
  bool wide = false; const u8* pad = nullptr;
  bool fresh = false; u32 freshJob = 0;
  // LDSTAB (round 5): TWO-byte cells in LDS, so that 59 frames' tables fit a CU instead of 31 — code | v << 6, where v = the state's
  // base >> width under a leading one at bit (tableLog - width): FSE's base is a multiple of 2^width and smaller than 2^tableLog, so the
  // leading one's place gives the width and the bits under it the base. The code's extra-bit count rides in the base-value tables.
  const u16* const cells16 = (const u16*)ldsTabs + (size_t)lane * ZRA_DEC_TBL_WORDS;
  u32 logLL = 0, logML = 0, logOF = 0;                  // LDSTAB: the frame's table logs (the cells' widths are relative to them)
  auto fetch_cells = [&]() {
    if (LDSTAB) { eL = cells16[ZRA_DEC_TBL_LL + sLL]; eM = cells16[ZRA_DEC_TBL_ML + sML]; eO = cells16[ZRA_DEC_TBL_OF + sOF]; }
    else { eL = T[ZRA_DEC_TBL_LL + sLL]; eM = T[ZRA_DEC_TBL_ML + sML]; eO = T[ZRA_DEC_TBL_OF + sOF]; }
  };
  // a cell's fields, whichever layout: code, extra bits of the code, state bits, state base
  auto c_code = [&](u32 e) -> u32 { return e & 63u; };
  auto c_sbits = [&](u32 e, u32 lg) -> u32 { if (LDSTAB) return lg - (31u - (u32)__builtin_clz(e >> 6)); return (e >> 16) & 0xF; };
  auto c_sbase = [&](u32 e, u32 lg) -> u32 {
    if (LDSTAB) { const u32 v = e >> 6, hb = 31u - (u32)__builtin_clz(v); return (v ^ (1u << hb)) << (lg - hb); }
    return e >> 20;
  };
  // base values of the length codes (the cells carry the code only); LDSTAB: the code's extra-bit count in the top byte
  baseLL[lane] = lane < 36 ? c_ll_base[lane] | (LDSTAB ? (u32)c_ll_bits[lane] << 24 : 0u) : 0u;
  baseML[lane] = lane < 53 ? c_ml_base[lane] | (LDSTAB ? (u32)c_ml_bits[lane] << 24 : 0u) : 0u;
  wsync();
  if (LDSTAB && (u32)lane >= ZRA_CHAIN_LDS_FRAMES) drained = true;

  auto finish = [&]() {
    // the last one to three sequences still sit in registers
    { const u32 r = valid & 3u, b = valid - r;
      if (r >= 1) sq[b] = q0;
      if (r >= 2) sq[b + 1] = q1;
      if (r >= 3) sq[b + 2] = q2; }
    // tail literals of the block (ZSTD_decompressSequences: "last literal segment")
    if (!err && !truncated) {
      if (regen - litPos > outCap - outPos) err = ZE_DSTSIZE_TOOSMALL;
    }
    F->chainErr = err; F->nSeqValid = valid; F->seqOut = validOut; F->seqLit = validLit; F->truncated = truncated;
    F->repOut[0] = rep0; F->repOut[1] = rep1; F->repOut[2] = rep2;
    have = false;
  };

  for (;;) {
    // ---- lanes without a job pull the next pending frame (one atomic per wave)
    const u64 want = __ballot(!have && !drained);
    if (want) {
      u32 base = 0;
      const u32 cnt = (u32)__builtin_popcountll(want);
      if (lane == (int)__builtin_ctzll(want)) base = atomicAdd(&a.counters[ZRA_DC_QCHAIN], cnt);
      base = bcast_u32(base, (u32)__builtin_ctzll(want));
      if (!have && !drained) {
        const u32 idx = base + (u32)__builtin_popcountll(want & ((1ull << lane) - 1ull));
        if (idx >= nPend) drained = true;
        else {
          const u32 j = a.pending[idx];
          const size_t gj = frame_of(a, j);               // (block-parallel pass: the job is a block, gj its frame)
          F = &a.frames[j];
          T = a.tables + (size_t)j * ZRA_DEC_TBL_WORDS;
          const u8* const blk = a.body + a.frameOff[gj * a.offStride] + F->bpos;
          nbSeq = F->nbSeq; regen = F->litRegen; produced0 = F->produced; longMode = F->longMode;
          outCap = a.outCap[gj] - produced0;
          limit = a.limit ? a.limit[gj] : 0xFFFFFFFFu;
          sq = a.seqs + F->seqBase;
          rep0 = F->rep[0]; rep1 = F->rep[1]; rep2 = F->rep[2];
          i = 0; outPos = 0; litPos = 0; err = 0; jErr = 0xFFFFFFFFu; valid = 0; validOut = 0; validLit = 0; truncated = 0;
          have = true; br.rLo = 0; br.pfState = 0;      // (a block without sequences never opens its stream)
          // what went wrong before the sequences, in the reference's order: the literal streams, then the sequences header / tables
          if (F->hufErr) { err = ZE_CORRUPTION; nbSeq = 0; finish(); }
          else if (F->lateErr) { err = F->lateErr; nbSeq = 0; finish(); }
          else if (nbSeq) {
            if (!br.init(blk + F->seqPos, F->bsize - F->seqPos)) { err = ZE_CORRUPTION; finish(); }
            else {
              logLL = F->llLog; logOF = F->ofLog; logML = F->mlLog;
              sLL = br.read(logLL); br.reload();
              sOF = br.read(logOF); br.reload();
              sML = br.read(logML); br.reload();
              wide = F->bsize - F->seqPos >= 8;
              pad = a.body + a.frameOff[gj * a.offStride];          // the frame's first bytes: always 8 readable ones
              if (LDSTAB) { fresh = true; freshJob = j; } else fetch_cells();
            }
          }
        }
      }
    }
    if (LDSTAB) {
      // the tables of the frames just taken: global scratch -> the lanes' LDS slots, 16 bytes per lane and step, the whole wave on each
      u64 nm = __ballot(fresh);
      while (nm) {
        const u32 l = (u32)__builtin_ctzll(nm); nm &= nm - 1;
        const u32 jj = bcast_u32(freshJob, l);
        const uint4* const g4 = (const uint4*)(a.tables + (size_t)jj * ZRA_DEC_TBL_WORDS);
        uint2* const d2 = (uint2*)((u16*)ldsTabs + (size_t)l * ZRA_DEC_TBL_WORDS);
        const ZraDecFrame* const Fj = &a.frames[jj];
        const u32 lgL = Fj->llLog, lgM = Fj->mlLog, lgO = Fj->ofLog;
        for (u32 i = (u32)lane; i < ZRA_DEC_TBL_WORDS / 4; i += DEC_THREADS) {
          const uint4 q = g4[i];
          const u32 lg = 4 * i < ZRA_DEC_TBL_ML ? lgL : 4 * i < ZRA_DEC_TBL_OF ? lgM : lgO;
          // (sym | extraBits << 8 | stateBits << 16 | base << 20) -> (sym | ((1 << (log - stateBits)) | base >> stateBits) << 6); cells beyond the
          // table's 2^log (never read) may hold anything: their shift is clamped
          auto pk = [&](u32 x) -> u32 { const u32 nb = min((x >> 16) & 0xFu, lg); return (x & 63u) | (((1u << (lg - nb)) | ((x >> 20) >> nb)) << 6); };
          d2[i] = make_uint2(pk(q.x) | (pk(q.y) << 16), pk(q.z) | (pk(q.w) << 16));
        }
      }
      if (__ballot(fresh)) wsync();
      if (fresh) { fetch_cells(); fresh = false; }
    }
    if (!__ballot(have)) break;

    // ---- one step of the lane's frame (no wave-level operation below: lanes are at different points of different frames)
    if (have) {
      bool go = true;
      if (LDSTAB) { br.ring_guard(); br.ring_advance(); }
      if (i >= nbSeq) {
        // all sequences decoded. Short loop: the stream must be consumed (over-read passes); long loop: no such check.
        if (!err && nbSeq && !longMode && br.reload() < Zds::COMPLETED) err = ZE_CORRUPTION;
        finish(); go = false;
      } else if (longMode) {
        // ZSTD_decompressSequencesLong: the loop condition looks at the stream BEFORE each decode and stops on an over-read; a
        // sequence is executed four iterations after it was decoded, so an earlier execution error only counts if the loop got there
        if (br.reload() > Zds::COMPLETED) {
          if (jErr == 0xFFFFFFFFu || i <= jErr + 4) err = ZE_CORRUPTION;
          finish(); go = false;
        } else if (jErr != 0xFFFFFFFFu && i > jErr + 4) { finish(); go = false; }
      }
      if (go) {
        // ZSTD_decodeSequence (64-bit path): offset bits, match-length bits, [reload], literal-length bits, then the three state
        // updates — always, the last sequence included
        u32 ll = baseLL[c_code(eL)], ml = baseML[c_code(eM)], off;
        const u32 ofBits = LDSTAB ? c_code(eO) : (eO >> 8) & 0xFF, mlBits = LDSTAB ? ml >> 24 : (eM >> 8) & 0xFF, llBits = LDSTAB ? ll >> 24 : (eL >> 8) & 0xFF;
        if (LDSTAB) { ll &= 0xFFFFFFu; ml &= 0xFFFFFFu; }
        bool badOff = false;
        if (ofBits > 1) {
          off = ((1u << ofBits) - 3u) + br.read_fast(ofBits);
          if (a.bpf > 1 && off >= ZRA_REP_MARK_LO) badOff = true;      // (block-parallel pass: beyond any frame it takes — and it would read as a marker)
          rep2 = rep1; rep1 = rep0; rep0 = off;
        } else {
          const u32 ll0 = (ll == 0);                       // the BASE value (code 0)
          if (ofBits == 0) {
            if (!ll0) off = rep0;
            else { off = rep1; rep1 = rep0; rep0 = off; }
          } else {
            const u32 idx = 1 + ll0 + br.read_fast(1);
            u32 t = idx == 3 ? rep0 - 1 : idx == 1 ? rep1 : rep2;
            t += !t;                                        // "0 is not valid; input is corrupted; force offset to 1"
            if (idx != 1) rep2 = rep1;
            rep1 = rep0; rep0 = off = t;
          }
        }
        if (mlBits) ml += br.read_fast(mlBits);
        if (llBits + mlBits + ofBits >= 57 - (9 + 9 + 8)) br.reload();
        if (llBits) ll += br.read_fast(llBits);
        sLL = c_sbase(eL, logLL) + br.read(c_sbits(eL, logLL));
        sML = c_sbase(eM, logML) + br.read(c_sbits(eM, logML));
        sOF = c_sbase(eO, logOF) + br.read(c_sbits(eO, logOF));
        fetch_cells();                                      // next sequence's cells and the container: one round trip
        if (!longMode) br.reload_quiet(wide, pad);
        // ZSTD_execSequence / ZSTD_execSequenceEnd, checks only (the execute kernel moves the bytes): destination room, literal
        // buffer, then — the literals now count as consumed — the offset
        if (jErr == 0xFFFFFFFFu) {
          u32 e = 0;
          if (ll + ml > outCap - outPos) e = ZE_DSTSIZE_TOOSMALL;
          else if (ll > regen - litPos) e = ZE_CORRUPTION;
          else if (badOff || (off > produced0 + outPos + ll && !(a.bpf > 1 && off >= ZRA_REP_MARK_LO))) e = ZE_CORRUPTION;   // (a marker: the execute stage checks the value it stands for)
          if (e) { jErr = i; err = e; if (!longMode) finish(); }
          else {
            const u64 qv = (u64)ll | ((u64)ml << 18) | ((u64)min(off, 0x0FFFFFFFu) << 36);
            const u32 slot = i & 3u;
            if (slot == 0) q0 = qv; else if (slot == 1) q1 = qv; else if (slot == 2) q2 = qv;
            else { uint4* d4 = (uint4*)(sq + (i - 3)); d4[0] = make_uint4((u32)q0, (u32)(q0 >> 32), (u32)q1, (u32)(q1 >> 32)); d4[1] = make_uint4((u32)q2, (u32)(q2 >> 32), (u32)qv, (u32)(qv >> 32)); }
            outPos += ll + ml; litPos += ll;
            valid = i + 1; validOut = outPos; validLit = litPos;
            if (produced0 + outPos >= limit) { truncated = 1; finish(); }     // random access: stop at the sequence that covers the last needed byte
          }
        }
        i++;
      }
    }
  }
}
}  // namespace

extern "C" __global__ void __launch_bounds__(DEC_THREADS)
zra_dec_chain_kernel(ZraDecodeArgs a) {
  __shared__ u32 baseLL[64], baseML[64];
  chain_body<false>(a, nullptr, baseLL, baseML);
}
// the same with the tables in LDS: one workgroup per CU (dynamic LDS: ZRA_CHAIN_LDS_FRAMES tables and bitstream rings + the two base-value
// tables), launched beside zra_dec_chain_kernel on another stream; both pull frames from the same queue. What this wave decodes asks
// nothing of L2 and the fabric but 16-byte pieces of its bitstream, ahead of their use, and its sequences.
// Round 5, measured (profiles/r05_experiments.md §10): this kernel ALONE takes 56-58 ms per 8 GiB (64 KiB frames) — 0.97 us per sequence
// and lane, with or without its sequence stores, with the bitstream from the ring or from memory: a lone wave per CU is bound by the
// ~500 instructions of a step, not by memory. Beside the HBM-table kernel (30 frames: 31 with rings no longer fit a CU next to that
// kernel's two waves, LDS comes in pieces of 1,280 bytes) it takes 18 % of the frames: stage 23.6 -> 19.4 ms (20.4 without the ring).
extern "C" __global__ void __launch_bounds__(DEC_THREADS)
zra_dec_chain_lds_kernel(ZraDecodeArgs a) {
  extern __shared__ u32 chainLds[];
  chain_body<true>(a, chainLds + 128, chainLds, chainLds + 64, chainLds + 128 + ZRA_CHAIN_LDS_FRAMES * (ZRA_DEC_TBL_WORDS / 2));
}

// =================================================================================================
// stage 3: execute. 64 sequences per step: wave scan of the lengths, per-lane literal runs, match copies in dependency rounds (a
// lane is ready once its source ends before the first unfinished destination). Copies issue all their loads before their first
// store and use as few memory instructions as possible (copy_le64): the kernel is bound by the CU's address unit and by the store
// drains between rounds. (Measured and dropped: chasing a match's source back through the step's sequences so that it reads final
// bytes only — sources mostly straddle sequence boundaries, 1.6 % of the matches could be followed.)
namespace {
struct __attribute__((aligned(16))) ExecShared {
  union {
    u8 slot[BATCH][64];     // exec_step_global: a lane's scratch for the first period of an overlapping match
    u8 win[XPRE + XWIN];    // exec_step: the step's output while it is being put together, behind the XPRE bytes that precede it
  };
  u32 endAt[BATCH];         // where each sequence of the step ends in the output (ascending) ...
  u32 matchAt[BATCH];       // ... and where its match begins (ascending): a match's source range is looked up in these
  u32 job;
};
}  // namespace

#ifdef ZRA_DEC_PROFILE
__device__ unsigned long long zra_dec_prof[16];
#define XCNT(k, v) { if (lane == 0) atomicAdd(&zra_dec_prof[k], (unsigned long long)(v)); }
#define XTIME(k) { __builtin_amdgcn_s_waitcnt(0); const u64 n_ = __builtin_amdgcn_s_memtime(); if (lane == 0) atomicAdd(&zra_dec_prof[k], n_ - xpt_); xpt_ = n_; }
extern "C" __attribute__((visibility("default"))) void ZraHipDebugReadDecProfile(unsigned long long* out16, int reset) {
  (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(zra_dec_prof), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(zra_dec_prof), z, sizeof(z)); }
}
#else
#define XCNT(k, v)
#define XTIME(k)
#endif

namespace {
// where a block's output and literals are, and how far its sequences have been executed (wave-uniform)
struct ExecCtx {
  u8* out; const u8* lit; u32 litKind; u8 rleByte;
  u32 outBase, litBase;
  const u8* src; u32 srcSize, produced0, regen;
  bool preValid;            // the LDS window holds the XPRE bytes in front of outBase
};
__device__ __forceinline__ ExecCtx exec_begin(const ZraDecodeArgs& a, const u32 j, const ZraDecFrame* const F) {
  ExecCtx c;
  const size_t gj = j;
  const u64 so = a.frameOff[gj * a.offStride], se = a.frameOff[gj * a.offStride + 1];
  c.src = a.body + so; c.srcSize = (u32)(se - so);
  c.produced0 = F->produced;
  c.out = a.out + a.outOff[j] + c.produced0;                    // this block's output start
  c.litKind = F->litKind; c.regen = F->litRegen;
  c.lit = c.litKind == 2 ? a.lits + F->litBase : c.src + F->litArg;
  c.rleByte = (u8)F->litArg;
  c.outBase = 0; c.litBase = 0; c.preValid = false;
  return c;
}
// up to 64 sequences, lane = sequence (act: the lane has one; oStart / lStart: where its literals go and come from)
__device__ __forceinline__ void exec_step_global(const ExecCtx& c, ExecShared& S, const u32 ll, const u32 ml, const u32 off, const u32 oStart, const u32 lStart,
                                                 const bool act, const int lane, const u32 debugSkip) {
  u8* const out = c.out; const u8* const lit = c.lit; const u32 litKind = c.litKind; const u8 rleByte = c.rleByte;
  const u32 mdst = oStart + ll;
#ifdef ZRA_DEC_PROFILE
  u64 xpt_ = __builtin_amdgcn_s_memtime();
#endif
  // -------- literal runs
  {
    u8* op = out + oStart;
    const bool longLit = ll > 32;
    if (!longLit && ll && !(debugSkip & 1)) {
      if (litKind == 1) for (u32 b = 0; b < ll; b++) op[b] = rleByte;
      else copy_le64(op, lit + lStart, ll);
    }
    u64 lm = (debugSkip & 8) ? 0ull : __ballot(longLit);
    while (lm) {
      const u32 k = (u32)__builtin_ctzll(lm); lm &= lm - 1;
      const u32 jl = bcast_u32(ll, k), jo = bcast_u32(oStart, k), js = bcast_u32(lStart, k);
      if (litKind == 1) fill_bytes(out + jo, rleByte, jl, lane, DEC_THREADS);
      else copy_bytes(out + jo, lit + js, jl, lane, DEC_THREADS);
    }
  }
  wsync();
  XTIME(2)
  // -------- the rest: dependency rounds. A match may go once every sequence of this step whose output its source range touches has
  // gone: the sequences [a, b) with a = the first one that ends behind the source's first byte, b = the first one whose match begins
  // at or behind the source's end — two binary searches in the step's (ascending) positions, once per step. (It was: once its source
  // ends before the FIRST unfinished match — 11 rounds per step on text, where most matches reach back a few sequences only.)
  {
    const i32 msrc = (i32)mdst - (i32)off;
    const i32 msrcEnd = min(msrc + (i32)ml, (i32)mdst);
    const u32 stepStart = bcast_u32(oStart, 0);
    S.endAt[lane] = mdst + ml; S.matchAt[lane] = mdst;
    wsync();
    u64 deps = 0;
    if (act && msrcEnd > (i32)stepStart) {
      u32 a = 0, b = 0;
#pragma unroll
      for (u32 st = BATCH / 2; st >= 1; st >>= 1) {
        a += (i32)S.endAt[a + st - 1] <= msrc ? st : 0u;
        b += (i32)S.matchAt[b + st - 1] < msrcEnd ? st : 0u;
      }
      a += (i32)S.endAt[a] <= msrc ? 1u : 0u;
      b += (i32)S.matchAt[b] < msrcEnd ? 1u : 0u;
      const u64 below_b = b >= 64 ? ~0ull : (1ull << b) - 1ull, below_a = a >= 64 ? ~0ull : (1ull << a) - 1ull;
      deps = below_b & ~below_a & ((1ull << lane) - 1ull);
    }
    u64 pending = __ballot(act);
    while (pending) {
      const bool mine = (pending >> lane) & 1;
      const bool ready = mine && !(deps & pending);
      const bool longM = ready && ml > 64;
      if (ready && !longM && !(debugSkip & 2)) {
        u8* dp = out + mdst; const u8* sp = dp - off;
        if (off >= ml) copy_le64(dp, sp, ml);      // no overlap: all loads, then all stores
        else copy_periodic_le64(dp, sp, ml, off, S.slot[lane]);  // overlapping match = period `off`: only the bytes in front of the destination are read
      }
      u64 lmk = __ballot(longM);
      while (lmk) {                            // long matches: the whole wave copies (period-safe modular source)
        const u32 k2 = (u32)__builtin_ctzll(lmk); lmk &= lmk - 1;
        const u32 jml = bcast_u32(ml, k2), jd = bcast_u32(mdst, k2), jof = bcast_u32(off, k2);
        u8* dp = out + jd; const u8* sp = dp - jof;
        if (jof >= jml) { for (u32 k = lane; k < jml; k += WAVE) dp[k] = sp[k]; }
        else { for (u32 k = lane; k < jml; k += WAVE) dp[k] = sp[k % jof]; }
      }
      pending &= ~__ballot(ready);
      if (!(debugSkip & 4)) wsync();
      XCNT(12, 1)
    }
  }
  XTIME(3)
}
// LDS traffic of the wave is complete and visible to its other lanes (global stores may still be in flight)
__device__ __forceinline__ void lsync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}
// The same step put together in LDS. A step's matches mostly copy from the step itself or from just in front of it, and every
// dependency round of exec_step_global is a round trip through the memory system (stores drained, then dependent loads): six to seven
// per step on text. Here the step's output — literal runs, then the matches in the same dependency rounds — is assembled in a window in
// LDS that also holds the XPRE bytes in front of the step (a match of up to 64 bytes that begins less than 64 bytes in front of the
// step lies in the window entirely; one that begins further back lies in finished output entirely), the rounds cost LDS round trips,
// and the window goes out in 16-byte stores, followed by the step's ONE drain. Steps larger than the window take exec_step_global.
__device__ __forceinline__ void exec_step(ExecCtx& c, ExecShared& S, const u32 ll, const u32 ml, const u32 off, const u32 oStart, const u32 lStart,
                                          const bool act, const int lane, const u32 debugSkip) {
  const u32 stepStart = bcast_u32(oStart, 0), stepLen = bcast_u32(oStart + ll + ml, 63) - stepStart;
  if (stepLen > XWIN || debugSkip) { exec_step_global(c, S, ll, ml, off, oStart, lStart, act, lane, debugSkip); c.preValid = false; return; }
  u8* const out = c.out; const u8* const lit = c.lit; const u32 litKind = c.litKind; const u8 rleByte = c.rleByte;
  u8* const W = S.win;                                    // output byte x of the window sits at W[x + wb] (x >= stepStart - XPRE; plain indices:
  const u32 wb = XPRE - stepStart;                        //  a biased LDS pointer would not survive being widened to a flat one)
  if (!c.preValid) {
    const i32 pos = (i32)stepStart - (i32)XPRE + lane;   // (bytes in front of the frame do not exist and are never a source)
    W[lane] = pos >= -(i32)c.produced0 ? out[pos] : (u8)0;
  }
  const u32 mdst = oStart + ll;
  // -------- literal runs
  {
    u8* const op = W + (oStart + wb);
    const bool longLit = ll > 32;
    if (!longLit && ll) {
      if (litKind == 1) for (u32 b = 0; b < ll; b++) op[b] = rleByte;
      else copy_le64(op, lit + lStart, ll);
    }
    u64 lm = __ballot(longLit);
    while (lm) {
      const u32 k = (u32)__builtin_ctzll(lm); lm &= lm - 1;
      const u32 jl = bcast_u32(ll, k), jo = bcast_u32(oStart, k), js = bcast_u32(lStart, k);
      if (litKind == 1) fill_bytes(W + (jo + wb), rleByte, jl, lane, DEC_THREADS);
      else copy_bytes(W + (jo + wb), lit + js, jl, lane, DEC_THREADS);
    }
  }
  // -------- matches, in dependency rounds (see exec_step_global)
  {
    const i32 msrc = (i32)mdst - (i32)off;
    const i32 msrcEnd = min(msrc + (i32)ml, (i32)mdst);
    S.endAt[lane] = mdst + ml; S.matchAt[lane] = mdst;
    lsync();
    u64 deps = 0;
    if (act && msrcEnd > (i32)stepStart) {
      u32 a = 0, b = 0;
#pragma unroll
      for (u32 st = BATCH / 2; st >= 1; st >>= 1) {
        a += (i32)S.endAt[a + st - 1] <= msrc ? st : 0u;
        b += (i32)S.matchAt[b + st - 1] < msrcEnd ? st : 0u;
      }
      a += (i32)S.endAt[a] <= msrc ? 1u : 0u;
      b += (i32)S.matchAt[b] < msrcEnd ? 1u : 0u;
      const u64 below_b = b >= 64 ? ~0ull : (1ull << b) - 1ull, below_a = a >= 64 ? ~0ull : (1ull << a) - 1ull;
      deps = below_b & ~below_a & ((1ull << lane) - 1ull);
    }
    const bool inWin = msrc >= (i32)stepStart - (i32)XPRE;       // the source lies in the window (else: in finished output, all of it, for ml <= 64)
    u64 pending = __ballot(act);
    while (pending) {
      const bool mine = (pending >> lane) & 1;
      const bool ready = mine && !(deps & pending);
      const bool longM = ready && ml > 64;
      if (ready && !longM) {
        u8* const dp = W + (mdst + wb);
        if (inWin) {
          const u8* const sp = W + ((u32)msrc + wb);
          if (off >= ml) copy_le64(dp, sp, ml);
          else if (off < 8) copy_periodic_le64(dp, sp, ml, off, nullptr);
          else { u32 pp = 0; for (u32 k = 0; k < ml; k++) { dp[k] = sp[pp]; pp = pp + 1 == off ? 0 : pp + 1; } }     // (the first period lies in front of dp, final)
        } else copy_le64(dp, out + msrc, ml);                    // (off > 64 >= ml: no overlap)
      }
      u64 lmk = __ballot(longM);
      while (lmk) {                            // long matches: the whole wave copies, byte by byte from wherever the byte is
        const u32 k2 = (u32)__builtin_ctzll(lmk); lmk &= lmk - 1;
        const u32 jml = bcast_u32(ml, k2), jd = bcast_u32(mdst, k2), jof = bcast_u32(off, k2);
        const i32 js = (i32)jd - (i32)jof;
        for (u32 k = lane; k < jml; k += WAVE) {
          const i32 q = js + (i32)(jof >= jml ? k : k % jof);
          u8 v;
          if (q >= (i32)stepStart - (i32)XPRE) v = W[(u32)q + wb]; else v = out[q];
          W[jd + k + wb] = v;
        }
        lsync();
      }
      pending &= ~__ballot(ready);
      lsync();
    }
  }
  // -------- the window goes out; its last XPRE bytes stay as the next step's front
  {
    const u8* const ws = S.win + XPRE;
    u8* const od = out + stepStart;
    for (u32 i = 16u * (u32)lane; i + 16 <= stepLen; i += 16u * WAVE) { const uint4 v = *(const uint4*)(ws + i); st128(od + i, v.x, v.y, v.z, v.w); }
    const u32 tail = stepLen & ~15u;
    if (tail + (u32)lane < stepLen) od[tail + lane] = ws[tail + lane];
    const u8 keep = S.win[stepLen + lane];
    lsync();
    S.win[lane] = keep;
    c.preValid = true;
  }
  wsync();
}

// block tail (remaining literals: the chain stage checked the room) and commit: a compressed block confirms its repeat offsets; then
// the frame ends, or takes another round. FUSED: a frame that goes on is not appended to the next round's list; returns 1 when it goes on.
template <bool FUSED>
__device__ __forceinline__ u32 exec_end(const ZraDecodeArgs& a, const u32 j, ZraDecFrame* const F, const ExecCtx& c, const int lane) {
  u32 more = 0;
  const u32 chainErr = F->chainErr, truncated = F->truncated;
  u32 blockOut = c.outBase;
  if (!chainErr && !truncated) {
    const u32 tail = c.regen - c.litBase;
    if (c.litKind == 1) fill_bytes(c.out + c.outBase, c.rleByte, tail, lane, DEC_THREADS);
    else copy_bytes(c.out + c.outBase, c.lit + c.litBase, tail, lane, DEC_THREADS);
    blockOut += tail;
  }
  wsync();
  const u32 produced = c.produced0 + blockOut, endPos = F->bpos + F->bsize;
  if (chainErr || truncated || F->blast) {
    frame_finish(a, j, c.src, c.srcSize, chainErr, produced, endPos, truncated != 0, F->fcsHave, F->fcsLo, F->fcsHi, F->hasChecksum, lane);
  } else {
    more = 1;
    if (lane == 0) {
      F->produced = produced; F->blkPos = endPos;
      F->rep[0] = F->repOut[0]; F->rep[1] = F->repOut[1]; F->rep[2] = F->repOut[2];
      if (!FUSED) a.nextActive[atomicAdd(&a.counters[ZRA_DC_NNEXT], 1u)] = j;
    }
  }
  wsync();
  return more;
}

// one job (frame j, the block the chain stage left sequences for) by one wave
template <bool FUSED>
__device__ __forceinline__ u32 exec_job(const ZraDecodeArgs& a, const u32 j, ExecShared& S, const int lane) {
  ZraDecFrame* const F = &a.frames[j];
  ExecCtx c = exec_begin(a, j, F);
  const u32 nSeq = F->nSeqValid;
  const u64* const sq = a.seqs + F->seqBase;
#ifdef ZRA_DEC_PROFILE
  u64 xpt_ = __builtin_amdgcn_s_memtime();
#endif
  XCNT(8, 1)
  u64 qNext = (u32)lane < nSeq ? sq[lane] : (1ull << 36);        // the next step's sequences are always in flight
  for (u32 first = 0; first < nSeq; first += BATCH) {
    const u32 cnt = min((u32)BATCH, nSeq - first);
    const bool act = (u32)lane < cnt;
    const u64 q = qNext;
    qNext = first + BATCH + (u32)lane < nSeq ? sq[first + BATCH + lane] : (1ull << 36);
    const u32 ll = (u32)q & 0x3FFFFu, ml = (u32)(q >> 18) & 0x3FFFFu, off = (u32)(q >> 36);
    const u32 tot = ll + ml;
    const u32 incT = dpp_scan_add(tot), incL = dpp_scan_add(ll);
    XTIME(0) XCNT(9, 1) XCNT(10, cnt)
    exec_step(c, S, ll, ml, off, c.outBase + incT - tot, c.litBase + incL - ll, act, lane, a.debugSkip);
#ifdef ZRA_DEC_PROFILE
    xpt_ = __builtin_amdgcn_s_memtime();
#endif
    c.outBase += bcast_u32(incT, 63); c.litBase += bcast_u32(incL, 63);
  }
  return exec_end<FUSED>(a, j, F, c, lane);
}
}  // namespace

// the same for the execute kernel (per-lane copies, waits on memory). Round 2's kernel: 4 waves per SIMD (114 VGPRs) 26.4 ms, 5 (96, 14 spilled) 22.9,
// 6 (80) 23.4. With the LDS-window step (4.6 KiB of LDS per workgroup): 5 / 6 / 8 waves -> 14.3 / 13.3 / 17.5 ms per 8 GiB.
#ifndef ZRA_EXEC_WAVES
#define ZRA_EXEC_WAVES 6
#endif
extern "C" __global__ void __launch_bounds__(DEC_THREADS, ZRA_EXEC_WAVES)
zra_dec_exec_kernel(ZraDecodeArgs a) {
  __shared__ ExecShared S;
  const int lane = threadIdx.x;
  const u32 nPend = a.counters[ZRA_DC_NPENDING];
  for (;;) {
    if (lane == 0) S.job = atomicAdd(&a.counters[ZRA_DC_QEXEC], 1u);
    wsync();
    const u32 qi = S.job;
    wsync();
    if (qi >= nPend) return;
    (void)exec_job<false>(a, a.pending[qi], S, lane);
  }
}

// ---- block-parallel pass (round 6): the execute stage of a WHOLE frame — its compressed blocks in order (zra_dec_parse_all_kernel made
// each one a job of the Huffman and chain stages; raw and RLE blocks are in place already). What the chain stage could not know is settled
// here: a block's initial repeat offsets (the previous block's last three: markers in the sequences and in repOut get their values, and a
// marker's offset is checked against the bytes that exist), and that every compressed block but the frame's last regenerated a whole
// block (its successors were placed on that assumption). Anything else than a clean frame — a stage reported an error, a check fails —
// puts the frame on the bail list: the classic rounds decode it again and report what the reference reports.
namespace {
__device__ __forceinline__ void exec_frame_all(const ZraDecodeArgs& a, const u32 j, ExecShared& S, const int lane) {
  ZraDecFrame* const F = &a.frames[j];
  const u32 nBlk = F->nBlk;
  const u64 so = a.frameOff[(size_t)j * a.offStride], se = a.frameOff[(size_t)j * a.offStride + 1];
  const u8* const src = a.body + so; const u32 srcSize = (u32)(se - so);
  u32 C0 = 1, C1 = 4, C2 = 8;                           // the repeat offsets at the start of the block being executed
  auto resolve = [&](u32 v) -> u32 {
    if (v < ZRA_REP_MARK_LO) return v;
    const u32 k = (v - ZRA_REP_MARK_LO) >> 24, d = ZRA_REP_MARK(k) - v, ck = k == 0 ? C0 : k == 1 ? C1 : C2;
    return ck > d ? ck - d : 1u;                        // "offset - 1" d times, each time "0 -> 1"
  };
  bool bail = false, truncated = false;
  u32 produced = 0;
  for (u32 b = 0; b < nBlk; b++) {
    ZraDecFrame* const R = &a.blkRecs[(size_t)j * a.bpf + b];
    if (R->hufErr | R->lateErr | R->chainErr) { bail = true; break; }
    ExecCtx c;
    c.src = src; c.srcSize = srcSize; c.produced0 = R->produced;
    c.out = a.out + a.outOff[j] + c.produced0;
    c.litKind = R->litKind; c.regen = R->litRegen;
    c.lit = c.litKind == 2 ? a.lits + R->litBase : c.src + R->litArg;
    c.rleByte = (u8)R->litArg;
    c.outBase = 0; c.litBase = 0; c.preValid = false;
    const u32 nSeq = R->nSeqValid;
    const u64* const sq = a.seqs + R->seqBase;
    bool bad = false;
    u64 qNext = (u32)lane < nSeq ? sq[lane] : (1ull << 36);
    for (u32 first = 0; first < nSeq; first += BATCH) {
      const u32 cnt = min((u32)BATCH, nSeq - first);
      const bool act = (u32)lane < cnt;
      const u64 q = qNext;
      qNext = first + BATCH + (u32)lane < nSeq ? sq[first + BATCH + lane] : (1ull << 36);
      const u32 ll = (u32)q & 0x3FFFFu, ml = (u32)(q >> 18) & 0x3FFFFu, offRaw = (u32)(q >> 36);
      const u32 tot = ll + ml;
      const u32 incT = dpp_scan_add(tot), incL = dpp_scan_add(ll);
      const u32 oStart = c.outBase + incT - tot;
      const u32 off = resolve(offRaw);
      if (__ballot(act && offRaw >= ZRA_REP_MARK_LO && off > c.produced0 + oStart + ll)) { bad = true; break; }
      exec_step(c, S, ll, ml, off, oStart, c.litBase + incL - ll, act, lane, a.debugSkip);
      c.outBase += bcast_u32(incT, 63); c.litBase += bcast_u32(incL, 63);
    }
    if (bad) { bail = true; break; }
    u32 blockOut = c.outBase;
    const u32 trunc = R->truncated;
    if (!trunc) {
      const u32 tail = c.regen - c.litBase;
      if (c.litKind == 1) fill_bytes(c.out + c.outBase, c.rleByte, tail, lane, DEC_THREADS);
      else copy_bytes(c.out + c.outBase, c.lit + c.litBase, tail, lane, DEC_THREADS);
      blockOut += tail;
    }
    wsync();
    if (!trunc && !R->blast && blockOut != ZRA_FMB_BLOCK) { bail = true; break; }      // (its successors were placed behind a whole block)
    produced = c.produced0 + blockOut;
    if (trunc) { truncated = true; break; }
    const u32 n0 = resolve(R->repOut[0]), n1 = resolve(R->repOut[1]), n2 = resolve(R->repOut[2]);
    C0 = n0; C1 = n1; C2 = n2;
  }
  if (bail) {
    if (lane == 0) a.nextActive[atomicAdd(&a.counters[ZRA_DC_NNEXT], 1u)] = j;
    wsync();
    return;
  }
  // raw / RLE blocks behind the last compressed one are counted in the frame record's (assumed, now verified) total
  const u32 total = max(produced, F->produced);
  frame_finish(a, j, src, srcSize, 0, total, F->endPos, truncated || F->parseTrunc != 0, F->fcsHave, F->fcsLo, F->fcsHi, F->hasChecksum, lane);
  wsync();
}
}  // namespace

extern "C" __global__ void __launch_bounds__(DEC_THREADS, ZRA_EXEC_WAVES)
zra_dec_exec_all_kernel(ZraDecodeArgs a) {
  __shared__ ExecShared S;
  const int lane = threadIdx.x;
  const u32 nExec = a.counters[ZRA_DC_NEXEC];
  for (;;) {
    if (lane == 0) S.job = atomicAdd(&a.counters[ZRA_DC_QEXECALL], 1u);
    wsync();
    const u32 qi = S.job;
    wsync();
    if (qi >= nExec) return;
    exec_frame_all(a, a.execList[qi], S, lane);
  }
}

// =================================================================================================
// One-launch decode for SMALL random-access batches (the latency path; reference call: DecompressRA, zra.cpp:258-296, one core
// answers a 4 KiB query in ~37 us). The four-kernel pipeline above is built for throughput: a lone frame pays four launches, a host
// synchronisation per round and a sequence chain that runs one lane at the pace of global-memory round trips. Here ONE workgroup of
// two waves takes a frame through all stages without leaving the kernel: wave 0 parses (the same parse_job), then wave 1 decodes the
// Huffman literals while wave 0 runs the FSE sequence chain out of LDS (tables and bitstream copied there: a table cell costs an LDS
// read instead of an L2 round trip), then wave 0 executes (the same exec_job). The chain here is the straight-line case of
// zra_dec_chain_kernel's step — the 64-bit sequence loop on a well-formed stream; whatever that loop would have to look at twice (a
// sequence that fails a check, the long-offset loop, a literal stream the decoder rejects, a stream that is not consumed exactly) makes
// the job BAIL: the host then takes the whole batch through the four-kernel pipeline, which is where every status of the reference is
// reproduced. Valid archives rarely bail: a sequence bitstream shorter than 8 bytes (a highly compressible frame with one to three
// sequences) and the long-offset mode do — such a batch pays the one-launch pass and the four-kernel pass.
namespace {
// chain ring of the one-launch kernel (chain_produce / chain_consume below)
constexpr u32 RING = 256;                        // ring entries (power of two)
struct ChainRing {
  uint4 e[RING];
  u32 head, tail, prodDone, stop, consReady, consDone, bail, pad;
};
constexpr u32 SMALL_SEQ_BYTES = 40u << 10;      // sequence bitstreams up to this size are copied to LDS (a 64 KiB frame's is ~10 KiB)
struct __attribute__((aligned(16))) SmallShared {
  ParseShared P;
  ExecShared X;
  u32 tabs[ZRA_LDS_TBL_WORDS];
  u16 hufTab[2048]; u8 hufW1[256];
  u8 bits[SMALL_SEQ_BYTES + 16];
  u32 ctl[4];                                   // [0] parse outcome, [1] bail, [2] frame goes on
  ChainRing ring;
};


// ---- the sequence chain of the one-launch kernel, split over two waves.
// PRODUCER (wave 0): BIT_DStream arithmetic only — per sequence the three table cells, the extra bits, the state updates, the reload —
// and hands {literal length, match length, offset code bits, raw offset bits} to an LDS ring. CONSUMER (wave 2): what the reference's
// sequence loop does with those numbers — repeat-offset resolution, the three checks of ZSTD_execSequence, positions, the stop at a
// random-access query's last byte — and the packed sequences for the execute stage. The chain's dependent path is the producer's; the
// consumer is the faster of the two and only ever waits.
__device__ __forceinline__ u32 ring_ld(const u32* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void ring_st(u32* p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

// lane i <- lane i-n inside a row of 16 (zero where there is no such lane); lanes 0..3 <-> 7..4 inside a group of 8
template <int N> __device__ __forceinline__ u32 dpp_row_shr(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + N, 0xF, 0xF, true); }
__device__ __forceinline__ u32 dpp_low4_from_mirror(u32 v) { return (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0x1, false); }

__device__ __forceinline__ void chain_produce(const ZraDecodeArgs& a, const u32 j, ZraDecFrame* const F, const u32* const T, u8* const bits, ChainRing& R, const int lane) {
  const u32 nbSeq = F->nbSeq;
  auto fail = [&]() { if (lane == 0) { ring_st(&R.bail, 1u); ring_st(&R.prodDone, 1u); } };
  if (F->lateErr || F->longMode) { fail(); return; }
  u32 i = 0;
  if (nbSeq) {
    const u32 n = F->bsize - F->seqPos;
    if (n < 8 || n > SMALL_SEQ_BYTES) { fail(); return; }
    {
      const u8* const g = a.body + a.frameOff[(size_t)j * a.offStride] + F->bpos + F->seqPos;
      for (u32 o = 8u * (u32)lane; o < n; o += 8u * DEC_THREADS) {
        if (o + 8 <= n) *(u64*)(bits + o) = ld64(g + o);
        else for (u32 k = o; k < n; k++) bits[k] = g[k];
      }
      wsync();
    }
    Zds br;
    if (!br.init(bits, n)) { fail(); return; }
    u32 sLL = br.read(F->llLog); br.reload();
    u32 sOF = br.read(F->ofLog); br.reload();
    u32 sML = br.read(F->mlLog); br.reload();
    bool stopped = false;
    // room in the ring / a stop request, looked at every sixteenth sequence
    auto gate = [&](u32 at) -> bool {
      if (ring_ld(&R.stop) | ring_ld(&R.bail)) return false;
      while (at + 16 - ring_ld(&R.tail) > RING) { if (ring_ld(&R.stop) | ring_ld(&R.bail)) return false; __builtin_amdgcn_s_sleep(1); }
      return true;
    };
    // ---- Far from the stream's start every reload is BIT_reloadDStream's first case and no read can leave the container, so the
    // reader is one number: `cur`, the bits still unread (container = the 8 bytes that end at bit cur rounded up to a byte, bitsConsumed =
    // the 0..7 bits of rounding). The six fields of a sequence — extra bits of OF, ML, LL, then the state bits of LL, ML, OF, in stream
    // order — sit on lanes 0,1,2 and 5,6,7 of the wave: each lane reads the cell of ITS stream, an 8-lane prefix sum of the widths places
    // the fields, one 64-bit shift per lane extracts them, and the new states go back to lanes 0..2 through a half-row mirror. The
    // dependent path of a sequence is one LDS round trip (cells and container together) and ~10 vector instructions instead of ~100.
    const u32 k8 = (u32)lane < 8u ? (u32)lane : 3u;
    const u32 strm = (k8 == 2 || k8 == 5) ? 0u : (k8 == 1 || k8 == 6) ? 1u : (k8 == 0 || k8 == 7) ? 2u : 3u;       // LL, ML, OF, idle
    const u8* const tk = (const u8*)(T + (strm == 0 ? ZRA_LDS_TBL_LL : strm == 1 ? ZRA_LDS_TBL_ML : ZRA_LDS_TBL_OF));
    const u32 shk = strm < 2 ? 3u : strm == 2 ? 2u : 0u;                         // cell size (idle lanes read the first OF cell, or junk)
    const u32 fOff = k8 < 3 ? 8u : 16u, fWid = k8 < 3 ? 8u : k8 >= 5 ? 4u : 0u;
    // what a lane leaves in the ring entry: lane 2 -> word 0 (literal length), lane 1 -> word 1 (match length), lane 0 -> word 2
    // ((1 << code) + extra bits); every other lane -> word 3, which nobody reads
    u32* const slot = (u32*)&R.e[0] + (k8 < 3 ? 2u - k8 : 3u);
    const u32 mBase = (k8 == 1 || k8 == 2) ? ~0u : 0u, mPow = k8 == 0 ? ~0u : 0u;
    u32 checkedLeft = 0;
    for (;;) {
      i32 cur = (i32)rfl(8u * br.ptr + 64u - br.bc);
      if (i < nbSeq && cur >= 256 && br.ptr >= 8 && br.bc < 8) {
        u32 st = strm == 0 ? sLL : strm == 1 ? sML : strm == 2 ? sOF : 0u;
        // ---- sixteen sequences at a time, straight-line: no loop counter, no ring arithmetic (the entry is an immediate offset), no
        // branch on a wide sequence — the widest one of the block is looked at once, behind it, and a block that had one is run again
        // through the checked steps below (nothing of it was published yet). 16 x 64 bits of room under the read position required.
        for (;;) {
          while (!checkedLeft && (i & 15u) == 0 && i + 16 <= nbSeq && cur >= 256 + 16 * 64) {
            if (i) { if (lane == 0) ring_st(&R.head, i); }
            if (!gate(i)) { stopped = true; break; }
            const u32 st0 = st; const i32 cur0 = cur;
            u32 widest = 0;
            u32* const blockSlot = slot + 4u * (i & (RING - 1));
#pragma unroll
            for (int r = 0; r < 16; r++) {
              const u32 p8 = ((u32)cur + 7u) >> 3, u = 8u * p8 - (u32)cur;             // container = the 8 bytes below byte p8; u = its 0..7 consumed bits
              const u32* const cell = (const u32*)(tk + (st << shk));
              const u32 x = cell[0], y = cell[1];
              const u64 c = ld64(bits + p8 - 8u);
              asm volatile("" :: "v"((u32)c), "v"((u32)(c >> 32)), "v"(y));             // (cells and container in flight together)
              const u32 w = __builtin_amdgcn_ubfe(x, fOff, fWid);
              u32 s = w + (u & mPow);                                                    // lane 0's field starts u bits below the container's top
              s += dpp_row_shr<1>(s); s += dpp_row_shr<2>(s); s += dpp_row_shr<4>(s);
              const u32 used = bcast_u32(s, 7);                                          // u + the sequence's bits
              widest = max(widest, used);
              const u32 val = __builtin_amdgcn_ubfe((u32)(c >> ((64u - s) & 63u)), 0u, w);
              st = dpp_low4_from_mirror((x >> 20) + val);
              blockSlot[4 * r] = val + ((1u << w) & mPow) + (y & mBase);
              cur -= (i32)(used - u);
            }
            if (widest > 64) { st = st0; cur = cur0; checkedLeft = 16; break; }
            i += 16;
          }
          if (stopped) break;
          // ---- one sequence at a time, each looked at before it is taken: up to the next block boundary, near the stream's start,
          // and through a block that holds a wide sequence
          bool again = false;
          while (i < nbSeq && cur >= 256) {
            if (!checkedLeft && (i & 15u) == 0 && i + 16 <= nbSeq && cur >= 256 + 16 * 64) { again = true; break; }
            if ((i & 15u) == 0) { if (i) { if (lane == 0) ring_st(&R.head, i); } if (!gate(i)) { stopped = true; break; } }
            const u32 p = (((u32)cur + 7u) >> 3) - 8u, room = (u32)cur - 8u * p;      // room = 64 - bitsConsumed
            const u32* const cell = (const u32*)(tk + (st << shk));
            const u32 x = cell[0], y = cell[1];
            const u64 c = ld64(bits + p);
            asm volatile("" :: "v"((u32)c), "v"((u32)(c >> 32)), "v"(y));
            const u32 w = __builtin_amdgcn_ubfe(x, fOff, fWid);
            u32 s = w; s += dpp_row_shr<1>(s); s += dpp_row_shr<2>(s); s += dpp_row_shr<4>(s);
            const u32 total = bcast_u32(s, 7);
            if (total > room) break;                                                    // (a sequence of more than 57 bits: the careful step)
            const u32 val = __builtin_amdgcn_ubfe((u32)(c >> ((room - s) & 63u)), 0u, w);
            st = dpp_low4_from_mirror((x >> 20) + val);
            slot[4u * (i & (RING - 1))] = val + ((1u << w) & mPow) + (y & mBase);
            cur -= (i32)total; i++;
            if (checkedLeft) checkedLeft--;
          }
          if (!again) break;
        }
        sLL = bcast_u32(st, 2); sML = bcast_u32(st, 1); sOF = bcast_u32(st, 0);
        br.ptr = (((u32)cur + 7u) >> 3) - 8u; br.bc = 8u * br.ptr + 64u - (u32)cur; br.c = ld64(bits + br.ptr);
        if (stopped) break;
      }
      if (i >= nbSeq) break;
      // ---- one careful step (every case of BIT_reloadDStream): the stream's last bytes, and sequences wider than a container
      if ((i & 15u) == 0) { if (i) { if (lane == 0) ring_st(&R.head, i); } if (!gate(i)) { stopped = true; break; } }
      const uint2 eL = *(const uint2*)(T + ZRA_LDS_TBL_LL + 2 * sLL), eM = *(const uint2*)(T + ZRA_LDS_TBL_ML + 2 * sML);
      const u32 eO = T[ZRA_LDS_TBL_OF + sOF];
      const u32 ofBits = (eO >> 8) & 0xFF, mlBits = (eM.x >> 8) & 0xFF, llBits = (eL.x >> 8) & 0xFF;
      u32 ll = eL.y, ml = eM.y, raw = 0;
      if (ofBits) raw = br.read_fast(ofBits);
      if (mlBits) ml += br.read_fast(mlBits);
      if (llBits + mlBits + ofBits >= 57 - (9 + 9 + 8)) br.reload();
      if (llBits) ll += br.read_fast(llBits);
      sLL = (eL.x >> 20) + br.read((eL.x >> 16) & 0xF);
      sML = (eM.x >> 20) + br.read((eM.x >> 16) & 0xF);
      sOF = (eO >> 20) + br.read((eO >> 16) & 0xF);
      br.reload_quiet(true, bits);
      if (lane == 0) R.e[i & (RING - 1)] = make_uint4(ll, ml, (1u << ofBits) + raw, 0u);
      i++;
    }
    if (!stopped && br.reload() < Zds::COMPLETED) { fail(); return; }     // the stream must be consumed exactly
  }
  if (lane == 0) { ring_st(&R.head, i); ring_st(&R.prodDone, 1u); }
}

// ring entry: {literal length, match length, (1 << offset code) + its extra bits}: 1 = repeat offset by literal length, 2 / 3 = the
// one-bit repeat codes, >= 4 = a new offset + 3
// ... and, as soon as the literals are there (*litFlag: 1 = ready, 2 = the literal decoder gave up), EXECUTES them: the execute stage of
// the block rides behind the chain instead of following it. `c` leaves with the positions the block's tail continues from.
__device__ __forceinline__ void chain_consume(const ZraDecodeArgs& a, const u32 j, ZraDecFrame* const F, ChainRing& R, ExecShared& X, const u32* const litFlag,
                                              ExecCtx& c, const int lane) {
  const u32 regen = F->litRegen, produced0 = F->produced;
  const u32 outCap = a.outCap[j] - produced0;
  const u32 limit = a.limit ? a.limit[j] : 0xFFFFFFFFu;
  bool litReady = false;
  u32 rep0 = rfl(F->rep[0]), rep1 = rfl(F->rep[1]), rep2 = rfl(F->rep[2]);
  u32 outPos = 0, litPos = 0, truncated = 0;
  u32 t = 0;
  bool bad = false;
  // up to 64 sequences at a time, lane = sequence: positions by prefix sums, the checks and the packing side by side; only the repeat
  // offsets are resolved one after the other (and only the sequences that use one: between two of them every sequence pushes a new offset)
  for (;;) {
    if (ring_ld(&R.bail)) return;
    const u32 done = ring_ld(&R.prodDone);
    const u32 h = ring_ld(&R.head);
    if (t == h) { if (done) break; __builtin_amdgcn_s_sleep(4); continue; }
    u32 nb = min(h - t, 64u);
    const bool on = (u32)lane < nb;
    const uint4 e4 = R.e[(t + (u32)lane) & (RING - 1)];
    const u32 ll = on ? e4.x : 0u, ml = on ? e4.y : 0u, ov = on ? e4.z : 4u;
    const u32 sOut = dpp_scan_add(ll + ml), sLit = dpp_scan_add(ll);
    const u32 outBefore = outPos + sOut - (ll + ml), litBefore = litPos + sLit - ll;
    // random access: stop at the sequence that covers the last needed byte
    const u64 tm = __ballot(on && produced0 + outPos + sOut >= limit);
    if (tm) { nb = (u32)__builtin_ctzll(tm) + 1u; truncated = 1; }
    const bool on2 = (u32)lane < nb;
    u32 off = ov - 3u;
    u64 m = __ballot(on2 && ov < 4u);
    u32 next = 0;                                          // first lane whose push is not in rep0..2 yet
    auto catch_up = [&](u32 L) {                           // the pushes of lanes next .. L-1 (new offsets, all of them)
      const u32 kk = L - next;
      if (kk >= 3) { rep0 = bcast_u32(off, L - 1); rep1 = bcast_u32(off, L - 2); rep2 = bcast_u32(off, L - 3); }
      else if (kk == 2) { rep2 = rep0; rep0 = bcast_u32(off, L - 1); rep1 = bcast_u32(off, L - 2); }
      else if (kk == 1) { rep2 = rep1; rep1 = rep0; rep0 = bcast_u32(off, L - 1); }
    };
    while (m) {
      const u32 L = (u32)__builtin_ctzll(m); m &= m - 1;
      catch_up(L);
      const u32 ovL = bcast_u32(ov, L), ll0 = bcast_u32(ll, L) == 0;
      u32 x;
      if (ovL == 1) {
        if (!ll0) x = rep0;
        else { x = rep1; rep1 = rep0; rep0 = x; }
      } else {
        const u32 idx = 1 + ll0 + (ovL - 2u);
        x = idx == 3 ? rep0 - 1 : idx == 1 ? rep1 : rep2;
        x += !x;
        if (idx != 1) rep2 = rep1;
        rep1 = rep0; rep0 = x;
      }
      off = wrlane_d(off, x, L);
      next = L + 1;
    }
    catch_up(nb);
    if (__ballot(on2 && (ll + ml > outCap - outBefore || ll > regen - litBefore || off > produced0 + outBefore + ll))) { bad = true; break; }
    if (!litReady) {
      u32 v;
      while ((v = ring_ld(litFlag)) == 0) __builtin_amdgcn_s_sleep(2);
      if (v == 2) { if (lane == 0) { ring_st(&R.bail, 1u); ring_st(&R.stop, 1u); } return; }
      litReady = true;
    }
    // a step cut short at the query's last byte ends where its last validated sequence ends: the lanes behind it carry that position
    // (exec_step takes the step's length from lane 63; the lengths of the unexecuted, unchecked sequences must not be in it)
    const u32 endOut = outPos + bcast_u32(sOut, nb - 1), endLit = litPos + bcast_u32(sLit, nb - 1);
    c.outBase = outPos; exec_step(c, X, on2 ? ll : 0u, on2 ? ml : 0u, on2 ? min(off, 0x0FFFFFFFu) : 1u, on2 ? outBefore : endOut, on2 ? litBefore : endLit, on2, lane, a.debugSkip);
    outPos += bcast_u32(sOut, nb - 1); litPos += bcast_u32(sLit, nb - 1);
    t += nb;
    if (lane == 0) ring_st(&R.tail, t);
    if (truncated) break;
  }
  if (bad || (!truncated && regen - litPos > outCap - outPos)) { if (lane == 0) { ring_st(&R.bail, 1u); ring_st(&R.stop, 1u); } return; }
  if (truncated && lane == 0) ring_st(&R.stop, 1u);
  c.outBase = outPos; c.litBase = litPos;
  if (lane == 0) {
    F->chainErr = 0; F->nSeqValid = t; F->seqOut = outPos; F->seqLit = litPos; F->truncated = truncated;
    F->repOut[0] = rep0; F->repOut[1] = rep1; F->repOut[2] = rep2;
  }
}
}  // namespace

// grid = jobs (one workgroup each), 192 threads = three waves: 0 parses, produces the chain and executes; 1 decodes the Huffman
// literals; 2 consumes the chain. *bail counts the jobs that have to go through the four-kernel pipeline.
extern "C" __global__ void __launch_bounds__(3 * DEC_THREADS)
zra_ra_small_kernel(ZraDecodeArgs a0, u32* bail, const u32* expect, u32 jobBase, unsigned long long* result) {
  __shared__ SmallShared S;
  const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
  const u32 j = blockIdx.x;
  ZraDecodeArgs a = a0;
  ZraDecFrame* const F = &a.frames[j];
  const u8* const lim = a.body + a.bodySize;
  SPROF_T0
#ifdef ZRA_SMALL_PROFILE
  if (threadIdx.x == 0) atomicAdd(&zra_small_prof[4], 1ull);
#endif
  bool bailed = false;
  for (u32 round = 0;; round++) {
    a.round = round;
    if (wave == 0) {
      const u32 oc = parse_job<true>(a, j, S.P, lane, S.tabs);
      if (lane == 0) {
        S.ctl[0] = oc; S.ctl[1] = 0; S.ctl[3] = 0;
        S.ring.head = 0; S.ring.tail = 0; S.ring.prodDone = 0; S.ring.stop = 0; S.ring.consReady = 0; S.ring.consDone = 0; S.ring.bail = 0;
      }
      SPROF(0)
    }
    __syncthreads();
    SPROF_RESET
    const u32 oc = S.ctl[0];
    if (oc == 0) break;                                    // the frame is finished (status and slices written by frame_finish)
    if (oc == 2) { bailed = true; break; }
    ExecCtx xc;
    if (wave == 1) {
      u32 ok = 1;
      if (F->litKind == 2) {
        huf_build_table(F, S.hufTab, S.hufW1, lane);
        wsync();
        const bool wide = huf_decode_wave(a, F, j, S.hufTab, lane, lim);
#ifdef ZRA_SMALL_PROFILE
        if (lane == 0) atomicAdd(&zra_small_prof[wide ? 14 : 15], 1ull);
#endif
        if (!wide)
          if ((u32)lane < F->litStreams && !huf_decode_stream(a, F, j, S.hufTab, S.hufW1, (u32)lane, lim)) S.ctl[1] = 1;
        wsync();
        ok = S.ctl[1] == 0;
      }
      __threadfence_block();
      if (lane == 0) ring_st(&S.ctl[3], ok ? 1u : 2u);             // the execute steps of the chain's consumer may read the literals
      SPROF(1)
    } else if (wave == 0) {
      chain_produce(a, j, F, S.tabs, S.bits, S.ring, lane);
      SPROF(2)
    } else {
      xc = exec_begin(a, j, F);
      chain_consume(a, j, F, S.ring, S.X, &S.ctl[3], xc, lane);
      SPROF(5)
    }
    __threadfence_block();
    __syncthreads();
    if (S.ctl[1] | S.ring.bail) { bailed = true; break; }
    SPROF_RESET
    if (wave == 2) { const u32 more = exec_end<true>(a, j, F, xc, lane); if (lane == 0) S.ctl[2] = more; SPROF(3) }
    __syncthreads();
    if (!S.ctl[2]) break;
  }
  // *bail counts DOWN from 0xFFFFFFFF (one memset presets it together with the result word)
  if (bailed) { if (threadIdx.x == 0) atomicSub(bail, 1u); return; }
  // ---- frame end, here instead of in two more launches: what zra_xxh64_verify_kernel and zra_first_error_kernel do behind the
  // four-kernel pipeline — content checksum over the regenerated bytes (four lanes), regenerated size against the slot, first error
  __threadfence_block();
  __syncthreads();
  if (wave == 1 && lane < 4) {
    const u32 st0 = a.status[j], meta = a.frameMeta[2 * (size_t)j];
    const bool live = st0 == 0 && meta != 2;            // 2: stopped early (random access), nothing to check
    const bool active = live && meta == 1;
    const u64 h = zra_xxh64_quad(active ? a.out + a.outOff[j] : a.out, active ? a.produced[j] : 0u, lane);
    if (lane == 0) {
      u32 st = st0;
      if (live) {
        if (active && (u32)h != a.frameMeta[2 * (size_t)j + 1]) st = ZE_CHECKSUM_WRONG;
        else if (expect && a.produced[j] != expect[j]) st = 255u;      // ZE_SIZE_MISMATCH (zra_engine.hip)
        if (st != st0) a.status[j] = st;
      }
      if (st) atomicMin(result, ((unsigned long long)(jobBase + j) << 8) | (st & 0xFF));
    }
  }
}
