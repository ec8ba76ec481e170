// zra_amd — ENCODE stage 2 for gfx950: entropy coding + block/frame assembly of zstd 1.4.9, bit-exact.
//
// One frame per workgroup (256 threads = 4 waves). Everything that is data-parallel runs on all lanes:
//   * literal gathering + byte histogram (wave prefix-scans for positions, LDS atomics for counts)
//   * Huffman literal streams: code lengths prefix-scanned across the workgroup, codes OR-ed into an LDS
//     staging tile (ballot-free, conflict-tolerant LDS atomics), flushed with coalesced stores
//   * sequence bitstream: the three FSE state chains (inherently serial) run on 3 lanes, then the
//     state/extra bits of 1024 sequences at a time are prefix-scanned and packed the same way
// The small serial pieces (Huffman tree, depth limiter, FSE normalisation, table descriptions) run on one
// lane per table, three tables concurrently on three waves. Tables, work arrays and staging live in LDS (20 KiB per workgroup).
// Rules restated from SURVEY.md Appendix A.4.2, A.4.4-A.4.8 (validated there against libzstd 1.4.9).
#include "zra_dev.h"
#include "zra_kernels.h"
#include "zra_encode_wave.h"

using namespace zra_dev;
using zra_wave::bcast;

namespace {

constexpr int ENT_THREADS = 256;
constexpr int SEQ_TILE = 512;      // sequences per packing tile
constexpr int SEQ_PER = SEQ_TILE / ENT_THREADS;   // ... per thread
constexpr int SYM_TILE = 4096;     // literal symbols per packing tile (16 per thread)
constexpr int STAGE_WORDS = 1536;  // 6 KiB LDS bit-staging buffer: 512 sequences x 90 bits, 4096 literals x 11 bits, 512 x 3 queue words
static_assert(STAGE_WORDS >= ((7 + SEQ_TILE * 90 + 28) >> 5) + 2 && STAGE_WORDS >= ((7 + SYM_TILE * 11 + 1) >> 5) + 2 && STAGE_WORDS >= 3 * SEQ_TILE, "staging tile too small");

__constant__ u8 c_LLcode[64] = {0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,16,17,17,18,18,19,19,20,20,20,20,21,21,21,21,22,22,22,22,22,22,22,22,
                                23,23,23,23,23,23,23,23,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24};
__constant__ u8 c_MLcode[128] = {0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,32,33,33,34,34,35,35,
                                 36,36,36,36,37,37,37,37,38,38,38,38,38,38,38,38,39,39,39,39,39,39,39,39,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,
                                 41,41,41,41,41,41,41,41,41,41,41,41,41,41,41,41,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,
                                 42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42};
__constant__ u8 c_LLbits[36] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16};
__constant__ u8 c_MLbits[53] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16};
__constant__ short c_LLdef[36] = {4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1};
__constant__ short c_MLdef[53] = {1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1};
__constant__ short c_OFdef[29] = {1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1};
// kInverseProbabilityLog256[i] = floor(-log2(i/256)*256), [0] = 0 (A.4.7 cost model)
__constant__ u16 c_invProb[256] = {0,2048,1792,1642,1536,1453,1386,1329,1280,1236,1197,1162,1130,1100,1073,1047,1024,1001,980,960,941,923,906,889,874,859,844,830,817,804,791,779,768,756,745,734,724,714,704,694,685,676,667,658,650,642,633,626,618,610,603,595,588,581,574,567,561,554,548,542,535,529,523,517,512,506,500,495,489,484,478,473,468,463,458,453,448,443,438,434,429,424,420,415,411,407,402,398,394,390,386,382,377,373,370,366,362,358,354,350,347,343,339,336,332,329,325,322,318,315,311,308,305,302,298,295,292,289,286,282,279,276,273,270,267,264,261,258,256,253,250,247,244,241,239,236,233,230,228,225,222,220,217,215,212,209,207,204,202,199,197,194,192,190,187,185,182,180,178,175,173,171,168,166,164,162,159,157,155,153,151,149,146,144,142,140,138,136,134,132,130,128,126,123,121,119,117,115,114,112,110,108,106,104,102,100,98,96,94,93,91,89,87,85,83,82,80,78,76,74,73,71,69,67,66,64,62,61,59,57,55,54,52,50,49,47,46,44,42,41,39,37,36,34,33,31,30,28,26,25,23,22,20,19,17,16,14,13,11,10,8,7,5,4,2,1};

__device__ __forceinline__ u32 ll_code(u32 v) { return v > 63 ? hb32(v) + 19 : c_LLcode[v]; }
__device__ __forceinline__ u32 ml_code(u32 mlBase) { return mlBase > 127 ? hb32(mlBase) + 36 : c_MLcode[mlBase]; }
__device__ __forceinline__ u32 hb32z(u32 x) { return x ? hb32(x) : 0; }

// LDS of one frame-block (~20 KiB: up to 7 workgroups per CU). The literal-phase arrays and the sequence-phase arrays are never
// live at the same time and share storage.
struct EncLitPhase {
  // (round 6) the literal histograms (gather phase, dead once the symbols are sorted) and the block's three sequence encoding tables (built
  // after that, alive to the block's end) share storage: with it the workgroup takes 14 of the CU's 1,280-byte LDS pieces instead of 19 —
  // room for a 19th match-finder wave beside it
  union {
    u32 hist[4][256];
    ZraFseCTable ct[3];      // 0 LL, 1 OF, 2 ML (next-block tables)
  };
  // Huffman construction (index 0 of node* is the sentinel "huffNode[-1]")
  u32 nodeCount[514];
  u16 nodeParent[514];
  u8 nodeBits[514];
  u8 nodeByte[256];
  u8 weights[256];
  u8 hufHdr[192];
  short wNorm[16];            // FSE description of the Huffman weights (tableLog <= 6)
  u8 wSpread[64];
  // small work arrays of the one-lane sections: in LDS, because a per-thread array that is indexed dynamically lives in scratch
  // (HBM-backed private memory) and every access then costs a memory round trip
  u32 wCount[16], rankLast[16];
  u16 nbPerRank[16], valPerRank[16];
  u8 ncTmp[192];
  // (the encoding table of the Huffman weights lives in the bit-staging tile between the table builds' arrays and the Huffman wave's scratch: ENT_WCT_OFF)
};
// work arrays of the three sequence-table builds: they live in the bit-staging tile, which nobody packs into while tables are built
struct EncSeqBuild {
  short norm[3][64];
  u8 spread[3][512];
  u32 cumul[3][56];           // fse_build_ctable work array (LDS, not scratch)
};
struct __attribute__((aligned(16))) EncShared {
  u32 stage[STAGE_WORDS];
  EncLitPhase lit;
  u32 seqCnt[3][64];         // code histograms of the block's sequences (counted while the literals are gathered: the sequences are in registers then)
  u32 tblReady[4];           // sequence table k built (lane 0 of wave k + 1 sets it; wave 1 walks the chains behind all three)
  // (the one-launch kernel's state chains keep their current 64 sequences — codes in, results out — at the start of the staging tile, where
  //  the table builds' arrays are dead by then: ENT_CHAIN_CODES / ENT_CHAIN_OUT)
  u8 ncount[3][192];         // table descriptions of the block's sequence section (written while the Huffman tree is built, emitted after the literals)
  u8 hNb[256];
  u16 hVal[256];
  u32 ncountSize[3], mode[3], nextRepeat[3], tblErr[3];
  u32 finalState[3];
  u32 wsum[16];
  u32 longCount;
  // scalars shared through LDS
  u32 sc[16];
};
static_assert(sizeof(EncSeqBuild) <= sizeof(u32) * STAGE_WORDS, "the table builds' work arrays must fit the staging tile");
// staging-tile tenants while tables are built (phase 2b): [0, 2592) EncSeqBuild (waves 1-3), [ENT_WCT_OFF, +1460) the weights' encoding
// table (wave 0), [4096, 6016) HufWaveScratch (wave 0); behind the builds, [0, 576) the one-launch kernel's chain arrays (wave 1)
constexpr u32 ENT_WCT_OFF = 2624;
static_assert(sizeof(EncSeqBuild) <= ENT_WCT_OFF && ENT_WCT_OFF + sizeof(ZraFseCTable) <= 4096, "the weights' table must fit between the table builds' arrays and the Huffman wave's scratch");
static_assert(sizeof(EncShared) <= 14 * 1280, "the entropy workgroup must fit 14 LDS pieces: 19 match-finder waves x 6 pieces + 14 = the CU's 128");

// ---- workgroup exclusive scan of one u32 per thread; returns exclusive prefix, *total = sum over the workgroup
__device__ __forceinline__ u32 block_excl_scan(EncShared& S, u32 v, u32* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32 inc = wave_incl_scan(v);
  __syncthreads();                       // protects wsum reuse between consecutive scans
  if (lane == 63) S.wsum[wave] = inc;
  __syncthreads();
  u32 base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 4; w++) { u32 s = S.wsum[w]; if (w < wave) base += s; tot += s; }
  *total = tot;
  return base + inc - v;
}
__device__ __forceinline__ u32 block_sum(EncShared& S, u32 v) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) S.wsum[8 + wave] = v;
  __syncthreads();
  return S.wsum[8] + S.wsum[9] + S.wsum[10] + S.wsum[11];
}
__device__ __forceinline__ u32 block_max(EncShared& S, u32 v) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = wave_max(v);
  __syncthreads();
  if (lane == 0) S.wsum[12 + wave] = v;
  __syncthreads();
  return max(max(S.wsum[12], S.wsum[13]), max(S.wsum[14], S.wsum[15]));
}

// ---- per-lane bit writer into the LDS staging tile (forward LSB-first stream, A.4.5/A.4.7 convention)
struct LaneBitW {
  u32* stage; u32 word; u32 n; u64 acc;
  __device__ __forceinline__ void init(u32* st, u32 startBit) { stage = st; word = startBit >> 5; n = startBit & 31; acc = 0; }
  __device__ __forceinline__ void add(u32 v, u32 nb) {   // nb <= 32, v already masked to nb bits
    acc |= (u64)v << n; n += nb;
    if (n >= 32) { atomicOr(&stage[word], (u32)acc); acc >>= 32; n -= 32; word++; }
  }
  __device__ __forceinline__ void finish() { if (n) atomicOr(&stage[word], (u32)acc); }
};

// flush the first nbytes of the staging tile to dst (cooperative, coalesced dword stores)
__device__ __forceinline__ void flush_stage(EncShared& S, u8* dst, u32 nbytes) {
  const u32 nw = nbytes >> 2;
  for (u32 i = threadIdx.x; i < nw; i += ENT_THREADS) st32(dst + 4 * i, S.stage[i]);
  const u32 tail = nbytes & 3;
  if (threadIdx.x < tail) dst[4 * nw + threadIdx.x] = (u8)(S.stage[nw] >> (8 * threadIdx.x));
}

// =============================================================================== FSE primitives (A.4.6), one lane
__device__ u32 fse_optimal_tablelog(u32 maxLog, u32 n, u32 maxSym, u32 minus) {
  u32 maxBitsSrc = hb32z(n - 1) - minus;         // unsigned wrap intended
  u32 t = maxLog;
  u32 a = hb32z(n) + 1, b = hb32z(maxSym) + 2;
  u32 minBits = a < b ? a : b;
  if (maxBitsSrc < t) t = maxBitsSrc;
  if (minBits > t) t = minBits;
  if (t < 5) t = 5;
  if (t > 12) t = 12;
  return t;
}

__device__ int fse_normalize_m2(short* norm, u32 t, const u32* cnt, u32 total, u32 maxSym, short low) {
  const short NYA = -2;
  u32 distributed = 0, toDist;
  u32 lowThr = total >> t, lowOne = (u32)(((u64)total * 3) >> (t + 1));
  for (u32 s = 0; s <= maxSym; s++) {
    if (cnt[s] == 0) { norm[s] = 0; continue; }
    if (cnt[s] <= lowThr) { norm[s] = low; distributed++; total -= cnt[s]; continue; }
    if (cnt[s] <= lowOne) { norm[s] = 1; distributed++; total -= cnt[s]; continue; }
    norm[s] = NYA;
  }
  toDist = (1u << t) - distributed;
  if (toDist == 0) return 0;
  if ((total / toDist) > lowOne) {
    lowOne = (u32)(((u64)total * 3) / ((u64)toDist * 2));
    for (u32 s = 0; s <= maxSym; s++)
      if (norm[s] == NYA && cnt[s] <= lowOne) { norm[s] = 1; distributed++; total -= cnt[s]; }
    toDist = (1u << t) - distributed;
  }
  if (distributed == maxSym + 1) {
    u32 maxV = 0, maxC = 0;
    for (u32 s = 0; s <= maxSym; s++) if (cnt[s] > maxC) { maxV = s; maxC = cnt[s]; }
    norm[maxV] += (short)toDist;
    return 0;
  }
  if (total == 0) {
    for (u32 s = 0; toDist > 0; s = (s + 1) % (maxSym + 1)) if (norm[s] > 0) { toDist--; norm[s]++; }
    return 0;
  }
  const u64 vLog = 62 - t, mid = (1ULL << (vLog - 1)) - 1;
  const u64 rStep = ((((u64)1 << vLog) * toDist) + mid) / total;
  u64 acc = mid;
  for (u32 s = 0; s <= maxSym; s++) {
    if (norm[s] == NYA) {
      u64 end = acc + (u64)cnt[s] * rStep;
      u32 w = (u32)(end >> vLog) - (u32)(acc >> vLog);
      if (w < 1) return -1;
      norm[s] = (short)w;
      acc = end;
    }
  }
  return 0;
}

// returns tableLog, 0 for the single-symbol case, <0 on error
__device__ int fse_normalize(short* norm, u32 t, const u32* cnt, u32 total, u32 maxSym, bool useLowProb) {
  const u32 rtb[8] = {0, 473195, 504333, 520860, 550000, 700000, 750000, 830000};
  const short low = useLowProb ? -1 : 1;
  const u64 scale = 62 - t, step = ((u64)1 << 62) / total, vStep = 1ULL << (scale - 20);
  int still = 1 << t;
  u32 largest = 0; short largestP = 0;
  const u32 lowThr = total >> t;
  for (u32 s = 0; s <= maxSym; s++) {
    if (cnt[s] == total) return 0;
    if (cnt[s] == 0) { norm[s] = 0; continue; }
    if (cnt[s] <= lowThr) { norm[s] = low; still--; }
    else {
      short p = (short)(((u64)cnt[s] * step) >> scale);
      if (p < 8) { u64 rest = vStep * rtb[p]; p += ((u64)cnt[s] * step) - ((u64)p << scale) > rest; }
      if (p > largestP) { largestP = p; largest = s; }
      norm[s] = p; still -= p;
    }
  }
  if (-still >= (norm[largest] >> 1)) { if (fse_normalize_m2(norm, t, cnt, total, maxSym, low)) return -1; }
  else norm[largest] += (short)still;
  return (int)t;
}

// table description writer (forward LSB-first); returns bytes, 0 on error. out must hold 192 bytes.
__device__ u32 fse_write_ncount(u8* out, const short* norm, u32 maxSym, u32 t) {
  u64 acc = 0; int nacc = 0; u32 pos = 0;
  const int tableSize = 1 << t;
  int remaining = tableSize + 1, thr = tableSize, nb = (int)t + 1;
  u32 sym = 0; const u32 alpha = maxSym + 1;
  bool prev0 = false;
  auto put = [&](u32 v, int n) { acc |= (u64)v << nacc; nacc += n; while (nacc >= 8) { out[pos++] = (u8)acc; acc >>= 8; nacc -= 8; } };
  put(t - 5, 4);
  while (sym < alpha && remaining > 1) {
    if (prev0) {
      u32 start = sym;
      while (sym < alpha && !norm[sym]) sym++;
      if (sym == alpha) break;
      while (sym >= start + 24) { start += 24; put(0xFFFFu, 16); }
      while (sym >= start + 3) { start += 3; put(3, 2); }
      put(sym - start, 2);
    }
    int c = norm[sym++];
    const int mx = (2 * thr - 1) - remaining;
    remaining -= c < 0 ? -c : c;
    c++;
    if (c >= thr) c += mx;
    put((u32)c, nb - (c < mx));
    prev0 = (c == 1);
    if (remaining < 1) return 0;
    while (remaining < thr) { nb--; thr >>= 1; }
    if (pos > 170) return 0;
  }
  if (remaining != 1) return 0;
  if (nacc > 0) out[pos++] = (u8)acc;
  return pos;
}

// (one wave, LDS in issue order: a wavefront-scope fence orders the compiler and waits for nothing)
__device__ __forceinline__ void wave_order() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

// The same table as fse_build_ctable, built by all 64 lanes of ONE wave (round 6; arguments wave-uniform, every lane calls). The serial
// build was 512 cells x two passes of dependent LDS round trips on one lane — the longest part of the entropy stage's one-lane sections
// beside the state chains. Here: the symbols' cell counts are prefix-scanned across lanes (lane = symbol, <= 53 of them); the spread's
// position sequence q(t) = t * step mod size is known for every t at once, the t-th VALID position (<= high: the low-probability symbols
// own the cells above) takes the symbol whose range of occurrences holds its rank (binary search through the lanes' inclusive sums);
// a cell's state index is its symbol's first index + its rank among the symbol's cells in cell order (one ballot per distinct symbol of
// 64 cells); the per-symbol transforms are one lane each.
__device__ int fse_build_ctable_wave(ZraFseCTable* ct, const short* norm, u32 maxSym, u32 t, u8* cell, int lane) {
  const u32 size = 1u << t, mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
  const u64 lt = (1ull << lane) - 1ull;
  const int p = (u32)lane <= maxSym ? (int)norm[lane] : 0;
  const u32 wAll = p == -1 ? 1u : (u32)(p > 0 ? p : 0), wN = p > 0 ? (u32)p : 0u;
  const u32 incAll = wave_incl_scan(wAll), incN = wave_incl_scan(wN);
  const u32 startAll = incAll - wAll;
  const u64 lowM = __ballot(p == -1);
  const u32 nLow = (u32)__popcll(lowM), high = size - 1 - nLow;
  const u32 totalN = bcast(incN, 63);
  if (lane == 0) { ct->tableLog = t; ct->maxSym = maxSym; ct->rle = 0; }
  if (totalN + nLow != size) return -1;                 // (fse_build_ctable: the spread does not come back to cell 0)
  if (p == -1) cell[size - 1 - (u32)__popcll(lowM & lt)] = (u8)lane;
  u32 carry = 0;
  for (u32 c = 0; c < size; c += 64) {
    const u32 tt = c + (u32)lane, q = (tt * step) & mask;
    const bool valid = tt < size && q <= high;
    const u64 vm = __ballot(valid);
    const u32 i = carry + (u32)__popcll(vm & lt);
    // symbol of occurrence i: the number of symbols whose inclusive sum is <= i (by every lane: the lanes read from must be executing)
    u32 sy = 0;
#pragma unroll
    for (u32 b = 32; b >= 1; b >>= 1) { const u32 v = (u32)__shfl((int)incN, (int)min(sy + b - 1, 63u), 64); if (sy + b <= 64 && v <= i) sy += b; }
    if (valid) cell[q] = (u8)sy;
    carry += (u32)__popcll(vm);
  }
  wave_order();
  u32 run = startAll;                                   // lane s: next state index of symbol s
  for (u32 c = 0; c < size; c += 64) {
    const u32 u = c + (u32)lane;
    const bool act = u < size;
    const u32 sy = act ? (u32)cell[u] : 0xFFu;
    u64 rem = __ballot(act);
    u32 idx = 0;
    while (rem) {
      const u32 l = (u32)__builtin_ctzll(rem), sv = bcast(sy, l);
      const u64 m = __ballot(act && sy == sv);
      const u32 base = bcast(run, sv);
      if (act && sy == sv) idx = base + (u32)__popcll(m & lt);
      if ((u32)lane == sv) run += (u32)__popcll(m);
      rem &= ~m;
    }
    if (act) ct->stateTable[idx] = (u16)(size + u);
  }
  if ((u32)lane <= maxSym) {
    u32 dnb; int dfs;
    if (p == 0) { dnb = ((t + 1) << 16) - (1u << t); dfs = 0; }
    else if (p == 1 || p == -1) { dnb = (t << 16) - (1u << t); dfs = (int)startAll - 1; }
    else { const u32 mbo = t - hb32((u32)p - 1); dnb = (mbo << 16) - ((u32)p << mbo); dfs = (int)startAll - p; }
    ct->deltaNbBits[lane] = dnb; ct->deltaFindState[lane] = dfs;
  }
  wave_order();
  return 0;
}
__device__ __forceinline__ u32 fse_init_state(const ZraFseCTable* ct, u32 sym) {
  if (ct->rle) return 0;
  const u32 d = ct->deltaNbBits[sym], nb = (d + (1u << 15)) >> 16, v = (nb << 16) - d;
  return ct->stateTable[(v >> nb) + ct->deltaFindState[sym]];
}
__device__ __forceinline__ u32 fse_encode(const ZraFseCTable* ct, u32& state, u32 sym, u32& bits) {
  if (ct->rle) { bits = 0; return 0; }
  const u32 nb = (state + ct->deltaNbBits[sym]) >> 16;
  bits = state & ((1u << nb) - 1);
  state = ct->stateTable[(state >> nb) + ct->deltaFindState[sym]];
  return nb;
}

// =============================================================================== Huffman (A.4.5)
// depth limiter on the sorted node arrays (positions 0..lastNonNull, +1 offset in LDS arrays); one lane
__device__ u32 huf_set_max_height(EncShared& S, u32 lastNonNull, u32 maxNbBits) {
  u8* nb = S.lit.nodeBits + 1; const u32* cnt = S.lit.nodeCount + 1;
  const u32 largestBits = nb[lastNonNull];
  if (largestBits <= maxNbBits) return largestBits;
  int totalCost = 0;
  const u32 baseCost = 1u << (largestBits - maxNbBits);
  int n = (int)lastNonNull;
  while (nb[n] > maxNbBits) { totalCost += (int)(baseCost - (1u << (largestBits - nb[n]))); nb[n] = (u8)maxNbBits; n--; }
  while (nb[n] == maxNbBits) n--;
  totalCost >>= (largestBits - maxNbBits);
  const u32 none = 0xF0F0F0F0u;
  u32* const rankLast = S.lit.rankLast;
  for (int i = 0; i < 14; i++) rankLast[i] = none;
  {
    u32 cur = maxNbBits;
    for (int pos = n; pos >= 0; pos--) { if (nb[pos] >= cur) continue; cur = nb[pos]; rankLast[maxNbBits - cur] = (u32)pos; }
  }
  while (totalCost > 0) {
    u32 d = hb32((u32)totalCost) + 1;
    for (; d > 1; d--) {
      const u32 hp = rankLast[d], lp = rankLast[d - 1];
      if (hp == none) continue;
      if (lp == none) break;
      if (cnt[hp] <= 2 * cnt[lp]) break;
    }
    while (d <= 12 && rankLast[d] == none) d++;
    totalCost -= 1 << (d - 1);
    if (rankLast[d - 1] == none) rankLast[d - 1] = rankLast[d];
    nb[rankLast[d]]++;
    if (rankLast[d] == 0) rankLast[d] = none;
    else { rankLast[d]--; if (nb[rankLast[d]] != maxNbBits - d) rankLast[d] = none; }
  }
  while (totalCost < 0) {
    if (rankLast[1] == none) {
      while (nb[n] == maxNbBits) n--;
      nb[n + 1]--;
      rankLast[1] = (u32)(n + 1);
      totalCost++;
      continue;
    }
    nb[rankLast[1] + 1]--;
    rankLast[1]++;
    totalCost++;
  }
  return maxNbBits;
}

// ---- the Huffman build of a block's literals by ONE wave (round 6). Rounds 1-5 ran it on one lane: ~700 k cycles of dependent LDS round
// trips per 64 KiB frame, as long as the three sequence-state chains beside it. Only the two-queue merge is inherently serial (<= 255
// steps, the queue heads kept in registers); depths come from pointer jumping over the internal nodes, code values from a rank among
// the symbols of equal length, the weight header's two FSE state chains run on two lanes and their bits are packed by a prefix scan.
// Scratch of the wave in the upper part of the bit-staging tile (the sequence-table builds of waves 1-3 use its first 2.6 KiB meanwhile).
struct HufWaveScratch { u16 P[256]; u16 D[256]; u16 wOut[256]; u32 wStage[64]; u32 rankCnt[16]; };
static_assert(sizeof(EncSeqBuild) <= 4096 && 4096 + sizeof(HufWaveScratch) <= sizeof(u32) * STAGE_WORDS, "Huffman wave scratch must fit behind the table builds' arrays");

// FSE-compress the weight string (all lanes of one wave; wave-uniform result). 0 = not compressible, 1 = single symbol.
__device__ u32 huf_compress_weights_wave(EncShared& S, HufWaveScratch& H, u8* dst, u32 cap, const u8* w, u32 n, int lane) {
  u32* const count = S.lit.wCount; short* const norm = S.lit.wNorm;
  const u64 lt = (1ull << lane) - 1ull;
  if (n <= 1) return 0;
  if (lane < 16) count[lane] = 0;
  wave_order();
  for (u32 i = (u32)lane; i < n; i += 64) atomicAdd(&count[w[i]], 1u);
  wave_order();
  const u32 cMine = lane < 13 ? count[lane] : 0u;
  const u32 maxSym = wave_max(cMine ? (u32)lane : 0u), maxCount = wave_max(cMine);
  if (maxCount == n) return 1;
  if (maxCount == 1) return 0;
  const u32 t = fse_optimal_tablelog(6, n, maxSym, 2);
  u8* const tmp = S.lit.ncTmp;
  u32 h = 0;
  if (lane == 0) {
    if (fse_normalize(norm, t, count, n, maxSym, false) > 0) { h = fse_write_ncount(tmp, norm, maxSym, t); if (h > cap) h = 0; }
  }
  h = (u32)__builtin_amdgcn_readfirstlane((int)h);
  if (!h) return 0;
  wave_order();
  for (u32 i = (u32)lane; i < h; i += 64) dst[i] = tmp[i];
  ZraFseCTable* const ct = (ZraFseCTable*)((u8*)S.stage + ENT_WCT_OFF);
  if (fse_build_ctable_wave(ct, norm, maxSym, t, S.lit.wSpread, lane)) return 0;
  if (n <= 2) return 0;
  // two interleaved states, symbols consumed from the end (A.4.5 "weight serialisation"): state 1 owns the even positions, state 2 the odd
  // ones; each starts at its highest position and encodes downwards; the bits leave in position order, n - 3 first. Lane 0 / 1 walk them.
  u32 state = 0;
  if (lane < 2) {
    int j = (int)n - 1; if ((j & 1) != lane) j--;
    state = fse_init_state(ct, w[j]);
    for (j -= 2; j >= 0; j -= 2) { u32 bits; const u32 nb = fse_encode(ct, state, w[j], bits); H.wOut[j] = (u16)(bits | (nb << 8)); }
  }
  for (u32 i = (u32)lane; i < 64; i += 64) H.wStage[i] = 0;
  wave_order();
  u32 total = 0;
  for (u32 c = 0; c + 2 < n; c += 64) {
    const u32 r = c + (u32)lane;                         // r-th encoded symbol = position n - 3 - r
    const bool act = r + 2 < n;
    const u32 e = act ? (u32)H.wOut[n - 3 - r] : 0u, nb = e >> 8;
    const u32 inc = wave_incl_scan(nb);
    if (act && nb) {
      const u32 at = total + inc - nb;
      const u64 v = (u64)(e & 0xFFu) << (at & 31);
      atomicOr(&H.wStage[at >> 5], (u32)v);
      if (v >> 32) atomicOr(&H.wStage[(at >> 5) + 1], (u32)(v >> 32));
    }
    total += bcast(inc, 63);
  }
  const u32 s1 = bcast(state, 0), s2 = bcast(state, 1);
  if (lane == 0) {
    const u64 fin = (u64)(s2 & ((1u << t) - 1)) | ((u64)(s1 & ((1u << t) - 1)) << t) | (1ull << (2 * t));
    const u64 v = fin << (total & 31);                   // 2 t + 1 <= 13 bits
    atomicOr(&H.wStage[total >> 5], (u32)v);
    if (v >> 32) atomicOr(&H.wStage[(total >> 5) + 1], (u32)(v >> 32));
  }
  total += 2 * t + 1;
  const u32 nbytes = (total + 7) >> 3, pos = h + nbytes;
  if (pos > cap) return 0;
  wave_order();
  for (u32 i = (u32)lane; i < nbytes; i += 64) dst[h + i] = (u8)(H.wStage[i >> 2] >> (8 * (i & 3)));
  (void)lt;
  return pos;
}

// HUF_writeCTable into S.lit.hufHdr (all lanes of one wave; wave-uniform result); returns size, 0 on failure
__device__ u32 huf_write_ctable_wave(EncShared& S, HufWaveScratch& H, u32 maxSym, u32 log, int lane) {
  u8* const w = S.lit.weights; u8* const dst = S.lit.hufHdr;
  for (u32 n = (u32)lane; n < maxSym; n += 64) w[n] = S.hNb[n] ? (u8)(log + 1 - S.hNb[n]) : 0;
  wave_order();
  const u32 h = huf_compress_weights_wave(S, H, dst + 1, 190, w, maxSym, lane);
  if (h > 1 && h < maxSym / 2) { if (lane == 0) dst[0] = (u8)h; return h + 1; }
  if (maxSym > 128) return 0;
  if (lane == 0) { dst[0] = (u8)(128 + (maxSym - 1)); w[maxSym] = 0; }
  wave_order();
  for (u32 n = 2 * (u32)lane; n < maxSym; n += 128) dst[(n / 2) + 1] = (u8)((w[n] << 4) + w[n + 1]);
  return ((maxSym + 1) / 2) + 1;
}

// The tree, the code lengths (depth-limited to `log`), the codes and the table description of the block's literals, from the symbols
// sorted by (count desc, symbol asc) in S.lit.nodeCount[1 + rank] / nodeByte[rank]. All lanes of one wave; results: S.hNb, S.hVal,
// S.lit.hufHdr, S.sc[0] = code length limit in use, S.sc[1] = header bytes (0: failure).
__device__ void huf_build_wave(EncShared& S, u32 maxSym, u32 log, int lane) {
  HufWaveScratch& H = *(HufWaveScratch*)((u8*)S.stage + 4096);
  u32* const cntN = S.lit.nodeCount + 1; u16* const par = S.lit.nodeParent + 1; u8* const nbN = S.lit.nodeBits + 1;
  const u64 lt = (1ull << lane) - 1ull;
  u32 nn = 0;
  for (u32 k = (u32)lane; k <= maxSym; k += 64) if (cntN[k]) nn = k;      // (sorted by count: the zeros are at the end)
  const int nonNull = (int)wave_max(nn);
  const int START = 256, nodeRoot = START + nonNull - 1;
  for (int k = START + 1 + lane; k <= nodeRoot; k += 64) cntN[k] = 1u << 30;
  if (lane == 0) S.lit.nodeCount[0] = 1u << 31;
  wave_order();
  if (lane == 0) {
    // two-queue merge (leaves from the smallest, made nodes in the order they were made); the heads of both queues stay in registers
    int lowS = nonNull, lowN = START, nodeNb = START;
    const u32 c0 = cntN[lowS] + cntN[lowS - 1];
    cntN[nodeNb] = c0; par[lowS] = par[lowS - 1] = (u16)nodeNb;
    nodeNb++; lowS -= 2;
    u32 cS = cntN[lowS], cN = c0;
    while (nodeNb <= nodeRoot) {
      int n1, n2; u32 c1, c2;
      if (cS < cN) { n1 = lowS--; c1 = cS; cS = cntN[lowS]; } else { n1 = lowN++; c1 = cN; cN = cntN[lowN]; }
      if (cS < cN) { n2 = lowS--; c2 = cS; cS = cntN[lowS]; } else { n2 = lowN++; c2 = cN; cN = cntN[lowN]; }
      cntN[nodeNb] = c1 + c2;
      par[n1] = par[n2] = (u16)nodeNb;
      if (lowN == nodeNb) cN = c1 + c2;                  // (the head of the node queue is the node just made: its count was read as "not made yet")
      nodeNb++;
    }
  }
  wave_order();
  // depth of the internal nodes START + i, i < nonNull (root = the last): pointer jumping, 2^r ancestors per round
  const u32 nInt = (u32)nonNull;
  for (u32 i = (u32)lane; i < nInt; i += 64) { const bool root = i + 1 == nInt; H.P[i] = root ? (u16)i : (u16)(par[START + i] - START); H.D[i] = root ? 0 : 1; }
  wave_order();
  for (u32 r = 1; r < nInt; r <<= 1) {
    u32 nd[4], np[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { const u32 i = (u32)lane + 64u * j; if (i < nInt) { const u32 pp = H.P[i]; nd[j] = H.D[i] + H.D[pp]; np[j] = H.P[pp]; } else { nd[j] = 0; np[j] = 0; } }
    wave_order();
#pragma unroll
    for (int j = 0; j < 4; j++) { const u32 i = (u32)lane + 64u * j; if (i < nInt) { H.D[i] = (u16)nd[j]; H.P[i] = (u16)np[j]; } }
    wave_order();
  }
  for (int k = lane; k <= nonNull; k += 64) nbN[k] = (u8)(H.D[par[k] - START] + 1);
  wave_order();
  u32 maxBits = 0;
  if (lane == 0) maxBits = huf_set_max_height(S, (u32)nonNull, log);
  maxBits = (u32)__builtin_amdgcn_readfirstlane((int)maxBits);
  if (lane < 16) H.rankCnt[lane] = 0;
  wave_order();
  for (int k = lane; k <= nonNull; k += 64) atomicAdd(&H.rankCnt[nbN[k]], 1u);
  for (u32 k = (u32)lane; k < 256; k += 64) S.hNb[k] = 0;
  wave_order();
  u16* const valPerRank = S.lit.valPerRank;
  if (lane == 0) {
    for (int k = 0; k < 14; k++) valPerRank[k] = 0;
    u32 mn = 0;
    for (int k = (int)maxBits; k > 0; k--) { valPerRank[k] = (u16)mn; mn += H.rankCnt[k]; mn >>= 1; }
  }
  for (u32 k = (u32)lane; k <= maxSym; k += 64) S.hNb[S.lit.nodeByte[k]] = k <= (u32)nonNull ? nbN[k] : 0;
  wave_order();
  // code of symbol k = first code of its length + its rank among the symbols of that length, in symbol order
  for (u32 c = 0; c <= maxSym; c += 64) {
    const u32 sym = c + (u32)lane;
    const bool act = sym <= maxSym;
    const u32 len = act ? (u32)S.hNb[sym] : 0xFFu;
    u64 rem = __ballot(act);
    while (rem) {
      const u32 l = (u32)__builtin_ctzll(rem), lv = bcast(len, l);
      const u64 m = __ballot(act && len == lv);
      const u32 base = valPerRank[lv];
      if (act && len == lv) S.hVal[sym] = (u16)(base + (u32)__popcll(m & lt));
      wave_order();
      if ((u32)lane == l) valPerRank[lv] = (u16)(base + (u32)__popcll(m));
      wave_order();
      rem &= ~m;
    }
  }
  wave_order();
  const u32 hs = huf_write_ctable_wave(S, H, maxSym, maxBits, lane);
  if (lane == 0) { S.sc[0] = maxBits; S.sc[1] = hs; }
}

// Encode `len` literal symbols (reverse order, A.4.5) as one Huffman stream into dst; all 256 threads. Returns bytes.
__device__ u32 huf_encode_stream(EncShared& S, const u8* lit, u32 len, const u8* nbTab, const u16* valTab, u8* dst) {
  u32 carryBits = 0, carryVal = 0, bytesOut = 0;
  for (u32 t0 = 0; t0 < len; t0 += SYM_TILE) {
    const u32 cntT = min((u32)SYM_TILE, len - t0);
    const bool lastTile = t0 + cntT == len;
    const u32 zw = ((7 + cntT * 11 + 1) >> 5) + 2;
    for (u32 i = threadIdx.x; i < zw; i += ENT_THREADS) S.stage[i] = (i == 0) ? carryVal : 0;
    u8 sym[16]; u32 nb = 0;
    const u32 r0 = t0 + 16 * threadIdx.x;
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const u32 r = r0 + k;
      sym[k] = r < t0 + cntT ? lit[len - 1 - r] : 0;
      nb += r < t0 + cntT ? nbTab[sym[k]] : 0;
    }
    u32 tot;
    const u32 ex = block_excl_scan(S, nb, &tot);      // contains the barriers that order the stage clear
    LaneBitW bw; bw.init(S.stage, carryBits + ex);
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const u32 r = r0 + k;
      if (r < t0 + cntT) bw.add(valTab[sym[k]], nbTab[sym[k]]);
    }
    bw.finish();
    u32 total = carryBits + tot;
    if (lastTile) { if (threadIdx.x == 0) atomicOr(&S.stage[total >> 5], 1u << (total & 31)); total += 1; }
    __syncthreads();
    const u32 nbytes = lastTile ? (total + 7) >> 3 : total >> 3;
    flush_stage(S, dst + bytesOut, nbytes);
    carryBits = lastTile ? 0 : (total & 7);
    carryVal = (S.stage[nbytes >> 2] >> (8 * (nbytes & 3))) & ((1u << carryBits) - 1);
    bytesOut += nbytes;
    __syncthreads();
  }
  return bytesOut;
}

// =============================================================================== sequence-table selection (A.4.7), one lane
__device__ u32 cross_entropy_cost(const short* norm, u32 accLog, const u32* count, u32 max) {
  const u32 shift = 8 - accLog; u64 cost = 0;
  for (u32 s = 0; s <= max; s++) { const u32 na = norm[s] != -1 ? (u32)norm[s] : 1; cost += (u64)count[s] * c_invProb[na << shift]; }
  return (u32)(cost >> 8);
}
__device__ u32 entropy_cost(const u32* count, u32 max, u32 total) {
  u32 cost = 0;
  for (u32 s = 0; s <= max; s++) {
    u32 norm = (256 * count[s]) / total;
    if (count[s] != 0 && norm == 0) norm = 1;
    cost += count[s] * c_invProb[norm];
  }
  return cost >> 8;
}
constexpr u32 COST_ERR = 0xFFFFFFFFu;
__device__ u32 fse_bit_cost(const ZraFseCTable* ct, const u32* count, u32 max) {
  if (ct->rle || ct->maxSym < max) return COST_ERR;
  const u32 tl = ct->tableLog; u64 cost = 0;
  for (u32 s = 0; s <= max; s++) {
    const u32 badCost = (tl + 1) << 8;
    const u32 minNb = ct->deltaNbBits[s] >> 16, thr = (minNb + 1) << 16;
    const u32 d = thr - (ct->deltaNbBits[s] + (1u << tl));
    const u32 bitCost = (minNb + 1) * 256 - ((d << 8) >> tl);
    if (!count[s]) continue;
    if (bitCost >= badCost) return COST_ERR;
    cost += (u64)count[s] * bitCost;
  }
  return (u32)(cost >> 8);
}

// mode: 0 predefined, 1 rle, 2 compressed, 3 repeat
__device__ u32 select_encoding(u32* repeatMode, const u32* count, u32 max, u32 mostFrequent, u32 nbSeq, u32 FSELog, const ZraFseCTable* prevCT,
                               const short* defNorm, u32 defLog, bool defaultAllowed, u32 strategy, short* normScratch, u8* ncScratch) {
  if (mostFrequent == nbSeq) { *repeatMode = 0; return (defaultAllowed && nbSeq <= 2) ? 0 : 1; }
  if (strategy < 4) {
    if (defaultAllowed) {
      const u32 mult = 10 - strategy, dynMin = ((1u << defLog) * mult) >> 3;
      if (*repeatMode == 2 && nbSeq < 1000) return 3;
      if (nbSeq < dynMin || mostFrequent < (nbSeq >> (defLog - 1))) { *repeatMode = 0; return 0; }
    }
  } else {
    const u64 INF = ~0ull;
    u64 basic = defaultAllowed ? cross_entropy_cost(defNorm, defLog, count, max) : INF;
    u64 repeat = INF;
    if (*repeatMode != 0) { u32 c = fse_bit_cost(prevCT, count, max); repeat = c == COST_ERR ? INF : c; }
    u64 nc = INF;
    {
      const u32 tl = fse_optimal_tablelog(FSELog, nbSeq, max, 2);
      if (fse_normalize(normScratch, tl, count, nbSeq, max, nbSeq >= 2048) > 0) {
        const u32 r = fse_write_ncount(ncScratch, normScratch, max, tl);
        if (r) nc = r;
      }
    }
    const u64 compressed = nc == INF ? INF : (nc << 3) + entropy_cost(count, max, nbSeq);
    if (basic <= repeat && basic <= compressed) { *repeatMode = 0; return 0; }
    if (repeat <= compressed) return 3;
  }
  *repeatMode = 1;
  return 2;
}

}  // namespace

// =================================================================================================
#ifdef ZRA_MF_PROFILE
__device__ unsigned long long zra_ent_prof[16];
#define EPROF(k) { __syncthreads(); if (threadIdx.x == 0) { __builtin_amdgcn_s_waitcnt(0); const u64 n_ = __builtin_amdgcn_s_memtime(); atomicAdd(&zra_ent_prof[k], n_ - ept_); ept_ = n_; } }
#else
#define EPROF(k)
#endif

// Register budget: beside the persistent match finder (112 KiB of a CU's LDS) two workgroups of this kernel fit on a CU, so nothing is
// gained by squeezing it under 72 VGPRs for 7 waves per SIMD (that cost 8 spilled VGPRs and 72 B of scratch in the one-lane sections).
#ifndef ZRA_ENT_WAVES
#define ZRA_ENT_WAVES 5
#endif
// One block of frame `f` of the batch: literals to `lits` (this workgroup's scratch), the encoded block to `slot`.
// `work`: this workgroup's scratch for the sequence section — code bytes [3][seqStride] (last sequence first), then the chains' output
// [3][seqStride] u16 (state bits << 0 | their count << 12)
// (the argument block is read where the kernel received it, in the constant address space: scalar loads, nothing copied into private memory)
typedef const __attribute__((address_space(4))) ZraEncArgs KArgs;
// PHASE (round 6, the split stage of the persistent pipeline): 0 = the whole block in one go; 1 = FRONT: everything up to the sequence
// section's table descriptions — literals section emitted, the three encoding tables handed to zra_ent_chain_kernel through the frame's
// record `rec`; 2 = BACK: the sequence bitstream from the chain kernel's output, block header, frame end. The state chains — three lanes
// of one wave for three quarters of a frame's time when they ran in here — are walked by zra_ent_chain_kernel between the two, lane =
// (frame, stream) over several frames per wave.
template <int PHASE>
__device__ __forceinline__ void entropy_frame(KArgs& a, u32 block, u32 f, u8* lits, u8* slot, u8* work, EncShared& S, ZraEntRec* rec) {
#ifdef ZRA_MF_PROFILE
  u64 ept_ = __builtin_amdgcn_s_memtime();
#endif
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const u64 fstart = (u64)(a.firstFrame + f) * a.frameSize;
  const u64 remaining = a.inSize - fstart;
  const u32 fsize = (u32)(remaining < a.frameSize ? remaining : a.frameSize);
  const bool fullP = fsize == a.frameSize;
  struct { u32 blockSize, strategy, windowLog, targetLength; } P;      // (field by field: the block sits in the constant address space)
  P.blockSize = fullP ? a.full.blockSize : a.tail.blockSize; P.strategy = fullP ? a.full.strategy : a.tail.strategy;
  P.windowLog = fullP ? a.full.windowLog : a.tail.windowLog; P.targetLength = fullP ? a.full.targetLength : a.tail.targetLength;
  const u32 bs = block * P.blockSize;
  if (bs >= fsize) return;
  const u32 be = min(fsize, bs + P.blockSize), L = be - bs;
  const bool first = block == 0, last = be == fsize;
  const u8* src = a.in + fstart;
  ZraEncFrameState* st = &a.state[f];
  const ZraEncBlockOut* bo = &a.blockOut[f];
  const u32 strategy = P.strategy;

  if (PHASE != 2 && first && tid == 0) {
    st32(slot, 0xFD2FB528u); slot[4] = a.checksum ? 4 : 0; slot[5] = (u8)((P.windowLog - 10) << 3);   // A.4.2 frame header
    st->outPos = 6; st->hufRepeat = 0; st->llRepeat = st->ofRepeat = st->mlRepeat = 0;
  }
  __syncthreads();
  const u32 outPos = first ? 6 : st->outPos;
  u8* const op0 = slot + outPos;          // 3-byte block header goes here
  u8* const blk = op0 + 3;                // block content
  u32 cSize = 0;
  bool newHuf = false; u32 newHufLog = 0, newHufMaxSym = 0;
  const u32 nbSeq = bo->nbSeq;

  if (!bo->skip) {
    const u64* seqs = a.seqs + (size_t)f * a.seqStride;

    u8* const codesG = work; u16* const chainG = (u16*)(work + 3 * a.seqStride);
    u8* op = blk; bool tblErr = false; u8* lastNCount = nullptr; u32 tlog[3] = {0, 0, 0}, fin[3] = {0, 0, 0};
    if constexpr (PHASE != 2) {
    // ------------------------------------------------------------ phase 1: gather literals + histogram
    for (int i = tid; i < 4 * 256; i += ENT_THREADS) (&S.lit.hist[0][0])[i] = 0;
    if (tid < 3 * 64) (&S.seqCnt[0][0])[tid] = 0;
    if (tid < 4) S.tblReady[tid] = 0;
    u32 litBase = 0, srcBase = bs;
    __syncthreads();
    for (u32 t0 = 0; t0 < nbSeq; t0 += SEQ_TILE) {
      u32 llv[SEQ_PER], mlv[SEQ_PER], tl = 0, tt = 0;
#pragma unroll
      for (int k = 0; k < SEQ_PER; k++) {
        const u32 i = t0 + SEQ_PER * tid + k;
        const u64 q = i < nbSeq ? seqs[i] : 0;
        llv[k] = (u32)q & 0xFFFFF; mlv[k] = (u32)(q >> 20) & 0xFFFFF;
        tl += llv[k]; tt += llv[k] + mlv[k];
        if (i < nbSeq) {                               // codes + code histograms of the sequence section (A.4.7): the sequence is in registers here
          const u32 cl = ll_code(llv[k]), co = hb32((u32)(q >> 40)), cm = ml_code(mlv[k] - 3);
          atomicAdd(&S.seqCnt[0][cl], 1u); atomicAdd(&S.seqCnt[1][co], 1u); atomicAdd(&S.seqCnt[2][cm], 1u);
          const u32 rl = nbSeq - 1 - i;                  // the chains run from the block's last sequence to its first
          codesG[rl] = (u8)cl; codesG[a.seqStride + rl] = (u8)co; codesG[2 * a.seqStride + rl] = (u8)cm;
        }
      }
      if (tid == 0) S.longCount = 0;
      u32 totL, totT;
      const u32 exL = block_excl_scan(S, tl, &totL);
      const u32 exT = block_excl_scan(S, tt, &totT);
      u32 lp = litBase + exL, sp = srcBase + exT;
#pragma unroll
      for (int k = 0; k < SEQ_PER; k++) {
        const u32 ll = llv[k];
        if (ll >= 32) {                              // long run: queue for the whole workgroup
          const u32 e = atomicAdd(&S.longCount, 1u);
          S.stage[3 * e] = lp; S.stage[3 * e + 1] = sp; S.stage[3 * e + 2] = ll;
        } else {
          // up to 31 bytes: four independent 8-byte loads in flight instead of a byte-by-byte chain of global round trips
          // (under the match finder's DRAM load a round trip costs microseconds); stores and histogram come from registers
          const u8* const se = src + fsize;
          u64 w[4];
#pragma unroll
          for (int q = 0; q < 4; q++) w[q] = 8u * q < ll ? ld64_safe(src + sp + 8 * q, se) : 0;
#pragma unroll
          for (int q = 0; q < 4; q++) {
            for (u32 b = 8u * q; b < ll && b < 8u * q + 8; b++) { const u8 c = (u8)(w[q] >> (8 * (b & 7))); lits[lp + b] = c; atomicAdd(&S.lit.hist[wave][c], 1u); }
          }
        }
        lp += ll; sp += ll + mlv[k];
      }
      __syncthreads();
      const u32 nLong = S.longCount;
      for (u32 e = (u32)wave; e < nLong; e += ENT_THREADS / 64) {      // a run per wave: four runs' round trips side by side
        const u32 lp2 = S.stage[3 * e], sp2 = S.stage[3 * e + 1], ll2 = S.stage[3 * e + 2];
        for (u32 b = (u32)lane; b < ll2; b += 64) { const u8 c = src[sp2 + b]; lits[lp2 + b] = c; atomicAdd(&S.lit.hist[wave][c], 1u); }
      }
      litBase += totL; srcBase += totT;
      __syncthreads();
    }
    const u32 lastLL = bo->lastLL;
    for (u32 b = tid; b < lastLL; b += ENT_THREADS) { const u8 c = src[be - lastLL + b]; lits[litBase + b] = c; atomicAdd(&S.lit.hist[wave][c], 1u); }
    const u32 n = litBase + lastLL;           // literal count of the block
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    EPROF(0)
    // (the three sequence tables are chosen and built on waves 1-3 WHILE wave 0 builds the Huffman tree — both are one-lane sections,
    // 28 % and 17 % of this kernel's time when they ran one after the other, and they share nothing; their work arrays sit in the staging tile)
    EncSeqBuild& SB = *(EncSeqBuild*)S.stage;
    EPROF(3)
    // ------------------------------------------------------------ phase 2: literals section (A.4.5)
    u32 litSec = 0;
    {
      // mode: 0 raw, 1 rle, 2 compressed (new table), 3 treeless (previous table)
      u32 mode = 0, hSize = 0, streams = 1, encSize = 0;
      u32 ssz[4] = {0, 0, 0, 0};
      const u8* nbTab = S.hNb; const u16* valTab = S.hVal;
      // ---- 2a (all threads): symbol counts, the cheap exits, the sort that the tree build starts from
      const bool litTry = n > 63 && !(strategy == 1 && P.targetLength > 0);      // fast strategy with an acceleration (negative levels): literals stay raw
      u32 cntS = 0, maxSym = 0, c = 0, repeat = 0, log = 0;   // c: compressed size candidate; 0 = not compressible
      bool useOld = false, candidate = false, needTree = false, fail = false;
      if (litTry) {
        cntS = S.lit.hist[0][tid] + S.lit.hist[1][tid] + S.lit.hist[2][tid] + S.lit.hist[3][tid];
        __syncthreads();
        S.lit.hist[0][tid] = cntS;
        const u32 largest = block_max(S, cntS);
        maxSym = block_max(S, cntS ? (u32)tid : 0u);
        if (largest == n) c = 1;
        else if (largest > (n >> 7) + 4) {
          candidate = true;
          repeat = first ? 0u : st->hufRepeat;
          if (repeat == 1) {
            const int bad = __syncthreads_or((u32)tid <= maxSym && cntS != 0 && st->hufNbBits[tid] == 0);
            if (bad) repeat = 0;
          }
          const bool preferRepeat = strategy < 4 && n <= 1024;
          if (preferRepeat && repeat != 0) useOld = true;
          else {
            // ---- new tree: sort by (count desc, symbol asc); the merge, the depth limit and the codes follow on wave 0 below
            needTree = true;
            log = fse_optimal_tablelog(11, n, maxSym, 1);
            if ((u32)tid <= maxSym) {
              u32 rank = 0;
              for (u32 t = 0; t <= maxSym; t++) { const u32 ct = S.lit.hist[0][t]; rank += (ct > cntS) || (ct == cntS && t < (u32)tid); }
              S.lit.nodeCount[1 + rank] = cntS; S.lit.nodeByte[rank] = (u8)tid;
            }
          }
        }
      }
      __syncthreads();
      // ---- 2b: wave 0 lane 0 builds the Huffman tree; waves 1-3 lane 0 choose and build the sequence tables (stream = wave - 1)
      if (wave == 0) {
        if (needTree) huf_build_wave(S, maxSym, log, lane);
      } else if (nbSeq) {
        // ---- wave 1: the three sequence tables and the three state chains, stream k on lane k. ONE wave (beside wave 0's tree build),
        // not three: this stage runs beside the match finder, and every wave that issues one-lane code at a raised priority takes issue
        // slots from it (round 5: with a wave per stream the finder lost 10-25 %, with round 4's single wave nothing measurable)
        // (the three tables are chosen and built side by side, stream k by lane 0 of wave k + 1 — short — and reported through an LDS
        //  flag; the three chains then run on lanes 0-2 of wave 1 alone)
        const int k = wave == 1 ? lane : wave - 1;
        {
          // (round 6: lane 0 chooses the encoding and normalises the counts — short —, the whole wave builds the table: fse_build_ctable_wave)
          const int kw = wave - 1;
          u32 doBuild = 0, buildT = 0, buildMax = 0;
          if (lane == 0) {
            const int k = kw;
            const u32 maxSymK = k == 0 ? 35 : k == 1 ? 31 : 52, FSELog = k == 1 ? 8 : 9, defLog = k == 1 ? 5 : 6, defMax = k == 0 ? 35 : k == 1 ? 28 : 52;
            const short* defNorm = k == 0 ? c_LLdef : k == 1 ? c_OFdef : c_MLdef;
            u32* count = S.seqCnt[k];
            u32 mx = 0, most = 0;
            for (u32 sy = 0; sy <= maxSymK; sy++) { if (count[sy]) mx = sy; if (count[sy] > most) most = count[sy]; }
            const u32 lastCode = codesG[(size_t)k * a.seqStride];          // (the block's last sequence is the chains' first)
            const ZraFseCTable* prevCT = k == 0 ? &st->ll : k == 1 ? &st->of : &st->ml;
            u32 repeatMode = first ? 0 : (k == 0 ? st->llRepeat : k == 1 ? st->ofRepeat : st->mlRepeat);
            const bool defaultAllowed = k != 1 || mx <= 28;
            const u32 modeK = select_encoding(&repeatMode, count, mx, most, nbSeq, FSELog, prevCT, defNorm, defLog, defaultAllowed, strategy, SB.norm[k], S.ncount[k]);
            S.mode[k] = modeK; S.nextRepeat[k] = repeatMode; S.ncountSize[k] = 0; S.tblErr[k] = 0;
            ZraFseCTable* ct = &S.lit.ct[k];
            if (modeK == 1) { ct->rle = 1; ct->tableLog = 0; ct->maxSym = mx; S.ncount[k][0] = (u8)mx; S.ncountSize[k] = 1; }
            else if (modeK == 0) { for (u32 sy = 0; sy <= defMax; sy++) SB.norm[k][sy] = defNorm[sy]; doBuild = 1; buildT = defLog; buildMax = defMax; }
            else if (modeK == 2) {
              u32 n1 = nbSeq;
              const u32 tl = fse_optimal_tablelog(FSELog, nbSeq, mx, 2);
              if (count[lastCode] > 1) { count[lastCode]--; n1--; }
              if (fse_normalize(SB.norm[k], tl, count, n1, mx, n1 >= 2048) <= 0) S.tblErr[k] = 1;
              else {
                const u32 h = fse_write_ncount(S.ncount[k], SB.norm[k], mx, tl);
                if (!h) S.tblErr[k] = 1; else { doBuild = 1; buildT = tl; buildMax = mx; }
                S.ncountSize[k] = h;
              }
            }
          }
          doBuild = (u32)__builtin_amdgcn_readfirstlane((int)doBuild); buildT = (u32)__builtin_amdgcn_readfirstlane((int)buildT);
          buildMax = (u32)__builtin_amdgcn_readfirstlane((int)buildMax);
          wave_order();
          if (doBuild && fse_build_ctable_wave(&S.lit.ct[kw], SB.norm[kw], buildMax, buildT, SB.spread[kw], lane)) { if (lane == 0) S.tblErr[kw] = 1; }
          if (lane == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __hip_atomic_store(&S.tblReady[kw], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
        if (wave == 1) {
        // wave 1 waits for the three tables (its own lane 0 built the first)
        while (__hip_atomic_load(&S.tblReady[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) + __hip_atomic_load(&S.tblReady[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) +
               __hip_atomic_load(&S.tblReady[2], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != 3u) __builtin_amdgcn_s_sleep(4);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");            // (the tables and decisions, for every lane of this wave)
        __builtin_amdgcn_wave_barrier();
        for (int kk = 0; kk < 3; kk++) {
          if (S.mode[kk] == 3) {
            // repeat mode: the previous block's table comes into LDS
            const u32* srcT = (const u32*)(kk == 0 ? &st->ll : kk == 1 ? &st->of : &st->ml);
            u32* dstT = (u32*)&S.lit.ct[kk];
            for (u32 i = (u32)lane; i < sizeof(ZraFseCTable) / 4; i += 64) dstT[i] = srcT[i];
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        // ---- the state chains over the whole block: state -> stateTable[(state >> nb) + dfs] -> state, one LDS round trip per sequence,
        // inherently serial (a step over a probable symbol maps a state next to itself: chains from different start states do not merge,
        // and a table kept in registers and walked with v_readlane costs more issue slots than the round trip — both measured in round
        // 5). It ran per tile with everybody else waiting: 35-47 % of this kernel's time. Now it runs HERE, beside the Huffman tree build
        // of wave 0, which takes about as long. 64 sequences at a time: their codes come in with one load per stream (the next 64
        // travel meanwhile) and are parked in LDS, lanes 0-2 walk the 64 steps of their streams (the symbol two steps ahead and its
        // parameters one step ahead are fetched beside the critical load), the results leave with one store per stream.
        if constexpr (PHASE != 1) {
          const bool runK = lane < 3 && !S.tblErr[k] && !S.lit.ct[lane < 3 ? k : 0].rle;
          const ZraFseCTable* const ct = &S.lit.ct[lane < 3 ? k : 0];
          u8 (*const chainCodes)[64] = (u8 (*)[64])S.stage; u16 (*const chainOut)[64] = (u16 (*)[64])((u8*)S.stage + 192);
          u8* const cdL = chainCodes[lane < 3 ? k : 0]; u16* const outL = chainOut[lane < 3 ? k : 0];
          u32 cV[3], cN[3] = {0, 0, 0};
#pragma unroll
          for (int kk = 0; kk < 3; kk++) cV[kk] = (u32)lane < nbSeq ? codesG[(size_t)kk * a.seqStride + lane] : 0u;
          u32 state = 0;
          for (u32 base = 0; base < nbSeq; base += 64) {
            const u32 nIn = min(64u, nbSeq - base);
#pragma unroll
            for (int kk = 0; kk < 3; kk++) {
              if (base + 64 < nbSeq) cN[kk] = base + 64 + (u32)lane < nbSeq ? codesG[(size_t)kk * a.seqStride + base + 64 + lane] : 0u;
              chainCodes[kk][lane] = (u8)cV[kk];
            }
            // (one wave, LDS in issue order: a wavefront-scope fence orders the compiler and waits for nothing — a workgroup-scope one
            //  would wait for the codes in flight and for the stores of the 64 results before)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (runK) {
              u32 i = 0;
              if (base == 0) { state = fse_init_state(ct, cdL[0]); outL[0] = 0; i = 1; }     // the block's last sequence: init only
              if (i < nIn) {
                const u32 last = nIn - 1;
                u32 sym1 = cdL[min(i + 1, last)];
                u32 dnb = ct->deltaNbBits[cdL[i]]; i32 dfs = ct->deltaFindState[cdL[i]];
                for (; i < nIn; i++) {
                  const u32 nb = (state + dnb) >> 16;
                  const u32 bits = state & ((1u << nb) - 1);
                  state = ct->stateTable[(state >> nb) + dfs];                              // critical load first
                  const u32 dnbN = ct->deltaNbBits[sym1]; const i32 dfsN = ct->deltaFindState[sym1];   // parameters of step i+1
                  const u32 sym2 = cdL[min(i + 2, last)];                                  // symbol of step i+2
                  outL[i] = (u16)((nb << 12) | bits);
                  dnb = dnbN; dfs = dfsN; sym1 = sym2;
                }
              }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int kk = 0; kk < 3; kk++) {
              // (a one-symbol table emits no state bits; after a table error the section is dropped: zeros either way)
              if ((u32)lane < nIn) chainG[(size_t)kk * a.seqStride + base + lane] = (S.lit.ct[kk].rle || S.tblErr[kk]) ? (u16)0 : chainOut[kk][lane];
              cV[kk] = cN[kk];
            }
          }
          if (lane < 3) S.finalState[k] = runK ? state : 0u;
        }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // (the chains' output in the work area, for every wave of the workgroup)
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      EPROF(4)
      // ---- 2c (all threads): old table or new, stream sizes, the section's mode
      if (candidate) {
        if (needTree) {
          newHufLog = S.sc[0]; hSize = S.sc[1]; newHufMaxSym = maxSym;
          if (!hSize) fail = true;
          else if (repeat != 0) {
            const u32 oldSize = block_sum(S, (u32)tid <= maxSym ? (u32)st->hufNbBits[tid] * cntS : 0u) >> 3;
            const u32 newSize = block_sum(S, (u32)tid <= maxSym ? (u32)S.hNb[tid] * cntS : 0u) >> 3;
            if (oldSize <= hSize + newSize || hSize + 12 >= n) useOld = true;
          }
          if (!fail && !useOld && hSize + 12 >= n) fail = true;   // "return 0": not compressible
        }
        if (!fail) {
          if (useOld) {
            // previous block's table -> LDS copies so both paths read the same arrays
            S.hNb[tid] = st->hufNbBits[tid]; S.hVal[tid] = st->hufVal[tid];
            hSize = 0;
            __syncthreads();
          }
          streams = n < 256 ? 1 : 4;
          // stream sizes from code lengths (prefix work is cheap: one pass over the literals)
          const u32 seg = streams == 1 ? n : (n + 3) / 4;
          u32 bits[4] = {0, 0, 0, 0};
          // (16 literals per load: one round trip per 4 KiB of literals instead of one per 256 bytes)
          for (u32 i0 = 16u * tid; i0 < n; i0 += 16u * ENT_THREADS) {
            const uint4 v = *(const uint4*)(lits + i0);
            const u32 w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (u32 b = 0; b < 16; b++) {
              const u32 i = i0 + b;
              if (i < n) { const u32 j = streams == 1 ? 0 : min(i / seg, 3u); bits[j] += nbTab[(w4[b >> 2] >> (8 * (b & 3))) & 255u]; }
            }
          }
          for (u32 j = 0; j < streams; j++) ssz[j] = (block_sum(S, bits[j]) + 1 + 7) >> 3;
          encSize = streams == 1 ? ssz[0] : 6 + ssz[0] + ssz[1] + ssz[2] + ssz[3];
          c = hSize + encSize;
          if (c >= n - 1) c = 0;
        }
      }
      if (litTry) {
        const u32 minGain = (n >> (strategy >= 8 ? strategy - 1 : 6)) + 2;
        if (c == 0 || c >= n - minGain) mode = 0;
        else if (c == 1) mode = 1;
        else mode = useOld ? 3 : 2;
      }
      EPROF(1)
      // ---- emit
      if (mode < 2) {
        const u32 fl = 1 + (n > 31) + (n > 4095);
        if (tid == 0) {
          if (fl == 1) blk[0] = (u8)(mode + (n << 3));
          else if (fl == 2) { const u32 v = mode + (1 << 2) + (n << 4); blk[0] = (u8)v; blk[1] = (u8)(v >> 8); }
          else st32(blk, mode + (3 << 2) + (n << 4));       // 4th byte is overwritten by the payload below
        }
        __syncthreads();
        if (mode == 1) { if (tid == 0) blk[fl] = lits[0]; litSec = fl + 1; }
        else { for (u32 i = tid; i < n; i += ENT_THREADS) blk[fl + i] = lits[i]; litSec = fl + n; }
      } else {
        const u32 lh = 3 + (n >= 1024) + (n >= 16384);
        const u32 c = hSize + encSize;
        if (tid == 0) {
          if (lh == 3) { const u32 v = mode + ((streams == 4) << 2) + (n << 4) + (c << 14); blk[0] = (u8)v; blk[1] = (u8)(v >> 8); blk[2] = (u8)(v >> 16); }
          else if (lh == 4) st32(blk, mode + (2 << 2) + (n << 4) + (c << 18));
          else { st32(blk, mode + (3 << 2) + (n << 4) + (c << 22)); blk[4] = (u8)(c >> 10); }
        }
        for (u32 i = tid; i < hSize; i += ENT_THREADS) blk[lh + i] = S.lit.hufHdr[i];
        u8* sp = blk + lh + hSize;
        if (streams == 4) {
          if (tid == 0) { sp[0] = (u8)ssz[0]; sp[1] = (u8)(ssz[0] >> 8); sp[2] = (u8)ssz[1]; sp[3] = (u8)(ssz[1] >> 8); sp[4] = (u8)ssz[2]; sp[5] = (u8)(ssz[2] >> 8); }
          sp += 6;
          const u32 seg = (n + 3) / 4;
          for (u32 j = 0; j < 4; j++) {
            const u32 len = j < 3 ? seg : n - 3 * seg;
            huf_encode_stream(S, lits + j * seg, len, nbTab, valTab, sp);
            sp += ssz[j];
          }
        } else huf_encode_stream(S, lits, n, nbTab, valTab, sp);
        litSec = lh + c;
        newHuf = mode == 2;
      }
    }
    __syncthreads();

    EPROF(2)
    // ------------------------------------------------------------ phase 3: sequences section (A.4.7)
    op = blk + litSec;
    if (tid == 0) {
      if (nbSeq < 128) op[0] = (u8)nbSeq;
      else if (nbSeq < 0x7F00) { op[0] = (u8)((nbSeq >> 8) + 0x80); op[1] = (u8)nbSeq; }
      else { op[0] = 0xFF; op[1] = (u8)(nbSeq - 0x7F00); op[2] = (u8)((nbSeq - 0x7F00) >> 8); }
    }
    op += nbSeq < 128 ? 1 : nbSeq < 0x7F00 ? 2 : 3;
    if (nbSeq) {
      u8* const seqHead = op++;
      tblErr = S.tblErr[0] | S.tblErr[1] | S.tblErr[2];
      if (tid == 0) seqHead[0] = (u8)((S.mode[0] << 6) + (S.mode[1] << 4) + (S.mode[2] << 2));
      for (int k = 0; k < 3; k++) {
        const u32 sz = S.ncountSize[k];
        if (S.mode[k] == 2) lastNCount = op;
        for (u32 i = tid; i < sz; i += ENT_THREADS) op[i] = S.ncount[k][i];
        op += sz;
      }
      for (int k = 0; k < 3; k++) { tlog[k] = S.lit.ct[k].tableLog; fin[k] = S.finalState[k]; }
    }
    if constexpr (PHASE == 1) {
      // FRONT ends here: the tables and where the bitstream goes, for the chain kernel and the BACK launch
      if (nbSeq && !tblErr) {
        for (int k = 0; k < 3; k++) {
          const u32* srcT = (const u32*)&S.lit.ct[k]; u32* dstT = (u32*)&rec->ct[k];
          for (u32 i = tid; i < sizeof(ZraFseCTable) / 4; i += ENT_THREADS) dstT[i] = srcT[i];
        }
      }
      if (tid == 0) {
        rec->nChain = (nbSeq && !tblErr) ? nbSeq : 0u;
        for (int k = 0; k < 3; k++) { rec->run[k] = (nbSeq && !tblErr && !S.lit.ct[k].rle) ? 1u : 0u; rec->tlog[k] = tlog[k]; rec->finalState[k] = 0; }
        rec->tblErr = tblErr ? 1u : 0u; rec->opOff = (u32)(op - blk); rec->lastNCountOff = lastNCount ? (u32)(lastNCount - blk) : 0xFFFFFFFFu;
        rec->newHuf = newHuf ? 1u : 0u; rec->newHufMaxSym = newHufMaxSym;
      }
      return;
    }
    } else {
      // BACK: where FRONT stopped
      op = blk + rec->opOff; tblErr = rec->tblErr != 0;
      lastNCount = rec->lastNCountOff == 0xFFFFFFFFu ? nullptr : blk + rec->lastNCountOff;
      for (int k = 0; k < 3; k++) { tlog[k] = rec->tlog[k]; fin[k] = rec->finalState[k]; }
      newHuf = rec->newHuf != 0; newHufMaxSym = rec->newHufMaxSym;
    }
    bool uncompressible = false;
    if (nbSeq) {
      // ---- pass B: parallel packing of the chains' output + the sequences' extra bits, 512 sequences per tile, last sequence first
      u32 carryBits = 0, carryVal = 0, bytesOut = 0;
      for (u32 t0 = 0; t0 < nbSeq && !tblErr; t0 += SEQ_TILE) {
        const u32 cntT = min((u32)SEQ_TILE, nbSeq - t0);
        const bool lastTile = t0 + cntT == nbSeq;
        const u32 zw = ((7 + cntT * 90 + 28) >> 5) + 2;
        for (u32 i = tid; i < zw; i += ENT_THREADS) S.stage[i] = (i == 0) ? carryVal : 0;
        u32 llv[SEQ_PER], mlb[SEQ_PER], ofv[SEQ_PER], cL[SEQ_PER], cO[SEQ_PER], cM[SEQ_PER];
#pragma unroll
        for (int k = 0; k < SEQ_PER; k++) {
          const u32 rl = SEQ_PER * tid + k;                 // reversed local index
          if (rl < cntT) {
            const u64 q = seqs[nbSeq - 1 - (t0 + rl)];
            llv[k] = (u32)q & 0xFFFFF; mlb[k] = ((u32)(q >> 20) & 0xFFFFF) - 3; ofv[k] = (u32)(q >> 40);
            cL[k] = chainG[t0 + rl]; cO[k] = chainG[a.seqStride + t0 + rl]; cM[k] = chainG[2 * a.seqStride + t0 + rl];
          } else { llv[k] = mlb[k] = ofv[k] = cL[k] = cO[k] = cM[k] = 0; }
        }
        EPROF(5)
        u32 nbits = 0;
#pragma unroll
        for (int k = 0; k < SEQ_PER; k++) {
          const u32 rl = SEQ_PER * tid + k;
          if (rl < cntT) nbits += (cL[k] >> 12) + (cO[k] >> 12) + (cM[k] >> 12) + c_LLbits[ll_code(llv[k])] + c_MLbits[ml_code(mlb[k])] + hb32(ofv[k]);
        }
        u32 tot;
        const u32 ex = block_excl_scan(S, nbits, &tot);
        LaneBitW bw; bw.init(S.stage, carryBits + ex);
#pragma unroll
        for (int k = 0; k < SEQ_PER; k++) {
          const u32 rl = SEQ_PER * tid + k;
          if (rl < cntT) {
            bw.add(cO[k] & 0xFFF, cO[k] >> 12); bw.add(cM[k] & 0xFFF, cM[k] >> 12); bw.add(cL[k] & 0xFFF, cL[k] >> 12);
            const u32 lb = c_LLbits[ll_code(llv[k])], mb = c_MLbits[ml_code(mlb[k])], ob = hb32(ofv[k]);
            bw.add(llv[k] & ((1u << lb) - 1), lb);
            bw.add(mlb[k] & ((1u << mb) - 1), mb);
            bw.add(ofv[k] & (ob >= 32 ? 0xFFFFFFFFu : ((1u << ob) - 1)), ob);
          }
        }
        bw.finish();
        u32 total = carryBits + tot;
        if (lastTile) {
          if (tid == 0) {
            LaneBitW fw; fw.init(S.stage, total);
            fw.add(fin[2] & ((1u << tlog[2]) - 1), tlog[2]);
            fw.add(fin[1] & ((1u << tlog[1]) - 1), tlog[1]);
            fw.add(fin[0] & ((1u << tlog[0]) - 1), tlog[0]);
            fw.add(1, 1);
            fw.finish();
          }
          total += tlog[0] + tlog[1] + tlog[2] + 1;
        }
        __syncthreads();
        const u32 nbytes = lastTile ? (total + 7) >> 3 : total >> 3;
        flush_stage(S, op + bytesOut, nbytes);
        carryBits = lastTile ? 0 : (total & 7);
        carryVal = (S.stage[nbytes >> 2] >> (8 * (nbytes & 3))) & ((1u << carryBits) - 1);
        bytesOut += nbytes;
        __syncthreads();
        EPROF(7)
      }
      op += bytesOut;
      if (tblErr) uncompressible = true;
      if (lastNCount && (op - lastNCount) < 4) uncompressible = true;
    }
    cSize = uncompressible ? 0 : (u32)(op - blk);
    const u32 minGainB = (L >> (strategy >= 8 ? strategy - 1 : 6)) + 2;
    if (cSize && cSize >= L - minGainB) cSize = 0;
    if (!first && cSize < 25) {
      // RLE block (never the first of a frame): all bytes of the block equal
      const u8 b0 = src[bs]; u32 diff = 0;
      for (u32 i = tid; i < L; i += ENT_THREADS) diff |= src[bs + i] != b0;
      if (!__syncthreads_or((int)diff)) { cSize = 1; if (tid == 0) blk[0] = b0; }
    }
  }
  if constexpr (PHASE == 1) { if (tid == 0) rec->nChain = 0; return; }      // (a block too small to compress: nothing for the chain kernel)
  __syncthreads();

  // ---------------------------------------------------------------- block header, state confirmation, frame tail
  u32 blockBytes;
  if (cSize == 0) {
    for (u32 i = tid; i < L; i += ENT_THREADS) blk[i] = src[bs + i];
    if (tid == 0) { const u32 h = (u32)last + (0u << 1) + (L << 3); op0[0] = (u8)h; op0[1] = (u8)(h >> 8); op0[2] = (u8)(h >> 16); }
    blockBytes = 3 + L;
  } else if (cSize == 1) {
    if (tid == 0) { const u32 h = (u32)last + (1u << 1) + (L << 3); op0[0] = (u8)h; op0[1] = (u8)(h >> 8); op0[2] = (u8)(h >> 16); }
    blockBytes = 4;
  } else {
    if (tid == 0) { const u32 h = (u32)last + (2u << 1) + (cSize << 3); op0[0] = (u8)h; op0[1] = (u8)(h >> 8); op0[2] = (u8)(h >> 16); }
    blockBytes = 3 + cSize;
    if (PHASE != 2 && !last) {      // (the split stage takes single-block frames only: nothing to confirm for a next block)
      // a compressed block confirms repcodes + entropy tables for the next block (A.4.2 / A.4.8)
      if (tid == 0) { st->rep[0] = bo->rep[0]; st->rep[1] = bo->rep[1]; st->rep[2] = bo->rep[2]; }
      if (newHuf) {
        st->hufNbBits[tid] = (u32)tid <= newHufMaxSym ? S.hNb[tid] : 0; st->hufVal[tid] = S.hVal[tid];
        if (tid == 0) { st->hufRepeat = 1; st->hufMaxSym = newHufMaxSym; }
      }
      if (nbSeq) {
        for (int k = 0; k < 3; k++) {
          u32* dstT = (u32*)(k == 0 ? &st->ll : k == 1 ? &st->of : &st->ml);
          const u32* srcT = (const u32*)&S.lit.ct[k];
          for (u32 i = tid; i < sizeof(ZraFseCTable) / 4; i += ENT_THREADS) dstT[i] = srcT[i];
        }
        if (tid == 0) { st->llRepeat = S.nextRepeat[0]; st->ofRepeat = S.nextRepeat[1]; st->mlRepeat = S.nextRepeat[2]; }
      }
    }
  }
  (void)newHufLog;
  EPROF(8)
#ifdef ZRA_MF_PROFILE
  if (tid == 0) atomicAdd(&zra_ent_prof[15], 1ull);
#endif
  if (tid == 0) {
    u32 pos = outPos + blockBytes;
    if (last) {
      if (a.checksum) { st32(slot + pos, a.contentCk[f]); pos += 4; }
      a.sizes[f] = pos;
    }
    st->outPos = pos;
  }
}

// (a real call: inlined into the kernel's queue loop the body spilled 66-74 vector registers under the same 5-waves budget)
template <int PHASE>
__device__ __attribute__((noinline)) void entropy_frame_call(KArgs& a, u32 block, u32 f, u8* lits, u8* slot, u8* work, EncShared& S, ZraEntRec* rec) {
  // (arguments arrive in vector registers: pin the wave-uniform ones to the scalar unit, or the frame's whole parameter set follows them)
  f = (u32)__builtin_amdgcn_readfirstlane((int)f); block = (u32)__builtin_amdgcn_readfirstlane((int)block);
  auto pin = [](u8* p) { const u64 v = (u64)p; return (u8*)(((u64)(u32)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)v)); };
  KArgs& au = *(KArgs*)(((u64)(u32)__builtin_amdgcn_readfirstlane((int)((u64)&a >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)(u64)&a));
  entropy_frame<PHASE>(au, block, f, pin(lits), pin(slot), pin(work), S, (ZraEntRec*)pin((u8*)rec));
}

// The entropy stage: workgroups that take frames from a queue, each with its own literal buffer and sequence work area.
//  * Batch path (multi-block frames, the other strategies): launched per block round behind the batch's match-finder launch; the queue
//    hands out the batch's frames, nothing to wait for (readyStamp == 0), a slot per frame.
//  * Persistent pipeline (round 5; single-block dfast frames): the workgroups stay resident BESIDE the persistent match finder — one
//    per CU is what a CU's LDS and registers hold next to 18-20 match-finder waves — and take frames in frame order: wait until the
//    match finder has published the frame (its stamp in the block record), encode it into the slot ring, count it for the host's scan +
//    gather of the sub-batch. Rounds 1-4 launched one workgroup per frame and sub-batch instead; whether a workgroup found room on a CU
//    then depended on how the match finder's waves had happened to land, and at one workgroup per CU the stage could not keep up.
//    Every wait gives up after ~10 s of the 100 MHz clock (pipeAbort): the call fails, the GPU does not hang.
// ---- the pipeline's seek-table build, done by the entropy stage's own workgroups (round 5: launches of a scan / gather kernel on a
// third stream were not placed while the two persistent kernels held the device — the first gather ran until the match finder left)
// exclusive scan of the sizes of sub-batch j -> a.offsets (absolute inside the body), *a.running += the sub-batch's bytes
__device__ __forceinline__ void pipe_scan_subbatch(KArgs& a, u32 j, EncShared& S) {
  const u32 j0 = j * a.entSubFrames, nbj = min(a.entSubFrames, a.nFrames - j0);
  const u32 per = (nbj + ENT_THREADS - 1) / ENT_THREADS, i0 = j0 + per * threadIdx.x, i1 = min(i0 + per, j0 + nbj);
  u32 sum = 0;
  for (u32 i = i0; i < i1; i++) sum += (u32)a.sizes[i];
  u32 tot;
  const u32 ex = block_excl_scan(S, sum, &tot);
  u64 off = *a.running + ex;
  for (u32 i = i0; i < i1; i++) { a.offsets[i] = off; off += a.sizes[i]; }
  __syncthreads();
  if (threadIdx.x == 0) *a.running += tot;
}
// frame `fi` of the launch: from its slot to body + offsets[fi]; seek-table entry and size
__device__ __forceinline__ void pipe_gather_frame(KArgs& a, u32 fi) {
  const u64 n = a.sizes[fi], off = a.offsets[fi];
  const u8* s = a.slots + (size_t)(fi % a.slotRing) * a.slotStride; u8* d = a.gBody + off;
  const u64 n16 = n >> 4;
  for (u64 i = threadIdx.x; i < n16; i += ENT_THREADS) {
    const uint4 v = ((const uint4*)s)[i];
    st64(d + 16 * i, (u64)v.x | ((u64)v.y << 32)); st64(d + 16 * i + 8, (u64)v.z | ((u64)v.w << 32));
  }
  for (u64 i = (n16 << 4) + threadIdx.x; i < n; i += ENT_THREADS) d[i] = s[i];
  if (threadIdx.x == 0) {
    if (a.gEntries) { u8* e = a.gEntries + (size_t)(a.firstFrame + fi) * 5; st32(e, (u32)off); e[4] = (u8)(off >> 32); }
    if (a.gSizesOut) a.gSizesOut[a.firstFrame + fi] = n;
  }
}

// The entropy stage: workgroups that take frames from a queue, each with its own literal buffer and sequence work area.
//  * Batch path (multi-block frames, the other strategies): launched per block round behind the batch's match-finder launch; the queue
//    hands out the batch's frames, nothing to wait for (readyStamp == 0), a slot per frame; the host launches scan and gather.
//  * Persistent pipeline (round 5; single-block dfast frames): the workgroups stay resident BESIDE the persistent match finder — one
//    per CU is what a CU's LDS and registers hold next to 18-20 match-finder waves — and take frames in frame order: wait until the
//    match finder has published the frame (its stamp in the block record), encode it into the slot ring, count it; the workgroup that
//    completes a sub-batch scans its sizes, and everybody copies encoded frames of scanned sub-batches into the archive between two
//    frames of their own. Rounds 1-4 launched one workgroup per frame and sub-batch instead; whether a workgroup found room on a CU
//    then depended on how the match finder's waves had happened to land, and at one workgroup per CU the stage could not keep up.
//    Nothing waits without doing the other work that is ready, and every wait gives up after ~10 s of the 100 MHz clock (pipeAbort):
//    the call fails, the GPU does not hang.
template <int PHASE>
__device__ __forceinline__ void entropy_kernel_body(u32 block) {
  KArgs& a = *(KArgs*)__builtin_amdgcn_kernarg_segment_ptr();     // (the kernel's first argument, where it arrived)
  __shared__ EncShared S;
  // This stage runs beside the match finder, which fills most issue slots; the one-lane serial sections here are latency-critical.
  // Raise the wave's issue priority so they are not queued behind match-finder waves (a.entPrio: bring-up knob ZRA_ENT_PRIO, default 3).
  switch (a.entPrio) { case 0: __builtin_amdgcn_s_setprio(0); break; case 1: __builtin_amdgcn_s_setprio(1); break; case 2: __builtin_amdgcn_s_setprio(2); break; default: __builtin_amdgcn_s_setprio(3); break; }
  const int tid = threadIdx.x;
  u8* const lits = a.lits + (size_t)blockIdx.x * a.litStride;
  u8* const workWg = a.entWork + (size_t)blockIdx.x * a.entWorkStride;      // (split stage: a work area per FRAME, below)
  const bool pipe = a.readyStamp != 0;
  const u32 SBF = a.entSubFrames, nSub = pipe ? (a.nFrames + SBF - 1) / SBF : 0u, ringSubs = max(1u, a.slotRing / SBF);
  // thread 0's bookkeeping
  u32 pend = 0xFFFFFFFFu; bool dry = false;
  u64 tStart = 0, tWait = 0, idleSince = 0; u32 nDone = 0;
  if (tid == 0 && a.mfTele) {
    const u32 hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
    atomicAdd((unsigned long long*)&a.mfTele[ZRA_TELE_ENT + 8 + ((xcc << 8) | (((hw >> 13) & 7u) << 5) | (((hw >> 12) & 1u) << 4) | ((hw >> 8) & 15u))], 1ull);
    tStart = wall_clock64();
  }
  for (;;) {
    __syncthreads();                                   // (S.sc of the step before is no longer read)
    if (tid == 0) {
      u32 act = 0, arg = 0;                            // 0 nothing ready (slept), 1 copy frame arg, 2 encode frame arg, 4 leave
      if (pipe) {
        // 1. an encoded frame of a scanned sub-batch to copy?
        u32 j = __hip_atomic_load(a.gatherJ, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (j < nSub && j < __hip_atomic_load(a.scanDone, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) {
          const u32 nbj = min(SBF, a.nFrames - j * SBF);
          const u32 g = __hip_atomic_load(&a.gQueue[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nbj ? atomicAdd(&a.gQueue[j], 1u) : nbj;
          if (g < nbj) { act = 1; arg = j * SBF + g; break; }
          atomicCAS(a.gatherJ, j, j + 1);
          j = __hip_atomic_load(a.gatherJ, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if (!act) {
        // 2. a frame to encode: the next of the queue, once its slot is free and the match finder has published it
        if (pend == 0xFFFFFFFFu && !dry) { const u32 f = atomicAdd(a.entQueue, 1u); if (f < a.nFrames) pend = f; else dry = true; }
        if (pend != 0xFFFFFFFFu) {
          bool okF = true;
          if (pipe) {
            const u32 j = pend / SBF;
            if (j >= ringSubs && __hip_atomic_load(a.gatherDone, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) + ringSubs <= j) okF = false;
            else if (__hip_atomic_load(&a.blockOut[pend].ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != a.readyStamp) okF = false;
          }
          if (okF) { act = 2; arg = pend; pend = 0xFFFFFFFFu; nDone++; }
        } else if (!pipe || __hip_atomic_load(a.gatherDone, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= nSub) act = 4;   // nothing left anywhere
      }
      if (pipe && __hip_atomic_load(a.pipeAbort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) act = 4;
      if (!act) {
        const u64 t = wall_clock64();
        if (!idleSince) idleSince = t;
        else if (t - idleSince > 1000000000ull) { atomicExch(a.pipeAbort, 2u); act = 4; }
        __builtin_amdgcn_s_sleep(64);
        tWait += wall_clock64() - t;
      } else idleSince = 0;
      S.sc[15] = act; S.sc[14] = arg;
    }
    __syncthreads();
    const u32 act = (u32)__builtin_amdgcn_readfirstlane((int)S.sc[15]), arg = (u32)__builtin_amdgcn_readfirstlane((int)S.sc[14]);
    if (act == 4) {
      if (tid == 0 && a.mfTele && nDone > 0) {
        u64* const t = a.mfTele + ZRA_TELE_ENT;
        atomicAdd((unsigned long long*)&t[0], 1ull); atomicAdd((unsigned long long*)&t[1], wall_clock64() - tStart);
        atomicAdd((unsigned long long*)&t[2], tWait); atomicAdd((unsigned long long*)&t[3], (unsigned long long)nDone);
      }
      return;
    }
    if (act == 0) continue;
    if (act == 1) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");             // the frame's slot, size and offset (other workgroups wrote them)
      pipe_gather_frame(a, arg);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");             // (the slot has been read: it may be written again once the sub-batch is through)
      __syncthreads();
      if (tid == 0) {
        const u32 j = arg / SBF, nbj = min(SBF, a.nFrames - j * SBF);
        if (atomicAdd(&a.gCopied[j], 1u) + 1 == nbj) {
          // sub-batches finish in any order: move the count of finished ones over every one that is complete
          for (;;) {
            const u32 d = __hip_atomic_load(a.gatherDone, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            if (d >= nSub || __hip_atomic_load(&a.gCopied[d], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != min(SBF, a.nFrames - d * SBF)) break;
            atomicCAS(a.gatherDone, d, d + 1);
          }
        }
      }
      continue;
    }
    // act == 2: encode frame `arg`
    if (pipe) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // what the match finder wrote for this frame (sequences, block record)
    entropy_frame_call<PHASE>(a, block, arg, lits, a.slots + (size_t)(arg % a.slotRing) * a.slotStride,
                              PHASE ? a.entWork + (size_t)arg * a.entWorkStride : workWg, S, PHASE ? a.entRec + arg : nullptr);
    if (pipe) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");             // the slot and the frame's size, before the count
      __syncthreads();
      const u32 j = arg / SBF, nbj = min(SBF, a.nFrames - j * SBF);
      if (tid == 0) S.sc[13] = atomicAdd(&a.entDone[j], 1u) + 1 == nbj ? 1u : 0u;
      __syncthreads();
      if (S.sc[13]) {
        // this workgroup encoded the sub-batch's last frame: it scans the sizes, behind the scan of the sub-batch before
        if (tid == 0) {
          const u64 t0 = wall_clock64();
          while (__hip_atomic_load(a.scanDone, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != j) {
            __builtin_amdgcn_s_sleep(16);
            if (__hip_atomic_load(a.pipeAbort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || wall_clock64() - t0 > 1000000000ull) { atomicExch(a.pipeAbort, 2u); break; }
          }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");           // every frame size of the sub-batch, the running offset
        pipe_scan_subbatch(a, j, S);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(a.scanDone, j + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

extern "C" __global__ void __launch_bounds__(ENT_THREADS, ZRA_ENT_WAVES)
zra_entropy_kernel(ZraEncArgs a_, u32 block) { (void)a_; entropy_kernel_body<0>(block); }
// the split stage of the persistent pipeline (round 6): FRONT, zra_ent_chain_kernel, BACK — one launch each per sub-batch
extern "C" __global__ void __launch_bounds__(ENT_THREADS, ZRA_ENT_WAVES)
zra_entropy_front_kernel(ZraEncArgs a_, u32 block) { (void)a_; entropy_kernel_body<1>(block); }
extern "C" __global__ void __launch_bounds__(ENT_THREADS, ZRA_ENT_WAVES)
zra_entropy_back_kernel(ZraEncArgs a_, u32 block) { (void)a_; entropy_kernel_body<2>(block); }

// The three FSE state chains of a block: state -> stateTable[(state >> nb) + dfs] -> state, one dependent LDS round trip per sequence and
// stream, inherently serial (chains from different start states do not merge; a table walked through v_readlane costs more than the
// round trip — round 5). Inside the frame's workgroup they kept three lanes busy for three quarters of the stage while 253 waited.
// Here lane = (frame, stream): ZRA_CHAIN_FRAMES frames per wave, their tables (FRONT left them in the frames' records) in 3.5 KiB of LDS
// each, the code bytes read 8 at a time and the results written 4 at a time per lane. One wave per workgroup; 21 KiB of LDS: one per CU
// beside the match finder's 18 waves, as many as fit otherwise.
#define ZRA_CHAIN_TBL_BYTES 3528u      /* per frame: state tables LL 512 + OF 256 + ML 512 u16; {deltaNbBits, deltaFindState} LL 36 + OF 32 + ML 53 */
// (G frames per wave. Beside the match finder a lone wave's step is bound by its issue slots, not by lanes: fewer frames per wave and more
//  waves per CU — same LDS — buy issue share; zra_encode.hip picks the instantiation: ZRA_CHAIN_G)
template <u32 ZRA_CHAIN_FRAMES>
__device__ __forceinline__ void ent_chain_body() {
  KArgs& a = *(KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  __shared__ __attribute__((aligned(16))) u8 T[ZRA_CHAIN_FRAMES * ZRA_CHAIN_TBL_BYTES];
  const int lane = threadIdx.x;
  switch (a.entPrio) { case 0: __builtin_amdgcn_s_setprio(0); break; case 1: __builtin_amdgcn_s_setprio(1); break; case 2: __builtin_amdgcn_s_setprio(2); break; default: __builtin_amdgcn_s_setprio(3); break; }
  const u32 f0 = blockIdx.x * ZRA_CHAIN_FRAMES;
  // ---- tables into LDS, all lanes: per frame and stream the state table (2^tableLog entries) and the per-symbol pairs
  for (u32 g = 0; g < ZRA_CHAIN_FRAMES; g++) {
    const u32 f = f0 + g;
    if (f >= a.nFrames) break;
    const ZraEntRec* const r = a.entRec + f;
    if (!r->nChain) continue;
    u8* const Tf = T + g * ZRA_CHAIN_TBL_BYTES;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      if (!r->run[k]) continue;
      const ZraFseCTable* const ct = &r->ct[k];
      u16* const stL = (u16*)(Tf + (k == 0 ? 0u : k == 1 ? 1024u : 1536u));
      u64* const symL = (u64*)(Tf + (k == 0 ? 2560u : k == 1 ? 2848u : 3104u));
      const u32 nState = min(1u << r->tlog[k], k == 1 ? 256u : 512u), nSym = k == 0 ? 36u : k == 1 ? 32u : 53u;
      for (u32 i = (u32)lane; i < nState / 2; i += 64) ((u32*)stL)[i] = ((const u32*)ct->stateTable)[i];
      for (u32 i = (u32)lane; i < nSym; i += 64) symL[i] = (u64)ct->deltaNbBits[i] | ((u64)(u32)ct->deltaFindState[i] << 32);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  // ---- the chains
  const u32 g = (u32)lane / 3u, k = (u32)lane % 3u, f = f0 + g;
  if (g >= ZRA_CHAIN_FRAMES || f >= a.nFrames) return;
  ZraEntRec* const r = a.entRec + f;
  const u32 n = r->nChain;
  if (!n) return;
  u8* const work = a.entWork + (size_t)f * a.entWorkStride;
  const u8* const codes = work + (size_t)k * a.seqStride;
  u16* const out = (u16*)(work + 3 * a.seqStride) + (size_t)k * a.seqStride;
  if (!r->run[k]) {
    // (a one-symbol table emits no state bits: zeros; the BACK launch reads them like any other stream's)
    for (u32 i = 0; i < n; i++) out[i] = 0;
    r->finalState[k] = 0;
    return;
  }
  const u8* const Tf = T + g * ZRA_CHAIN_TBL_BYTES;
  const u16* const stL = (const u16*)(Tf + (k == 0 ? 0u : k == 1 ? 1024u : 1536u));
  const u64* const symL = (const u64*)(Tf + (k == 0 ? 2560u : k == 1 ? 2848u : 3104u));
  // code bytes: 8 per load (the work area is 256-byte aligned per frame, seqStride even: 2-byte aligned streams -> byte-assembled head)
  u64 cw = 0; u32 cHave = 0, cPos = 0;
  auto next_code = [&]() -> u32 {
    if (!cHave) {
      const u32 left = n - cPos;
      if (left >= 8 && (((uintptr_t)(codes + cPos)) & 7u) == 0) { cw = *(const u64*)(codes + cPos); cHave = 8; }
      else { cw = codes[cPos]; cHave = 1; }
      cPos += cHave;
    }
    const u32 c = (u32)cw & 255u; cw >>= 8; cHave--;
    return c;
  };
  u32 sym = next_code();
  u32 state;
  {
    const u64 sp = symL[sym];
    const u32 d = (u32)sp, nb = (d + (1u << 15)) >> 16, v = (nb << 16) - d;
    state = stL[(v >> nb) + (u32)(i32)(u32)(sp >> 32)];
  }
  u64 ow = 0; u32 oHave = 1;                            // out[0] = 0: the block's last sequence only initialises the states
  u32 oPos = 0;
  auto put_out = [&](u32 v16) {
    ow |= (u64)v16 << (16 * oHave); oHave++;
    if (oHave == 4) { st64((u8*)(out + oPos), ow); oPos += 4; ow = 0; oHave = 0; }
  };
  if (n > 1) {
    u32 symN = next_code();
    u64 sp = symL[symN];
    for (u32 i = 1; i < n; i++) {
      const u32 dnb = (u32)sp; const i32 dfs = (i32)(u32)(sp >> 32);
      const u32 nb = (state + dnb) >> 16;
      const u32 bits = state & ((1u << nb) - 1);
      state = stL[(state >> nb) + dfs];                  // critical load first
      if (i + 1 < n) { symN = next_code(); sp = symL[symN]; }
      put_out((nb << 12) | bits);
    }
  }
  for (u32 j = 0; j < oHave; j++) out[oPos + j] = (u16)(ow >> (16 * j));
  r->finalState[k] = state;
}
extern "C" __global__ void __launch_bounds__(64) zra_ent_chain_kernel(ZraEncArgs a_) { (void)a_; ent_chain_body<6>(); }
extern "C" __global__ void __launch_bounds__(64) zra_ent_chain5_kernel(ZraEncArgs a_) { (void)a_; ent_chain_body<5>(); }   // 17,640 B: beside 19 finder waves
extern "C" __global__ void __launch_bounds__(64) zra_ent_chain3_kernel(ZraEncArgs a_) { (void)a_; ent_chain_body<3>(); }
extern "C" __global__ void __launch_bounds__(64) zra_ent_chain2_kernel(ZraEncArgs a_) { (void)a_; ent_chain_body<2>(); }
extern "C" __global__ void __launch_bounds__(64) zra_ent_chain1_kernel(ZraEncArgs a_) { (void)a_; ent_chain_body<1>(); }

#ifdef ZRA_MF_PROFILE
extern "C" __attribute__((visibility("default"))) void ZraHipDebugReadEntProfile(unsigned long long* out16, int reset) {
  (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(zra_ent_prof), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(zra_ent_prof), z, sizeof(z)); }
}
#endif
