// zra_amd — argument blocks shared between the host engine and the gfx950 kernels.
#pragma once
#include <stdint.h>
#include <stddef.h>

// ------------------------------------------------------------------------------------------------ decode
// The decoder is four stages per ROUND (one compressed block of every unfinished frame per round; raw / RLE blocks and the frame
// ends are consumed by the first kernel on the way); zra_decode.hip's header describes them:
//   zra_dec_parse_kernel      wave per frame    headers, Huffman tree descriptions, FSE tables -> per-frame table scratch
//   zra_dec_huf_kernel        wave per frame    Huffman literal streams (16 runs per stream side by side) -> literal scratch
//   zra_dec_chain_kernel      LANE per frame    the serial FSE sequence chains, every check of the reference's sequence loop
//   zra_dec_chain_lds_kernel  (the same, tables in LDS, beside it)                                              -> sequence scratch
//   zra_dec_exec_kernel       wave per frame    data movement: literal + match copies, frame end, random-access slices
//   zra_ra_small_kernel       workgroup per frame, all stages in one launch (small random-access batches)
// frames whose decode tables one workgroup of zra_dec_huf_kernel holds (it takes them one after the other, all lanes on each); its LDS = this many 4.25 KiB tables
#ifndef ZRA_HUF_FRAMES
#define ZRA_HUF_FRAMES 2        /* A/B on one box, 8 GiB decode, lane-per-stream decoder: 16 / 8 / 4 / 2 / 1 frames per wave -> Huffman stage 20.0 / 18.4 / 14.7 / 13.3 / 15.4 ms; wave-wide decoder: 1 / 2 / 4 -> 12.9 / 9.8 / 10.1 */
#endif
#define ZRA_DEC_TBL_LL 0u       // 512 cells x 4 B  sym | extraBits<<8 | stateBits<<16 | nextBase<<20 (the base value of a length code comes from a
#define ZRA_DEC_TBL_ML 512u     // 512 cells x 4 B   constant table: 5 KiB of cells per frame instead of 9, so that the frames in flight fit the caches)
#define ZRA_DEC_TBL_OF 1024u    // 256 cells x 4 B  (base value = 1 << code)
#define ZRA_DEC_TBL_WORDS 1280u // u32 words of table scratch per frame
// the one-launch kernel's copy in LDS keeps the base value next to the cell (one read per cell on its dependent path)
#define ZRA_LDS_TBL_LL 0u       // 512 cells x 8 B  {cell, baseValue}
#define ZRA_LDS_TBL_ML 1024u    // 512 cells x 8 B
#define ZRA_LDS_TBL_OF 2048u    // 256 cells x 4 B
#define ZRA_LDS_TBL_WORDS 2304u
// zra_dec_chain_lds_kernel: frames (= lanes) whose tables one workgroup keeps in LDS (31 x 5 KiB + 512 B of 160 KiB; the rest leaves
// room for two workgroups of zra_dec_chain_kernel on the same CU)
#define ZRA_CHAIN_LDS_FRAMES 59u   // lanes of the LDS-table chain kernel's wave that take frames: 59 x (2,560 B of two-byte cells + 144 B of ring) + 512 B beside the other chain kernel's two waves
#define ZRA_CHAIN_RING_WORDS 36u   // per frame of the LDS-table chain kernel: a 128-byte ring of its sequence bitstream + 16 (zra_decode.hip: CHAIN_RING)

// per-frame decode state; lives in HBM scratch for the whole call (frames take one round per compressed block)
struct ZraDecFrame {
  uint32_t blkPos, produced, done;
  uint32_t hufValid, hufMaxBits, hufNSym, hufX2;       // kept Huffman description (treeless blocks rebuild the table from it)
  uint32_t llValid, mlValid, ofValid, llLog, mlLog, ofLog, ofShare;   // ofShare: ZSTD_getLongOffsetsShare of the current OF table
  uint32_t rep[3];
  uint32_t fcsLo, fcsHi, fcsHave, hasChecksum, bigWindow;
  // the compressed block handed from the parse kernel to the chain and execute kernels of the same round
  uint32_t bpos, bsize, blast;
  uint32_t litKind, litRegen, litArg;                  // 0: raw bytes at frame position litArg; 1: RLE of byte litArg; 2: literal scratch at litBase
  uint32_t litStreams, streamOff[4], streamLen[4];     // Huffman streams (positions relative to the block content)
  uint32_t hufErr, lateErr;                            // literal-stream failure (Huffman kernel); error of the sections behind the literals (parse kernel)
  uint32_t nbSeq, seqPos, longMode;
  uint64_t litBase, seqBase;
  // results of the chain kernel
  uint32_t chainErr, nSeqValid, seqOut, seqLit, truncated;
  uint32_t repOut[3];
  // block-parallel pass (round 6; frame records only): compressed blocks handed on as jobs, where the frame's last block ends, the
  // random-access limit was reached before a block
  uint32_t nBlk, endPos, parseTrunc, pad1_;
  alignas(8) uint8_t weights[256];      // kept Huffman description: weight and first decode-table cell of every symbol
  alignas(8) uint16_t hufStart[256];
};
// Block-parallel pass: a block's repeat offsets at its start are not known while its sequences are decoded (they are the previous block's
// last three) — the chain stage runs on MARKERS, the execute stage, which walks a frame's blocks in order, puts the values in. A marker =
// centre of band k (k = which of the block's three initial offsets) minus the number of times "offset - 1" was applied to it
// (ZSTD_decodeSequence's third repeat code); real offsets of such a frame lie below the bands (frames of at most 128 MiB take this pass).
#define ZRA_REP_MARK_LO 0x0C000000u
#define ZRA_REP_MARK(k) (ZRA_REP_MARK_LO + ((uint32_t)(k) << 24) + 0x00800000u)
#define ZRA_FMB_BLOCK (128u << 10)     /* what every compressed block but a frame's last regenerates in frames this pass finishes itself (zstd's block size; checked) */

// one random-access slice: `len` bytes at `srcOff` inside decoded frame job `job` go to raOut + dstOff
struct ZraRaPiece { uint64_t dstOff; uint32_t srcOff, len; };

struct ZraDecodeArgs {
  const uint8_t* body;       // compressed bytes (all frames)
  uint64_t bodySize;         // readable bytes at body
  const uint64_t* frameOff;  // frame f spans [frameOff[f*offStride], frameOff[f*offStride+1]) inside body
  uint32_t offStride;        // 1: consecutive frames share boundaries; 2: independent (start,end) pairs
  uint8_t* out;              // destination base
  const uint64_t* outOff;    // [nFrames] destination offset of each frame
  const uint32_t* outCap;    // [nFrames] destination capacity of each frame
  uint32_t nFrames;
  // random access (nullptr otherwise): decode may stop once limit[f] bytes exist; finished frames copy their slices out
  const uint32_t* limit;
  const uint32_t* pieceBase; // [job+1] first slice of each job
  const ZraRaPiece* pieces;
  uint8_t* raOut;
  // round machinery (device scratch)
  const uint32_t* active;    // jobs of this round (nullptr in round 0: all of them)
  uint32_t nActive;
  const uint32_t* nActivePtr; // rounds enqueued behind another without a host synchronisation: the number of jobs is read here (nActive: upper bound)
  uint32_t round;
  uint32_t* counters;        // see ZRA_DC_* below
  uint32_t* nextActive;      // jobs that need another round
  uint32_t* pending;         // jobs with a compressed block waiting for the chain / execute kernels
  uint32_t* hufJobs;         // ... of those, the ones with Huffman-coded literals
  ZraDecFrame* frames;       // [nFrames]
  uint32_t* tables;          // [nFrames * ZRA_DEC_TBL_WORDS]
  uint8_t* lits; uint64_t litCap;      // literal scratch (bump-allocated per round)
  uint64_t* seqs; uint64_t seqCap;     // sequence scratch: litLength | matchLength<<18 | offset<<36
  uint32_t* status;          // [nFrames] zstd error code per frame (0 = ok)
  uint32_t* produced;        // [nFrames] bytes regenerated
  uint32_t* frameMeta;       // [2*nFrames] {1 = has checksum / 2 = stopped early (no frame-end checks), stored checksum}
  uint32_t debugSkip;        // bring-up timing knob (ZRA_DEC_SKIP): execute kernel stage ablation; 0 in production
  // block-parallel pass (round 6, frames of several blocks): the Huffman and chain stages take BLOCKS as jobs — job = frame * bpf + the
  // compressed block's ordinal, its record in `frames`, its tables in `tables` (the host points both at the block arrays for those two
  // launches) —, zra_dec_parse_all_kernel writes them from the frame records, zra_dec_exec_all_kernel walks a frame's blocks in order
  uint32_t bpf;              // block jobs per frame (0 / 1: jobs are frames)
  ZraDecFrame* blkRecs;      // [nFrames * bpf]
  uint32_t* blkTables;       // [nFrames * bpf * ZRA_DEC_TBL_WORDS]
  uint32_t* execList;        // frames whose blocks were all handed on (count: counters[ZRA_DC_NEXEC]); the others are in nextActive (bail list)
};
// counters[]: u32 words, zeroed before every round
#define ZRA_DC_QPARSE 0
#define ZRA_DC_QCHAIN 1
#define ZRA_DC_QEXEC 2
#define ZRA_DC_NPENDING 3
#define ZRA_DC_NNEXT 4
#define ZRA_DC_NHUF 5
#define ZRA_DC_QHUF 10
#define ZRA_DC_LITCUR 6      // u64 (words 6,7)
#define ZRA_DC_SEQCUR 8      // u64 (words 8,9)
#define ZRA_DC_NEXEC 11
#define ZRA_DC_QEXECALL 12
#define ZRA_DC_WORDS 16

// ------------------------------------------------------------------------------------------------ encode
// effective zstd 1.4.9 compression parameters of one frame size class
struct ZraEncParams {
  uint32_t windowLog, chainLog, hashLog, searchLog, minMatch, targetLength, strategy;
  uint32_t blockSize;     // min(128 KiB, 1 << windowLog)
};

// optimal parsers (btopt / btultra / btultra2, levels 13-22): per-frame statistics of the price model (they live as long as the frame)
// and the work arrays of the forward parse; sits in the frame's table slot behind the hash, tree and 3-byte hash tables
#define ZRA_OPT_NUM 4096u
struct ZraOptCell { int32_t price; uint32_t off, mlen, litlen, rep[3]; };
struct ZraOptMatch { uint32_t off, len; };
struct ZraOptState {
  uint32_t litFreq[256], litLengthFreq[36], matchLengthFreq[53], offCodeFreq[32];
  uint32_t litSum, litLengthSum, matchLengthSum, offCodeSum;
  uint32_t litSumBasePrice, litLengthSumBasePrice, matchLengthSumBasePrice, offCodeSumBasePrice;
  uint32_t predef, pad;
  ZraOptCell table[ZRA_OPT_NUM + 1];
  ZraOptMatch matches[ZRA_OPT_NUM + 1];
};

// FSE encoding table in the layout the kernels use (A.4.6)
struct ZraFseCTable {
  uint16_t stateTable[512];
  uint32_t deltaNbBits[53];
  int32_t deltaFindState[53];
  uint32_t tableLog, maxSym, rle;
};

// per-frame state carried from block to block inside a frame (A.4.8); lives in HBM scratch
struct alignas(16) ZraEncFrameState {
  uint32_t rep[3];
  uint32_t nextToUpdate;
  // wave-cooperative hash-chain finder: it inserts whole windows ahead of the parse. insEnd = first index not inserted yet (at the end
  // of a block up to 66 indices beyond nextToUpdate); ring[i & 127] = what index i found in its chain slot when it was inserted. If
  // the next block starts with the reference's "limited update after a very long match", the indices inserted ahead are taken out
  // of the tables again, newest first, with these values.
  uint32_t insEnd;
  uint32_t pad0_[3];
  uint32_t ring[128][4];           // (the four links of the slot: zra_encode_mf.hip HCW; 16-byte aligned)
  uint32_t idxShift;               // optimal parsers: table index = position + 1 + idxShift (btultra2 moves the window base after its statistics pass)
  uint32_t outPos;                 // bytes of the frame already written to its slot
  uint32_t hufRepeat;              // 0 none, 1 check, 2 valid
  uint32_t llRepeat, ofRepeat, mlRepeat;
  uint32_t hufMaxSym;
  uint8_t hufNbBits[256];
  uint16_t hufVal[256];
  ZraFseCTable ll, of, ml;
};
// (the hash-chain finder copies ring[] as uint4: every element of an array of these, and the ring inside it, sits on 16 bytes)
static_assert(sizeof(ZraEncFrameState) % 16 == 0 && offsetof(ZraEncFrameState, ring) % 16 == 0, "ZraEncFrameState::ring must be 16-byte aligned in arrays");

// split entropy stage of the persistent pipeline (round 6): what the FRONT launch leaves per frame for zra_ent_chain_kernel and the BACK launch
struct ZraEntRec {
  ZraFseCTable ct[3];        // 0 LL, 1 OF, 2 ML: the block's encoding tables (valid when nChain != 0)
  uint32_t nChain;           // sequences whose state chains are to be walked (0: none — raw or skipped block, no sequences, table error)
  uint32_t run[3];           // stream k has a real table (not a one-symbol one)
  uint32_t finalState[3];    // written by the chain kernel
  uint32_t tlog[3];
  uint32_t tblErr;
  uint32_t opOff, lastNCountOff;   // offsets in the block content: where the sequence bitstream starts; the last table description (0xFFFFFFFF: none)
  uint32_t newHuf, newHufMaxSym;
  uint32_t pad_;
};

// stage 1 -> stage 2 hand-off for the block being processed
struct ZraEncBlockOut {
  uint32_t nbSeq, lastLL, skip;    // skip: block shorter than 7 bytes, emitted raw without entropy stage
  uint32_t rep[3];                 // repcodes after the block (confirmed only if emitted compressed)
  uint32_t ready;                  // persistent pipeline: ZraEncArgs::readyStamp once the match finder has published the frame (release)
  uint32_t pad_;
};

struct ZraEncArgs {
  const uint8_t* in;       // input bytes of this call
  uint64_t inSize;         // bytes at `in`
  uint32_t frameSize;
  uint32_t firstFrame;     // index (within this call) of the first frame of the batch
  uint32_t nFrames;        // frames in the batch
  uint32_t checksum;
  uint32_t mfFilter;       // dfast kernel LDS geometry: filterShiftLong | filterShiftShort<<4 | log2(dupSlots)<<8
  uint32_t serialAll;      // frames larger than the level's window: every frame of the batch goes through the serial finders (zra_mf_kernel)
  ZraEncParams full, tail; // parameters of full-size frames / of the short last frame
  uint32_t* tables;        // nFrames * tableStride u32
  uint64_t tableStride;    // words per frame
  uint64_t* seqs;          // nFrames * seqStride packed sequences: ll | ml<<20 | offsetValue<<40
  uint64_t seqStride;
  uint8_t* lits;           // per entropy-stage workgroup: litStride literal bytes of the block it is encoding
  uint64_t litStride;
  uint8_t* entWork;        // per entropy-stage workgroup: the sequence section's codes [3][seqStride] u8 + chain output [3][seqStride] u16
  uint64_t entWorkStride;
  uint8_t* slots;          // slotRing * slotStride: the encoded frames, frame f in slot f % slotRing
  uint64_t slotStride;
  ZraEncFrameState* state; // [nFrames]
  ZraEncBlockOut* blockOut;// [nFrames]
  uint32_t* contentCk;     // [nFrames] XXH64 low 32 bits of each frame's input
  uint64_t* sizes;         // [nFrames] final frame sizes (written with the last block)
  // persistent pipeline (single-block dfast frames): TWO persistent kernels and nothing else. The match finder's waves pull frame indices
  // from `mfQueue`, use the hash-table slot of their workgroup (tables = nSlots * tableStride) and publish a finished frame by writing
  // `readyStamp` into its block record (ZraEncBlockOut::ready). The entropy stage's workgroups (zra_entropy_kernel) pull frame indices
  // from `entQueue`, wait for the frame's stamp, encode it into slot (frame % slotRing) and count it in entDone[frame / entSubFrames];
  // the workgroup that completes a sub-batch scans its frame sizes (offsets[], *running; sub-batches in order: scanDone), and every
  // workgroup, between two frames, copies encoded frames of scanned sub-batches to their place in the archive (gQueue / gCopied per
  // sub-batch, gatherJ = the sub-batch being handed out, gatherDone = sub-batches whose slots are free again).
  // mfQueue == nullptr: one workgroup per frame, tables per frame (batch path; its entropy launches use entQueue alone, readyStamp 0).
  uint32_t* mfQueue;
  uint32_t* mfStarted;     // counts the match finder's waves as they start (the entropy stage is launched once all of them are resident)
  uint32_t* mfDone;        // [sub-batches] frames the match finder has finished (pipeMode 1: the host releases a sub-batch's entropy launch by a stream wait on it); may be nullptr
  uint32_t* entQueue;
  uint32_t* entDone;       // [sub-batches]
  uint32_t* scanDone;
  uint32_t* gatherJ;
  uint32_t* gQueue;        // [sub-batches]
  uint32_t* gCopied;       // [sub-batches]
  uint32_t* gatherDone;
  uint32_t* pipeAbort;     // non-zero: give up waiting (error exit of the host, or a wait that ran out of patience)
  uint64_t* offsets;       // [nFrames] offset of each frame inside the body (written by the scans)
  uint64_t* running;       // body bytes so far (carried from launch to launch)
  uint8_t* gBody;          // the archive's body
  uint8_t* gEntries;       // seek-table entries (5 bytes per frame of the call) or nullptr
  uint64_t* gSizesOut;     // u64 size per frame of the call, or nullptr
  uint32_t entSubFrames;
  uint32_t slotRing;       // frames the slot buffer holds
  uint32_t readyStamp;
  uint32_t entPrio;        // issue priority of the entropy stage's waves (s_setprio 0..3)
  // launch telemetry of the persistent match finder (nullptr: none), written by lane 0 of every wave: where the wave sat (XCD / SE / CU),
  // its shader cycles against the constant 100 MHz clock (the effective shader clock of the launch), frames taken per XCD
  uint64_t* mfTele;        // ZRA_TELE_WORDS u64
  ZraEntRec* entRec;       // split entropy stage: [nFrames] records (nullptr: the one-launch stage); entWork is then per FRAME, not per workgroup
};
// mfTele layout (u64 words): [0] sum of shader cycles over waves, [1] sum of 100 MHz ticks over waves, [2] waves, [3] longest wave in ticks,
// [4] earliest wave start (ticks, stored inverted for an atomic max), [5] latest wave end, [6] latest wave START, [7] earliest wave end (inverted),
// [8..15] waves per XCD, [16..23] frames taken per XCD, [24..31] ticks spent per XCD, [32 + k] waves on CU key k (k = xcc << 8 | se << 5 | sh << 4 | cu)
#define ZRA_TELE_CUKEYS 2048u
#define ZRA_TELE_HEAD (32u + ZRA_TELE_CUKEYS)
#define ZRA_TELE_WAVES 8192u                               /* then per wave (workgroup index): start cycles, start ticks, XCD, frames taken */
#define ZRA_TELE_ENT (ZRA_TELE_HEAD + 4u * ZRA_TELE_WAVES)   /* then the entropy stage's persistent workgroups: [0] workgroups that took a frame, [1] their resident ticks, [2] ticks spent waiting for a frame or a slot, [3] frames, [8 + k] workgroups on CU key k */
#define ZRA_TELE_WORDS (ZRA_TELE_ENT + 8u + ZRA_TELE_CUKEYS)
// bucket flags of the dfast match finder (round 5): every wave computes its frame's flags itself (df_later_flags) into
// flags + workgroup * flagStride, over ldsWords words of its LDS, ahead of the parse
struct ZraFlagArgs {
  uint8_t* flags; uint64_t flagStride; uint32_t ldsWords;
  uint32_t spanBytes;      // source span of the parse in LDS behind the filter (mf_dfast_lean; 0: none): spanBytes + 16 + (spanBytes / 64 + 1) * 16 bytes
  uint32_t epochBits;      // round 6: bits of a cell's tag field that hold the writing frame's epoch (0: none, the slot is cleared per frame); a wave clears its slot once per 2^epochBits frames
};


