// zra_amd — argument blocks shared between the host engine and the gfx950 kernels.
#pragma once
#include <stdint.h>
#include <stddef.h>

#define ZRA_LIT_STRIDE ((size_t)(128u << 10) + 64)   // per-workgroup Huffman literal scratch (one block max)

// one launch of zra_decode_frames_kernel decodes nFrames independent zstd frames
struct ZraDecodeArgs {
  const uint8_t* body;       // compressed bytes (all frames)
  uint64_t bodySize;         // readable bytes at body
  const uint64_t* frameOff;  // frame f spans [frameOff[f*offStride], frameOff[f*offStride+1]) inside body
  uint32_t offStride;        // 1: consecutive frames share boundaries; 2: independent (start,end) pairs
  uint8_t* out;              // destination base
  const uint64_t* outOff;    // [nFrames] destination offset of each frame
  const uint32_t* outCap;    // [nFrames] destination capacity of each frame
  const uint32_t* outExpect; // [nFrames] bytes the frame must regenerate
  uint32_t nFrames;
  uint32_t* queue;           // frame queue head (zeroed before the launch)
  uint8_t* litScratch;       // gridDim.x * ZRA_LIT_STRIDE bytes
  uint32_t* status;          // [nFrames] zstd error code per frame (0 = ok)
  uint32_t* produced;        // [nFrames] bytes regenerated
  uint32_t* frameMeta;       // [2*nFrames] {has checksum, stored checksum}
  uint32_t debugSkip;        // bring-up timing knob (ZRA_DEC_SKIP): 1 skip match copies, 2 skip literal copies, 4 skip Huffman decode; 0 in production
};

// ------------------------------------------------------------------------------------------------ encode
// effective zstd 1.4.9 compression parameters of one frame size class
struct ZraEncParams {
  uint32_t windowLog, chainLog, hashLog, searchLog, minMatch, targetLength, strategy;
  uint32_t blockSize;     // min(128 KiB, 1 << windowLog)
};

// FSE encoding table in the layout the kernels use (A.4.6)
struct ZraFseCTable {
  uint16_t stateTable[512];
  uint32_t deltaNbBits[53];
  int32_t deltaFindState[53];
  uint32_t tableLog, maxSym, rle;
};

// per-frame state carried from block to block inside a frame (A.4.8); lives in HBM scratch
struct ZraEncFrameState {
  uint32_t rep[3];
  uint32_t nextToUpdate;
  uint32_t outPos;                 // bytes of the frame already written to its slot
  uint32_t hufRepeat;              // 0 none, 1 check, 2 valid
  uint32_t llRepeat, ofRepeat, mlRepeat;
  uint32_t hufMaxSym;
  uint8_t hufNbBits[256];
  uint16_t hufVal[256];
  ZraFseCTable ll, of, ml;
};

// stage 1 -> stage 2 hand-off for the block being processed
struct ZraEncBlockOut {
  uint32_t nbSeq, lastLL, skip;    // skip: block shorter than 7 bytes, emitted raw without entropy stage
  uint32_t rep[3];                 // repcodes after the block (confirmed only if emitted compressed)
};

struct ZraEncArgs {
  const uint8_t* in;       // input bytes of this call
  uint64_t inSize;         // bytes at `in`
  uint32_t frameSize;
  uint32_t firstFrame;     // index (within this call) of the first frame of the batch
  uint32_t nFrames;        // frames in the batch
  uint32_t checksum;
  uint32_t mfFilter;       // dfast kernel LDS geometry: filterShiftLong | filterShiftShort<<4 | log2(dupSlots)<<8
  ZraEncParams full, tail; // parameters of full-size frames / of the short last frame
  uint32_t* tables;        // nFrames * tableStride u32
  uint64_t tableStride;    // words per frame
  uint64_t* seqs;          // nFrames * seqStride packed sequences: ll | ml<<20 | offsetValue<<40
  uint64_t seqStride;
  uint8_t* lits;           // nFrames * litStride literal bytes of the current block
  uint64_t litStride;
  uint8_t* slots;          // nFrames * slotStride: the encoded frames, one per slot
  uint64_t slotStride;
  ZraEncFrameState* state; // [nFrames]
  ZraEncBlockOut* blockOut;// [nFrames]
  uint32_t* contentCk;     // [nFrames] XXH64 low 32 bits of each frame's input
  uint64_t* sizes;         // [nFrames] final frame sizes (written with the last block)
  // persistent match-finder launch (single-block frames): workgroups pull frame indices from `mfQueue`, use the hash-table slot
  // of their workgroup (tables = nSlots * tableStride) and count finished frames per sub-batch of `mfSubFrames` frames in
  // mfDone[] (the entropy stage of a sub-batch is released by a stream wait on its counter). mfQueue == nullptr: one workgroup
  // per frame, tables per frame.
  uint32_t* mfQueue;
  uint32_t* mfDone;
  uint32_t mfSubFrames;
};
