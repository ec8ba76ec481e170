"""Register / scratch budgets of the hot kernels, from the compiler's own resource remarks (zra_amd/build.py keeps them in
zra_amd/build/kernel_resources.json). A kernel that shares a translation unit with heavier code can silently inherit its budget: the
hash-chain kernel went from 37 to 119 VGPRs + 140 B of scratch (and to half its throughput) when the serial block parser with the
optimal parsers was called from it — no test noticed, because the bytes stayed right. These limits are what the measured numbers in
DESIGN.md were obtained with."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

# kernel -> (max VGPRs, max scratch bytes per lane)
BUDGET = {
    "zra_mf_dfast_kernel": (72, 0),      # 18 resident waves per CU with 6 KiB LDS each (the allocation granule is 8 registers: 65-72 cost the same 7 waves per SIMD)
    "zra_mf_dfast_fl_kernel": (80, 64),  # the same parse behind the wave own bucket-flag sweep (the headline kernel since round 5): 80 registers so that five of its waves leave room for the entropy stage wave on a SIMD (two registers spilled for it, in the sweep)
    "zra_mf_dfast_ls_kernel": (96, 0),   # the same parse over a copy of the frame in LDS (calls of a few hundred frames): one or two waves per CU by LDS, registers are not its limit
    "zra_mf_hc_kernel": (64, 0),         # one wave per frame, as many waves per CU as the hardware holds
    "zra_mf_fast_kernel": (64, 384),     # lane = frame; the scratch is the per-lane frame descriptor + a copy of the argument block
    "zra_mf_kernel": (96, 392),          # generic one-lane finder without the optimal parsers (btlazy2, frames beyond the window, odd tails)
    "zra_mf_opt_kernel": (136, 480),     # ... with them (levels 13-22; its scratch holds the parser's small arrays)
    "zra_dec_chain_kernel": (80, 0),     # lane = frame FSE chains (two waves per CU are launched: registers are not its limit; 78 since round 6's block jobs — a job's frame is job / bpf, marker-aware offset check)
    "zra_dec_parse_all_kernel": (96, 256),  # round 6, block-parallel pass: the parse of every block of a frame in one go (same body, same budget as the parse kernel)
    "zra_dec_exec_all_kernel": (80, 160),   # ... and the execute stage over a frame's blocks in order (the execute kernel's step, same budget)
    "zra_dec_chain_lds_kernel": (112, 0), # the same with its frames' tables and bitstream rings in LDS, ONE wave per CU beside it (no occupancy to protect: 98 with the ring's piece in flight)
    "zra_dec_huf_kernel": (88, 0),       # wave-wide literal decode (two stream readers while a restarted lane looks for its previous path); 8.5 KiB of LDS per workgroup is its occupancy limit
    "zra_dec_parse_kernel": (96, 256),   # 5 waves per SIMD asked for: 40 spilled VGPRs bought 2.1x on the stage (frames in flight are what it needs); round 5: 63 with libzstd's FSE_readNCount restated in full
    "zra_dec_exec_kernel": (80, 160),    # 6 waves per SIMD (the LDS-window step and the in-memory one side by side: 37 spilled VGPRs, 13.3 vs 14.3 ms at 5 waves)
    "zra_ra_small_kernel": (256, 0),     # one-launch path for small batches: all stages of a frame in one workgroup, occupancy is not its point
    "zra_entropy_front_kernel": (96, 512),  # round 6, the split stage of the persistent pipeline: the same body up to the table descriptions ...
    "zra_entropy_back_kernel": (96, 512),   # ... and from the sequence bitstream on; same budget as the one-launch kernel (one wave of it per SIMD beside the finder's five)
    "zra_ent_chain_kernel": (64, 0),        # lane = (frame, stream) state chains between the two: one wave per workgroup, its LDS (21 KiB of tables) is the occupancy limit
    "zra_entropy_kernel": (96, 512),     # the entropy stage's queue-driven workgroups: one wave of it per SIMD next to five of the match finder's (5 x 80 + 96 <= 512); the scratch is the frame body's call frame (its copy of the argument block) + a few spilled registers      # 5 waves per SIMD asked for: no spills (7 cost 8 spilled VGPRs + 72 B scratch and 2 % of the bench)
}


def test_hot_kernels_stay_inside_their_register_and_scratch_budgets():
    from zra_amd import build
    build.build()
    with open(build.RESOURCES) as f:
        res = json.load(f)
    for name, (vg, scratch) in BUDGET.items():
        assert name in res, "kernel %s missing from the build (renamed?)" % name
        r = res[name]
        assert r["vgprs"] <= vg, (name, r)
        assert r["scratch_bytes"] <= scratch, (name, r)
        assert r["vgpr_spill"] == 0 or scratch >= 64, (name, r)        # spills only where the budget line above says they were bought
