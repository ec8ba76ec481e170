#!/bin/bash
# round 6, session 3: wave-cooperative Huffman / FSE table builds in the entropy stage — the whole GPU suite, then one-box A/B against
# round 5's library in the default pipeline and with the stages in sequence (ZRA_PIPE=0: the entropy stage alone shows as "ent")
export TMPDIR=/tmp; mkdir -p gpurun_out
( timeout 2400 python3 -m pytest tests -q -x -m gpu -p no:cacheprovider < /dev/null 2>&1 | tail -8 ) > gpurun_out/r06_gputest_a.txt; cat gpurun_out/r06_gputest_a.txt
bash tools/ab.sh -v r5 -v A -v r5:ZRA_PIPE=0 -v A:ZRA_PIPE=0 -r 3 -o r06_ab_ent_a.txt
