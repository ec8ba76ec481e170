#!/bin/bash
# round 6, session 2: the epoch A/B again, five processes each (session 1's three were noisy: 857-925 ms for the new build)
bash tools/ab.sh -v r5 -v A -v A:ZRA_MF_EPOCH=0 -r 5 -o r06_ab_epoch_b.txt
