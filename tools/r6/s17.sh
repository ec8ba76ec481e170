#!/bin/bash
# round 6, session 17: with the entropy workgroup at 14 LDS pieces, finder waves of 5 pieces (6,400 B) leave room for 20-22 of them:
# one box, alternating against the default (19 waves of 6 pieces)
bash tools/ab.sh -v A -v A:ZRA_MF_WAVES=20+ZRA_MF_FILTER=1,4,8+ZRA_MF_SPAN=0 -v A:ZRA_MF_WAVES=22+ZRA_MF_FILTER=1,4,8+ZRA_MF_SPAN=0 \
  -v A:ZRA_MF_WAVES=22+ZRA_MF_FILTER=1,3,7+ZRA_MF_SPAN=512 -v A:ZRA_MF_WAVES=21+ZRA_MF_FILTER=1,4,8+ZRA_MF_SPAN=0 -r 2 -o r06_ab_w22.txt
