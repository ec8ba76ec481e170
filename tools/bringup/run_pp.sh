cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ZRA_MF_FLAGS=1 ZRA_PP_MIN=1 timeout 120 python tools/bringup/gpu_flags_check.py 2>&1 | grep -v amdgpu.ids | tail -2
echo -n "link parity: "; ZRA_MF_LK=1 timeout 300 python tools/bringup/gpu_lk_check.py 0 2>&1 | grep "TOTAL BAD"
echo -n "flags parity: "; ZRA_MF_FLAGS=1 ZRA_PP_MIN=1 timeout 300 python tools/bringup/gpu_lk_check.py 0 2>&1 | grep "TOTAL BAD"
for cfg in "ZRA_MF_FLAGS=1 ZRA_PP_SAMESTREAM=1 ZRA_PP_CUS=64"; do
  tag=pp
  rm -rf /tmp/kst_$tag
  ( cd /tmp && env $cfg timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst_$tag -o k -- python3 $GRAFT_REPO_ROOT/tools/bringup/gpu_speed.py 1 3 65536 3 > /tmp/out_$tag.txt 2>/dev/null < /dev/null )
  echo "== $cfg"; grep compress /tmp/out_$tag.txt | tail -1 | cut -c1-100
  f=$(find /tmp/kst_$tag -name '*kernel_stats.csv' | head -1)
  if [ -n "$f" ]; then grep "dfast\|prepass" "$f" | cut -c1-120; fi
done
