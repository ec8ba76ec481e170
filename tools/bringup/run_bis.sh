cd $GRAFT_REPO_ROOT
sed -i 's/    data = {"bench": corp\[:3 << 20\], "C": C.gen_C(1 << 20), "E": C.gen_E(1 << 20).*/    data = {"C": C.gen_C(1 << 20), "log": C.gen_loglike(1 << 20)}/' tools/bringup/gpu_lk_check.py
sed -i '/"F": C.gen_struct(1 << 19), "log"/d' tools/bringup/gpu_lk_check.py
for v in X2 X3 X3a X3b X3c X4 X5; do echo -n "$v: "; ZRA_AMD_BRINGUP=1 ZRA_AMD_LIB=$GRAFT_REPO_ROOT/zra_amd/libzra_amd_$v.so timeout 200 python tools/bringup/gpu_lk_check.py 0 2>&1 | grep "TOTAL BAD"; done
