cd $GRAFT_REPO_ROOT
timeout 300 python tools/bringup/gpu_lk_check.py 1 > gpurun_out/lk_check2.log 2>&1; tail -6 gpurun_out/lk_check2.log
for w in 1 8; do echo "== profile ZRA_LK_WAVES=$w"; ZRA_LK_WAVES=$w ZRA_AMD_BRINGUP=1 ZRA_AMD_LIB=$GRAFT_REPO_ROOT/zra_amd/libzra_amd_prof.so timeout 200 python tools/bringup/gpu_lk_profile.py 0.25 2>&1 | grep -v amdgpu.ids; done > gpurun_out/lk_prof2.log 2>&1
cat gpurun_out/lk_prof2.log
