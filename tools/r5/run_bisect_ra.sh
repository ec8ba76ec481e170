#!/bin/bash
# round 5: FSE_readNCount restated with libzstd's end-of-input behaviour — the soak's seed, the decode-side tests, fresh damaged-archive seeds
root=$(pwd); out=$root/gpurun_out/r5_ncount.txt; : > $out; export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "random_access_on_damaged or ra_vs_bruteforce or randomised_corruption or randomised_header or damaged_frame_size or inflated_frame or differential_decode or golden_frames or libzstd_frames or small_inputs_through" 2>&1 | tail -3 >> $out
echo "== seeds 91410..91420" >> $out
timeout 200 python3 tools/bringup/gpu_soak_ra_damage.py 91410 91420 2>&1 | grep -v amdgpu.ids | grep "MISMATCH\|soak:" | tail -3 >> $out
cat $out
SOAK_SEEDS=0.4 SOAK_TIMEOUT=300 bash tools/soak.sh -b 95000 -o r5_soak_r.txt corrupt ra_damage headers
SOAK_SEEDS=0.3 SOAK_TIMEOUT=300 bash tools/soak.sh -b 96000 -o r5_soak_s.txt -e ZRA_DEC_SMALL_MAX=0+ZRA_DEC_CHAIN_LDS_MIN=1 corrupt ra_damage
