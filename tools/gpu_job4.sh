#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3j4
mkdir -p $O
cd $R
export EVH_BENCH_CACHE=/tmp/evh_bench_cache
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "fast or orb_keypoints or tied or shared or hint or random_geometry or odd_and_small or other_key_point or pair_batch or other_baseline" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
bash tools/ab/run_ab.sh a 2>&1 | tee $O/ab.txt
AB_FLAGS=--sync-solve bash tools/ab/run_ab.sh a 2>&1 | tee $O/ab_sync.txt
