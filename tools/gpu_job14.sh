#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3j14
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sift.py -x -q -k "homography or stream or multi_type or find_h or config or reference_default" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
EVH_RANSAC_PROF=1 timeout -k 10 300 python tools/types_probe.py 400x224 > $O/types_prof.log 2>&1
grep -E "prof\]" $O/types_prof.log | awk 'NR%4==0' 
for v in a tree; do
  if [ $v = tree ]; then unset EVHIP_LIBRARY; else export EVHIP_LIBRARY=$R/tools/ab/$v.so; fi
  echo "== $v"
  python tools/stream_probe.py 1280x720:500:0 1280x720:2000:0 3840x2160:4000:0 2>/dev/null | grep -E "pairs_per_s|ransac_final_ms|x" | tr -d '\n '; echo
  python tools/types_probe.py 400x224 2>/dev/null | grep -E "pairs_per_s|x" | tr -d '\n '; echo
done
