#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3j11
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sift.py -x -q -k "fixed_iter or config or failure or sharded or two_phase or multi_stream or find_h or static" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for v in a tree; do
  if [ $v = tree ]; then unset EVHIP_LIBRARY; else export EVHIP_LIBRARY=$R/tools/ab/$v.so; fi
  echo "== $v"
  timeout -k 10 300 python tools/stream_probe.py 1280x720:2000:1 1280x720:500:1 3840x2160:4000:1 2>/dev/null | grep -E "pairs_per_s|ransac_final_ms|ransac_static|x" | tr -d '\n '; echo
done
