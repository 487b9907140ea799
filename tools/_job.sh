#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout -k 10 1000 python -m pytest tests/test_gpu_sift.py -x -q 2>&1 | tail -4
for v in a tree; do
  if [ $v = tree ]; then unset EVHIP_LIBRARY; else export EVHIP_LIBRARY=$R/tools/ab/$v.so; fi
  echo "== $v"
  python tools/types_probe.py 400x224 2>/dev/null | grep -E "pairs_per_s|x" | tr -d '\n '; echo
  python tools/types_probe.py 1280x720 2>/dev/null | grep -E "pairs_per_s|x" | tr -d '\n '; echo
done
