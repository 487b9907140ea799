#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for rep in 1 2; do
for v in tree mw256 mw128; do
  if [ $v = tree ]; then unset EVHIP_LIBRARY; else export EVHIP_LIBRARY=$R/tools/ab/$v.so; fi
  echo "== $v"
  python tools/stream_probe.py 1280x720:500:0 1280x720:2000:0 2>/dev/null | grep -E "pairs_per_s|ransac_final_ms|x" | tr -d '\n '; echo
  python tools/types_probe.py 400x224 2>/dev/null | grep -E "pairs_per_s|x" | tr -d '\n '; echo
done
done
