#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "static or pair_batch or stream or failure or geometry or config" 2>&1 | tail -3
for rep in 1 2; do
for v in a tree; do
  if [ $v = tree ]; then unset EVHIP_LIBRARY; else export EVHIP_LIBRARY=$R/tools/ab/$v.so; fi
  python bench.py --steps 10 --warmup 2 --cpu-pairs 0 --skip-no-temporal 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stage_ms']
print('$v', d['value'], d['ms_per_step'], {k: round(x,2) for k,x in s.items()})"
done
done
