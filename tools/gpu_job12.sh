#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3j12
mkdir -p $O
cd $R
EVH_PYR_TWO=1 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "pyramid or unaligned or geometry_sweep or orb_detect or pair_batch or resize" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for rep in 1 2; do
  for v in walk two; do
    if [ $v = two ]; then export EVH_PYR_TWO=1; else unset EVH_PYR_TWO; fi
    for fl in "" "--sync-solve"; do
    python bench.py --steps 10 --warmup 2 --cpu-pairs 0 --skip-no-temporal $fl 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stage_ms']
print('$v', '$fl', d['value'], d['ms_per_step'], {k: round(x,2) for k,x in s.items()})"
    done
  done
done
