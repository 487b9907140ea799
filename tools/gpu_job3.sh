#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3j3
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "pyramid or random_geometry or unaligned or odd_and_small or fused_ingest or other_baseline or orb_keypoints" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
EVH_PYR_OLD=1 timeout -k 10 300 python bench.py --steps 10 --cpu-pairs 0 --skip-no-temporal > $O/bench_old.json 2> $O/bench_old.err
timeout -k 10 300 python bench.py --steps 10 --cpu-pairs 0 --skip-no-temporal > $O/bench_new.json 2> $O/bench_new.err
EVH_PYR_OLD=1 timeout -k 10 300 python bench.py --steps 10 --cpu-pairs 0 --skip-no-temporal --sync-solve > $O/bench_old_sync.json 2> $O/bench_old_sync.err
timeout -k 10 300 python bench.py --steps 10 --cpu-pairs 0 --skip-no-temporal --sync-solve > $O/bench_new_sync.json 2> $O/bench_new_sync.err
python3 - <<PY
import json
for n in ("old","new","old_sync","new_sync"):
    d=json.loads(open("$O/bench_%s.json"%n).read().strip().splitlines()[-1])
    print(n, d["value"], d["ms_per_step"], d["roofline"]["stage_ms"])
PY
