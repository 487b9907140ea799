#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
export EVH_BENCH_CACHE=/tmp/evh_bench_cache
timeout -k 10 300 python -m pytest tests/test_gpu_sift.py -x -q -k "fused_resize_at_the_reference" 2>&1 | tail -2
for rep in 1 2; do
for args in "--config 1" "--config 1 --contexts 2" "--config 2" "--config 2 --contexts 2" "--config 2 --contexts 3"; do
  python bench.py $args --cpu-pairs 0 --skip-no-temporal 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$args', d['value'], d['ms_per_step'])"
done; done
