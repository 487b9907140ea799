#!/bin/bash
# A/B of builds of libevhip.so on the SAME GPU box (box-to-box spread is several percent):
#   cp evenvizion_amd/libevhip.so tools/ab/a.so      # baseline, before a change
#   ... edit, make ...                               # candidate = the in-tree library ("tree")
#   gpurun -- 'bash tools/ab/run_ab.sh a'            # any number of names from tools/ab/*.so; "tree" is always run
#   AB_FLAGS=--sync-solve gives per-stage times free of the overlapped RANSAC kernels of the previous step
set -e
for rep in 1 2; do
  for v in "$@" tree; do
    if [ $v = tree ]; then unset EVHIP_LIBRARY; else export EVHIP_LIBRARY=$PWD/tools/ab/$v.so; fi
    python bench.py --steps 10 --warmup 2 --cpu-pairs 0 --skip-no-temporal $AB_FLAGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stage_ms']
print('$v', d['value'], d['ms_per_step'], {k: round(x,2) for k,x in s.items()})"
  done
done
