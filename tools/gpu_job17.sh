#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3j17
mkdir -p $O
cd $R
EVH_RANSAC_PROF=1 timeout -k 10 300 python tools/types_probe.py 400x224 > $O/types_prof.log 2>&1
grep -E "prof\]" $O/types_prof.log | awk 'NR%4==0' 
