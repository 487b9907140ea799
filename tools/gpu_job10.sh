#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3j10
mkdir -p $O
cd $R
EVH_RANSAC_PROF=1 timeout -k 10 300 python tools/stream_probe.py 1280x720:2000:1 > $O/probe_prof.log 2>&1
grep -E "prof\]" $O/probe_prof.log | tail -3
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof -o forced -- python $R/tools/stream_probe.py 1280x720:2000:1 > $O/rocprof.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("$O/prof/**/forced_kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]: print(r["Name"][:60],r["Calls"],r["TotalDurationNs"],r["AverageNs"])
PY
