#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_sift.py -x -q 2>&1 | tail -2
python tools/types_probe.py 400x224 2>/dev/null | grep -E "pairs_per_s|x" | tr -d '\n '; echo
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r3j9
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3j9/stats -o types --output-format csv -- python3 $R/tools/types_probe.py 400x224 > $R/gpurun_out/r3j9/stats.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/r3j9/stats/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:12]: print(r["Name"][:60].ljust(60), r["Calls"].rjust(6), r["TotalDurationNs"].rjust(12), r["AverageNs"].rjust(10), r["Percentage"])
PY
