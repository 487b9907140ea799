#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3j2
mkdir -p $O
cd $R
timeout -k 10 300 python tools/types_probe.py 400x224 > $O/types_probe.json 2> $O/types_probe.err || { tail -20 $O/types_probe.err; exit 1; }
cat $O/types_probe.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats -o types --output-format csv -- python3 $R/tools/types_probe.py 400x224 > $O/stats.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$O/stats/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:22]: print(r["Name"][:60].ljust(60), r["Calls"].rjust(6), r["TotalDurationNs"].rjust(12), r["AverageNs"].rjust(10), r["Percentage"])
PY
