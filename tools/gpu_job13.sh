#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3j13
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_sift.py tests/test_gpu_parity.py -x -q -k "knn or float or multi_type or surf or alternative or reference_default" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
EVH_RANSAC_PROF=1 timeout -k 10 300 python tools/types_probe.py 400x224 > $O/types_prof.log 2>&1
grep -E "prof\]" $O/types_prof.log | tail -4
grep -E "pairs_per_s|x" $O/types_prof.log | tr -d '\n '; echo
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof -o types -- python $R/tools/types_probe.py 400x224 > $O/rocprof.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("$O/prof/**/types_kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:10]: print(r["Name"][:70],r["Calls"],r["TotalDurationNs"],r["AverageNs"],r["Percentage"])
PY
