#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3j16
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_sift.py tests/test_gpu_parity.py -x -q -k "sift or surf or knn or float or multi_type or reference_default" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for v in a tree; do
  if [ $v = tree ]; then unset EVHIP_LIBRARY; else export EVHIP_LIBRARY=$R/tools/ab/$v.so; fi
  echo "== $v"
  python tools/types_probe.py 400x224 2>/dev/null | grep -E "pairs_per_s|x" | tr -d '\n '; echo
done
unset EVHIP_LIBRARY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof -o types --output-format csv -- python3 $R/tools/types_probe.py 400x224 > $O/rocprof.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$O/prof/**/types_kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:9]: print(r["Name"][:70],r["Calls"],r["AverageNs"],r["Percentage"])
PY
