#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3j5
mkdir -p $O
cd $R
export EVH_BENCH_CACHE=/tmp/evh_bench_cache
EVHIP_LIBRARY=$R/tools/ab/two.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "fast or orb_keypoints or tied or shared or hint or random_geometry or odd_and_small or other_key_point or pair_batch" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
bash tools/ab/run_ab.sh two 2>&1 | tee $O/ab.txt
AB_FLAGS=--sync-solve bash tools/ab/run_ab.sh two 2>&1 | tee $O/ab_sync.txt
cd /tmp && export TMPDIR=/tmp
for v in tree two; do
  if [ $v = tree ]; then unset EVHIP_LIBRARY; else export EVHIP_LIBRARY=$R/tools/ab/$v.so; fi
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU -d $O/sq_$v -o sq --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-pairs 0 --gen-procs 1 --skip-no-temporal > $O/sq_$v.log 2>&1
  python3 - <<PY
import csv,glob,collections
f=glob.glob("$O/sq_$v/**/*counter_collection.csv",recursive=True)[0]
acc=collections.defaultdict(float); n=collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    if "k_fast_main" in r["Kernel_Name"]:
        acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
print("$v", {k:(v/len(n[k])) for k,v in acc.items()})
PY
done
