#!/bin/bash
# round-3 job 1: GPU tests, FETCH_SIZE calibration on FAST's access pattern, counter list, baseline bench
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3j1
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/cal -o cal --output-format csv -- $R/tools/ubench/fetch_cal > $O/cal.log 2>&1
cat $O/cal.log | grep pattern
(rocprofv3 -L > $O/counters.txt 2>&1 || true)
grep -i "TCC_EA0_RDREQ\|TCC_REQ\|FETCH_SIZE\|TCC_EA0_RD" $O/counters.txt | head -40 || true
cd $R && timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json
