#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3j6
mkdir -p $O
cd $R
# BASELINE configs[2] at its full length: 10 032 pairs of one 720p stream (209 steps of 48 pairs), ORB 2000
timeout -k 10 600 python bench.py --config 3 --steps 209 --cpu-pairs 0 > $O/cfg3_full.json 2> $O/cfg3_full.err
tail -1 $O/cfg3_full.json | cut -c1-300
# configs[4] shape at its full length on one GPU: 1 008 pairs of one 4K stream, ORB 4000
timeout -k 10 600 python bench.py --config 5 --steps 63 --cpu-pairs 0 > $O/cfg5_full.json 2> $O/cfg5_full.err
tail -1 $O/cfg5_full.json | cut -c1-300
timeout -k 10 300 python tools/stream_probe.py > $O/stream_probe.json 2> $O/stream_probe.err
cat $O/stream_probe.json | head -60
timeout -k 10 300 python tools/e2e_probe.py > $O/e2e_probe.json 2> $O/e2e_probe.err || true
cat $O/e2e_probe.json
timeout -k 10 300 python tools/multi_stream_probe.py > $O/multi_stream_probe.json 2> $O/multi_stream_probe.err || true
cat $O/multi_stream_probe.json
timeout -k 10 300 python tools/types_probe.py 400x224 > $O/types_probe.json 2> $O/types_probe.err
cat $O/types_probe.json
