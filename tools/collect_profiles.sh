#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash tools/collect_profiles.sh' and, as a second call, 'bash tools/collect_profiles.sh stream'): rocprofv3 kernel stats + separate PMC passes of the
# default bench workload (under the profiler the synthetic pairs are generated in-process: --gen-procs 1, no child processes).  Raw output -> gpurun_out/prof/ ; tools/summarize_profiles.py turns it into profiles/*.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export EVH_BENCH_CACHE=/tmp/evh_bench_cache     # the 64 synthetic pairs are generated once (in parallel, outside the profiler)
if [ "$1" = "stream" ]; then      # second call (gpurun's 1200 s limit): only the PMC passes of the stream workloads
(cd $R && python bench.py --config 3 --steps 1 --warmup 1 --cpu-pairs 0 --skip-no-temporal > $O/prime3.json 2> $O/prime3.err)
# FETCH_SIZE / WRITE_SIZE passes of the stream workloads (configs 3 and 5): per-step HBM bytes of the scan kernel and of the whole step
(cd $R && python bench.py --config 5 --steps 1 --warmup 1 --cpu-pairs 0 --skip-no-temporal > $O/prime5.json 2> $O/prime5.err)
for c in 3 5; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch$c -o fetch$c --output-format csv -- python3 $R/bench.py --config $c --steps 2 --warmup 1 --cpu-pairs 0 --gen-procs 1 --skip-no-temporal > $O/fetch$c.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write$c -o write$c --output-format csv -- python3 $R/bench.py --config $c --steps 2 --warmup 1 --cpu-pairs 0 --gen-procs 1 --skip-no-temporal > $O/write$c.log 2>&1
done
echo stream-passes-done
exit 0
fi
(cd $R && python bench.py --steps 2 --warmup 1 --cpu-pairs 0 --skip-no-temporal > $O/prime.json 2> $O/prime.err)
(cd $R && python bench.py --config 3 --steps 1 --warmup 1 --cpu-pairs 0 > $O/prime3.json 2> $O/prime3.err)   # primes the config-3 cache outside the profiler
rocprofv3 --kernel-trace --stats -d $O/stats -o stats --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --gen-procs 1 --unique 64 --skip-no-temporal > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o fetch --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-pairs 0 --gen-procs 1 --unique 64 --skip-no-temporal > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o write --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-pairs 0 --gen-procs 1 --unique 64 --skip-no-temporal > $O/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS -d $O/sq -o sq --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-pairs 0 --gen-procs 1 --unique 64 --skip-no-temporal > $O/sq.log 2>&1
# exact read requests by size (cross-check of the FETCH_SIZE correction: bytes = 32*n32 + 64*n64 + 128*n128); may be refused
# when the three do not fit one pass -- then the calibrated factor of profiles/r03_fetch_calibration.txt stands alone
(rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d $O/rdreq -o rdreq --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-pairs 0 --gen-procs 1 --unique 64 --skip-no-temporal > $O/rdreq.log 2>&1 || true)
# the stream workload (BASELINE configs[2]): kernel stats only
rocprofv3 --kernel-trace --stats -d $O/stats3 -o stats3 --output-format csv -- python3 $R/bench.py --config 3 --steps 4 --warmup 1 --cpu-pairs 0 --gen-procs 1 > $O/stats3.log 2>&1
cd $R && python bench.py > $O/bench.json 2> $O/bench.err
for c in 2 3 4 5; do python bench.py --config $c > $O/bench_cfg$c.json 2> $O/bench_cfg$c.err || true; done
python bench.py --config 3 --force-max-iters > $O/bench_cfg3_forced.json 2> $O/bench_cfg3_forced.err || true
# probes: stream cases (adaptive / forced), several streams per call, host frames in / dict out, the three detector types
python tools/stream_probe.py > $O/stream_probe.json 2> $O/stream_probe.err || true
python tools/stream_probe.py 3840x2160:4000:0 3840x2160:4000:1 > $O/stream_probe_4k.json 2>> $O/stream_probe.err || true
python tools/multi_stream_probe.py > $O/multi_stream_probe.json 2> $O/multi_stream_probe.err || true
python tools/e2e_probe.py > $O/e2e_probe.json 2> $O/e2e_probe.err || true
python tools/types_probe.py 400x224 > $O/types_probe.json 2> $O/types_probe.err || true
python tools/types_probe.py 1280x720 > $O/types_probe_720p.json 2> $O/types_probe_720p.err || true
(cd /tmp && rocprofv3 --kernel-trace --stats -d $O/types_stats -o types --output-format csv -- python3 $R/tools/types_probe.py 400x224 > $O/types_stats.log 2>&1 || true)
tail -1 $O/bench.json
