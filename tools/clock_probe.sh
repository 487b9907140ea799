#!/bin/bash
# samples GPU clock / power with rocm-smi while the default bench runs (is the part power- or clock-limited here?)
python bench.py --steps 1500 --warmup 5 --cpu-pairs 0 --skip-no-temporal > gpurun_out/clk_bench.json 2>/dev/null &
BP=$!
: > gpurun_out/clk.log
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | sed -e 's/.*sclk clock level: [^(]*(\([0-9]*\)Mhz).*/sclk \1/' -e 's/.*Power (W): \([0-9.]*\).*/power \1/' | tr '\n' ' ' >> gpurun_out/clk.log
  echo >> gpurun_out/clk.log
  sleep 0.4
done
sort gpurun_out/clk.log | uniq -c | sort -k3 -n | tail -25
tail -c 200 gpurun_out/clk_bench.json
