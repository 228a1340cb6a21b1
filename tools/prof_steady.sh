#!/bin/bash
# Steady-state per-step kernel profile of bench.py on the GPU box:  tools/prof_steady.sh <name> [bench args...]
# writes gpurun_out/<name>/steady.csv (the full trace is deleted: it exceeds the gpurun_out size cap)
set -e
name=$1; shift
root=$(pwd)
out=$root/gpurun_out/$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$name -- python3 $root/bench.py --steps 100 --warmup 5 --burnin 4895 --psnr-iters 0 --no-cpu-baseline "$@" > $out/bench.log 2>&1
trace=$(find /tmp/prof_$name -name '*kernel_trace.csv' | head -1)
python3 $root/tools/trace_tail.py $trace --anchor composite_backward_wave --steps 100 --split grid_update_kernel --out $out/steady.csv --timeline $out/timeline.csv > $out/steady.txt
rm -rf /tmp/prof_$name
