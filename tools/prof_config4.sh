#!/bin/bash
# Steady-state kernel table of the config-4-style run (light-conditioned field + BARF + HDR loss):  tools/prof_config4.sh <name> [pose_refine args]
# last 100 steps of 3000 through tools/trace_tail.py -> gpurun_out/<name>/steady.csv
set -e
name=$1; shift
root=$(pwd); out=$root/gpurun_out/$name; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/pc4_$name -- python3 $root/tools/pose_refine.py --iters 3000 --hdr --log-every 0 "$@" > $out/run.log 2>&1
trace=$(find /tmp/pc4_$name -name '*kernel_trace.csv' | head -1)
python3 $root/tools/trace_tail.py $trace --anchor composite_backward_wave --steps 100 --split grid_update_kernel --out $out/steady.csv > $out/steady.txt
rm -rf /tmp/pc4_$name
tail -2 $out/run.log | cut -c1-300
