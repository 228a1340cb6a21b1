#!/bin/bash
# One counter pass over bench.py's steady state:  tools/pmc_bench.sh <name> "<COUNTERS>" "<kernel filters>" [bench flags]
set -e
name=$1; ctrs=$2; match=$3; shift 3
root=$(pwd); out=$root/gpurun_out/$name; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 ${PMC_TIMEOUT:-420} rocprofv3 --pmc $ctrs --output-format csv -d /tmp/pmcb_$name -- python3 $root/tools/pmc_bench.py $out --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --psnr-iters 0 --no-probe "$@" > $out/run.log 2>&1 || echo "[pmc_bench] exit code $?" >> $out/run.log
f=$(find /tmp/pmcb_$name -name '*counter_collection.csv' | head -1)
[ -n "$f" ] && python3 $root/tools/pmc_summary.py $f --last 30 --match "$match" --out $out/pmc.csv > /dev/null
rm -rf /tmp/pmcb_$name
tail -3 $out/run.log | cut -c1-300
echo "[pmc_bench] $name done"
