#!/bin/bash
# One counter pass over tools/grid_bench.py:  tools/pmc_run.sh <name> "<COUNTER ...>" [kernel-name filters, comma separated] [extra grid_bench flags]
# (separate passes per counter group: gfx950 has 8 SQ / 4 TCC slots; never combined with --sys-trace etc.)
set -e
name=$1; ctrs=$2; match=${3:-}; extra=${4:-}
root=$(pwd); out=$root/gpurun_out/$name; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 ${PMC_TIMEOUT:-240} rocprofv3 --pmc $ctrs --output-format csv -d /tmp/pmc_$name -- python3 $root/tools/grid_bench.py --iters 8 $extra > $out/run.log 2>&1
f=$(find /tmp/pmc_$name -name '*counter_collection.csv' | head -1)
python3 $root/tools/pmc_summary.py $f --last 6 --match "$match" --out $out/pmc.csv > /dev/null
rm -rf /tmp/pmc_$name
echo "[pmc] $name done"
