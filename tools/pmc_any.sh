#!/bin/bash
# One counter pass over any driver script:  tools/pmc_any.sh <name> "<COUNTERS>" "<kernel filters>" <script> [args...]
set -e
name=$1; ctrs=$2; match=$3; shift 3
root=$(pwd); out=$root/gpurun_out/$name; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 ${PMC_TIMEOUT:-240} rocprofv3 --pmc $ctrs --output-format csv -d /tmp/pmc_$name -- python3 $root/"$@" > $out/run.log 2>&1
f=$(find /tmp/pmc_$name -name '*counter_collection.csv' | head -1)
python3 $root/tools/pmc_summary.py $f --last 6 --match "$match" --out $out/pmc.csv > /dev/null
rm -rf /tmp/pmc_$name
echo "[pmc] $name done"
