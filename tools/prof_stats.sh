#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench command (whole run, burn-in included):  tools/prof_stats.sh <name>
set -e
name=$1; shift
root=$(pwd); out=$root/gpurun_out/$name; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps_$name -- python3 $root/bench.py --no-cpu-baseline "$@" > $out/bench.log 2>&1
f=$(find /tmp/ps_$name -name '*kernel_stats.csv' | head -1)
python3 - "$f" > $out/kernel_stats.csv <<'PY'
import csv, sys
print("kernel,calls,total_ms,avg_us,percent")
for r in list(csv.DictReader(open(sys.argv[1])))[:32]:
    print(f"\"{r['Name'][:90]}\",{r['Calls']},{float(r['TotalDurationNs']) / 1e6:.2f},{float(r['AverageNs']) / 1e3:.2f},{r['Percentage']}")
PY
rm -rf /tmp/ps_$name
