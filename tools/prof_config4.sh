#!/bin/bash
# rocprofv3 --kernel-trace --stats of the config-4-style run (light-conditioned field + BARF + HDR loss):  tools/prof_config4.sh <name>
set -e
name=$1; shift
root=$(pwd); out=$root/gpurun_out/$name; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc4_$name -- python3 $root/tools/pose_refine.py --iters 1500 --hdr --log-every 0 "$@" > $out/run.log 2>&1
f=$(find /tmp/pc4_$name -name '*kernel_stats.csv' | head -1)
python3 - "$f" > $out/kernel_stats.csv <<'PY'
import csv, sys
print("kernel,calls,total_ms,avg_us,percent")
for r in list(csv.DictReader(open(sys.argv[1])))[:28]:
    print(f"\"{r['Name'][:70]}\",{r['Calls']},{float(r['TotalDurationNs']) / 1e6:.2f},{float(r['AverageNs']) / 1e3:.2f},{r['Percentage']}")
PY
rm -rf /tmp/pc4_$name
tail -3 $out/run.log | cut -c1-300
