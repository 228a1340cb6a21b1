#!/bin/bash
# rocprofv3 --kernel-trace --stats of the config-4-style step (tools/pose_refine.py):  tools/prof_config4.sh <name>
set -e
name=${1:-config4}
root=$(pwd); out=$root/gpurun_out/$name; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc_$name -- python3 $root/tools/pose_refine.py --iters 600 --log-every 0 "${@:2}" > $out/run.log 2>&1
f=$(find /tmp/pc_$name -name '*kernel_stats.csv' | head -1)
python3 - "$f" > $out/kernel_stats.csv <<'PY'
import csv, sys
print("kernel,calls,total_ms,avg_us,percent")
for r in list(csv.DictReader(open(sys.argv[1])))[:40]:
    print(f"\"{r['Name'][:110]}\",{r['Calls']},{float(r['TotalDurationNs']) / 1e6:.2f},{float(r['AverageNs']) / 1e3:.2f},{r['Percentage']}")
PY
rm -rf /tmp/pc_$name
grep '^{' $out/run.log | tail -1 > $out/run.json; cat $out/run.json
