#!/bin/bash
# per-kernel averages of tools/grid_bench.py (kernels run alone, back to back):  tools/prof_grid.sh <name>
set -e
name=$1
root=$(pwd); out=$root/gpurun_out/$name; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pg_$name -- python3 $root/tools/grid_bench.py --iters 20 "${@:2}" > $out/run.log 2>&1
f=$(find /tmp/pg_$name -name '*kernel_stats.csv' | head -1)
python3 - "$f" > $out/kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:10]:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs']) / 1e3:8.1f}")
PY
rm -rf /tmp/pg_$name
cat $out/kernel_stats.csv
