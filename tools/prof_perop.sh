#!/bin/bash
# rocprofv3 --kernel-trace --stats of the per-op (autograd) path:  tools/prof_perop.sh <name> [bench args]
set -e
name=$1; shift
root=$(pwd); out=$root/gpurun_out/$name; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp_$name -- python3 $root/bench.py --autograd --no-cpu-baseline --psnr-iters 0 --burnin 600 --steps 300 "$@" > $out/bench.log 2>&1
f=$(find /tmp/pp_$name -name '*kernel_stats.csv' | head -1)
python3 - "$f" > $out/kernel_stats.csv <<'PY'
import csv, sys
print("kernel,calls,total_ms,avg_us,percent")
for r in list(csv.DictReader(open(sys.argv[1])))[:36]:
    print(f"\"{r['Name'][:100]}\",{r['Calls']},{float(r['TotalDurationNs']) / 1e6:.2f},{float(r['AverageNs']) / 1e3:.2f},{r['Percentage']}")
PY
rm -rf /tmp/pp_$name
tail -1 $out/bench.log | cut -c1-300
