#!/bin/bash
# same-box A/B of bench.py under two environments: tools/ab_env.sh OUT "VAR=1 ..." "VAR=0 ..." [repeats] [bench args...]
out=$1; a=$2; b=$3; n=${4:-2}; shift 4
: > "$out"
for i in $(seq 1 "$n"); do
  for v in "$a" "$b"; do
    line=$(env $v python bench.py --no-secondary --no-cpu-baseline "$@" 2>/dev/null | tail -1)
    python - "$v" "$line" >> "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
r, f = d.get("roofline") or {}, d.get("roofline_forward") or {}
print(f"{sys.argv[1]!r:40s} ms/step {d['ms_per_step']:.4f}  Mrays/s {d['value'] / 1e6:.3f}  psnr {d.get('psnr', {}).get('value')}  "
      f"table-bwd {r.get('avg_us')} us  fwd {f.get('avg_us')} us")
PY
  done
done
cat "$out"
