#!/bin/bash
# same-box A/B of bench.py variants: tools/ab.sh OUT "ARGS A" "ARGS B" [repeats]   (each run: --no-secondary --no-cpu-baseline)
out=$1; a=$2; b=$3; n=${4:-2}
: > "$out"
for i in $(seq 1 "$n"); do
  for v in "$a" "$b"; do
    line=$(python bench.py --no-secondary --no-cpu-baseline $v 2>/dev/null | tail -1)
    python - "$v" "$line" >> "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
r, f = d.get("roofline") or {}, d.get("roofline_forward") or {}
print(f"{sys.argv[1]!r:40s} ms/step {d['ms_per_step']:.4f}  Mrays/s {d['value'] / 1e6:.3f}  psnr {d.get('psnr', {}).get('value')}  "
      f"table-bwd {r.get('avg_us')} us  fwd {f.get('avg_us')} us  samples {d['config']['samples_per_step']}  scaler {d['config'].get('loss_scaler')}")
PY
  done
done
cat "$out"
