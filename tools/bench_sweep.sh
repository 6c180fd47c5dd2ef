#!/bin/bash
# usage: tools/bench_sweep.sh OUT "ENV=.. ENV=.." ...   - one bench run per environment setting, value / value_depth1 per line
out=$1; shift
for e in "$@"; do
  env $e python bench.py --steps ${SWEEP_STEPS:-300} --warmup ${SWEEP_WARMUP:-10} --no-cpu-baseline 2>/dev/null > /tmp/sweep_line.json || exit 1
  python - "$e" >> "$out" <<'PY'
import json, sys
d = json.loads(open('/tmp/sweep_line.json').read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"]), round(d.get("value_depth1") or 0))
PY
done
