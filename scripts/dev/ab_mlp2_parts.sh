#!/bin/bash
# the sweep step with the fused transform cut into 4 (default) / 5 / 6 row parts per job (dev tool)
cd "$(dirname "$0")/../.."
for rep in 1 2; do
for p in 4 5 6; do
  WDG_GEMM_PARTS=$p python bench.py --steps 200 --warmup 20 --secondary 0 --full-metrics 0 --cpu-budget 0 --cold 0 2>/dev/null | tail -1 > /tmp/ab_p.json
  python - "$p" <<'PY'
import json, sys
d = json.loads(open("/tmp/ab_p.json").read())
print("WDG_GEMM_PARTS=" + sys.argv[1], "edges/s %.3e" % d["value"], "ms/step %.4f" % d["ms_per_step"])
PY
done
done
