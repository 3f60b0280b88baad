#!/bin/bash
# the sweep step with the fused transform on the fp32 matrix pipe (0) / on split bf16 operands (1), same box (dev tool)
cd "$(dirname "$0")/../.."
for rep in 1 2; do
for s in 0 1; do
  WDG_MLP2_SPLIT=$s python bench.py --steps 200 --warmup 20 --secondary 0 --full-metrics 0 --cpu-budget 0 --cold 0 2>/dev/null | tail -1 > /tmp/ab_$s.json
  python - "$s" <<'PY'
import json, sys
d = json.loads(open(f"/tmp/ab_{sys.argv[1]}.json").read())
print("WDG_MLP2_SPLIT=" + sys.argv[1], "edges/s %.3e" % d["value"], "ms/step %.4f" % d["ms_per_step"], "aggregation launch %.1f us" % d["roofline"]["avg_launch_us"])
PY
done
done
