#!/bin/bash
# kernel trace of the step with the split-operand transform (dev tool)
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
cd /tmp && export TMPDIR=/tmp
export WDG_MLP2_SPLIT=${1:-1}
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/mlp2_split_$WDG_MLP2_SPLIT" -- python3 "$ROOT/bench.py" --steps 100 --warmup 10 --secondary 0 --full-metrics 0 --cpu-budget 0 --cold 0 > /dev/null 2>&1
f=$(ls -t $(find "$ROOT/gpurun_out/mlp2_split_$WDG_MLP2_SPLIT" -name "*kernel_stats.csv") | head -1)
head -6 "$f" | cut -c1-160
