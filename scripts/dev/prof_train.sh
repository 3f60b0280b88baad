#!/bin/bash
# kernel trace of the batched training epochs (dev tool)
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/train_prof" -- python3 "$ROOT/scripts/dev/time_train_batch.py" > "$ROOT/gpurun_out/train_prof.log" 2>&1
f=$(ls -t $(find "$ROOT/gpurun_out/train_prof" -name "*kernel_stats.csv") | head -1)
cp "$f" "$ROOT/gpurun_out/train_kernel_stats.csv"
tail -5 "$ROOT/gpurun_out/train_prof.log"
