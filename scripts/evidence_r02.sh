#!/bin/bash
# Everything profiles/ holds for round 2, from one box (run ON the GPU box from the repo root): bash scripts/evidence_r02.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
bash scripts/profile_r02.sh r02_c3 > gpurun_out/ev_c3.log 2>&1
bash scripts/profile_r02.sh r02_c2 --k 2 --seeds 10 > gpurun_out/ev_c2.log 2>&1
timeout 300 python3 scripts/bench_configs.py --out gpurun_out/r02_configs.jsonl > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
for cfg in C1 C4 C5; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/r02_${cfg}_stats" -- python3 "$ROOT/scripts/bench_configs.py" --only $cfg > /dev/null 2>&1
  find "$ROOT/gpurun_out/r02_${cfg}_stats" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$ROOT/gpurun_out/r02_${cfg}_kernel_stats.csv"
done
cd "$ROOT"
bash scripts/pmc_configs.sh r02_c5_pmc C5 > gpurun_out/ev_c5_pmc.log 2>&1
bash scripts/pmc_configs.sh r02_c4_pmc C4 > gpurun_out/ev_c4_pmc.log 2>&1
cut -c1-260 gpurun_out/r02_configs.jsonl
tail -3 gpurun_out/ev_c3.log | cut -c1-300
