#!/bin/bash
# Everything profiles/ holds for round 3, from one box (run ON the GPU box from the repo root): bash scripts/evidence_r03.sh
# (one rank only: the profiled process never spawns workers - scripts/profile_pmc.sh refuses --gpus > 1)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
bash scripts/profile_r02.sh r03_c3 --cold 0 > gpurun_out/ev3_c3.log 2>&1                      # bench line + kernel stats + PMC passes, C3 headline
python3 bench.py > gpurun_out/r03_bench_full.json 2> gpurun_out/r03_bench_full.err           # the whole default line (secondary, sweep_full, sweep_cold, cpu)
timeout 600 python3 scripts/bench_configs.py --out gpurun_out/r03_configs.jsonl > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/r03_kr_stats" -- python3 "$ROOT/scripts/dev/time_kr_batch.py" 5 100 > "$ROOT/gpurun_out/r03_kr.log" 2>&1
find "$ROOT/gpurun_out/r03_kr_stats" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$ROOT/gpurun_out/r03_kr_kernel_stats.csv"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/r03_cold_stats" -- python3 "$ROOT/bench.py" --steps 10 --warmup 2 --secondary 0 --full-metrics 0 --cpu-budget 0 > /dev/null 2>&1
find "$ROOT/gpurun_out/r03_cold_stats" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$ROOT/gpurun_out/r03_cold_kernel_stats.csv"
for cfg in C1m C4m C5 C5m C3-literal; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/r03_${cfg}_stats" -- python3 "$ROOT/scripts/bench_configs.py" --only $cfg > /dev/null 2>&1
  find "$ROOT/gpurun_out/r03_${cfg}_stats" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$ROOT/gpurun_out/r03_${cfg}_kernel_stats.csv"
done
cd "$ROOT"
bash scripts/pmc_configs.sh r03_c5_pmc C5 > /dev/null 2>&1          # C5: both variants; C5cal: the FETCH_SIZE calibration run
bash scripts/pmc_configs.sh r03_c5cal_pmc C5cal > /dev/null 2>&1
cut -c1-200 gpurun_out/r03_configs.jsonl | head -30
tail -3 gpurun_out/ev3_c3.log | cut -c1-300
