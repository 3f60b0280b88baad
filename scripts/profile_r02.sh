#!/bin/bash
# Round-2 evidence for the bench command (run ON the GPU box, from the repo root): the JSON line, the rocprofv3 kernel-trace
# statistics of the same command, and the PMC passes (scripts/profile_pmc.sh).  usage: bash scripts/profile_r02.sh <tag> [bench args]
set -u
TAG=${1:-r02}; shift || true
# a profiled process has the GPU initialised by the profiler's preload: it must not spawn torch workers (bench.py --gpus N > 1)
case " $* " in *" --gpus "[2-9]*|*" --gpus=[2-9]"*) echo "profile one rank only: --gpus > 1 is refused under rocprofv3" >&2; exit 2;; esac
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
python3 bench.py "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" --steps 60 --warmup 10 --cpu-budget 0 --secondary 0 --full-metrics 0 "$@" > "$OUT/stats.log" 2>&1
echo "stats rc=$?"
find "$OUT/stats" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats.csv"
cd "$ROOT"
bash scripts/profile_pmc.sh "$TAG/pmc" --secondary 0 --full-metrics 0 "$@" > "$OUT/pmc.log" 2>&1
cp "$OUT/pmc/summary.txt" "$OUT/pmc_summary.txt" 2>/dev/null
head -c 600 "$OUT/bench.json"; echo; head -8 "$OUT/kernel_stats.csv"
