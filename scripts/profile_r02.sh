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
STEPS=400
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" --steps $STEPS --warmup 10 --cpu-budget 0 --secondary 0 --full-metrics 0 "$@" > "$OUT/stats.log" 2>&1
echo "stats rc=$?"
find "$OUT/stats" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats.csv"
# the agreement the bench contract asks for: the dominant kernel's average duration in the trace - over the launches of the TIMED
# region, i.e. the last $STEPS of the trace (the statistics file averages the tuner's ~250 launches with other tape cuts and the
# untuned launches in as well) - beside the HIP-event figure the same (profiled) process printed on its line
python3 - "$OUT" $STEPS > "$OUT/trace_vs_events.txt" <<'PY'
import csv, glob, json, sys
out, steps = sys.argv[1], int(sys.argv[2])
rows = []
for f in glob.glob(out + "/stats/**/*kernel_trace.csv", recursive=True):
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
line = [l for l in open(out + "/stats.log") if l.startswith("{")]
rl = json.loads(line[-1])["roofline"] if line else {}
name = rl.get("kernel", "spmm_quad_kernel").split("<")[0]
mine = [d for _t, d, n in rows if name in n]
timed = mine[-steps:]
print(f"kernel {rl.get('kernel')}: {len(mine)} launches in the trace (5 first uses + 24 untuned + the tuner's + warm-up + {steps} timed)")
print(f"  trace, all launches        : {sum(mine) / max(len(mine), 1) / 1e3:8.2f} us")
print(f"  trace, the last {steps} (timed): {sum(timed) / max(len(timed), 1) / 1e3:8.2f} us   (min {min(timed) / 1e3:.2f}, max {max(timed) / 1e3:.2f})")
print(f"  HIP events, same process   : {rl.get('avg_launch_us')} us over {rl.get('launches_timed')} launches -> frac {rl.get('frac')}")
PY
cat "$OUT/trace_vs_events.txt"
cd "$ROOT"
bash scripts/profile_pmc.sh "$TAG/pmc" --secondary 0 --full-metrics 0 "$@" > "$OUT/pmc.log" 2>&1
cp "$OUT/pmc/summary.txt" "$OUT/pmc_summary.txt" 2>/dev/null
head -c 600 "$OUT/bench.json"; echo; head -8 "$OUT/kernel_stats.csv"
