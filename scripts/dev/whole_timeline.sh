#!/bin/bash
# Kernel timeline of one rank's whole-sweep pass (dev tool): how much of the wall clock has a kernel running, per kernel family.
# usage (on the GPU box, from the repo root): bash scripts/dev/whole_timeline.sh [world] [rank] [adjacencies per shard]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
W=${1:-1}; R=${2:-0}; PS=${3:-70}
OUT=$ROOT/gpurun_out/whole_timeline_${W}_${PS}
cd /tmp && export TMPDIR=/tmp
rm -rf "$OUT"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -- python3 "$ROOT/scripts/dev/profile_whole_rank.py" $W $R $PS 2>&1 | grep "rows in"
python3 - "$OUT" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# everything from the first regression launch on: the script's warm-up pass + five passes (busy time per pass = total / ~5.03)
kr = [i for i, r in enumerate(rows) if "kr_solve" in r[2]]
part = rows[kr[0]:]
t0, t1 = part[0][0], max(r[1] for r in part)
busy, end = 0, t0
for a, b, _ in part:
    if b > end:
        busy += b - max(a, end)
        end = b
by = {}
for a, b, n in part:
    k = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0][:40]
    by[k] = by.get(k, 0) + (b - a)
print(f"all passes: {len(part)} kernels; span {1e-6 * (t1 - t0):.1f} ms; some kernel running {1e-6 * busy:.1f} ms ({100 * busy / (t1 - t0):.1f} %); sum of durations {1e-6 * sum(b - a for a, b, _ in part):.1f} ms")
for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:14]:
    print(f"  {k:42s} {1e-6 * v:8.1f} ms")
PY
rm -rf "$OUT"
