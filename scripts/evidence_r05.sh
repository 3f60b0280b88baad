#!/bin/bash
# Everything profiles/ holds for round 5, from one box (run ON the GPU box from the repo root): bash scripts/evidence_r05.sh
# (one rank only: the profiled process never spawns workers - scripts/profile_pmc.sh refuses --gpus > 1).  Results land under
# gpurun_out/r05_collect/ with the names they keep under profiles/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
C=$ROOT/gpurun_out/r05_collect
mkdir -p "$C"
LEAN="--cold 0 --configs 0 --train 0 --projection 0 --whole 0"
bash scripts/profile_r02.sh r05_c3 $LEAN > gpurun_out/ev5_c3.log 2>&1                        # bench line + kernel stats + PMC passes, C3 headline
cp gpurun_out/r05_c3/bench.json "$C/r05_c3_bench.json"; cp gpurun_out/r05_c3/kernel_stats.csv "$C/r05_c3_bench_kernel_stats.csv"
cp gpurun_out/r05_c3/pmc/summary.txt "$C/r05_c3_pmc_summary.txt"; cp gpurun_out/r05_c3/pmc/summary.json "$C/r05_c3_pmc_summary.json"
bash scripts/profile_pmc.sh r05_c2/pmc --k 2 --seeds 10 --secondary 0 --full-metrics 0 $LEAN > gpurun_out/ev5_c2.log 2>&1    # C2: traffic of the `secondary` line
cp gpurun_out/r05_c2/pmc/summary.txt "$C/r05_c2_pmc_summary.txt"; cp gpurun_out/r05_c2/pmc/summary.json "$C/r05_c2_pmc_summary.json"
bash scripts/profile_pmc.sh r05_c3lit/pmc --nodes 4000 --secondary 0 --full-metrics 0 $LEAN > gpurun_out/ev5_c3lit.log 2>&1  # the literal N = 4000 shard
cp gpurun_out/r05_c3lit/pmc/summary.txt "$C/r05_c3lit_pmc_summary.txt"; cp gpurun_out/r05_c3lit/pmc/summary.json "$C/r05_c3lit_pmc_summary.json"
python3 bench.py > "$C/r05_bench_full.json" 2> gpurun_out/r05_bench_full.err                  # the whole default line
python3 bench.py --steps 20 --warmup 5 > "$C/r05_bench_driver_flags.json" 2> /dev/null         # with the driver's flags
python3 scripts/bench_configs.py --out "$C/r05_configs.jsonl" > /dev/null 2> gpurun_out/r05_configs.err   # one line per BASELINE config (+ binding resource)
{ python3 scripts/dev/ablate_quad.py 10 5 0 1 4 5 2 8 12 13; N=4000 python3 scripts/dev/ablate_quad.py 10 5 0 1 4 5 2 8 12 13; } 2>&1 | grep -v amdgpu.ids > "$C/r05_quad_ablations.txt"
build/store_shape > "$C/r05_store_shape.txt" 2>&1
python3 scripts/dev/ab_chunked_step.py 200 2>&1 | grep -v amdgpu.ids > "$C/r05_chunked_step.txt"
cd /tmp && export TMPDIR=/tmp
for part in train configs whole; do                                                            # kernel statistics of the other blocks of the line
  flags="--steps 10 --warmup 2 --secondary 0 --full-metrics 0 --cpu-budget 0 --cold 0 --configs 0 --train 0 --projection 0 --whole 0"
  flags=${flags/--$part 0/--$part 1}
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/r05_${part}_stats" -- python3 "$ROOT/bench.py" $flags > /dev/null 2>&1
  find "$ROOT/gpurun_out/r05_${part}_stats" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$C/r05_${part}_kernel_stats.csv"
done
# the whole sweep's timeline: how much of the wall clock has a kernel running (union of the kernel intervals of the trace)
python3 - "$ROOT/gpurun_out/r05_whole_stats" > "$C/r05_whole_timeline.txt" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# the timed sweep = the last contiguous stretch of kr_solve launches (warm-up pass first): take everything after the largest gap
gaps = sorted(((rows[i + 1][0] - max(r[1] for r in rows[:i + 1]), i) for i in range(len(rows) - 1)), reverse=True)[:1]
cut = gaps[0][1] + 1 if gaps else 0
part = rows[cut:]
t0, t1 = part[0][0], max(r[1] for r in part)
busy, end = 0, t0
for a, b, _ in part:
    if b > end:
        busy += b - max(a, end)
        end = b
by = {}
for a, b, n in part:
    k = n.split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")[:40]
    by[k] = by.get(k, 0) + (b - a)
print(f"kernels of the timed pass: {len(part)}; span {1e-6 * (t1 - t0):.1f} ms; some kernel running {1e-6 * busy:.1f} ms ({100 * busy / (t1 - t0):.1f} %); sum of kernel durations {1e-6 * sum(b - a for a, b, _ in part):.1f} ms (two streams overlap)")
for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:12]:
    print(f"  {k:42s} {1e-6 * v:8.1f} ms")
PY
cd "$ROOT"
bash scripts/dev/pmc_kr.sh r05_kr > gpurun_out/ev5_kr.log 2>&1
cp gpurun_out/r05_kr_summary.txt "$C/r05_kr_pmc_summary.txt"
python3 scripts/make_traffic_json.py \
  "10:50:$C/r05_c3_pmc_summary.json:profiles/r05_c3_pmc_summary.txt (round 5, bench.py headline under rocprofv3 --pmc, separate passes; Y tiled by feature group)" \
  "2:100:$C/r05_c2_pmc_summary.json:profiles/r05_c2_pmc_summary.txt (round 5, bench.py --k 2 --seeds 10)" \
  "10@4000:50:$C/r05_c3lit_pmc_summary.json:profiles/r05_c3lit_pmc_summary.txt (round 5, bench.py --nodes 4000: HALF slabs)" > "$C/traffic.json" 2> gpurun_out/ev5_traffic.err
tail -3 gpurun_out/ev5_c3.log | cut -c1-300
cat "$C/r05_whole_timeline.txt"; ls -la "$C"
