#!/bin/bash
# Everything profiles/ holds for round 6, from one box (run ON the GPU box from the repo root): bash scripts/evidence_r06.sh
# (one rank only: the profiled process never spawns workers - scripts/profile_pmc.sh refuses --gpus > 1).  Results land under
# gpurun_out/r06_collect/ with the names they keep under profiles/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
C=$ROOT/gpurun_out/r06_collect
mkdir -p "$C"
LEAN="--cold 0 --configs 0 --train 0 --projection 0 --whole 0 --gnb 0"
bash scripts/profile_r02.sh r06_c3 $LEAN > gpurun_out/ev6_c3.log 2>&1                        # bench line + kernel stats + PMC passes, C3 headline
cp gpurun_out/r06_c3/bench.json "$C/r06_c3_bench.json"; cp gpurun_out/r06_c3/kernel_stats.csv "$C/r06_c3_bench_kernel_stats.csv"; cp gpurun_out/r06_c3/trace_vs_events.txt "$C/r06_c3_trace_vs_events.txt"
cp gpurun_out/r06_c3/pmc/summary.txt "$C/r06_c3_pmc_summary.txt"; cp gpurun_out/r06_c3/pmc/summary.json "$C/r06_c3_pmc_summary.json"
bash scripts/profile_pmc.sh r06_c2/pmc --k 2 --seeds 10 --secondary 0 --full-metrics 0 $LEAN > gpurun_out/ev6_c2.log 2>&1    # C2: traffic of the `secondary` line
cp gpurun_out/r06_c2/pmc/summary.txt "$C/r06_c2_pmc_summary.txt"; cp gpurun_out/r06_c2/pmc/summary.json "$C/r06_c2_pmc_summary.json"
bash scripts/profile_pmc.sh r06_c3lit/pmc --nodes 4000 --secondary 0 --full-metrics 0 $LEAN > gpurun_out/ev6_c3lit.log 2>&1  # the literal N = 4000 shard
cp gpurun_out/r06_c3lit/pmc/summary.txt "$C/r06_c3lit_pmc_summary.txt"; cp gpurun_out/r06_c3lit/pmc/summary.json "$C/r06_c3lit_pmc_summary.json"
python3 bench.py > "$C/r06_bench_full_line.json" 2> gpurun_out/r06_bench_full.err              # the whole default run: the ONE compact line ..
cp gpurun_out/bench_detail.json "$C/r06_bench_full_detail.json"                                # .. and the full record it points at
python3 bench.py --steps 20 --warmup 5 > "$C/r06_bench_driver_flags_line.json" 2> /dev/null     # with the driver's flags
cp gpurun_out/bench_detail.json "$C/r06_bench_driver_flags_detail.json"
python3 scripts/bench_configs.py --out "$C/r06_configs.jsonl" > /dev/null 2> gpurun_out/r06_configs.err   # one line per BASELINE config (+ binding resource)
cd /tmp && export TMPDIR=/tmp
for part in train configs whole; do                                                            # kernel statistics of the other blocks of the line
  flags="--steps 10 --warmup 2 --secondary 0 --full-metrics 0 --cpu-budget 0 --cold 0 --configs 0 --train 0 --projection 0 --whole 0 --gnb 0"
  flags=${flags/--$part 0/--$part 1}
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/r06_${part}_stats" -- python3 "$ROOT/bench.py" $flags > /dev/null 2>&1
  find "$ROOT/gpurun_out/r06_${part}_stats" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$C/r06_${part}_kernel_stats.csv"
done
# the whole sweep's timeline: how much of the wall clock has a kernel running (union of the kernel intervals of the trace)
python3 - "$ROOT/gpurun_out/r06_whole_stats" > "$C/r06_whole_timeline.txt" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# the timed sweep = the last contiguous stretch of launches (warm-up passes first): take everything after the largest gap
gaps = sorted(((rows[i + 1][0] - max(r[1] for r in rows[:i + 1]), i) for i in range(len(rows) - 1)), reverse=True)[:1]
cut = gaps[0][1] + 1 if gaps else 0
part = rows[cut:]
t0, t1 = part[0][0], max(r[1] for r in part)
busy, end = 0, t0
for a, b, _ in part:
    if b > end:
        busy += b - max(a, end)
        end = b
by, lib = {}, 0
for a, b, n in part:
    k = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0][:40]
    by[k] = by.get(k, 0) + (b - a)
    if "at::native" in n or "rocclr" in n:
        lib += b - a
tot = sum(b - a for a, b, _ in part)
print(f"kernels after the largest gap of the trace (bench.py's whole-sweep passes: warm-up piece, first pass, timed pass): {len(part)}; span {1e-6 * (t1 - t0):.1f} ms; some kernel running {1e-6 * busy:.1f} ms ({100 * busy / (t1 - t0):.1f} %); sum of kernel durations {1e-6 * tot:.1f} ms (streams overlap); library kernels (torch + rocclr copies) {1e-6 * lib:.1f} ms = {100 * lib / tot:.1f} % of that sum")
for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:14]:
    print(f"  {k:42s} {1e-6 * v:8.1f} ms")
PY
cd "$ROOT"
bash scripts/dev/pmc_kr.sh r06_kr > gpurun_out/ev6_kr.log 2>&1
cp gpurun_out/r06_kr_summary.txt "$C/r06_kr_pmc_summary.txt"
python3 -m pytest tests/test_gpu_kr_epochs.py -m gpu -q -s 2>&1 | grep "kr epochs\|passed\|failed" > "$C/r06_kr_epochs_vs_reference.txt"
cp gpurun_out/kr_ridge_vs_pinv.json "$C/r06_kr_deflation_vs_pinv.json"
python3 scripts/dev/time_kr_batch.py 2>&1 | grep -v amdgpu.ids > "$C/r06_kr_solver_time.txt"; WDG_KR_DEFLATE=0 python3 scripts/dev/time_kr_batch.py 2>&1 | grep -v amdgpu.ids >> "$C/r06_kr_solver_time.txt"
python3 scripts/dev/rank_phases.py 8 0 2>&1 | grep -v amdgpu.ids | tail -14 > "$C/r06_rank_phases.txt"
python3 -m pytest tests/test_gpu_kernels.py -m gpu -q -s -k half_slab 2>&1 | grep "half slab order\|passed" > "$C/r06_half_slab_conflicts.txt"
python3 scripts/make_traffic_json.py \
  "10:50:$C/r06_c3_pmc_summary.json:profiles/r06_c3_pmc_summary.txt (round 6, bench.py headline under rocprofv3 --pmc, separate passes; Y tiled by feature group)" \
  "2:100:$C/r06_c2_pmc_summary.json:profiles/r06_c2_pmc_summary.txt (round 6, bench.py --k 2 --seeds 10)" \
  "10@4000:50:$C/r06_c3lit_pmc_summary.json:profiles/r06_c3lit_pmc_summary.txt (round 6, bench.py --nodes 4000: HALF slabs)" > "$C/traffic.json" 2> gpurun_out/ev6_traffic.err
tail -3 gpurun_out/ev6_c3.log | cut -c1-300
cat "$C/r06_whole_timeline.txt"; ls -la "$C"
