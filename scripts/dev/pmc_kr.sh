#!/bin/bash
# PMC passes over the kernel-regression launches of one C3 shard (run ON the GPU box from the repo root): bash scripts/dev/pmc_kr.sh <tag>
# kernel-trace + one counter group per run (no sys/hip/hsa tracing with --pmc on this pool); summary -> gpurun_out/<tag>_summary.txt
set -u
TAG=${1:-r05_kr}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$ROOT/scripts/dev/time_kr_batch.py" 5 100 > "$OUT/$name.log" 2>&1
  echo "$name rc=$?"; }
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run l2 TCC_REQ_sum TCC_READ_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA
run sq2 SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/scripts/dev/time_kr_batch.py" 5 100 > "$OUT/stats.log" 2>&1
cd "$ROOT"
python3 - "$OUT" <<'PY' > "$ROOT/gpurun_out/${TAG}_summary.txt"
import csv, glob, collections, sys
out = sys.argv[1]
want = ("kr_solve", "kr_select", "kr_sample", "gram_split", "gram_diag")
for d in sorted(glob.glob(out + "/*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc, n = collections.defaultdict(float), collections.Counter()
        for r in csv.DictReader(open(f)):
            k = next((w for w in want if w in r["Kernel_Name"]), None)
            if k:
                acc[(k, r["Counter_Name"])] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for key in sorted(acc):
            print(f"{key[0]:12s} {key[1]:28s} per launch {acc[key] / n[key]:16.1f}   launches {n[key]}")
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(w in r["Name"] for w in want):
            print(f"stats {r['Name'][:60]:60s} calls {r['Calls']} avg_us {float(r['AverageNs']) / 1e3:10.1f} min_us {float(r['MinNs']) / 1e3:10.1f}")
for f in sorted(glob.glob(out + "/*.log")):
    for line in open(f):
        if "regressions in" in line:
            print(f.split("/")[-1], line.strip())
PY
cat "$ROOT/gpurun_out/${TAG}_summary.txt"
