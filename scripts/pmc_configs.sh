#!/bin/bash
# PMC passes for one config of scripts/bench_configs.py (run ON the GPU box):  bash scripts/pmc_configs.sh <tag> <C1|C4|C5>
set -u
TAG=${1:-pmc_cfg}; CFG=${2:-C5}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift
  timeout 180 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$ROOT/scripts/bench_configs.py" --only $CFG --reps 4 > "$OUT/$name.log" 2>&1
  echo "$name rc=$?"; }
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run l2 TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_EA0_RDREQ_32B_sum
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
run sq2 SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum
# (a pass over TCP_PENDING_STALL_CYCLES / TA_* counters hung rocprofv3 on this pool: not collected; every pass has a timeout)
cd "$ROOT"
python3 scripts/summarise_pmc.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
