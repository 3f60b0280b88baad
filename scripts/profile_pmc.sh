#!/bin/bash
# PMC passes for the bench command (run ON the GPU box, from the repo root):  bash scripts/profile_pmc.sh <tag> [bench args]
# One counter group per rocprofv3 run, kernel-trace only (no sys/hip/hsa tracing with --pmc on this pool).
set -u
TAG=${1:-pmc}; shift || true
# a profiled process has the GPU initialised by the profiler's preload: it must not spawn torch workers (bench.py --gpus N > 1)
case " $* " in *" --gpus "[2-9]*|*" --gpus=[2-9]"*) echo "profile one rank only: --gpus > 1 is refused under rocprofv3" >&2; exit 2;; esac
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 6 --warmup 2 --cpu-budget 0 $*"
run() { # name counters...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/$name.log" 2>&1
  echo "$name rc=$?"
}
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run l2 TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_EA0_RDREQ_32B_sum
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
run sq2 SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum
cd "$ROOT"
python3 scripts/summarise_pmc.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
