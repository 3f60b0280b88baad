#!/bin/bash
# (every pass runs under `timeout 300`: some counter passes hang rocprofv3 on this pool)
# PMC passes for the quad-row SpMM alone (run ON the GPU box, from the repo root):  bash scripts/pmc_quad.sh <tag> <k> <seeds> <ablate code>
set -u
TAG=${1:-pmcq}; K=${2:-10}; SEEDS=${3:-5}; AB=${4:-0}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$ROOT/scripts/dev/ablate_quad.py" $K $SEEDS $AB > "$OUT/$name.log" 2>&1
  echo "$name rc=$?"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
run sq2 SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM
run sq3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_SENDMSG
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum
cd "$ROOT"
python3 scripts/summarise_pmc.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
