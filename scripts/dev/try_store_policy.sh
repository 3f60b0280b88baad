#!/bin/bash
# cache policy of the quad-row kernel's Y stores (run ON the GPU box): rebuilds spmm_quad.hip per policy, times launch + step
cd ${GRAFT_REPO_ROOT:-.}
for pol in "" "sc1" "nt" "sc0 sc1" "sc0"; do
  touch when-do-gnns-help_amd/csrc/spmm_quad.hip
  make -s EXTRA="-DWDG_Q_STORE_POLICY='\"$pol\"'" 2>&1 | grep -i " error"
  echo "== policy [$pol]"
  timeout 300 python3 scripts/dev/ablate_quad.py 10 5 0 | grep ablate=
  timeout 300 python3 scripts/dev/ablate_quad.py 2 10 0 | grep ablate=
  timeout 300 python3 bench.py --cpu-budget 0 --full-metrics 0 --cold 0 --steps 300 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step', round(d['ms_per_step'],4), 'spmm', round(d['roofline']['avg_launch_us'],1), 'secondary step', round(d['secondary']['ms_per_step'],4), round(d['secondary']['roofline']['avg_launch_us'],1))"
done
touch when-do-gnns-help_amd/csrc/spmm_quad.hip; make -s 2>&1 | grep -i " error"
