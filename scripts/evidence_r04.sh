#!/bin/bash
# Everything profiles/ holds for round 4, from one box (run ON the GPU box from the repo root): bash scripts/evidence_r04.sh
# (one rank only: the profiled process never spawns workers - scripts/profile_pmc.sh refuses --gpus > 1)
# Needs the -DWDG_Q_PROFILE builds of scripts/dev/build_quad_variants.sh under when-do-gnns-help_amd/lib/variants/prof_{dyn,static,noadds}.so
# (QUAD_EXTRA="-DWDG_Q_PROFILE [-DWDG_Q_STATIC_DEAL | -DWDG_Q_ABLATE_ADDS]" scripts/dev/build_quad_variants.sh "1024 1", renamed).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
LEAN="--cold 0 --configs 0 --train 0 --projection 0 --whole 0"
bash scripts/profile_r02.sh r04_c3 $LEAN > gpurun_out/ev4_c3.log 2>&1                       # bench line + kernel stats + PMC passes, C3 headline
bash scripts/profile_pmc.sh r04_c2/pmc --k 2 --seeds 10 --secondary 0 --full-metrics 0 $LEAN > gpurun_out/ev4_c2.log 2>&1   # C2: traffic of the `secondary` line
python3 bench.py > gpurun_out/r04_bench_full.json 2> gpurun_out/r04_bench_full.err          # the whole default line
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_driver_flags.json 2> /dev/null # with the driver's flags
bash scripts/profile_pmc.sh r04_c3lit/pmc --nodes 4000 --secondary 0 --full-metrics 0 $LEAN > gpurun_out/ev4_c3lit.log 2>&1  # the literal N = 4000 shard (HALF slabs)
python3 scripts/bench_configs.py --out gpurun_out/r04_configs.jsonl > /dev/null 2> gpurun_out/r04_configs.err  # one line per BASELINE config
for v in static dyn noadds; do                                                                # per-wave clocks of the pipelined loop
  [ -f when-do-gnns-help_amd/lib/variants/prof_$v.so ] && WDG_LIB_PATH=$ROOT/when-do-gnns-help_amd/lib/variants/prof_$v.so python3 scripts/dev/quad_profile.py 10 5
done 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_quad_wave_profile.txt
for v in dyn noadds; do                                                                       # ... and of the HALF-slab loop (N = 4000)
  [ -f when-do-gnns-help_amd/lib/variants/prof_$v.so ] && N=4000 WDG_LIB_PATH=$ROOT/when-do-gnns-help_amd/lib/variants/prof_$v.so python3 scripts/dev/quad_profile.py 10 5
done 2>&1 | grep -v amdgpu.ids >> gpurun_out/r04_quad_wave_profile.txt
{ python3 scripts/dev/ablate_quad.py 10 5 0 1 4 5 2 8 12 13; N=4000 python3 scripts/dev/ablate_quad.py 10 5 0 1 4 5 2 8 12 13; } 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_quad_ablations.txt
{ python3 scripts/dev/segment_spans.py; for o in 0 1; do WDG_SELL_ORDER=$o python3 scripts/dev/half_slab.py; done; } 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_segment_spans_half_slab.txt
cd /tmp && export TMPDIR=/tmp
for part in train configs whole; do
  flags="--steps 10 --warmup 2 --secondary 0 --full-metrics 0 --cpu-budget 0 --cold 0 --configs 0 --train 0 --projection 0 --whole 0"
  flags=${flags/--$part 0/--$part 1}
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/r04_${part}_stats" -- python3 "$ROOT/bench.py" $flags > /dev/null 2>&1
  find "$ROOT/gpurun_out/r04_${part}_stats" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$ROOT/gpurun_out/r04_${part}_kernel_stats.csv"
done
cd "$ROOT"
tail -3 gpurun_out/ev4_c3.log | cut -c1-300
head -12 gpurun_out/r04_quad_wave_profile.txt
