python -m pytest tests/test_gpu_batched_build.py tests/test_gpu_sweep.py -x -q 2>&1 | tail -4 | cut -c1-300
python3 scripts/dev/time_cold_build.py 2>&1 | head -12 | cut -c1-200
python3 bench.py --steps 20 --warmup 5 --configs 0 --train 0 --secondary 0 --projection 0 --full-metrics 0 > gpurun_out/r05_bench_f.json 2> gpurun_out/r05_bench_f.err
python3 - <<'PY'
import json
b=json.load(open('gpurun_out/r05_bench_f.json'))
print(b['ms_per_step'], 'whole', b['sweep_whole']['seconds'])
c=b['sweep_cold']
for k in ('six_scalars','nine_scalars'): print(k, {kk:round(v,2) for kk,v in c[k].items() if isinstance(v,float)}, c[k]['pipelined']['graphs_per_s'])
PY
