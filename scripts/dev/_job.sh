python -m pytest tests/test_gpu_kernels.py -x -q -k "propagated" 2>&1 | tail -4 | cut -c1-300
python3 bench.py --steps 20 --warmup 5 --configs 0 --train 0 --secondary 0 --cold 0 --full-metrics 0 > gpurun_out/r05_bench_g.json 2> gpurun_out/r05_bench_g.err
python3 - <<'PY'
import json
b=json.load(open('gpurun_out/r05_bench_g.json'))
print(b['ms_per_step'], 'whole', b['sweep_whole']['seconds'], b['sweep_whole']['ms_per_base_rank0'])
print(json.dumps(b['scaling_projection']['whole_sweep']['worlds']))
print(b['sweep_whole']['mean_metrics'])
PY
