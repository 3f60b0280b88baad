python -m pytest tests/test_gpu_kr_epochs.py -x -q -k at_scale 2>&1 | tail -40 > gpurun_out/r05_scale_test.log
python -m pytest tests/test_gpu_batched_build.py tests/test_gpu_sweep.py tests/test_integration_doc.py tests/test_gpu_api.py -x -q 2>&1 | tail -8 > gpurun_out/r05_tests_b.log
build/store_shape > gpurun_out/r05_store_shape.txt 2>&1
python3 bench.py --steps 20 --warmup 5 --configs 0 --train 0 --cold 0 --secondary 0 > gpurun_out/r05_bench_b.json 2> gpurun_out/r05_bench_b.err
WDG_KR_SAMPLER_SORT=1 python3 bench.py --steps 20 --warmup 5 --configs 0 --train 0 --cold 0 --secondary 0 --projection 0 > gpurun_out/r05_bench_b_sort.json 2> /dev/null
bash scripts/dev/pmc_kr.sh r05_kr > gpurun_out/r05_kr_pmc.log 2>&1
tail -5 gpurun_out/r05_scale_test.log; tail -3 gpurun_out/r05_tests_b.log; cat gpurun_out/r05_store_shape.txt
