python -X faulthandler -m pytest tests/test_gpu_kr_epochs.py tests/test_gpu_sweep.py -x -q > gpurun_out/r05_tests_f_full.log 2>&1
grep -n "Fatal\|Segmentation\|passed\|failed\|Error" gpurun_out/r05_tests_f_full.log | head; grep -n "Current thread\|most recent call first" -A 12 gpurun_out/r05_tests_f_full.log | head -60 | cut -c1-200
