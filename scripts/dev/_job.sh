python -X faulthandler -m pytest tests -m gpu -x -q > gpurun_out/r05_tests_h_full.log 2>&1
tail -3 gpurun_out/r05_tests_h_full.log | cut -c1-300
