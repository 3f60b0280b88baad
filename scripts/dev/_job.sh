python3 scripts/dev/profile_whole_rank.py 1 0 2>&1 | grep -v amdgpu | head -6 | cut -c1-160
python3 scripts/dev/profile_whole_rank.py 8 0 2>&1 | grep -v amdgpu | head -6 | cut -c1-160
python3 scripts/dev/line_profile_init.py 2>&1 | grep -v amdgpu | head -14 | cut -c1-200
python -m pytest tests/test_gpu_sweep.py tests/test_gpu_models_cli.py -x -q 2>&1 | tail -2 | cut -c1-200
