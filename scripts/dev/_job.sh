python -m pytest tests/test_gpu_sweep.py tests/test_gpu_batched_build.py tests/test_gpu_kernels.py tests/test_quad_segments.py -x -q 2>&1 | tail -2 | cut -c1-200
python3 scripts/dev/profile_whole_rank.py 1 0 2>&1 | grep -v amdgpu | head -5 | cut -c1-160
python3 scripts/dev/profile_whole_rank.py 8 0 2>&1 | grep -v amdgpu | head -5 | cut -c1-160
python3 scripts/dev/line_profile_init.py 2>&1 | grep -v amdgpu | head -12 | cut -c1-200
