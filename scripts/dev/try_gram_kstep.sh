#!/bin/bash
# the split-operand Gram with 16-k steps (smaller LDS tiles: four workgroups per CU instead of three) against 32-k steps
cd "$(dirname "$0")/../.."
for k in 16 32; do
  touch when-do-gnns-help_amd/csrc/kernel_reg.hip
  make EXTRA="-DWDG_SGBK=$k" > /dev/null 2>&1 || { echo "build failed"; exit 1; }
  echo "WDG_SGBK=$k"
  python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "gram" 2>&1 | tail -1
  python scripts/dev/time_gram.py 2>/dev/null | grep "WDG_GRAM" | cut -c1-110
done
