#!/bin/bash
# the split-operand transform without its A loads / without its products (timing only; rebuilds the library three times)
cd "$(dirname "$0")/../.."
for a in 1 2 0; do
  touch when-do-gnns-help_amd/csrc/gemm.hip
  make EXTRA="-DWDG_SPLIT_ABLATE=$a" > /dev/null 2>&1 || { echo "build failed"; exit 1; }
  echo "WDG_SPLIT_ABLATE=$a"
  python scripts/dev/time_mlp2.py 2>/dev/null | grep "parts=auto split=1"
done
