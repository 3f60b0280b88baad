#!/bin/bash
# the split-operand Gram without its tile loads / products / stores / map (timing only; rebuilds the library each time)
cd "$(dirname "$0")/../.."
for a in 1 2 3 4 0; do
  touch when-do-gnns-help_amd/csrc/kernel_reg.hip
  make EXTRA="-DWDG_SG_ABLATE=$a" > /dev/null 2>&1 || { echo "build failed"; exit 1; }
  echo "WDG_SG_ABLATE=$a"
  python scripts/dev/time_gram.py 2>/dev/null | grep "WDG_GRAM" | cut -c1-110
done
