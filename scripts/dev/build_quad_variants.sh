#!/bin/bash
# Builds libwdg_hip_<variant>.so under when-do-gnns-help_amd/lib/variants/ for the quad-row kernel's (threads, depth) variants:
# the shipped objects of every other file + spmm_quad.hip compiled with -DWDG_Q_FAST_THREADS / -DWDG_Q_DEPTH (+ extra flags).
# usage: scripts/dev/build_quad_variants.sh "1024 1" "768 2" ...   (run `make` first); A/B: scripts/dev/ab_quad_variants.py
set -e
cd "$(dirname "$0")/../.."
PKG=when-do-gnns-help_amd
mkdir -p build/variants $PKG/lib/variants
for v in "$@"; do
  set -- $v
  n=t$1d$2
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -I$PKG/csrc -DWDG_Q_FAST_THREADS=$1 -DWDG_Q_DEPTH=$2 $QUAD_EXTRA \
      -c $PKG/csrc/spmm_quad.hip -o build/variants/spmm_quad_$n.o 2> build/variants/$n.log &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/lib/variants/libwdg_hip_$n.so \
      $(ls build/*.o | grep -v spmm_quad.o) build/variants/spmm_quad_$n.o && echo built $n ) &
done
wait
