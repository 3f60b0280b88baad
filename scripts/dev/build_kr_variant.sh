#!/bin/bash
# Builds when-do-gnns-help_amd/lib/variants/libwdg_hip_<name>.so: the shipped objects of every other file + kernel_reg.hip compiled
# with extra flags (e.g. -DK2_PROFILE: one workgroup prints its shader clocks per phase).  usage: build_kr_variant.sh <name> "<flags>"
set -e
cd "$(dirname "$0")/../.."
PKG=when-do-gnns-help_amd
mkdir -p build/variants $PKG/lib/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -I$PKG/csrc -mllvm -pragma-unroll-threshold=200000 $2 \
  -c $PKG/csrc/kernel_reg.hip -o build/variants/kernel_reg_$1.o 2> build/variants/kr_$1.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/lib/variants/libwdg_hip_kr_$1.so $(ls build/*.o | grep -v kernel_reg.o) build/variants/kernel_reg_$1.o
echo built $PKG/lib/variants/libwdg_hip_kr_$1.so
