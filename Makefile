# Builds libwdg_hip.so (gfx950 only) and the CPU oracle.  `python -c "import __graft_entry__ as g; g.build()"` calls this.
PKG      := when-do-gnns-help_amd
CSRC     := $(PKG)/csrc
HIPCC    ?= /opt/rocm/bin/hipcc
HIPFLAGS := --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -I$(CSRC) -Wall -Wno-unused-function -Wno-unused-value $(if $(STAMPS),-DWDG_STAMPS,) $(if $(DEPTH),-DWDG_PREFETCH_DEPTH=$(DEPTH),) $(EXTRA)
SRCS     := $(wildcard $(CSRC)/*.hip)
# per-file flags: the 32 steps of kernel_reg.hip's in-wave substitution must unroll completely (their register indices and
# branch conditions are compile-time only then) - with the default budget the compiler peels 11 steps and rolls the rest
FLAGS_kernel_reg := -mllvm -pragma-unroll-threshold=200000
# gnb.hip reproduces numpy's fp32 sums bit for bit: no multiply-add may be fused (its source says so as well: #pragma clang fp contract(off))
FLAGS_gnb := -ffp-contract=off
OBJS     := $(patsubst $(CSRC)/%.hip,build/%.o,$(SRCS))
LIB      := $(PKG)/lib/libwdg_hip.so

all: $(LIB) oracle

$(LIB): $(OBJS)
	@mkdir -p $(dir $@)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o $@ $(OBJS)

# (the compiler's per-kernel resource report - registers, spills, scratch - is kept beside the object: tests/test_abi.py
# checks that no shipped kernel spills vector registers)
build/%.o: $(CSRC)/%.hip $(wildcard $(CSRC)/*.h) include/wdg.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) $(FLAGS_$*) -Rpass-analysis=kernel-resource-usage -c $< -o $@ 2> build/$*.rsrc || (cat build/$*.rsrc; exit 1)
	@grep -E "warning:|error:" build/$*.rsrc || true

oracle: oracle/_build/libwdg_oracle.so

oracle/_build/libwdg_oracle.so: oracle/wdg_oracle.c
	@mkdir -p oracle/_build
	gcc -O2 -ffp-contract=off -shared -fPIC -o $@ $< -lm

clean:
	rm -rf build $(LIB) oracle/_build

.PHONY: all oracle clean
