# Build libdlwpmi.so (HIP kernels + C ABI, gfx950 only).  The oracle (oracle/*.py) is numpy / torch-CPU code: nothing to compile.
#   make            -> dlwp_benchmark_amd/libdlwpmi.so
HIPCC      ?= /opt/rocm/bin/hipcc
ARCH       ?= gfx950
CSRC       := dlwp_benchmark_amd/csrc
HIPFLAGS   := -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -munsafe-fp-atomics -ffp-contract=fast \
              -Wall -Wno-unused-function -Iinclude
SRCS       := $(wildcard $(CSRC)/*.hip)
OBJS       := $(patsubst $(CSRC)/%.hip,build/%.o,$(SRCS))
LIB        := dlwp_benchmark_amd/libdlwpmi.so

all: $(LIB)

build/%.o: $(CSRC)/%.hip $(wildcard $(CSRC)/*.hip.h) $(wildcard $(CSRC)/*.h) include/dlwpmi.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) $(OBJS) -ldl -o $@

# diagnostic build with in-kernel phase stamps (DLWP_STAMP in common.hip.h); never loaded by the package
STAMP_OBJS := $(patsubst $(CSRC)/%.hip,build_stamps/%.o,$(SRCS))
build_stamps/%.o: $(CSRC)/%.hip $(wildcard $(CSRC)/*.hip.h) $(wildcard $(CSRC)/*.h) include/dlwpmi.h
	@mkdir -p build_stamps
	$(HIPCC) $(HIPFLAGS) -DDLWP_STAMPS -c $< -o $@
stamps: $(STAMP_OBJS)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) $(STAMP_OBJS) -ldl -o dlwp_benchmark_amd/libdlwpmi_stamps.so

clean:
	rm -rf build build_stamps $(LIB) dlwp_benchmark_amd/libdlwpmi_stamps.so

.PHONY: all clean stamps
