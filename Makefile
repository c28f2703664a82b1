# Build libdlwpmi.so (HIP kernels + C ABI, gfx950 only) and the CPU oracle helpers.
#   make            -> dlwp_benchmark_amd/libdlwpmi.so
#   make oracle     -> oracle/_build/liboracle_fno.so   (plain C restatement, CPU)
HIPCC      ?= /opt/rocm/bin/hipcc
ARCH       ?= gfx950
CSRC       := dlwp_benchmark_amd/csrc
HIPFLAGS   := -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -munsafe-fp-atomics -ffp-contract=fast \
              -Wall -Wno-unused-function -Iinclude
SRCS       := $(wildcard $(CSRC)/*.hip)
OBJS       := $(patsubst $(CSRC)/%.hip,build/%.o,$(SRCS))
LIB        := dlwp_benchmark_amd/libdlwpmi.so

all: $(LIB)

build/%.o: $(CSRC)/%.hip $(wildcard $(CSRC)/*.cuh) $(wildcard $(CSRC)/*.h) include/dlwpmi.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) $(OBJS) -o $@

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf build $(LIB)

.PHONY: all oracle clean
