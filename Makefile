# Builds libcvaegan_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
SRC_DIR := cvae_gan_amd/csrc
LIB     := cvae_gan_amd/lib/libcvaegan_hip.so
SRCS    := $(SRC_DIR)/api.hip $(SRC_DIR)/igemm.hip $(SRC_DIR)/conv_patch.hip $(SRC_DIR)/elementwise.hip
OBJS    := $(SRCS:.hip=.o)
FLAGS   := --offload-arch=$(ARCH) -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -Iinclude

all: $(LIB)

$(SRC_DIR)/%.o: $(SRC_DIR)/%.hip $(SRC_DIR)/common.h $(SRC_DIR)/gemm_util.h $(SRC_DIR)/conv_patch.h include/cvaegan.h
	$(HIPCC) $(FLAGS) -c $< -o $@

$(LIB): $(OBJS)
	@mkdir -p $(dir $(LIB))
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

clean:
	rm -f $(OBJS) $(LIB)
.PHONY: all clean

# Instrumented A/B build (phase timestamps in patch_fwd; tools/layer_bench.py and tools/stem_timing.py print them when
# CVG_LIB points at it) and the two MFMA-rate microbenchmarks.  Not used by the package.
timing: $(LIB)
	@mkdir -p build_exp
	$(HIPCC) $(FLAGS) -DEXP_TIMING -c $(SRC_DIR)/conv_patch.hip -o build_exp/conv_patch_timing.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o build_exp/lib_timing.so $(SRC_DIR)/api.o $(SRC_DIR)/igemm.o build_exp/conv_patch_timing.o $(SRC_DIR)/elementwise.o
	$(HIPCC) --offload-arch=$(ARCH) -O3 -Wno-unused-value tools/mfma_peak.hip -o build_exp/mfma_peak
	$(HIPCC) --offload-arch=$(ARCH) -O3 -Wno-unused-value tools/mfma_shape.hip -o build_exp/mfma_shape
.PHONY: timing

# Host-side sanitizer configuration: the library's host code (validation, planning, launch set-up) compiled with
# AddressSanitizer + UndefinedBehaviorSanitizer (device code as usual) and linked into tests/native/abi_check.cpp,
# which calls every planning query over the networks' shapes and the rejection paths; runs on the CPU.
ASAN_DIR := build_exp/asan
ASAN_FLAGS := $(FLAGS) -g -fno-omit-frame-pointer -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-sanitize-recover=undefined
$(ASAN_DIR)/%.o: $(SRC_DIR)/%.hip $(SRC_DIR)/common.h $(SRC_DIR)/gemm_util.h $(SRC_DIR)/conv_patch.h include/cvaegan.h
	@mkdir -p $(ASAN_DIR)
	$(HIPCC) $(ASAN_FLAGS) -c $< -o $@
$(ASAN_DIR)/abi_check.o: tests/native/abi_check.cpp include/cvaegan.h
	@mkdir -p $(ASAN_DIR)
	/opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++17 -Iinclude -fsanitize=address,undefined -fno-sanitize-recover=undefined -c $< -o $@
$(ASAN_DIR)/abi_check: $(patsubst $(SRC_DIR)/%.hip,$(ASAN_DIR)/%.o,$(SRCS)) $(ASAN_DIR)/abi_check.o
	$(HIPCC) --offload-arch=$(ARCH) -fsanitize=address,undefined -fno-gpu-sanitize $^ -o $@
asan: $(ASAN_DIR)/abi_check
	ASAN_OPTIONS=detect_leaks=0 $(ASAN_DIR)/abi_check
.PHONY: asan
