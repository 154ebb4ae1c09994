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

# Instrumented A/B build (phase timestamps in patch_fwd; tools/layer_bench.py prints them when
# CVG_LIB points at it).  Not used by the package.
timing: $(LIB)
	@mkdir -p build_exp
	$(HIPCC) $(FLAGS) -DEXP_TIMING -c $(SRC_DIR)/conv_patch.hip -o build_exp/conv_patch_timing.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o build_exp/lib_timing.so $(SRC_DIR)/api.o $(SRC_DIR)/igemm.o build_exp/conv_patch_timing.o $(SRC_DIR)/elementwise.o
	$(HIPCC) --offload-arch=$(ARCH) -O3 -Wno-unused-value tools/mfma_peak.hip -o build_exp/mfma_peak
.PHONY: timing
