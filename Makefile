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
