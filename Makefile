# Builds the C-ABI HIP library (gfx950 only) and the oracle-side helpers.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
SRC   := avlen_amd/csrc
OBJ   := build/obj
LIB   := avlen_amd/lib/libavlen_hip.so
HIPFLAGS := --offload-arch=$(ARCH) -O3 -fPIC -std=c++17 -Iinclude -Wall -Wno-unused-function -ffp-contract=off
SRCS  := $(wildcard $(SRC)/*.hip)
OBJS  := $(patsubst $(SRC)/%.hip,$(OBJ)/%.o,$(SRCS))

all: $(LIB)

$(OBJ)/%.o: $(SRC)/%.hip $(SRC)/common.h $(SRC)/internal.h $(SRC)/tower_util.h include/avlen_hip.h
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	@mkdir -p $(dir $(LIB))
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -ldl

# kernels that issue loads / stores from inline asm with their own waits: walk the gfx950 ISA of exactly what is shipped (same
# compiler, same flags) for a register the compiler touched too early (tools/asm_wait_check.py)
asmcheck:
	HIPCC="$(HIPCC)" HIPFLAGS="$(HIPFLAGS)" python3 tools/asm_wait_check.py $(SRC)/clip_tower.hip $(SRC)/tower_x3.hip $(SRC)/chain.hip

clean:
	rm -rf build $(LIB)
.PHONY: all clean asmcheck
