# Build the gfx950 kernel library (product) and, for the CPU test tier only, the same
# kernel sources against the host-side SIMT simulator.
HIPCC   ?= /opt/rocm/bin/hipcc
HOSTCXX ?= /opt/rocm/lib/llvm/bin/clang++
ARCH    ?= gfx950
CSRC    := dpc_amd/csrc
SRCS    := $(wildcard $(CSRC)/*.hip)
HDRS    := $(wildcard $(CSRC)/*.h) include/dpc_hip.h
OBJS    := $(patsubst $(CSRC)/%.hip,build/hip/%.o,$(SRCS))
EOBJS   := $(patsubst $(CSRC)/%.hip,build/emu/%.o,$(SRCS))
EMU     := tests/simt_emu

all: dpc_amd/libdpc_hip.so

dpc_amd/libdpc_hip.so: $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

build/hip/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p build/hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Iinclude -c $< -o $@

emu: $(EMU)/libdpc_emu.so

$(EMU)/libdpc_emu.so: $(EOBJS) build/emu/simt_emu.o
	$(HOSTCXX) -shared -fPIC -o $@ $(EOBJS) build/emu/simt_emu.o

build/emu/%.o: $(CSRC)/%.hip $(HDRS) $(EMU)/simt_emu.h
	@mkdir -p build/emu
	$(HOSTCXX) -x c++ -DDPC_SIMT_EMU -O2 -g -std=c++17 -fPIC -Wno-unused-value -I$(EMU) -Iinclude -c $< -o $@

build/emu/simt_emu.o: $(EMU)/simt_emu.cpp $(EMU)/simt_emu.h
	@mkdir -p build/emu
	$(HOSTCXX) -O2 -g -std=c++17 -fPIC -I$(EMU) -c $< -o $@

# timing probes only (scripts/probes/*.py): the two loader/compute kernels rebuilt with -DDPC_WS_PROBE (phases can be left out
# through DPC_WS_DBG; results are then wrong by design), linked with the product objects
PROBE_SRCS := conv_igemm_ws conv_wgrad_patch conv_halo conv_wgrad_stem
probe: all
	@mkdir -p build/probe
	for f in $(PROBE_SRCS); do $(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Iinclude -DDPC_WS_PROBE -c $(CSRC)/$$f.hip -o build/probe/$$f.o || exit 1; done
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o scripts/probes/libdpc_probe.so $(patsubst %,build/probe/%.o,$(PROBE_SRCS)) \
		$(filter-out $(patsubst %,build/hip/%.o,$(PROBE_SRCS)),$(OBJS))

# the round-3 hazard re-created on purpose (conv_igemm_ws.hip WS_RETIRE_TAIL_READS left out): the positive control of
# scripts/probes/squat_probe.py and tests/test_cotenant_gpu.py -- never loaded by the product
nofix: all
	@mkdir -p build/nofix
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Iinclude -DDPC_WS_NOFIX -c $(CSRC)/conv_igemm_ws.hip -o build/nofix/conv_igemm_ws.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o scripts/probes/libdpc_nofix.so build/nofix/conv_igemm_ws.o \
		$(filter-out build/hip/conv_igemm_ws.o,$(OBJS))

clean:
	rm -rf build dpc_amd/libdpc_hip.so $(EMU)/libdpc_emu.so

.PHONY: all emu probe nofix clean
