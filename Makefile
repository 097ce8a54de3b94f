# Top-level build: host C objects (gcc) + HIP kernels (hipcc, gfx950 only) -> one in-tree shared
# library, the `biscuit_align` CLI, and (test infrastructure) the oracle libraries.
ROCM    ?= /opt/rocm
HIPCC   ?= $(ROCM)/bin/hipcc
CC      := gcc
CFLAGS  := -O3 -DNDEBUG -std=gnu11 -fPIC -Wall -Wno-unused-function -fvisibility=hidden -Iinclude -Ibiscuit_amd/csrc/host
HIPFLAGS:= --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Iinclude -Ibiscuit_amd/csrc/host -Ibiscuit_amd/csrc/hip
BUILD   := build
HOSTSRC := $(wildcard biscuit_amd/csrc/host/*.c)
HOSTOBJ := $(patsubst biscuit_amd/csrc/host/%.c,$(BUILD)/host_%.o,$(HOSTSRC))
HIPSRC  := $(wildcard biscuit_amd/csrc/hip/*.hip)
HIPOBJ  := $(patsubst biscuit_amd/csrc/hip/%.hip,$(BUILD)/hip_%.o,$(HIPSRC))
LIB     := biscuit_amd/libbiscuit_amd.so
CLI     := biscuit_amd/biscuit_align
PORT    := oracle/liboracle_port.so

ORACLE_CLI := oracle/oracle_align
all: $(LIB) $(PORT) $(CLI) $(ORACLE_CLI)

$(BUILD)/host_%.o: biscuit_amd/csrc/host/%.c $(wildcard biscuit_amd/csrc/host/*.h) include/bsx.h
	@mkdir -p $(BUILD)
	$(CC) $(CFLAGS) -c $< -o $@

$(BUILD)/hip_%.o: biscuit_amd/csrc/hip/%.hip $(wildcard biscuit_amd/csrc/hip/*.h) $(wildcard biscuit_amd/csrc/hip/*.hpp) include/bsx.h
	@mkdir -p $(BUILD)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(HOSTOBJ) $(HIPOBJ)
	$(HIPCC) --offload-arch=gfx950 -shared -o $@ $(HOSTOBJ) $(HIPOBJ) -lz -lm -lpthread

$(CLI): biscuit_amd/csrc/cli_main.c $(LIB)
	$(CC) -O2 -Iinclude -o $@ $< -Lbiscuit_amd -lbiscuit_amd -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,$(ROCM)/lib

# test infrastructure: CPU restatement of the device kernels (never linked into $(LIB))
$(PORT): oracle/port.c $(LIB)
	$(CC) $(CFLAGS) -shared -o $@ oracle/port.c -Lbiscuit_amd -lbiscuit_amd -Wl,-rpath,'$$ORIGIN/../biscuit_amd' -lm -lpthread -ldl

$(ORACLE_CLI): oracle/oracle_align_main.c $(PORT)
	$(CC) -O2 -o $@ $< -Loracle -loracle_port -Lbiscuit_amd -lbiscuit_amd -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,'$$ORIGIN/../biscuit_amd' -Wl,-rpath,$(ROCM)/lib

ref:
	@if [ -d /root/reference/lib/aln ]; then $(MAKE) -C oracle; else echo "reference sources absent: using prebuilt oracle/_ref if present"; fi

clean:
	rm -rf $(BUILD) $(LIB) $(CLI) $(PORT) $(ORACLE_CLI)
.PHONY: all clean ref
