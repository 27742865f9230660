# Top-level build: product library (HIP, gfx950), CPU oracle, and the test-only emulation build.
HIPCC   ?= /opt/rocm/bin/hipcc
CXX     ?= g++
CSRC    = speedseq_amd/csrc
KHDRS   = $(wildcard $(CSRC)/*.h) include/ssgpu.h
HIPFLAGS = --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-variable

all: lib oracle emu

lib: speedseq_amd/libssgpu.so
speedseq_amd/libssgpu.so: $(CSRC)/ssgpu_core.cpp $(CSRC)/sam_format.cpp $(KHDRS)
	$(HIPCC) $(HIPFLAGS) -x hip -c $(CSRC)/ssgpu_core.cpp -o $(CSRC)/ssgpu_core.o
	$(CXX) -O2 -std=c++17 -fPIC -c $(CSRC)/sam_format.cpp -o $(CSRC)/sam_format.o
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC $(CSRC)/ssgpu_core.o $(CSRC)/sam_format.o -o $@

oracle:
	$(MAKE) -C oracle

emu: tests/emu/libssgpu_emu.so
tests/emu/libssgpu_emu.so: $(CSRC)/ssgpu_core.cpp $(CSRC)/sam_format.cpp tests/emu/emu.cpp tests/emu/emu.h $(KHDRS)
	$(CXX) -O2 -g -std=c++17 -fPIC -ffp-contract=off -DSSG_EMU -Itests/emu -I$(CSRC) -Wall -Wno-unused-function -Wno-unused-variable \
		$(CSRC)/ssgpu_core.cpp $(CSRC)/sam_format.cpp tests/emu/emu.cpp -shared -o $@ -lpthread

clean:
	rm -f speedseq_amd/libssgpu.so tests/emu/libssgpu_emu.so
	$(MAKE) -C oracle clean
.PHONY: all lib oracle emu clean
