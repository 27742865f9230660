# Top-level build: product library + executables (HIP, gfx950), CPU oracle, and the test-only emulation build.
HIPCC   ?= /opt/rocm/bin/hipcc
CXX     ?= g++
CSRC    = speedseq_amd/csrc
HOST    = speedseq_amd/host
KHDRS   = $(wildcard $(CSRC)/*.h) include/ssgpu.h
HOSTHDRS = $(wildcard $(HOST)/*.h) include/ssgpu.h $(CSRC)/ssg_types.h
HIPFLAGS = --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-variable

all: lib tools oracle emu synth

lib: speedseq_amd/libssgpu.so
# Every object's prerequisites come from the compiler (-MMD): no hand-kept header lists, so no object can be stale against a shared
# declaration (round 3 shipped variant libraries linked from objects of different ages).
HIPOBJS = $(CSRC)/ssgpu_core.o $(CSRC)/ssg_index_build.o $(CSRC)/ssg_seed.o $(CSRC)/ssg_bgzf.o $(CSRC)/ssg_bam.o $(CSRC)/ssg_coll.o
$(HIPOBJS): $(CSRC)/%.o: $(CSRC)/%.cpp
	$(HIPCC) $(HIPFLAGS) -MMD -MP -x hip -c $< -o $@
$(CSRC)/sam_format.o: $(CSRC)/sam_format.cpp
	$(CXX) -O2 -std=c++17 -fPIC -MMD -MP -c $< -o $@
-include $(wildcard $(CSRC)/*.d)
LIBOBJS = $(HIPOBJS) $(CSRC)/sam_format.o
speedseq_amd/libssgpu.so: $(LIBOBJS)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC $(LIBOBJS) -o $@ -lz -ldl

# instrumented build (device phase counters, tools/dbg/phase.py); never the default library
tune: speedseq_amd/libssgpu_tune.so
speedseq_amd/libssgpu_tune.so: $(LIBOBJS)
	$(MAKE) variant NAME=tune VFLAGS="-DSSG_TUNE -DSSG_C2A_WAVES_PER_SIMD=2"

# bench utility: synthetic reference generator (one kernel launch)
synth: tools/synth/libsynthref.so tools/synth/libsynthreads.so
tools/synth/libsynthref.so: tools/synth/synth_ref.cpp
	$(HIPCC) --offload-arch=gfx950 -O3 -shared -fPIC $< -o $@
# soak utility: interleaved FASTQ records of simulated pairs, one kernel launch per chunk (tools/soak.py --stream)
tools/synth/libsynthreads.so: tools/synth/synth_reads.cpp
	$(HIPCC) --offload-arch=gfx950 -O3 -shared -fPIC $< -o $@

# random 64-byte-line gather probe (the roofline denominator of the FM-index kernels; tools/profile_round.sh runs it)
probe: tools/dbg/gather_probe tools/dbg/valu_probe tools/dbg/libm_probe
tools/dbg/libm_probe: tools/dbg/libm_probe.cpp
	$(HIPCC) --offload-arch=gfx950 -O3 -ffp-contract=off $< -o $@
tools/dbg/valu_probe: tools/dbg/valu_probe.cpp
	$(HIPCC) --offload-arch=gfx950 -O3 $< -o $@
tools/dbg/gather_probe: tools/dbg/gather_probe.cpp
	$(HIPCC) --offload-arch=gfx950 -O3 $< -o $@

# the executables speedseq.config names (reference bin/speedseq.config:13-14)
tools: bin/bwa bin/samblaster bin/sambamba bin/bamkit
# the helpers `speedseq realign` runs through $$PYTHON (bin/pyrun), under the names speedseq.config gives them
bin/bamkit: $(HOST)/bamkit_main.cpp $(HOST)/bamio.h
	$(CXX) -O2 -std=c++17 $(HOST)/bamkit_main.cpp -o $@ -lz -lpthread
	for t in bamtofastq bamheadrg bamcleanheader bamlibs; do ln -sf bamkit bin/$$t.py; done
bin/sambamba: $(HOST)/sambamba_main.cpp $(HOSTHDRS) speedseq_amd/libssgpu.so
	$(CXX) -O2 -std=c++17 $(HOST)/sambamba_main.cpp -o $@ -Lspeedseq_amd -lssgpu -lz -lpthread -Wl,-rpath,'$$ORIGIN/../speedseq_amd'
bin/bwa: $(HOST)/bwa_main.cpp $(HOSTHDRS) speedseq_amd/libssgpu.so
	$(CXX) -O2 -std=c++17 $(HOST)/bwa_main.cpp -o $@ -Lspeedseq_amd -lssgpu -lz -lpthread -Wl,-rpath,'$$ORIGIN/../speedseq_amd'
bin/samblaster: $(HOST)/samblaster_main.cpp $(HOSTHDRS) speedseq_amd/libssgpu.so
	$(CXX) -O2 -std=c++17 $(HOST)/samblaster_main.cpp -o $@ -Lspeedseq_amd -lssgpu -lz -lpthread -Wl,-rpath,'$$ORIGIN/../speedseq_amd'

oracle:
	$(MAKE) -C oracle

# host emulation of the HIP execution model: same kernel + host sources, CPU-side tests only
emu: tests/emu/libssgpu_emu.so tests/emu/bwa_emu tests/emu/samblaster_emu tests/emu/sambamba_emu tests/emu/fq_dump tests/emu/fi_test tests/emu/fi_mt_test tests/emu/scan_test
tests/emu/fi_mt_test: tools/dbg/fi_mt_test.cpp $(HOST)/fast_inflate.h $(HOST)/fast_inflate_mt.h
	$(CXX) -O2 -std=c++17 tools/dbg/fi_mt_test.cpp -o $@ -lz -lpthread
tests/emu/scan_test: tools/dbg/scan_test.cpp $(HOST)/ranksplit.h $(HOST)/ranks.h $(HOST)/fastq.h
	$(CXX) -O2 -std=c++17 -I$(HOST) tools/dbg/scan_test.cpp -o $@ -lz -lpthread
tests/emu/fi_test: tools/dbg/fi_test.cpp $(HOST)/fast_inflate.h
	$(CXX) -O2 -std=c++17 tools/dbg/fi_test.cpp -o $@ -lz
tests/emu/fq_dump: tools/dbg/fq_dump.cpp $(HOST)/fastq.h $(HOST)/fast_inflate.h
	$(CXX) -O2 -std=c++17 tools/dbg/fq_dump.cpp -o $@ -lz -lpthread
tests/emu/libssgpu_emu.so: $(CSRC)/ssgpu_core.cpp $(CSRC)/ssg_index_build.cpp $(CSRC)/ssg_seed.cpp $(CSRC)/ssg_bgzf.cpp $(CSRC)/ssg_bam.cpp $(CSRC)/ssg_coll.cpp $(CSRC)/sam_format.cpp tests/emu/emu.cpp tests/emu/emu.h $(KHDRS)
	$(CXX) -O2 -g -std=c++17 -fPIC -ffp-contract=off -DSSG_EMU -Itests/emu -I$(CSRC) -Wall -Wno-unused-function -Wno-unused-variable \
		$(CSRC)/ssgpu_core.cpp $(CSRC)/ssg_index_build.cpp $(CSRC)/ssg_seed.cpp $(CSRC)/ssg_bgzf.cpp $(CSRC)/ssg_bam.cpp $(CSRC)/ssg_coll.cpp $(CSRC)/sam_format.cpp tests/emu/emu.cpp -shared -o $@ -lpthread -lz
tests/emu/bwa_emu: $(HOST)/bwa_main.cpp $(HOSTHDRS) tests/emu/libssgpu_emu.so
	$(CXX) -O2 -std=c++17 $(HOST)/bwa_main.cpp -o $@ -Ltests/emu -lssgpu_emu -lz -lpthread -Wl,-rpath,'$$ORIGIN'
tests/emu/samblaster_emu: $(HOST)/samblaster_main.cpp $(HOSTHDRS) tests/emu/libssgpu_emu.so
	$(CXX) -O2 -std=c++17 $(HOST)/samblaster_main.cpp -o $@ -Ltests/emu -lssgpu_emu -lz -lpthread -Wl,-rpath,'$$ORIGIN'

tests/emu/sambamba_emu: $(HOST)/sambamba_main.cpp $(HOSTHDRS) tests/emu/libssgpu_emu.so
	$(CXX) -O2 -std=c++17 $(HOST)/sambamba_main.cpp -o $@ -Ltests/emu -lssgpu_emu -lz -lpthread -Wl,-rpath,'$$ORIGIN'

clean:
	rm -rf build; rm -f speedseq_amd/libssgpu.so speedseq_amd/libssgpu_*.so $(CSRC)/*.d tests/emu/libssgpu_emu.so bin/bwa bin/samblaster bin/sambamba tests/emu/bwa_emu tests/emu/samblaster_emu tests/emu/sambamba_emu tests/emu/fq_dump tests/emu/fi_test tests/emu/fi_mt_test tests/emu/scan_test $(CSRC)/*.o
	$(MAKE) -C oracle clean
.PHONY: all lib tools oracle emu clean variant tune

# A/B builds of the device library with other compile-time parameters (bench / tests: SSGPU_LIB=speedseq_amd/libssgpu_$(NAME).so); never the
# default.  The units named in VUNITS (default: all) are compiled with VFLAGS into build/$(NAME)/; the others are the product build's
# objects, which `lib` has just brought up to date against every header they include (-MMD), so no stale object can be linked.
#   make variant NAME=x VFLAGS="-D..."                   all translation units
#   make variant NAME=x VFLAGS="-D..." VUNITS=ssg_seed   only that unit (seconds)
VUNITS ?= ssgpu_core ssg_index_build ssg_seed ssg_bgzf ssg_bam ssg_coll
variant: lib
	mkdir -p build/$(NAME) && rm -f build/$(NAME)/*.o
	for u in ssgpu_core ssg_index_build ssg_seed ssg_bgzf ssg_bam ssg_coll; do \
	  case " $(VUNITS) " in *" $$u "*) $(HIPCC) $(HIPFLAGS) $(VFLAGS) -x hip -c $(CSRC)/$$u.cpp -o build/$(NAME)/$$u.o || exit 1;; \
	  *) cp $(CSRC)/$$u.o build/$(NAME)/$$u.o;; esac; done
	cp $(CSRC)/sam_format.o build/$(NAME)/sam_format.o
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC build/$(NAME)/*.o -o speedseq_amd/libssgpu_$(NAME).so -lz -ldl
	rm -rf build/$(NAME)
