# Top-level build.  `make all` = everything that can be built on this machine.
#   hip     mecat_amd/lib/libmecat_hip.so   HIP kernels + C-ABI (gfx950)
#   host    mecat_amd/bin/mecat2pw          C++ host driver (drop-in CLI) on top of the C-ABI
#   synth   mecat_amd/lib/libsynth.so + mecat_amd/bin/synth_reads   synthetic read generator
#   oracle  oracle/liboracle.so             CPU restatement (test infrastructure)
#   ref     oracle/_ref/*                   unmodified reference (only where /root/reference exists)
HIPCC ?= /opt/rocm/bin/hipcc
CC ?= gcc
CXX ?= g++
ARCH ?= gfx950
LIBDIR := mecat_amd/lib
BINDIR := mecat_amd/bin
CSRC := mecat_amd/csrc
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -I$(CSRC) -Wall -Wno-unused-function
HIP_SRCS := $(wildcard $(CSRC)/*.hip)
HIP_HDRS := $(wildcard $(CSRC)/*.h) $(wildcard $(CSRC)/*.hpp) include/mecat_hip.h
HIP_OBJS := $(patsubst $(CSRC)/%.hip,build/%.o,$(HIP_SRCS))
HOST_SRCS := $(wildcard mecat_amd/host/*.cpp)

.PHONY: all hip host synth oracle ref clean
all: synth oracle hip host
	@if [ -d /root/reference/src ]; then $(MAKE) ref; fi

hip: $(LIBDIR)/libmecat_hip.so
build/%.o: $(CSRC)/%.hip $(HIP_HDRS)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@
$(LIBDIR)/libmecat_hip.so: $(HIP_OBJS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(HIP_OBJS) -o $@

host: $(BINDIR)/mecat2pw $(BINDIR)/mecat2cns_partition $(BINDIR)/valu_peak $(BINDIR)/gather_peak $(BINDIR)/mecat2asmpw
# issue-rate calibration for bench.py's dw roofline (measures, computes nothing of the path)
$(BINDIR)/valu_peak: mecat_amd/tools/valu_peak.hip
	@mkdir -p $(BINDIR)
	$(HIPCC) --offload-arch=$(ARCH) -O3 $< -o $@
# random short-run fetch ceiling for the bucket walks of the seeding kernels (measures, computes nothing of the path)
$(BINDIR)/gather_peak: mecat_amd/tools/gather_peak.hip
	@mkdir -p $(BINDIR)
	$(HIPCC) --offload-arch=$(ARCH) -O3 $< -o $@
$(BINDIR)/mecat2pw: $(HOST_SRCS) $(wildcard mecat_amd/host/*.h) include/mecat_hip.h $(LIBDIR)/libmecat_hip.so
	@mkdir -p $(BINDIR)
	$(CXX) -O2 -std=c++17 -pthread -Wall -Iinclude $(HOST_SRCS) -L$(LIBDIR) -lmecat_hip -Wl,-rpath,'$$ORIGIN/../lib' -o $@

# mecat2canu's overlappers for corrected reads: one program under the four names the canu job scripts start
$(BINDIR)/mecat2asmpw: mecat_amd/asmpw/asmpw_main.cpp include/mecat_hip.h $(LIBDIR)/libmecat_hip.so
	@mkdir -p $(BINDIR)
	$(CXX) -O3 -std=c++17 -pthread -Wall -Iinclude mecat_amd/asmpw/asmpw_main.cpp -L$(LIBDIR) -lmecat_hip -Wl,-rpath,'$$ORIGIN/../lib' -o $@
	for n in mecat2trimpw mecat2asmpw50 mecat2trimpw50; do cp -f $@ $(BINDIR)/$$n; done

$(BINDIR)/mecat2cns_partition: mecat_amd/tools/partition_main.cpp mecat_amd/host/partition.cpp mecat_amd/host/partition.h
	@mkdir -p $(BINDIR)
	$(CXX) -O2 -std=c++17 -pthread -Wall mecat_amd/tools/partition_main.cpp mecat_amd/host/partition.cpp -o $@

synth: $(LIBDIR)/libsynth.so $(BINDIR)/synth_reads
$(LIBDIR)/libsynth.so: mecat_amd/tools/synth_reads.c
	@mkdir -p $(LIBDIR)
	$(CC) -O2 -fopenmp -fPIC -shared $< -o $@
$(BINDIR)/synth_reads: mecat_amd/tools/synth_reads.c
	@mkdir -p $(BINDIR)
	$(CC) -O2 -fopenmp -DSYNTH_MAIN $< -o $@

oracle:
	$(MAKE) -C oracle oracle
ref:
	$(MAKE) -C oracle ref

clean:
	rm -rf build $(LIBDIR) $(BINDIR)
	$(MAKE) -C oracle clean

# development variant of the library with dw_extend2's section clocks and row statistics compiled in (tools/dev/dw_breakdown.sh)
dwstats: $(LIBDIR)/libmecat_hip_dwstats.so
$(LIBDIR)/libmecat_hip_dwstats.so: $(HIP_SRCS) $(HIP_HDRS)
	@mkdir -p build/dwstats $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -DMECAT_DW_STATS -x hip -c $(CSRC)/align.hip -o build/dwstats/align.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC build/dwstats/align.o $(filter-out build/align.o,$(HIP_OBJS)) -o $@

# development variant with the index build's section clocks compiled in (tools/dev/idx_time.py with MECAT_HIP_LIB set)
ixstats: $(LIBDIR)/libmecat_hip_ixstats.so
$(LIBDIR)/libmecat_hip_ixstats.so: $(HIP_SRCS) $(HIP_HDRS)
	@mkdir -p build/ixstats $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -DMECAT_IX_STATS -x hip -c $(CSRC)/index_part.hip -o build/ixstats/index_part.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC build/ixstats/index_part.o $(filter-out build/index_part.o,$(HIP_OBJS)) -o $@

# development variant with section clocks in seed_filter_wide / seed_emit (tools/dev/ont_cell.py with MECAT_HIP_LIB set)
wfprof: $(LIBDIR)/libmecat_hip_wfprof.so
$(LIBDIR)/libmecat_hip_wfprof.so: $(HIP_SRCS) $(HIP_HDRS)
	@mkdir -p build/wfprof $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -DWF_PROF -x hip -c $(CSRC)/seed.hip -o build/wfprof/seed.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC build/wfprof/seed.o $(filter-out build/seed.o,$(HIP_OBJS)) -o $@

ixknock: $(LIBDIR)/libmecat_hip_ixknock.so
$(LIBDIR)/libmecat_hip_ixknock.so: $(HIP_SRCS) $(HIP_HDRS)
	@mkdir -p build/ixknock $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -DMECAT_IX_KNOCK -x hip -c $(CSRC)/index_part.hip -o build/ixknock/index_part.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC build/ixknock/index_part.o $(filter-out build/index_part.o,$(HIP_OBJS)) -o $@

# development variant with xd_extend_w's section clocks compiled in (tools/dev/xd_breakdown.py)
xdstats: $(LIBDIR)/libmecat_hip_xdstats.so
$(LIBDIR)/libmecat_hip_xdstats.so: $(HIP_SRCS) $(HIP_HDRS)
	@mkdir -p build/xdstats $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -DMECAT_XD_STATS -x hip -c $(CSRC)/xalign.hip -o build/xdstats/xalign.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC build/xdstats/xalign.o $(filter-out build/xalign.o,$(HIP_OBJS)) -o $@
