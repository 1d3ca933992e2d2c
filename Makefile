# Builds every native piece in-tree:
#   ky_amd/lib/libkyhip.so   the product: gfx950 kernels + C ABI (include/kyhip.h)
#   ky_amd/lib/libkyhost.so  host-side C++ API (ky_amd/host/ky.hpp) behind a small C API for Python
#   oracle/libkyoracle.so    the CPU checker (test infrastructure only)
HIPCC    ?= hipcc
CXX      ?= g++
ROCM     ?= /opt/rocm
# -no-hip-rt: libkyhip.so does NOT pin a HIP runtime.  A process must hold exactly one runtime: Python callers get the
# one torch bundles (ky_amd/_abi.py loads it RTLD_GLOBAL first), C++ callers link $(ROCM)/lib/libamdhip64.so themselves.
HIPFLAGS ?= --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-bitwise-instead-of-logical -no-hip-rt -fno-slp-vectorize -fno-hip-fp32-correctly-rounded-divide-sqrt
LIBDIR   := ky_amd/lib

all: $(LIBDIR)/libkyhip.so $(LIBDIR)/libkyhost.so oracle examples

# the device headers as text inside the library: what its run-time instantiations compile (kyhip.hip, kyjit)
RTC_INC := ky_amd/csrc/ky_rtc_sources.inc
$(RTC_INC): ky_amd/csrc/ky_device.hpp ky_amd/csrc/ky_render.hpp include/kyhip.h tools/embed_sources.py
	python3 tools/embed_sources.py $@ ky_device.hpp=ky_amd/csrc/ky_device.hpp ky_render.hpp=ky_amd/csrc/ky_render.hpp ../../include/kyhip.h=include/kyhip.h

$(LIBDIR)/libkyhip.so: ky_amd/csrc/kyhip.hip ky_amd/csrc/ky_device.hpp ky_amd/csrc/ky_queue.hpp ky_amd/csrc/ky_smallpt.hpp ky_amd/csrc/ky_measure.hpp ky_amd/csrc/ky_render.hpp include/kyhip.h $(RTC_INC)
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ ky_amd/csrc/kyhip.hip

$(LIBDIR)/libkyhost.so: ky_amd/host/ky_capi.cpp ky_amd/host/ky.hpp include/kyhip.h $(LIBDIR)/libkyhip.so
	$(CXX) -O2 -std=c++17 -fPIC -Wall -shared -o $@ ky_amd/host/ky_capi.cpp -L$(LIBDIR) -lkyhip -Wl,-rpath,'$$ORIGIN'

oracle:
	$(MAKE) -C oracle

EXAMPLES := $(patsubst examples/%.cpp,examples/bin/%,$(wildcard examples/*.cpp))
examples: $(EXAMPLES)
examples/bin/%: examples/%.cpp ky_amd/host/ky.hpp include/kyhip.h $(LIBDIR)/libkyhip.so
	@mkdir -p examples/bin
	$(CXX) -O2 -std=c++17 -Wall -o $@ $< -L$(LIBDIR) -lkyhip -L$(ROCM)/lib -lamdhip64 -Wl,-rpath,'$$ORIGIN/../../$(LIBDIR)' -Wl,-rpath,$(ROCM)/lib

# the micro-benchmarks tools/final_profiles.sh runs (not part of `all`; build_variants/ is scratch that travels to the GPU box)
UBENCH := $(patsubst tools/ubench/%.hip,build_variants/%,$(wildcard tools/ubench/*.hip))
ubench: $(UBENCH)
build_variants/%: tools/ubench/%.hip
	@mkdir -p build_variants
	$(HIPCC) --offload-arch=gfx950 -O2 -Wno-unused-value -o $@ $<

clean:
	rm -rf $(LIBDIR) examples/bin $(RTC_INC)
	$(MAKE) -C oracle clean
.PHONY: all oracle examples ubench clean
