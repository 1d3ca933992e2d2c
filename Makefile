# Builds every native piece in-tree:
#   ky_amd/lib/libkyhip.so   the product: gfx950 kernels + C ABI (include/kyhip.h), five translation units under ky_amd/csrc
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

# the device headers as text inside the library: what its run-time instantiations compile (ky_jit.cpp)
CSRC    := ky_amd/csrc
RTC_INC := $(CSRC)/ky_rtc_sources.inc
DEVICE_HDRS := $(CSRC)/ky_scene.hpp $(CSRC)/ky_shard.hpp $(CSRC)/ky_device.hpp $(CSRC)/ky_render.hpp
$(RTC_INC): $(DEVICE_HDRS) include/kyhip.h tools/embed_sources.py
	python3 tools/embed_sources.py $@ ky_scene.hpp=$(CSRC)/ky_scene.hpp ky_shard.hpp=$(CSRC)/ky_shard.hpp ky_device.hpp=$(CSRC)/ky_device.hpp ky_render.hpp=$(CSRC)/ky_render.hpp ../../include/kyhip.h=include/kyhip.h

# libkyhip.so = five translation units (round 5; one 2100-line kyhip.hip before):
#   ky_launch.hip  the render kernels' table + launch path, film kernels, fp64 smallpt      ky_kat.hip   KAT kernels + entries
#   ky_pack.cpp    host: params, scene packing, occluder proof, policies, HostPool           ky_jit.cpp   run-time instantiations' code cache
#   ky_seam.cpp    host-film calls (kyhip_render / kyhip_render_multi)
# (ky_pack.cpp and ky_jit.cpp make no HIP call: `make sanitize` builds the same files with g++ -fsanitize=...)
HOST_HDRS := $(CSRC)/ky_host.hpp $(CSRC)/ky_ctx.hpp $(CSRC)/ky_scene.hpp $(CSRC)/ky_shard.hpp include/kyhip.h
OBJDIR  := build/obj
KYHIP_OBJS := $(OBJDIR)/ky_launch.o $(OBJDIR)/ky_kat.o $(OBJDIR)/ky_pack.o $(OBJDIR)/ky_jit.o $(OBJDIR)/ky_seam.o
$(OBJDIR)/ky_launch.o: $(CSRC)/ky_launch.hip $(DEVICE_HDRS) $(CSRC)/ky_queue.hpp $(CSRC)/ky_smallpt.hpp $(CSRC)/ky_measure.hpp $(HOST_HDRS)
$(OBJDIR)/ky_kat.o: $(CSRC)/ky_kat.hip $(DEVICE_HDRS) $(CSRC)/ky_measure.hpp $(HOST_HDRS)
$(OBJDIR)/ky_pack.o: $(CSRC)/ky_pack.cpp $(HOST_HDRS)
$(OBJDIR)/ky_jit.o: $(CSRC)/ky_jit.cpp $(HOST_HDRS) $(RTC_INC)
$(OBJDIR)/ky_seam.o: $(CSRC)/ky_seam.cpp $(HOST_HDRS)
$(OBJDIR)/%.o:
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) $(KYFLAGS) -c -o $@ $<
$(LIBDIR)/libkyhip.so: $(KYHIP_OBJS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=gfx950 -fPIC -no-hip-rt -shared -o $@ $(KYHIP_OBJS)

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

# ---- sanitizers on the CPU build (never on the GPU: the pool has no GPU sanitizers) --------------------------------------------------------
# The host code that needs no GPU -- ky_pack.cpp (scene packing, occluder proof, HostPool, seam lock order), ky_jit.cpp (code cache, posix_spawn) plus
# ky_hostcheck.cpp (entry points for them) -- the host mirror's C API (ky.hpp's scene graph) and the oracle, built by g++ with
# -fsanitize=address,undefined, and the first three again with -fsanitize=thread.  `make sanitize-build` builds them (tests/test_sanitize.py does that
# on demand and runs its checks in sanitized child processes); `make sanitize` also runs the CPU test suite under the address build and writes
# profiles/<round>_sanitize_summary.txt (tools/sanitize/run.sh).
SANDIR   := build/san
SANFLAGS := -O1 -g -std=c++17 -fPIC -fno-omit-frame-pointer -Wall -Wno-unused-function -D__HIP_PLATFORM_AMD__ -I$(ROCM)/include
HOSTSAN_SRC := $(CSRC)/ky_pack.cpp $(CSRC)/ky_jit.cpp $(CSRC)/ky_hostcheck.cpp
$(SANDIR)/libkyhip_host_asan.so: $(HOSTSAN_SRC) $(HOST_HDRS) $(RTC_INC)
	@mkdir -p $(SANDIR)
	$(CXX) $(SANFLAGS) -fsanitize=address,undefined -shared -o $@ $(HOSTSAN_SRC) -lpthread
$(SANDIR)/libkyhip_host_tsan.so: $(HOSTSAN_SRC) $(HOST_HDRS) $(RTC_INC)
	@mkdir -p $(SANDIR)
	$(CXX) $(SANFLAGS) -fsanitize=thread -shared -o $@ $(HOSTSAN_SRC) -lpthread
$(SANDIR)/libkyhost_asan.so: ky_amd/host/ky_capi.cpp ky_amd/host/ky.hpp include/kyhip.h
	@mkdir -p $(SANDIR)
	$(CXX) $(SANFLAGS) -fsanitize=address,undefined -shared -o $@ ky_amd/host/ky_capi.cpp -Wl,-z,lazy   # kyhip_render* stay unresolved until called (nothing in the CPU suite renders)
$(SANDIR)/libkyoracle_asan.so: oracle/ky_oracle.cpp oracle/smallpt_oracle.cpp oracle/smallpt_rewrite_oracle.cpp include/kyhip.h
	@mkdir -p $(SANDIR)
	$(CXX) $(SANFLAGS) -ffp-contract=off -fopenmp -Wno-unused-parameter -fsanitize=address,undefined -shared -o $@ oracle/ky_oracle.cpp oracle/smallpt_oracle.cpp oracle/smallpt_rewrite_oracle.cpp
$(SANDIR)/stress_tsan: tools/sanitize/stress.cpp $(HOSTSAN_SRC) $(HOST_HDRS) $(RTC_INC)
	@mkdir -p $(SANDIR)
	$(CXX) $(SANFLAGS) -fsanitize=thread -o $@ tools/sanitize/stress.cpp $(HOSTSAN_SRC) -lpthread
$(SANDIR)/stress_asan: tools/sanitize/stress.cpp $(HOSTSAN_SRC) $(HOST_HDRS) $(RTC_INC)
	@mkdir -p $(SANDIR)
	$(CXX) $(SANFLAGS) -fsanitize=address,undefined -o $@ tools/sanitize/stress.cpp $(HOSTSAN_SRC) -lpthread
sanitize-build: $(SANDIR)/libkyhip_host_asan.so $(SANDIR)/libkyhip_host_tsan.so $(SANDIR)/libkyhost_asan.so $(SANDIR)/libkyoracle_asan.so $(SANDIR)/stress_tsan $(SANDIR)/stress_asan
sanitize: sanitize-build
	bash tools/sanitize/run.sh

clean:
	rm -rf $(LIBDIR) examples/bin $(RTC_INC) $(OBJDIR) $(SANDIR)
	$(MAKE) -C oracle clean
.PHONY: all oracle examples ubench clean sanitize sanitize-build
