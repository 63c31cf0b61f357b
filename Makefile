# Builds libaomhip.so (the product: HIP kernels + C ABI, gfx950 only) and the CPU oracle
# (test infrastructure).  `make -j4` here cross-compiles without a GPU.
HIPCC ?= /opt/rocm/bin/hipcc
PKG := aom-av1-psy_amd
CSRC := $(PKG)/csrc
LIBDIR := $(PKG)/lib
HIPFLAGS ?= --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -I$(CSRC) -Wall -Wno-unused-function
SRCS := $(wildcard $(CSRC)/*.hip)
CPPS := $(wildcard $(CSRC)/*.cpp)
HOSTC := $(wildcard $(PKG)/host/*.c)
OBJS := $(patsubst $(CSRC)/%.hip,build/%.o,$(SRCS)) $(patsubst $(CSRC)/%.cpp,build/%.o,$(CPPS)) $(patsubst $(PKG)/host/%.c,build/host_%.o,$(HOSTC))

# Objects also depend on a stamp named after the compiler version + flags: a build/ directory left behind by another toolchain or another
# flag set is rebuilt, not reused (timestamps alone would accept it).
STAMP := build/.toolchain_$(shell (echo '$(HIPCC) $(HIPFLAGS)'; $(HIPCC) --version 2>/dev/null | head -2; gcc --version | head -1) | md5sum | cut -c1-16)
$(STAMP):
	@mkdir -p build
	@rm -f build/.toolchain_*
	@touch $@

all: lib oracle
lib: $(LIBDIR)/libaomhip.so
oracle:
	$(MAKE) -C oracle

build/%.o: $(CSRC)/%.hip $(wildcard $(CSRC)/*.h) $(wildcard $(CSRC)/*.inc) include/aomhip.h $(STAMP)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

build/%.o: $(CSRC)/%.cpp $(wildcard $(CSRC)/*.h) include/aomhip.h $(STAMP)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

# the host-side modules are plain C99 (the reference is a C code base): compiled with gcc, no HIP in them
build/host_%.o: $(PKG)/host/%.c include/aomhip.h $(STAMP)
	@mkdir -p build
	gcc -std=c99 -pedantic -Wall -Wextra -Werror -O2 -fPIC -Iinclude -c $< -o $@

$(LIBDIR)/libaomhip.so: $(OBJS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o $@ $(OBJS) -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib

clean:
	rm -rf build $(LIBDIR)/libaomhip.so
	$(MAKE) -C oracle clean
# a plain-C99 host program against the C ABI (the reference is a C code base): compiles with gcc, links only libaomhip.so
demo: build/c_host_demo
build/c_host_demo: examples/c_host_demo.c include/aomhip.h $(LIBDIR)/libaomhip.so
	gcc -std=c99 -pedantic -Wall -Wextra -Werror -O2 -Iinclude $< -L$(LIBDIR) -laomhip -Wl,-rpath,'$$ORIGIN/../$(LIBDIR)' -o $@

.PHONY: all lib oracle clean demo prof exp

# phase timing build of the strip-walking SAD kernel (tools/gpu_sb_prof.py): the product library with sad_sb.hip compiled -DAOMHIP_SB_PROF
prof: build/prof/libaomhip_prof.so
build/prof/libaomhip_prof.so: $(OBJS) $(CSRC)/sad_sb.hip
	@mkdir -p build/prof
	$(HIPCC) $(HIPFLAGS) -DAOMHIP_SB_PROF -DAOMHIP_SB_DBG_KNOBS=1 -c $(CSRC)/sad_sb.hip -o build/prof/sad_sb.o
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o $@ $(filter-out build/sad_sb.o,$(OBJS)) build/prof/sad_sb.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib

# kernel experiments on sad_sb.hip without the 3-minute rebuild: only the 16x16 instantiations (AOMHIP_LIB=build/exp/libaomhip_exp.so
# python tools/gpu_ab_sadsb.py ...; the _prof one for tools/gpu_sb_prof.py).  Not the product library.
exp: build/exp/libaomhip_exp.so build/exp/libaomhip_exp_prof.so
build/exp/libaomhip_exp.so: $(CSRC)/sad_sb.hip $(filter-out build/sad_sb.o,$(OBJS))
	@mkdir -p build/exp
	$(HIPCC) $(HIPFLAGS) -DAOMHIP_SB_ONLY_16 -DAOMHIP_SB_DBG_KNOBS=1 $(EXPFLAGS) -c $(CSRC)/sad_sb.hip -o build/exp/sad_sb.o
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o $@ $(filter-out build/sad_sb.o,$(OBJS)) build/exp/sad_sb.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
build/exp/libaomhip_exp_prof.so: $(CSRC)/sad_sb.hip $(filter-out build/sad_sb.o,$(OBJS))
	@mkdir -p build/exp
	$(HIPCC) $(HIPFLAGS) -DAOMHIP_SB_ONLY_16 -DAOMHIP_SB_DBG_KNOBS=1 $(EXPFLAGS) -DAOMHIP_SB_PROF -c $(CSRC)/sad_sb.hip -o build/exp/sad_sb_prof.o
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o $@ $(filter-out build/sad_sb.o,$(OBJS)) build/exp/sad_sb_prof.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
