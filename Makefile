# Builds libsparsex.so (host preprocessor + HIP executor for gfx950) and the
# parity oracle.  hipcc cross-compiles the device code without a GPU present.
ROCM      ?= /opt/rocm
HIPCC     ?= $(ROCM)/bin/hipcc
CXX       ?= g++
CC        ?= gcc
ARCH      ?= gfx950

CSRC      := sparsex_amd/csrc
LIBDIR    := sparsex_amd/lib
OBJDIR    := build/obj
LIB       := $(LIBDIR)/libsparsex.so
SYNLIB    := $(LIBDIR)/libspxsynth.so

HOST_SRCS := common.cpp config.cpp partition.cpp stats.cpp encoder.cpp input.cpp reorder.cpp \
             csx_emit.cpp gpu_emit.cpp stream_index.cpp xwindows.cpp sxplan.cpp dist.cpp api.cpp
HOST_OBJS := $(HOST_SRCS:%.cpp=$(OBJDIR)/%.o)
HIP_OBJ   := $(OBJDIR)/spmv_kernels.o $(OBJDIR)/spmv_xw_kernels.o $(OBJDIR)/spmv_sx_kernels.o $(OBJDIR)/vec_kernels.o $(OBJDIR)/dist_kernels.o

CXXFLAGS  := -std=c++17 -O2 -g -fPIC -Wall -Iinclude -I$(CSRC) -pthread
HIPFLAGS  := --offload-arch=$(ARCH) -std=c++17 -O3 -fPIC -munsafe-fp-atomics \
             -Iinclude -I$(CSRC)

.PHONY: all lib oracle clean print-host-srcs print-hip-objs
all: lib oracle

lib: $(LIB) $(SYNLIB)

# input generators of the nlpkkt stand-in and of syn-kkt2f (bench/test data only, no SpMV code)
$(SYNLIB): tools/synth/nlpkkt_gen.c tools/synth/kkt2f_gen.c tools/synth/mm_write.c
	@mkdir -p $(LIBDIR)
	$(CC) -O3 -shared -fPIC -o $@ $^ -lm

$(OBJDIR)/%.o: $(CSRC)/%.cpp $(wildcard $(CSRC)/*.hpp) $(wildcard $(CSRC)/*.h) \
               $(wildcard include/sparsex/*.h) include/sparsex_hip.h
	@mkdir -p $(OBJDIR)
	$(CXX) $(CXXFLAGS) -c $< -o $@

$(OBJDIR)/%.o: $(CSRC)/%.hip $(wildcard $(CSRC)/*.hpp) $(wildcard $(CSRC)/*.h) \
               $(wildcard include/sparsex/*.h) include/sparsex_hip.h
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(HOST_OBJS) $(HIP_OBJ)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^ -pthread -ldl

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf build $(LIBDIR)/*.so oracle/*.so oracle/_ref

# (for tools/build_asan.sh)
print-host-srcs:
	@echo $(HOST_SRCS)
print-hip-objs:
	@echo $(HIP_OBJ)
