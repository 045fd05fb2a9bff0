# Build everything in-tree (built .so/.o are git-ignored but travel to the GPU box with gpurun).
#   make            -> oracle + tools + product library + CLIs
#   make oracle     -> oracle/liboracle.so   (CPU checker; test infrastructure only)
#   make product    -> unicore_amd/libunicore_cluster.so, bin/unicore, bin/foldseek (shim)
HIPCC   ?= /opt/rocm/bin/hipcc
CC      ?= gcc
CXX     ?= g++
ARCH    ?= gfx950
OPT     ?= -O3

.PHONY: all oracle oracle-asan tools product clean
all: oracle tools product

# ---------------------------------------------------------------- oracle (plain C, OpenMP)
oracle: oracle/liboracle.so
# -march=x86-64-v3 (AVX2), not -march=native: the .so is built in the dev container and runs on the GPU box's host CPU
oracle/liboracle.so: oracle/uc_oracle.c oracle/uc_simd.c oracle/uc_oracle.h
	$(CC) -std=c11 $(OPT) -march=x86-64-v3 -fopenmp -fPIC -shared -Wall -Wextra -D_POSIX_C_SOURCE=200809L -o $@ oracle/uc_oracle.c oracle/uc_simd.c -lm

# address + undefined-behaviour sanitizer build of the checker (tests/test_oracle_kat.py runs a pipeline through it once)
oracle-asan: oracle/_asan/liboracle_asan.so
oracle/_asan/liboracle_asan.so: oracle/uc_oracle.c oracle/uc_simd.c oracle/uc_oracle.h
	@mkdir -p oracle/_asan
	$(CC) -std=c11 -O1 -g -march=x86-64-v3 -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -fPIC -shared -D_POSIX_C_SOURCE=200809L -o $@ oracle/uc_oracle.c oracle/uc_simd.c -lm

# ---------------------------------------------------------------- tools
tools: bin/gen_synth
bin/gen_synth: tools/gen_synth.c
	@mkdir -p bin
	$(CC) -std=gnu11 -O2 -Wall -pthread -o $@ $< -lm

# ---------------------------------------------------------------- product (C++17 host + HIP kernels, gfx950 only)
CSRC    := unicore_amd/csrc
HOSTSRC := $(wildcard $(CSRC)/*.cpp)
HIPSRC  := $(wildcard $(CSRC)/*.hip)
HOSTOBJ := $(HOSTSRC:.cpp=.o)
HIPOBJ  := $(HIPSRC:.hip=.o)
HDRS    := $(wildcard $(CSRC)/*.h $(CSRC)/*.hpp include/*.h)
CXXFLAGS := -std=c++17 $(OPT) -fPIC -Wall -Wextra -Iinclude -I$(CSRC) -pthread
HIPFLAGS := --offload-arch=$(ARCH) -std=c++17 $(OPT) -fPIC -Iinclude -I$(CSRC) -Wall -Wno-unused-parameter

product: unicore_amd/libunicore_cluster.so bin/unicore bin/foldseek

$(CSRC)/%.o: $(CSRC)/%.cpp $(HDRS)
	$(HIPCC) -x c++ -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include $(CXXFLAGS) -c $< -o $@

# the MFMA kernels keep their accumulators in VGPRs (gfx950's register file is unified): without this flag the compiler parks
# them in AGPRs and pays a v_accvgpr_read/write for every VALU touch of an accumulator (352 copies per attention block)
$(CSRC)/uc_t5_kernels.o: HIPFLAGS += -mllvm -amdgpu-mfma-vgpr-form

$(CSRC)/%.o: $(CSRC)/%.hip $(HDRS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

unicore_amd/libunicore_cluster.so: $(HOSTOBJ) $(HIPOBJ)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^ -pthread -L/opt/rocm/lib -lrccl

bin/unicore: $(CSRC)/cli/unicore_main.cpp unicore_amd/libunicore_cluster.so $(HDRS)
	@mkdir -p bin
	$(CXX) -std=c++17 $(OPT) -Iinclude -I$(CSRC) -o $@ $< -Lunicore_amd -lunicore_cluster -Wl,-rpath,'$$ORIGIN/../unicore_amd' -pthread

bin/foldseek: $(CSRC)/cli/foldseek_shim.cpp unicore_amd/libunicore_cluster.so $(HDRS)
	@mkdir -p bin
	$(CXX) -std=c++17 $(OPT) -Iinclude -I$(CSRC) -o $@ $< -Lunicore_amd -lunicore_cluster -Wl,-rpath,'$$ORIGIN/../unicore_amd' -pthread

clean:
	rm -f oracle/*.so $(CSRC)/*.o unicore_amd/*.so bin/unicore bin/foldseek bin/gen_synth
