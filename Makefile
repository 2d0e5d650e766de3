# Build of the product (HIP gfx950 filter, C++ host plumbing, `bucketmap` CLI) and of the test
# infrastructure (C oracle, oracle-backed CLI).  `python -c "import __graft_entry__ as g; g.build()"`
# drives the same targets.  The reference itself cannot be built here (SeqAn3/Sharg absent): no _ref.
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
HIPFLAGS = --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function
CXX     ?= g++
CXXFLAGS = -O3 -std=c++17 -fPIC -Wall -Wextra -pthread
CC      ?= gcc
CFLAGS   = -O3 -std=c11 -fPIC -shared -Wall -Wextra

PKG  = bucket-map_amd
HOST = $(PKG)/host
HOST_HDRS = $(wildcard $(HOST)/*.h) include/bmf.h include/bml.h

PRODUCT = $(PKG)/libbmf.so $(PKG)/libbmhost.so $(PKG)/bucketmap
TESTINFRA = oracle/libbm_oracle.so tests/cpp/bucketmap_oracle tests/cpp/umm_order

all: $(PRODUCT) $(TESTINFRA)

# one product library: the candidate-bucket filter (bmf_*) and the locator scan (bml_*)
$(PKG)/libbmf.so: $(PKG)/csrc/bmf_api.hip $(PKG)/csrc/bmf_kernels.hip.h $(PKG)/csrc/bml_api.hip $(PKG)/csrc/bml_kernels.hip.h include/bmf.h include/bml.h
	$(HIPCC) $(HIPFLAGS) -Wno-unused-result -o $@ $(PKG)/csrc/bmf_api.hip $(PKG)/csrc/bml_api.hip

$(PKG)/libbmhost.so: $(HOST)/bm_host_api.cpp $(HOST_HDRS)
	$(CXX) $(CXXFLAGS) -shared -o $@ $(HOST)/bm_host_api.cpp

# the command-line tool: GPU mapper behind bm::mapper
$(PKG)/bucketmap: $(HOST)/main.cpp $(HOST)/make_mapper_gpu.cpp $(HOST_HDRS) $(PKG)/libbmf.so
	$(CXX) $(CXXFLAGS) -o $@ $(HOST)/main.cpp $(HOST)/make_mapper_gpu.cpp -L$(PKG) -lbmf -Wl,-rpath,'$$ORIGIN'

oracle/libbm_oracle.so: oracle/bm_oracle.c oracle/bm_oracle.h oracle/bm_locator_oracle.c oracle/bm_locator_oracle.h
	$(CC) $(CFLAGS) -o $@ oracle/bm_oracle.c oracle/bm_locator_oracle.c -lm

# TEST ONLY: same main.cpp / locator / SAM code with the CPU oracle plugged in behind bm::mapper
tests/cpp/%.o: oracle/%.c oracle/bm_oracle.h oracle/bm_locator_oracle.h
	$(CC) -O3 -std=c11 -fPIC -Wall -Wextra -c -o $@ $<

tests/cpp/bucketmap_oracle: $(HOST)/main.cpp tests/cpp/make_mapper_oracle.cpp tests/cpp/bm_oracle.o tests/cpp/bm_locator_oracle.o $(HOST_HDRS) $(PKG)/libbmf.so
	$(CXX) $(CXXFLAGS) -o $@ $(HOST)/main.cpp tests/cpp/make_mapper_oracle.cpp tests/cpp/bm_oracle.o tests/cpp/bm_locator_oracle.o -L$(PKG) -lbmf -Wl,-rpath,'$$ORIGIN/../../$(PKG)' -lm

# TEST ONLY: checks the one assumption the locator oracle imports from libstdc++ (equal_range order)
tests/cpp/umm_order: tests/cpp/umm_order.cpp
	$(CXX) -O2 -std=c++17 -o $@ $<

clean:
	rm -f $(PRODUCT) $(TESTINFRA)

.PHONY: all clean
