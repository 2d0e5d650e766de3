# Build of the product (HIP gfx950 filter, C++ host plumbing, `bucketmap` CLI) and of the test
# infrastructure (C oracle, oracle-backed CLI).  `python -c "import __graft_entry__ as g; g.build()"`
# drives the same targets.  The reference itself cannot be built here (SeqAn3/Sharg absent): no _ref.
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
HIPFLAGS = --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function
CXX     ?= g++
CXXFLAGS = -O3 -std=c++17 -fPIC -Wall -Wextra -pthread
CC      ?= gcc
CFLAGS   = -O3 -std=c11 -fPIC -shared -Wall -Wextra

PKG  = bucket-map_amd
HOST = $(PKG)/host
HOST_HDRS = $(wildcard $(HOST)/*.h) include/bmf.h include/bml.h include/bmv.h

PRODUCT = $(PKG)/libbmf.so $(PKG)/libbmhost.so $(PKG)/bucketmap $(PKG)/bucketmap_align $(PKG)/mapper_test
TESTINFRA = oracle/libbm_oracle.so tests/cpp/bucketmap_oracle tests/cpp/bucketmap_align_oracle tests/cpp/umm_order

all: $(PRODUCT) $(TESTINFRA)

# one product library: the candidate-bucket filter (bmf_*), the locator scan (bml_*), the verifier (bmv_*)
CSRC = $(PKG)/csrc
$(CSRC)/%.o: $(CSRC)/%.hip $(wildcard $(CSRC)/*.h) include/bmf.h include/bml.h include/bmv.h
	$(HIPCC) $(HIPFLAGS) -Wno-unused-result -c -o $@ $<

$(PKG)/libbmf.so: $(CSRC)/bmf_api.o $(CSRC)/bml_api.o $(CSRC)/bmv_api.o $(CSRC)/bmv_variants.o $(CSRC)/bmv_variants2.o $(CSRC)/bmv_variants3.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -o $@ $^

$(PKG)/libbmhost.so: $(HOST)/bm_host_api.cpp $(HOST_HDRS)
	$(CXX) $(CXXFLAGS) -shared -o $@ $(HOST)/bm_host_api.cpp

# the command-line tool: GPU mapper behind bm::mapper
$(PKG)/bucketmap: $(HOST)/main.cpp $(HOST)/make_mapper_gpu.cpp $(HOST_HDRS) $(PKG)/libbmf.so
	$(CXX) $(CXXFLAGS) -o $@ $(HOST)/main.cpp $(HOST)/make_mapper_gpu.cpp -L$(PKG) -lbmf -Wl,-rpath,'$$ORIGIN'

# same sources with -DBM_ALIGN: every located candidate goes through the alignment verifier
# (the reference builds `bucketmap_align` the same way, bucket_map/CMakeLists.txt:138)
$(PKG)/bucketmap_align: $(HOST)/main.cpp $(HOST)/make_mapper_gpu.cpp $(HOST_HDRS) $(PKG)/libbmf.so
	$(CXX) $(CXXFLAGS) -DBM_ALIGN -o $@ $(HOST)/main.cpp $(HOST)/make_mapper_gpu.cpp -L$(PKG) -lbmf -Wl,-rpath,'$$ORIGIN'

# the reference's mapper benchmark (mapper_test.cpp): _query_file + _check_ground_truth on the GPU filter
$(PKG)/mapper_test: $(HOST)/mapper_test.cpp $(HOST_HDRS) $(PKG)/libbmf.so
	$(CXX) $(CXXFLAGS) -o $@ $(HOST)/mapper_test.cpp -L$(PKG) -lbmf -Wl,-rpath,'$$ORIGIN'

ORACLE_SRC = oracle/bm_oracle.c oracle/bm_locator_oracle.c oracle/bm_align_oracle.c
ORACLE_HDR = oracle/bm_oracle.h oracle/bm_locator_oracle.h oracle/bm_align_oracle.h
oracle/libbm_oracle.so: $(ORACLE_SRC) $(ORACLE_HDR)
	$(CC) $(CFLAGS) -o $@ $(ORACLE_SRC) -lm

# TEST ONLY: same main.cpp / locator / SAM code with the CPU oracle plugged in behind bm::mapper
tests/cpp/%.o: oracle/%.c $(ORACLE_HDR)
	$(CC) -O3 -std=c11 -fPIC -Wall -Wextra -c -o $@ $<

ORACLE_OBJ = tests/cpp/bm_oracle.o tests/cpp/bm_locator_oracle.o tests/cpp/bm_align_oracle.o
tests/cpp/bucketmap_oracle: $(HOST)/main.cpp tests/cpp/make_mapper_oracle.cpp $(ORACLE_OBJ) $(HOST_HDRS) $(PKG)/libbmf.so
	$(CXX) $(CXXFLAGS) -o $@ $(HOST)/main.cpp tests/cpp/make_mapper_oracle.cpp $(ORACLE_OBJ) -L$(PKG) -lbmf -Wl,-rpath,'$$ORIGIN/../../$(PKG)' -lm

tests/cpp/bucketmap_align_oracle: $(HOST)/main.cpp tests/cpp/make_mapper_oracle.cpp $(ORACLE_OBJ) $(HOST_HDRS) $(PKG)/libbmf.so
	$(CXX) $(CXXFLAGS) -DBM_ALIGN -o $@ $(HOST)/main.cpp tests/cpp/make_mapper_oracle.cpp $(ORACLE_OBJ) -L$(PKG) -lbmf -Wl,-rpath,'$$ORIGIN/../../$(PKG)' -lm

# TEST ONLY, not part of `all`: the oracle-backed tools under AddressSanitizer + UBSan (CPU build; GPU sanitizers are
# not available on the pool).  tests/test_sanitizers.py builds them and runs them on the SAM fixture's inputs.
SANITIZE = -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -std=c++17 -pthread
tests/cpp/bucketmap_oracle_asan: $(HOST)/main.cpp tests/cpp/make_mapper_oracle.cpp $(ORACLE_SRC) $(HOST_HDRS) $(PKG)/libbmf.so
	$(CXX) $(SANITIZE) -o $@ $(HOST)/main.cpp tests/cpp/make_mapper_oracle.cpp $(ORACLE_SRC) -L$(PKG) -lbmf -Wl,-rpath,'$$ORIGIN/../../$(PKG)' -lm

tests/cpp/bucketmap_align_oracle_asan: $(HOST)/main.cpp tests/cpp/make_mapper_oracle.cpp $(ORACLE_SRC) $(HOST_HDRS) $(PKG)/libbmf.so
	$(CXX) $(SANITIZE) -DBM_ALIGN -o $@ $(HOST)/main.cpp tests/cpp/make_mapper_oracle.cpp $(ORACLE_SRC) -L$(PKG) -lbmf -Wl,-rpath,'$$ORIGIN/../../$(PKG)' -lm

# TEST ONLY: checks the one assumption the locator oracle imports from libstdc++ (equal_range order)
tests/cpp/umm_order: tests/cpp/umm_order.cpp
	$(CXX) -O2 -std=c++17 -o $@ $<

# TEST ONLY, build container only (the reference does not travel): the reference-side binding compiled against the
# reference's REAL mapper.h.  The binary travels to the GPU box like the other built files.
REF ?= /root/reference
integration/_build/ref_binding: integration/ref_binding_main.cpp integration/gpu_q_gram_mapper.h include/bmf.h $(PKG)/libbmf.so
	mkdir -p integration/_build
	$(CXX) -O2 -std=c++17 -Wall -Wextra -o $@ integration/ref_binding_main.cpp -I$(REF)/bucket_map -Iintegration -Iinclude \
	    -L$(PKG) -lbmf -Wl,-rpath,'$$ORIGIN/../../$(PKG)'

clean:
	rm -f $(PRODUCT) $(TESTINFRA) $(CSRC)/*.o tests/cpp/*.o

.PHONY: all clean
