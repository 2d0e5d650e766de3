# Build of the product libraries (HIP gfx950 filter + C++ host plumbing) and of the test oracle (C).
# `python -c "import __graft_entry__ as g; g.build()"` drives the same targets.
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
HIPFLAGS = --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function
CXX     ?= g++
CXXFLAGS = -O3 -std=c++17 -fPIC -Wall -Wextra -pthread
CC      ?= gcc
CFLAGS   = -O3 -std=c11 -fPIC -shared -Wall -Wextra

PKG  = bucket-map_amd
HOST = $(PKG)/host
HOST_HDRS = $(wildcard $(HOST)/*.h) include/bmf.h

all: $(PKG)/libbmf.so $(PKG)/libbmhost.so oracle/libbm_oracle.so

$(PKG)/libbmf.so: $(PKG)/csrc/bmf_api.hip $(PKG)/csrc/bmf_kernels.hip.h include/bmf.h
	$(HIPCC) $(HIPFLAGS) -o $@ $(PKG)/csrc/bmf_api.hip

$(PKG)/libbmhost.so: $(HOST)/bm_host_api.cpp $(HOST_HDRS)
	$(CXX) $(CXXFLAGS) -shared -o $@ $(HOST)/bm_host_api.cpp

oracle/libbm_oracle.so: oracle/bm_oracle.c oracle/bm_oracle.h
	$(CC) $(CFLAGS) -o $@ oracle/bm_oracle.c -lm

clean:
	rm -f $(PKG)/libbmf.so $(PKG)/libbmhost.so oracle/libbm_oracle.so

.PHONY: all clean
