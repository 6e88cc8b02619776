# Builds everything native in this repo for gfx950 (MI355X):
#   hwang_amd/libhwang_amd.so   product: host C++ + HIP kernels, C ABI
#   oracle/_build/*             CPU oracle (test infrastructure)
#   oracle/_ref/*               reference sources compiled in place (only
#                               where /root/reference exists)
# hipcc cross-compiles without a GPU.  `make -j8` is fine on this box.
HIPCC      ?= /opt/rocm/bin/hipcc
ARCH       ?= gfx950
CXXFLAGS   := -std=c++17 -O3 -march=x86-64-v3 -fPIC -Wall -Wno-unused-function -pthread
HIPFLAGS   := --offload-arch=$(ARCH) $(CXXFLAGS)
CSRC       := hwang_amd/csrc
BUILD      := build/obj

CPP_SRCS   := $(wildcard $(CSRC)/*.cpp $(CSRC)/h264/*.cpp $(CSRC)/hevc/*.cpp $(CSRC)/amd/*.cpp)
HIP_SRCS   := $(wildcard $(CSRC)/amd/*.hip)
OBJS       := $(patsubst %.cpp,$(BUILD)/%.o,$(CPP_SRCS)) \
              $(patsubst %.hip,$(BUILD)/%.hip.o,$(HIP_SRCS))
HDRS       := $(wildcard include/*.h $(CSRC)/*.h $(CSRC)/*/*.h)

LIB        := hwang_amd/libhwang_amd.so
# pybind11 module hwang._python (the reference's Python boundary) over the same C++ classes
PYMOD      := hwang/_python$(shell python3-config --extension-suffix)
PYBIND_INC := $(shell python3 -m pybind11 --includes)

.PHONY: all lib oracle ref corpus clean pymod
all: lib pymod oracle corpus ref

pymod: $(PYMOD)

$(PYMOD): $(CSRC)/pybind/hwang_python.cpp $(LIB) $(HDRS)
	g++ -std=c++17 -O2 -fPIC -shared -fvisibility=hidden $(PYBIND_INC) $< -o $@ -Lhwang_amd -lhwang_amd -Wl,-rpath,'$$ORIGIN/../hwang_amd'

lib: $(LIB)

$(BUILD)/%.o: %.cpp $(HDRS)
	@mkdir -p $(dir $@)
	$(HIPCC) $(CXXFLAGS) -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c $< -o $@

$(BUILD)/%.hip.o: %.hip $(HDRS)
	@mkdir -p $(dir $@)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

# -Bsymbolic: the library's own references to its C++ classes bind inside it.  Those classes carry the reference's
# names (hwang::DecoderAutomata, hwang::VideoDecoderFactory, ...); a host program that contains hwang's own classes -
# which is what INTEGRATION.md describes - would otherwise interpose them (found by tests/test_integration_shim.py:
# hwang_decoder_create() landed in the HOST's make_from_config and recursed).
$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -Wl,-Bsymbolic -o $@ $(OBJS) -pthread

oracle:
	$(MAKE) -C oracle

corpus:
	$(MAKE) -C corpus

ref:
	$(MAKE) -C oracle ref

clean:
	rm -rf build $(LIB) $(PYMOD) oracle/_build oracle/_ref corpus/_build
