# Builds, in-tree:
#   raytracer_amd/lib/libmirt.so        the HIP engine behind the C ABI of include/mirt.h (gfx950)
#   raytracer_amd/lib/libmirt_host.so   the C++ drop-in surface (Scene/Camera/Config, loader, BVH, PPM)
#   raytracer_amd/bin/raytracer         the command line of the reference, rendering on the GPU
#   oracle/liboracle.so, oracle/_ref/*  the CPU checker and the reference build (test infrastructure)

HIPCC ?= /opt/rocm/bin/hipcc
CXX ?= g++
ARCH ?= gfx950

LIB := raytracer_amd/lib
BIN := raytracer_amd/bin
CSRC := raytracer_amd/csrc
HOST := raytracer_amd/host

ENGINE_SRCS := $(wildcard $(CSRC)/*.hip)
ENGINE_HDRS := $(wildcard $(CSRC)/*.h) include/mirt.h include/mirt_rng.h
HOST_SRCS := $(HOST)/mirt_math.cpp $(HOST)/scene_loader.cpp $(HOST)/image_io.cpp $(HOST)/flatten.cpp \
             $(HOST)/render.cpp $(HOST)/capi.cpp $(HOST)/synthetic.cpp
HOST_HDRS := $(wildcard $(HOST)/*.h) include/mirt.h

# -disable-machine-licm: the kernels are one long loop (the walk) around every intersector; hoisting each
# double-precision literal and address out of it costs more registers than a kernel has, and the allocator
# then parks them in vector-register lanes and reads them back inside the loop (k_trace: 168 registers
# and scratch with it, 139 and none without)
HIPFLAGS := --offload-arch=$(ARCH) -std=c++17 -O3 -fPIC -Iinclude -I$(CSRC) -Wall -Wno-unused-function \
            -ffp-contract=off -munsafe-fp-atomics -fvisibility=hidden -mllvm -disable-machine-licm $(EXTRA_HIPFLAGS)
# no FMA contraction on the host: the matrices and boxes computed here are compared bit for bit
# with the reference's
HOSTFLAGS := -std=c++17 -O2 -ffp-contract=off -fPIC -pthread -Wall -Wextra -Iinclude -I$(HOST)

.PHONY: all engine host cli oracle clean test-all benchmark $(addprefix test-,1 2 3 4 5 6 7)
all: engine host cli oracle

engine: $(LIB)/libmirt.so
host: $(LIB)/libmirt_host.so
cli: $(BIN)/raytracer

$(LIB) $(BIN):
	mkdir -p $@

# device code by hipcc; the engine's host side by the host compiler without FMA contraction, with
# hidden visibility so that only the C ABI is exported
ENGINE_HOSTFLAGS := $(HOSTFLAGS) -fvisibility=hidden -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I$(CSRC)
BUILD := $(LIB)/obj

$(BUILD): | $(LIB)
	mkdir -p $@
$(BUILD)/kernels.o: $(CSRC)/kernels.hip $(ENGINE_HDRS) Makefile | $(BUILD)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<
$(BUILD)/engine.o: $(CSRC)/engine.cpp $(ENGINE_HDRS) $(HOST)/mirt_math.h Makefile | $(BUILD)
	$(CXX) $(ENGINE_HOSTFLAGS) -c -o $@ $<
$(BUILD)/multi.o: $(CSRC)/multi.cpp $(ENGINE_HDRS) Makefile | $(BUILD)
	$(CXX) $(ENGINE_HOSTFLAGS) -c -o $@ $<
$(BUILD)/bvh_device.o: $(CSRC)/bvh_device.cpp $(ENGINE_HDRS) Makefile | $(BUILD)
	$(CXX) $(ENGINE_HOSTFLAGS) -c -o $@ $<
$(BUILD)/engine_math.o: $(HOST)/mirt_math.cpp $(HOST)/mirt_math.h | $(BUILD)
	$(CXX) $(ENGINE_HOSTFLAGS) -c -o $@ $<
$(LIB)/libmirt.so: $(BUILD)/kernels.o $(BUILD)/engine.o $(BUILD)/multi.o $(BUILD)/bvh_device.o $(BUILD)/engine_math.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^ -ldl -pthread

$(LIB)/libmirt_host.so: $(HOST_SRCS) $(HOST_HDRS) $(LIB)/libmirt.so | $(LIB)
	$(CXX) $(HOSTFLAGS) -shared -o $@ $(HOST_SRCS) -L$(LIB) -lmirt -Wl,-rpath,'$$ORIGIN'

$(BIN)/raytracer: $(HOST)/main.cpp $(LIB)/libmirt_host.so | $(BIN)
	$(CXX) $(HOSTFLAGS) -o $@ $(HOST)/main.cpp -L$(LIB) -lmirt_host -lmirt -Wl,-rpath,'$$ORIGIN/../lib'

# (after the engine: oracle/Makefile links the reference's command line against libmirt.so, see integration/)
oracle: engine
	$(MAKE) -C oracle all

# ---- render recipes: the settings of the reference's `make test-1 ... test-7`, `test-all` and `benchmark`
# (reference Makefile:161-163,185,222-301), through the drop-in command line on the GPU.  Run from fixtures/
# (the renderer looks for Textures/ in its working directory); images go to $(OUT_DIR).
OUT_DIR ?= $(CURDIR)/Output
RAYTRACER := $(CURDIR)/$(BIN)/raytracer
TEST_1 := --resolution 1920 1080 --light-intensity 0.02 --ambient-light 0.8 --samples 16 --soft-shadows 4 --glossy-reflection 4 --max-depth 5
TEST_2 := --resolution 1920 1080 --light-intensity 0.008 --ambient-light 1 --samples 4 --soft-shadows 4 --glossy-reflection 4 --max-depth 2
TEST_3 := --resolution 720 400 --light-intensity 0.02 --ambient-light 0.8 --samples 4 --soft-shadows 2 --max-depth 2
TEST_4 := --resolution 720 400 --light-intensity 0.02 --ambient-light 1 --samples 4 --soft-shadows 16 --glossy-reflection 16 --max-depth 16
TEST_5 := --resolution 720 400 --light-intensity 0.02 --ambient-light 1 --samples 4 --soft-shadows 8 --glossy-reflection 8 --motion-blur 12 --max-depth 64
TEST_6 := --resolution 720 400 --light-intensity 0.05 --ambient-light 1 --samples 16 --soft-shadows 4 --glossy-reflection 4 --max-depth 5
TEST_7 := --resolution 800 400 --samples 4 --light-intensity 0.1 --ambient-light 1 --soft-shadows 4 --glossy-reflection 4 --max-depth 12
$(OUT_DIR):
	mkdir -p $@
define RECIPE
test-$(1): cli | $$(OUT_DIR)
	cd fixtures && $$(RAYTRACER) --scene ASCII/Test$(1).txt --output $$(OUT_DIR)/rendered_bvh_Test$(1).ppm $$(TEST_$(1)) $$(RENDER_FLAGS)
endef
$(foreach n,1 2 3 4 5 6 7,$(eval $(call RECIPE,$(n))))
test-all: test-1 test-2 test-3 test-4 test-5 test-6 test-7
SCENE_FILE ?= Test5.txt
benchmark: cli | $(OUT_DIR)
	cd fixtures && time $(RAYTRACER) --scene ASCII/$(SCENE_FILE) --output $(OUT_DIR)/benchmark.ppm $(RENDER_FLAGS)

clean:
	rm -rf $(LIB) $(BIN)
	$(MAKE) -C oracle clean
