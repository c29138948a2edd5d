# Builds libvo_hip.so (HIP kernels + C ABI, gfx950 only) in-tree.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC := visual_odometry_amd/csrc
SRCS := $(CSRC)/vo_context.hip $(CSRC)/vo_orb.hip $(CSRC)/vo_match.hip $(CSRC)/vo_ransac.hip $(CSRC)/vo_pipeline.hip
HDRS := $(CSRC)/vo_internal.h $(CSRC)/orb_pattern.inc include/vo_hip.h
OBJS := $(SRCS:.hip=.o)
FLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function

all: visual_odometry_amd/libvo_hip.so visual_odometry_amd/libvisual_odometry_hip.so tests/cpp/test_dropin tests/cpp/test_callers visual_odometry_amd/vo_main visual_odometry_amd/vo_benchmark

%.o: %.hip $(HDRS)
	$(HIPCC) $(FLAGS) -c $< -o $@

visual_odometry_amd/libvo_hip.so: $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

# C++20 drop-in layer (the reference's interfaces over the C ABI, plus the callers either side
# of the path: image loader, trajectory, batched sequence helper) and its test binaries
CXX ?= g++
HOST_SRCS := $(CSRC)/host/vo_cpp_api.cpp $(CSRC)/host/vo_trajectory.cpp $(CSRC)/host/vo_image_loader.cpp $(CSRC)/host/vo_jpeg.cpp $(CSRC)/host/vo_draw.cpp
visual_odometry_amd/libvisual_odometry_hip.so: $(HOST_SRCS) include/visual_odometry/*.hpp include/visual_odometry/compat/*.hpp include/vo_hip.h visual_odometry_amd/libvo_hip.so
	$(CXX) -std=c++20 -O2 -fPIC -shared -Iinclude -o $@ $(HOST_SRCS) -Lvisual_odometry_amd -lvo_hip -lz -Wl,-rpath,'$$ORIGIN'

LINK_DROPIN := -Lvisual_odometry_amd -lvisual_odometry_hip -lvo_hip -Wl,-rpath,'$$ORIGIN/../../visual_odometry_amd' -Wl,-rpath-link,visual_odometry_amd
tests/cpp/test_dropin: tests/cpp/test_dropin.cpp visual_odometry_amd/libvisual_odometry_hip.so
	$(CXX) -std=c++20 -O2 -Iinclude -o $@ tests/cpp/test_dropin.cpp $(LINK_DROPIN)

# host-only checks of trajectory and image loader (no GPU call)
tests/cpp/test_callers: tests/cpp/test_callers.cpp visual_odometry_amd/libvisual_odometry_hip.so
	$(CXX) -std=c++20 -O2 -Iinclude -o $@ tests/cpp/test_callers.cpp $(LINK_DROPIN)

# command-line front end (the reference tool's flags and output)
visual_odometry_amd/vo_main: apps/vo_main.cpp visual_odometry_amd/libvisual_odometry_hip.so
	$(CXX) -std=c++20 -O2 -Iinclude -pthread -o $@ apps/vo_main.cpp -Lvisual_odometry_amd -lvisual_odometry_hip -lvo_hip -Wl,-rpath,'$$ORIGIN' -Wl,-rpath-link,visual_odometry_amd

visual_odometry_amd/vo_benchmark: apps/vo_benchmark.cpp visual_odometry_amd/libvisual_odometry_hip.so
	$(CXX) -std=c++20 -O2 -Iinclude -o $@ apps/vo_benchmark.cpp -Lvisual_odometry_amd -lvisual_odometry_hip -lvo_hip -Wl,-rpath,'$$ORIGIN' -Wl,-rpath-link,visual_odometry_amd

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(OBJS) visual_odometry_amd/libvo_hip.so visual_odometry_amd/libvisual_odometry_hip.so tests/cpp/test_dropin tests/cpp/test_callers visual_odometry_amd/vo_main visual_odometry_amd/vo_benchmark

.PHONY: all oracle clean
