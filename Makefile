# Builds libvo_hip.so (HIP kernels + C ABI, gfx950 only) in-tree.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC := visual_odometry_amd/csrc
SRCS := $(CSRC)/vo_context.hip $(CSRC)/vo_orb.hip $(CSRC)/vo_match.hip $(CSRC)/vo_ransac.hip $(CSRC)/vo_pipeline.hip
HDRS := $(CSRC)/vo_internal.h $(CSRC)/orb_pattern.inc include/vo_hip.h
OBJS := $(SRCS:.hip=.o)
FLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function

all: visual_odometry_amd/libvo_hip.so

%.o: %.hip $(HDRS)
	$(HIPCC) $(FLAGS) -c $< -o $@

visual_odometry_amd/libvo_hip.so: $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(OBJS) visual_odometry_amd/libvo_hip.so

.PHONY: all oracle clean
