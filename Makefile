# Builds libbwt_mi355x.so (HIP kernels for gfx950 + the C host layer that
# exports the reference's API), the command-line drivers and the CPU oracle.
# hipcc cross-compiles gfx950 without a GPU present.
ROCM    ?= /opt/rocm
HIPCC   ?= $(ROCM)/bin/hipcc
CC      ?= gcc
ARCH    ?= gfx950
HIPFLAGS = --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Iinclude
CFLAGS   = -O2 -std=gnu11 -fPIC -Wall -Wextra -pthread -Iinclude

CSRC  = bwt_amd/csrc
HOST  = bwt_amd/host
DRV   = bwt_amd/drivers
LIB   = bwt_amd/libbwt_mi355x.so

HIP_SRCS = $(wildcard $(CSRC)/*.hip)
HIP_OBJS = $(HIP_SRCS:.hip=.o)
HOST_SRCS = $(wildcard $(HOST)/*.c)
HOST_OBJS = $(HOST_SRCS:.c=.o)
DRIVERS  = $(patsubst %.c,%,$(wildcard $(DRV)/*.c))

all: $(LIB) drivers oracle

$(CSRC)/%.o: $(CSRC)/%.hip $(wildcard $(CSRC)/*.hpp) include/bwt_gpu.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(HOST)/%.o: $(HOST)/%.c $(wildcard include/*.h)
	$(CC) $(CFLAGS) -c $< -o $@

$(LIB): $(HIP_OBJS) $(HOST_OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -Wl,-Bsymbolic -o $@ $^ -lpthread

drivers: $(DRIVERS)

$(DRV)/%: $(DRV)/%.c $(LIB) $(wildcard include/*.h)
	$(CC) $(CFLAGS) -o $@ $< -Lbwt_amd -lbwt_mi355x -Wl,-rpath,'$$ORIGIN/..' -lpthread

oracle: $(LIB)
	$(MAKE) -C oracle all refprogs

clean:
	rm -f $(HIP_OBJS) $(HOST_OBJS) $(LIB) $(DRIVERS)
	$(MAKE) -C oracle clean

.PHONY: all drivers oracle clean
