#!/bin/bash
# Build an A/B variant of libsrhip.so with extra -D flags:  scripts/build_variant.sh NAME -DSOME_SWITCH=1 ...
# -> exp/libsrhip_NAME.so (select it with SRHIP_LIB=exp/libsrhip_NAME.so).  Only the kernel object is rebuilt.
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -pthread "$@" -x hip -c rusty_sr_amd/csrc/sr_kernels.hip -o exp/sr_kernels_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -pthread exp/sr_kernels_$NAME.o rusty_sr_amd/build/sr_aux.hip.o rusty_sr_amd/build/sr_api.cpp.o rusty_sr_amd/build/sr_comm.cpp.o -ldl -o exp/libsrhip_$NAME.so
echo exp/libsrhip_$NAME.so
