#!/bin/bash
# Build an A/B variant of libsrhip.so with extra -D flags:  scripts/build_variant.sh NAME -DSOME_SWITCH=1 ...
# -> rusty_sr_amd/build/variants/libsrhip_NAME.so (select it with SRHIP_LIB=...; build/ travels to the GPU box, exp/ does not).
# Only the stage-kernel object is rebuilt; the other objects are those of the last build_lib().
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
V=rusty_sr_amd/build/variants
mkdir -p $V
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -pthread "$@" -x hip -c rusty_sr_amd/csrc/sr_kernels.hip -o $V/sr_kernels_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -pthread $V/sr_kernels_$NAME.o rusty_sr_amd/build/sr_aux.hip.o rusty_sr_amd/build/sr_api.cpp.o rusty_sr_amd/build/sr_comm.cpp.o -ldl -o $V/libsrhip_$NAME.so
rm -f $V/sr_kernels_$NAME.o
echo $V/libsrhip_$NAME.so
