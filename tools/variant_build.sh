#!/bin/bash
# builds a variant of the library with extra compiler flags next to the default one:
#   tools/variant_build.sh <name> <flags...>   ->  volxel_amd/libvolxel_hip_<name>.so  (use with VOLXEL_HIP_LIB=...)
set -e
name=$1; shift
cd "$(dirname "$0")/../volxel_amd/csrc"
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function "$@" -c vx_api.hip -o $tmp/vx_api.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvolxel_hip_$name.so $tmp/vx_api.o brick_builder.o dicom_reader.o -lpthread
rm -rf $tmp
ls -la ../libvolxel_hip_$name.so
