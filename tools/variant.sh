#!/bin/bash
# Build a variant of the library with one translation unit recompiled with extra flags:  tools/variant.sh NAME kernel_x.hip -DFLAG ...
# -> tools/lib_NAME.so (use with BNMTF_LIB=...; never committed)
name=$1; src=$2; shift 2
cd $(dirname $0)/../bnmtf_amd/csrc
obj=/tmp/variant_${name}_$(basename $src .hip).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include "$@" -c $src -o $obj || exit 1
objs=$(ls build/*.o | grep -v "_timing.o" | grep -v "build/$(basename $src .hip).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/lib_$name.so $objs $obj -ldl
