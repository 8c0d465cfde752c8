#!/bin/bash
# kernel experiments: tools/build_variant.sh NAME FILE.hip "-DFLAG=..."  ->  kaldi-lattice-utils_amd/lib/variants/libklat_NAME.so
# (the library with FILE.hip recompiled under the flags; timings only, run with KLAT_LIB_OVERRIDE)
set -e
PKG=$(dirname "$0")/../kaldi-lattice-utils_amd
name=$1; file=$2; flags=$3
mkdir -p $PKG/lib/variants $PKG/build/variants
obj=$PKG/build/variants/${file%.hip}_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $flags -c $PKG/csrc/$file -o $obj
others=$(ls $PKG/build/*.o | grep -v "/${file%.hip}.o" | grep -v klat_dist.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/lib/variants/libklat_$name.so $obj $others -lpthread
echo built $PKG/lib/variants/libklat_$name.so
