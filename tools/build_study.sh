#!/bin/bash
# A study build of the library: ONE source recompiled with extra -D flags, the rest taken from the regular objects.
#   tools/build_study.sh <name> <source.hip> [f16] -DFLAG...   ->  tcvom_amd/lib/libtcvom_hip_<name>.so   (load it with TCVOM_LIB=)
set -e
name=$1; src=$2; shift; shift
dir=build; extra=""
if [ "$1" = "f16" ]; then dir=build_f16; extra="-DTCVOM_F16"; shift; fi
cd "$(dirname "$0")/../tcvom_amd/csrc"
make -s -j8
mkdir -p build_study
obj=build_study/${name}_$(basename $src .hip).o
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-value"
if [ "$src" = "wsconv.hip" ]; then FL="$FL -fno-slp-vectorize"; fi
hipcc $FL $extra "$@" -c $src -o $obj
others=$(ls $dir/*.o | grep -v "/$(basename $src .hip).o")
hipcc --offload-arch=gfx950 -shared -fPIC $obj $others -o ../lib/libtcvom_hip_${name}.so
echo ../lib/libtcvom_hip_${name}.so
