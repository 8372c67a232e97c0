#!/bin/bash
# Dev helper (build host): a timing / A-B variant of the library with extra -D flags on conv_p32.hip -> build/ab/<name>.so
# usage: build_variant.sh <name> [-DP32_ABLATE=32 ...]
set -e
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/build/ab
cd $R/deepemia_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c conv_p32.hip -o /tmp/conv_p32_$name.o
objs=$(ls obj/*.o | grep -v "/conv_p32.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/ab/$name.so $objs /tmp/conv_p32_$name.o
echo built $R/build/ab/$name.so
