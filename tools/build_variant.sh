#!/bin/bash
# A variant of the library for an A/B or a probe: tools/build_variant.sh NAME [-DFLAG=1 ...]  ->  mp3-steganography-lib_amd/build/ab/NAME.so
# (only csrc/mp3s_device.hip is compiled again with the flags; the host objects are the tree's).  Variants travel to the GPU box with gpurun
# (build/ is git-ignored, not gpurun-ignored); delete build/ab when done.
set -e
cd "$(dirname "$0")/../mp3-steganography-lib_amd"
name=$1; shift
make -j8 >/dev/null
mkdir -p build/ab
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-result -Wno-unused-value -pthread"
/opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS "$@" -c -o build/ab/$name.dev.o csrc/mp3s_device.hip
objs=$(ls build/*.o | grep -v mp3s_device.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS -shared -o build/ab/$name.so build/ab/$name.dev.o $objs
rm build/ab/$name.dev.o
ls -la build/ab/$name.so
