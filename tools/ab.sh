#!/bin/bash
# A/B of two builds of the library on ONE box (clocks differ from box to box by more than most kernel changes): the resident steps with
# mp3-steganography-lib_amd/build/ab/A.so and B.so in turn, twice.  usage (through gpurun): bash tools/ab.sh [bench args]
cd "$GRAFT_REPO_ROOT"
L=mp3-steganography-lib_amd/mp3stego/libmp3s_hip.so
cp $L /tmp/keep.so
for r in 1 2; do for v in A B; do cp mp3-steganography-lib_amd/build/ab/$v.so $L; echo "== $v"; bash tools/kb.sh "$@"; done; done
cp /tmp/keep.so $L
