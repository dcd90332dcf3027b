#!/bin/bash
# several builds of the library on ONE box, the kernels alone: mp3-steganography-lib_amd/build/ab/<tag>.so in turn.  usage (through gpurun): bash tools/abc.sh A B C ...
cd "$GRAFT_REPO_ROOT"
L=mp3-steganography-lib_amd/mp3stego/libmp3s_hip.so
cp $L /tmp/keep.so
for v in "$@"; do cp mp3-steganography-lib_amd/build/ab/$v.so $L; echo "== $v"; bash tools/kb.sh --no-overlap; done
cp /tmp/keep.so $L
