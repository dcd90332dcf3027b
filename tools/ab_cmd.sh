#!/bin/bash
# A/B of builds of the library on ONE box under any command: build/ab/<tag>.so in turn (tools/build_variant.sh), ROUNDS times alternating.
# usage (through gpurun): ROUNDS=2 bash tools/ab_cmd.sh "python3 tools/dec_only.py 10000 float32" base tw4 ...
cd "$GRAFT_REPO_ROOT"
L=mp3-steganography-lib_amd/mp3stego/libmp3s_hip.so
cp $L /tmp/keep.so
cmd=$1; shift
for r in $(seq ${ROUNDS:-2}); do
  for v in "$@"; do
    cp mp3-steganography-lib_amd/build/ab/$v.so $L
    echo "== $v (round $r)"; $cmd
  done
done
cp /tmp/keep.so $L
