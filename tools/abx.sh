#!/bin/bash
# A/B/C... of builds of the library on ONE box (the boxes' clocks differ by more than most kernel changes): build/ab/<tag>.so in turn (tools/build_variant.sh),
# ROUNDS times alternating, the resident steps in the step's own arrangement (and alone with ABX_ALONE=1).  A variant is TAG or TAG@VAR=VALUE (an
# environment setting for that run).  usage (through gpurun): bash tools/abx.sh o5@MP3S_RATE_PERSIST=0 o5 o6 ...
cd "$GRAFT_REPO_ROOT"
L=mp3-steganography-lib_amd/mp3stego/libmp3s_hip.so
cp $L /tmp/keep.so
for r in $(seq ${ROUNDS:-2}); do
  for spec in "$@"; do
    v=${spec%%@*}; e=""; [ "$spec" != "$v" ] && e=${spec#*@}
    cp mp3-steganography-lib_amd/build/ab/$v.so $L
    echo "== $spec (round $r)"; env $e bash tools/kb.sh ${ABX_ARGS}
    [ -n "$ABX_ALONE" ] && { echo "-- $spec alone"; env $e bash tools/kb.sh --no-overlap; }
  done
done
cp /tmp/keep.so $L
