#!/bin/bash
# AddressSanitizer + UBSan build of the HOST side of the library (parser, scanner, formatter, WAV / message code) with
# g++ -- GPU sanitizers are not available on this pool, and the host side is where untrusted bytes are parsed.
#   tools/asan_host/build.sh <outdir>;  <outdir>/parse_mutants <dir of files>;  <outdir>/files_messages [iterations scale]
set -e
here=$(cd "$(dirname "$0")" && pwd); root=$(cd "$here/../.." && pwd); out=${1:-/tmp/mp3s_asan}
mkdir -p "$out"
src="$root/mp3-steganography-lib_amd/csrc"
common="-std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -I$src -I$root/include"
host="$src/mp3s_host_decode.cpp $src/mp3s_host_encode.cpp $src/mp3s_host_files.cpp $src/mp3s_tables.cpp"
g++ $common "$here/parse_mutants.cpp" $host -o "$out/parse_mutants"
g++ $common "$here/files_messages.cpp" $host -o "$out/files_messages"
