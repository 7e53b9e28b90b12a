#!/bin/bash
# on the GPU box: run a command with a library variant in place of libmi355_lzss.so, then put the original back
# usage: scripts/with_lib.sh compress_amd/variants/lib_x.so <command ...>
cd "$(dirname "$0")/.."
v="$1"; shift
cp compress_amd/libmi355_lzss.so /tmp/lib_orig.so && cp "$v" compress_amd/libmi355_lzss.so || exit 1
"$@"; rc=$?
cp /tmp/lib_orig.so compress_amd/libmi355_lzss.so
exit $rc
