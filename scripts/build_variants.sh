#!/bin/bash
# builds libmi355_lzss.so variants with different sa_build tuning macros into compress_amd/variants/
# usage: scripts/build_variants.sh "NAME:-DFOO=1 -DBAR=2" ...
set -e
cd "$(dirname "$0")/../compress_amd/csrc"
make -s
mkdir -p ../variants
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  extra=""
  if [ "${VSRC:-sa_build.hip}" = sa_build.hip ]; then extra="-mllvm -amdgpu-atomic-optimizer-strategy=None"; fi  # as in the Makefile
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $extra $flags -c ${VSRC:-sa_build.hip} -o /tmp/var_$name.o
  objs=""
  for o in sa_build.o sa_large.o match.o emit.o decompress.o capi.o; do
    if [ "$o" = "$(basename ${VSRC:-sa_build.hip} .hip).o" ]; then objs="$objs /tmp/var_$name.o"; else objs="$objs $o"; fi
  done
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/lib_$name.so $objs
  echo built $name
done
