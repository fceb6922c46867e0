#!/bin/bash
# Builds variants of the library that differ in the -D flags ldpc.hip is compiled with (kernel experiments):
#   tools/build_variants.sh name1:"-DX=1 -DY=0" name2:"..."   ->  mercury_amd/_variants/lib_<name>.so
# (the normal build must exist: the other objects are taken from mercury_amd/_build)
set -e
cd "$(dirname "$0")/.."
python -c "from mercury_amd import build as b; b.build()"
mkdir -p mercury_amd/_variants
B=mercury_amd/_build
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-atomic-optimizer-strategy=None -Wno-unused-result -Wno-deprecated-declarations $flags -I mercury_amd/csrc -c mercury_amd/csrc/ldpc.hip -o $B/ldpc.$name.o &
done
wait
for spec in "$@"; do
  name=${spec%%:*}
  objs=$(ls $B/*.o | grep -v "ldpc\.[a-zA-Z0-9_]*\.o$" | grep -v "/ldpc.hip.o$")
  hipcc --offload-arch=gfx950 -shared -fPIC -o mercury_amd/_variants/lib_$name.so $objs $B/ldpc.$name.o -lpthread -lrt
  echo built mercury_amd/_variants/lib_$name.so
done
