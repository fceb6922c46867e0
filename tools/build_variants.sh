#!/bin/bash
# Builds variants of the library that differ in the -D flags ONE translation unit is compiled with (kernel experiments):
#   tools/build_variants.sh name1:"-DX=1 -DY=0" name2@frontend.hip:"-DFE_Z=1" ...   ->  mercury_amd/_variants/lib_<name>.so
# (default unit: ldpc.hip; the normal build must exist: the other objects are taken from mercury_amd/_build).
# Run one with MERCURY_GPU_LIB=mercury_amd/_variants/lib_<name>.so <command>.
set -e
cd "$(dirname "$0")/.."
python -c "from mercury_amd import build as b; b.build()"
mkdir -p mercury_amd/_variants
B=mercury_amd/_build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-atomic-optimizer-strategy=None -Wno-unused-result -Wno-deprecated-declarations"
for spec in "$@"; do
  head=${spec%%:*}; flags=${spec#*:}
  name=${head%%@*}; unit=ldpc.hip; [ "$head" != "$name" ] && unit=${head#*@}
  hipcc $FLAGS $flags -I mercury_amd/csrc -c mercury_amd/csrc/$unit -o $B/variant.$name.o &
done
wait
for spec in "$@"; do
  head=${spec%%:*}
  name=${head%%@*}; unit=ldpc.hip; [ "$head" != "$name" ] && unit=${head#*@}
  objs=$(ls $B/*.o | grep -v "/variant\." | grep -v "/$unit.o$")
  hipcc --offload-arch=gfx950 -shared -fPIC -o mercury_amd/_variants/lib_$name.so $objs $B/variant.$name.o -lpthread -lrt
  echo built mercury_amd/_variants/lib_$name.so
done
