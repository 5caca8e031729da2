#!/bin/bash
# Diagnostic: build a variant of the library that differs from the product build only in
# the flags flagger_ring.hip is compiled with.   usage: tools/build_ring_variant.sh <name> [-D...]
# Needs an up-to-date product build (python -m katsdpsigproc_amd.build_native) for the
# other objects. Output: build/variants/lib_<name>.so (time it with tools/time_fused.py).
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p build/variants
OBJ=build/variants/flagger_ring_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off \
  -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function -Wno-unused-variable "$@" \
  -c katsdpsigproc_amd/csrc/flagger_ring.hip -o $OBJ
OTHERS=$(ls katsdpsigproc_amd/_native/*.o | grep -v flagger_ring.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/lib_$NAME.so $OBJ $OTHERS
echo built build/variants/lib_$NAME.so
