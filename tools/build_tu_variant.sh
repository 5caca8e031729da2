#!/bin/bash
# Diagnostic: build a variant of the library that differs from the product build only in the
# flags ONE translation unit is compiled with.
#   usage: tools/build_tu_variant.sh <name> <unit (e.g. percentile)> [-D...]
# Needs an up-to-date product build for the other objects. Output: build/variants/lib_<name>.so
# (KSP_LIB=build/variants/lib_<name>.so tools/time_ops.py and the like).
set -e
cd "$(dirname "$0")/.."
NAME=$1; UNIT=$2; shift; shift
mkdir -p build/variants
OBJ=build/variants/${UNIT}_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off \
  -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function -Wno-unused-variable "$@" \
  -c katsdpsigproc_amd/csrc/$UNIT.hip -o $OBJ
OTHERS=$(ls katsdpsigproc_amd/_native/*.o | grep -v "/$UNIT.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/lib_$NAME.so $OBJ $OTHERS
echo built build/variants/lib_$NAME.so
