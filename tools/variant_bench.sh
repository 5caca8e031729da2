#!/bin/bash
# Diagnostic: time prebuilt library variants (build/variants/lib_*.so) on the bench shape,
# alternating so that clock drift shows up as scatter instead of bias.
for rep in 1 2; do
for f in build/variants/lib_*.so; do
  cp $f katsdpsigproc_amd/_native/libkatsdpsigproc_hip.so
  echo -n "$(basename $f): "
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*'
done
done
