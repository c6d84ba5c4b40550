#!/bin/bash
# Build A/B variants of the library: tools/build_variants.sh name "-DFLAG ..." [name flags]...
# Every .hip file is recompiled with the flags (objects under build/variants/<name>/).
set -e
cd "$(dirname "$0")/../reactranker_amd/csrc"
while [ $# -gt 0 ]; do
  name=$1; flags=$2; shift 2
  d=../../build/variants/$name
  mkdir -p $d
  for f in gather elementwise loss linear ffn plan; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $flags -c $f.hip -o $d/$f.o 2>/dev/null &
  done
  wait
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $d/gather.o $d/elementwise.o $d/loss.o pack.o $d/linear.o $d/ffn.o $d/plan.o collective.o -ldl -o ../../build/variants/lib_$name.so
  rm -rf $d
  echo built $name
done
