#!/bin/bash
# Build A/B variants of the library: tools/build_variants.sh name "-DFLAG ..." [name flags]...
set -e
cd "$(dirname "$0")/../reactranker_amd/csrc"
mkdir -p ../../build/variants
while [ $# -gt 0 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $flags -c linear.hip -o ../../build/variants/linear_$name.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 gather.o elementwise.o loss.o pack.o ../../build/variants/linear_$name.o -o ../../build/variants/lib_$name.so
  echo built $name
done
