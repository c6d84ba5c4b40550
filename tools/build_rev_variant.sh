#!/bin/bash
# Build the library as of a git revision into build/variants/lib_<name>.so (same-box A/B against the working tree):
#   tools/build_rev_variant.sh <git-rev> <name> [extra hipcc flags]
set -e
rev=$1; name=$2; flags=$3
root=$(cd "$(dirname "$0")/.." && pwd)
d=$root/build/variants/src_$name
rm -rf $d && mkdir -p $d/reactranker_amd/csrc $d/include
for f in gather.hip elementwise.hip loss.hip linear.hip plan.hip pack.cpp rr_common.h; do
  git -C $root show $rev:reactranker_amd/csrc/$f > $d/reactranker_amd/csrc/$f
done
git -C $root show $rev:include/reactranker_hip.h > $d/include/reactranker_hip.h
cd $d/reactranker_amd/csrc
for f in gather elementwise loss linear plan; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $flags -c $f.hip -o $f.o 2>/dev/null &
done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -x hip -c pack.cpp -o pack.o 2>/dev/null &
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 gather.o elementwise.o loss.o pack.o linear.o plan.o -o $root/build/variants/lib_$name.so
rm -rf $d
echo built $name from $rev
