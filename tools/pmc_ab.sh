#!/bin/bash
# HBM traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes) of the bench's kernels for library variants:
#   tools/pmc_ab.sh name1 name2 ...   (build/variants/lib_<name>.so)  ->  gpurun_out/pmc_ab_<name>.json
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  export RR_LIB_PATH=$root/build/variants/lib_$v.so
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $out/pmcab_$c
    rocprofv3 --pmc $c --output-format csv -d $out/pmcab_$c -- python3 $root/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile --no-fwd-only --no-epoch --no-presets --no-f32-path > /dev/null 2> $out/pmcab_${v}_$c.err
  done
  (cd $root && python3 tools/traffic_from_pmc.py $out/pmcab_FETCH_SIZE $out/pmcab_WRITE_SIZE $out/pmc_ab_$v.json > /dev/null)
  rm -rf $out/pmcab_FETCH_SIZE $out/pmcab_WRITE_SIZE
  echo "== $v"
  python3 - <<PY
import json
k=json.load(open("$out/pmc_ab_$v.json"))["kernels"]
for n,v in k.items():
    if "gather" in n: print(f"{n:40s} {v['hbm_bytes_per_launch']/1e6:8.1f} MB/launch  x{v['launches']}")
PY
done
