#!/bin/bash
# HBM traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes) of ISOLATED launches of the split GEMM forms
# (tools/linear_modes_bench.py) under environment variants:   tools/pmc_kernel_ab.sh "" RR_NO_PERSIST=1 ...
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$root
for v in "$@"; do
  tag=$(echo "${v:-base}" | tr '= ' '__')
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $out/pk_$c
    env $v rocprofv3 --pmc $c --output-format csv -d $out/pk_$c -- python3 $root/tools/linear_modes_bench.py > /dev/null 2> $out/pk_${tag}_$c.err
  done
  python3 - "$out/pk_FETCH_SIZE" "$out/pk_WRITE_SIZE" "$tag" <<'PY'
import collections, csv, glob, re, sys
def load(d, counter):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    agg, cnt = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter: continue
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*", "", n).replace(", ", ",")
        key = (n, r["Grid_Size"])
        agg[key] += float(r["Counter_Value"]); cnt[key] += 1
    return {k: agg[k] / cnt[k] for k in agg}, cnt
fe, cf = load(sys.argv[1], "FETCH_SIZE"); wr, _ = load(sys.argv[2], "WRITE_SIZE")
print("==", sys.argv[3])
for k in sorted(fe):
    if "linear_split" in k[0]:
        print(f"{k[0]:48s} grid {k[1]:>8s} x{cf[k]:4d}  fetch(2x raw) {2*fe[k]*1024/1e6:8.1f} MB  write {wr.get(k,0)*1024/1e6:8.1f} MB  sum {(2*fe[k]+wr.get(k,0))*1024/1e6:8.1f} MB")
PY
  rm -rf $out/pk_FETCH_SIZE $out/pk_WRITE_SIZE
done
