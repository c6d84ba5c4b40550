#!/bin/bash
# L2 (TCC) requests / hits / misses of the bond <- bonds gather against the number of sources per row (tools/gather_k_scan.py under
# rocprofv3 --pmc): tools/gather_k_pmc.sh  -> gpurun_out/gather_k_pmc.txt
set -e
root=$(pwd); out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf $out/pmc_gk
rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $out/pmc_gk -- python3 $root/tools/gather_k_scan.py > /dev/null 2> $out/gather_k_pmc.err
cd $root
python3 - $out/pmc_gk > $out/gather_k_pmc.txt <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "gather_sum_kernel" in r["Kernel_Name"]]
by = collections.OrderedDict()
for r in rows:
    by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(by)
# the scan launches K = 1..4 (sources within 34 rows), then K = 1..4 (own row), 155 launches each (5 + 5 x 30)
per = len(ids) // 8
print(f"{len(ids)} gather launches, {per} per case; per launch: L2 requests, hits, misses (128-byte lines), EA read requests")
for c in range(8):
    sel = ids[c * per + per // 2: c * per + per // 2 + 20]
    avg = {k: sum(by[i].get(k, 0.0) for i in sel) / len(sel) for k in ("TCC_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum")}
    print(f"K {c % 4 + 1} {'sources within 34 rows' if c < 4 else 'own row             '}: req {avg['TCC_REQ_sum']/1e6:7.2f} M  hit {avg['TCC_HIT_sum']/1e6:7.2f} M  miss {avg['TCC_MISS_sum']/1e6:7.2f} M  EA rd {avg['TCC_EA0_RDREQ_sum']/1e6:7.2f} M")
PY
rm -rf $out/pmc_gk
