"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (sums over launches)."""
import collections, csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
first = None
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    m = re.search(r'(linear_fast_kernel<\d+, \d+>|linear_kernel<\d+, \d+>|wgrad_fast_kernel<\w+, \w+, \d>|wgrad_kernel|gather_sum_kernel<\d>|gather_diff_kernel<\d>|relu_bwd_kernel|colsum_\w+_kernel|wgrad_reduce_kernel|segment_\w+|listmle_\w+|pack_weight_kernel)', n)
    k = m.group(1) if m else n[:40]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    first = first or r["Counter_Name"]
    if r["Counter_Name"] == first: cnt[k] += 1
names = sorted({c for v in agg.values() for c in v})
print("kernel".ljust(32), "n".rjust(5), " ".join(c[-22:].rjust(22) for c in names))
key = "GRBM_GUI_ACTIVE" if "GRBM_GUI_ACTIVE" in names else names[0]
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get(key, 0))[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print(k.ljust(32), str(cnt[k]).rjust(5), " ".join(("%.4g" % (v.get(c, 0) / cnt[k])).rjust(22) for c in names))
