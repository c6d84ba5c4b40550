"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate passes as
MI355X_MICROARCH.md prescribes).  gfx950 correction: FETCH_SIZE reports 1/2 of a wide coalesced stream's
bytes -> doubled; WRITE_SIZE is exact.  Units in the CSV are KiB.  Writes profiles/<tag>_traffic.json."""
import collections, csv, glob, json, os, re, sys
fetch_dir, write_dir, out = sys.argv[1], sys.argv[2], sys.argv[3]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_hash  # noqa: E402  (bench.py trusts the file only while the kernel sources still hash to this)


def load(d, counter):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    agg, cnt = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        n = re.sub(r"^void ", "", n)
        n = re.sub(r"\(.*", "", n).replace(", ", ",")
        agg[n] += float(r["Counter_Value"])
        cnt[n] += 1
    return {k: agg[k] / cnt[k] for k in agg}, cnt


fe, cf = load(fetch_dir, "FETCH_SIZE")
wr, cw = load(write_dir, "WRITE_SIZE")
res = {}
for k in sorted(set(fe) | set(wr)):
    f, w = fe.get(k, 0.0), wr.get(k, 0.0)
    res[k] = dict(fetch_kib_raw=round(f, 1), write_kib=round(w, 1), launches=int(cf.get(k, cw.get(k, 0))),
                  hbm_bytes_per_launch=int((2.0 * f + w) * 1024))
json.dump(dict(note="avg per launch over the bench's kernel mix; FETCH_SIZE doubled (gfx950 half-count of wide reads)",
               csrc_hash=csrc_hash(), kernels=res), open(out, "w"), indent=1)
print(json.dumps({k: v["hbm_bytes_per_launch"] for k, v in list(res.items())[:8]}))
