"""Per-stream kernel sequence of ONE training step from a rocprofv3 --kernel-trace CSV (start offset, duration, name),
plus per-stream busy time.  Usage: python tools/step_timeline.py <kernel_trace.csv> [step_from_end=2] [--brief] [--marks-per-step N] [--mark KERNEL]
(N = 2 for losses whose forward and backward are the same kernel symbol: ListNet, evidential)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 2
brief = "--brief" in sys.argv
ev = []
for r in rows:
    n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Stream_Id") or r.get("Queue_Id")))
ev.sort()
if "--mark" in sys.argv:                                   # e.g. --mark ranking_metrics_kernel: one per validation step
    key = sys.argv[sys.argv.index("--mark") + 1]
    marks = [e[0] for e in ev if key in e[2]]
else:
    marks = [e[0] for e in ev if "listmle_fwd" in e[2] or "listmle_step" in e[2] or "listnet_kernel" in e[2] or "ranknet_fwd" in e[2] or "evidential_kernel" in e[2]]
if "--marks-per-step" in sys.argv:
    marks = marks[::int(sys.argv[sys.argv.index("--marks-per-step") + 1])]
lo, hi = marks[-1 - back], marks[-back]
print(f"step window {(hi - lo) / 1e6:.3f} ms")
streams = {}
for s, e, n, q in ev:
    if lo <= s < hi:
        streams.setdefault(q, []).append((s, e, n))
for q, l in streams.items():
    busy = sum(e - s for s, e, _ in l)
    print(f"== stream {q}: {len(l)} kernels, busy {busy / 1e6:.3f} ms")
    agg = {}
    for s, e, n in l:
        agg[n] = agg.get(n, 0) + (e - s)
        if not brief:
            print(f"  {(s - lo) / 1e3:9.1f} +{(e - s) / 1e3:7.1f}  {n[:70]}")
    for n, v in sorted(agg.items(), key=lambda kv: -kv[1])[:14]:
        print(f"    {v / 1e3:8.1f} us  {n[:70]}")
