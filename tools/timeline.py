"""Occupancy timeline from a rocprofv3 --kernel-trace CSV: how much of the wall time has an MFMA kernel
(linear_* / wgrad_*) in flight, only memory-bound kernels in flight, or nothing in flight.
Usage: python tools/timeline.py <kernel_trace.csv> [skip_fraction]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ev = []
for r in rows:
    name = r.get("Kernel_Name") or r.get("Name")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    mf = ("linear_" in name) or ("wgrad_fast" in name) or ("wgrad_kernel" in name)
    ev.append((s, e, mf, name))
marks = sorted(s for s, _, _, n in ev if "listmle_fwd" in n)          # one per training step
nsteps = min(6, len(marks) - 1)
lo, hi = marks[-1 - nsteps], marks[-1]                                # the last nsteps whole steps
print(f"{nsteps} steps, {(hi - lo) / nsteps / 1e6:.3f} ms/step under the profiler")
pts = []
busy = {}
for s, e, mf, n in ev:
    if e <= lo or s >= hi: continue
    s, e = max(s, lo), min(e, hi)
    pts.append((s, 1, mf)); pts.append((e, -1, mf))
    key = n.split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")[:60]
    busy[key] = busy.get(key, 0) + (e - s)
pts.sort()
n_mf = n_other = 0
last = lo
acc = {"mfma": 0, "mem_only": 0, "idle": 0, "mfma_and_mem": 0}
for t, d, mf in pts:
    dt = t - last
    if dt > 0:
        if n_mf and n_other: acc["mfma_and_mem"] += dt
        elif n_mf: acc["mfma"] += dt
        elif n_other: acc["mem_only"] += dt
        else: acc["idle"] += dt
    last = t
    if mf: n_mf += d
    else: n_other += d
tot = sum(acc.values())
print(f"window {tot / 1e6:.2f} ms")
for k, v in acc.items():
    print(f"  {k:14s} {v / 1e6:8.3f} ms  {100 * v / tot:5.1f} %")
print("kernel time inside the window (sum of durations, overlapping kernels both count), ms/step:")
for k, v in sorted(busy.items(), key=lambda kv: -kv[1])[:24]:
    print(f"  {k:60s} {v / 1e6 / nsteps:7.3f}")
