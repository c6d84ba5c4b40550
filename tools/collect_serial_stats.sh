#!/bin/bash
# Per-kernel totals of one training step with every kernel on ONE stream (no side / aux stream: durations do not overlap),
# for the three-bf16-term and the two-f16-term GEMM paths, side by side:  tools/collect_serial_stats.sh <tag>
set -e
tag=${1:-r04}
root=$(pwd); out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for mode in 0 1; do
  export RR_F16X2=$mode
  rm -rf $out/ss$mode
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/ss$mode -- python3 $root/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-profile --no-fwd-only --no-epoch --no-presets --no-f32-path --no-side-stream --no-aux-stream > /dev/null 2> $out/${tag}_ss$mode.err
done
cd $root
python3 - $out <<'PY' > $out/${tag}_serial_stats.txt
import csv, glob, sys
out = sys.argv[1]
tot = [{}, {}]
for mode in (0, 1):
    f = glob.glob(f"{out}/ss{mode}/**/*kernel_stats.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if name.startswith(("linear_split_kernel", "wgrad_split_kernel")):
            name = name.replace(", true>", ">").replace(", false>", ">") if name.count(",") >= (5 if name[0] == "l" else 3) else name
        d = tot[mode].setdefault(name, [0, 0.0])
        d[0] += int(r["Calls"]); d[1] += float(r["TotalDurationNs"]) / 1e3
names = sorted(set(tot[0]) | set(tot[1]), key=lambda n: -max(tot[0].get(n, [0, 0])[1], tot[1].get(n, [0, 0])[1]))
steps = 16.0
print(f"{'kernel':58s} {'bf16x3 us/step':>15s} {'f16x2 us/step':>15s}   calls/step")
sa = sb = 0.0
for n in names[:45]:
    a, b = tot[0].get(n, [0, 0.0]), tot[1].get(n, [0, 0.0])
    sa += a[1]; sb += b[1]
    print(f"{n[:58]:58s} {a[1] / steps:15.1f} {b[1] / steps:15.1f}   {a[0] / steps:.1f} / {b[0] / steps:.1f}")
print(f"{'sum of the rows above':58s} {sa / steps:15.1f} {sb / steps:15.1f}")
PY
rm -rf $out/ss0 $out/ss1
