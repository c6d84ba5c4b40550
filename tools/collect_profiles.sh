#!/bin/bash
# Collect the judged evidence on a GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh <tag>      e.g. r02
# 1. rocprofv3 --kernel-trace --stats of the bench command   -> gpurun_out/<tag>_bench_kernel_stats.csv
# 2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE)      -> gpurun_out/<tag>_traffic.json (+ raw per-kernel tables);
#    the JSON records the hash of csrc/ it was measured on - bench.py ignores it once a kernel source changes
# 3. default bench (JSON line incl. cpu_baseline, epoch_stream, presets) -> gpurun_out/<tag>_bench_default.json
# Copy what should be judged from gpurun_out/ into profiles/.
set -e
tag=${1:-r02}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
lib=$root/reactranker_amd/csrc/libreactranker_hip.so
for f in $root/reactranker_amd/csrc/*.hip $root/reactranker_amd/csrc/*.cpp $root/reactranker_amd/csrc/*.h; do
  if [ "$f" -nt "$lib" ]; then echo "[collect] $f is newer than the built library: rebuild first" >&2; exit 1; fi
done
cd /tmp && export TMPDIR=/tmp
echo "[collect] stats pass"
# a PURE plan-step profile (round-4 review item 7): the timed region only, no per-op event passes, no comparison legs - its
# percentages are a training step's
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_stats -- python3 $root/bench.py --plan-only --steps 30 --warmup 5 > $out/${tag}_stats_bench.json 2> $out/${tag}_stats.err
cp $(find $out/prof_stats -name "*kernel_stats.csv" | head -1) $out/${tag}_bench_kernel_stats.csv
cp $(find $out/prof_stats -name "*domain_stats.csv" | head -1) $out/${tag}_bench_domain_stats.csv 2>/dev/null || true
rm -rf $out/prof_stats
for c in FETCH_SIZE WRITE_SIZE; do
  echo "[collect] pmc pass $c"
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -- python3 $root/bench.py --plan-only --steps 6 --warmup 2 > /dev/null 2> $out/${tag}_pmc_$c.err
done
cd $root
python3 tools/traffic_from_pmc.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/${tag}_traffic.json
python3 tools/pmc_summary.py $out/pmc_FETCH_SIZE > $out/${tag}_pmc_fetch_size.txt 2>/dev/null || true
python3 tools/pmc_summary.py $out/pmc_WRITE_SIZE > $out/${tag}_pmc_write_size.txt 2>/dev/null || true
rm -rf $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
cp $out/${tag}_traffic.json profiles/${tag}_traffic.json          # bench.py reads roofline.traffic from profiles/
# a traffic file measured on other kernel sources must not survive as the newest one
python3 - <<PY
import glob, json, sys
sys.path.insert(0, "$root")
from bench import csrc_hash
newest = sorted(glob.glob("$root/profiles/*_traffic.json"))[-1]
if json.load(open(newest)).get("csrc_hash") != csrc_hash():
    sys.exit(f"[collect] {newest} does not match the current kernel sources")
PY
echo "[collect] default bench"
python3 bench.py > $out/${tag}_bench_default.json 2> $out/${tag}_bench_default.err
echo "[collect] done"
