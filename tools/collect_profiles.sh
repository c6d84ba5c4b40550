#!/bin/bash
# Collect the judged evidence on a GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh <tag>      e.g. r01
# 1. default bench (JSON line incl. cpu_baseline)            -> gpurun_out/<tag>_bench_default.json
# 2. rocprofv3 --kernel-trace --stats of the same command    -> gpurun_out/<tag>_bench_kernel_stats.csv
# 3. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE)      -> gpurun_out/<tag>_traffic.json (+ raw per-kernel tables)
# Copy what should be judged from gpurun_out/ into profiles/.
set -e
tag=${1:-r01}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
echo "[collect] stats pass"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_stats -- python3 $root/bench.py --no-cpu-baseline --no-fwd-only > $out/${tag}_stats_bench.json 2> $out/${tag}_stats.err
cp $(find $out/prof_stats -name "*kernel_stats.csv" | head -1) $out/${tag}_bench_kernel_stats.csv
cp $(find $out/prof_stats -name "*domain_stats.csv" | head -1) $out/${tag}_bench_domain_stats.csv 2>/dev/null || true
rm -rf $out/prof_stats
for c in FETCH_SIZE WRITE_SIZE; do
  echo "[collect] pmc pass $c"
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -- python3 $root/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile --no-fwd-only > /dev/null 2> $out/${tag}_pmc_$c.err
done
cd $root
python3 tools/traffic_from_pmc.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/${tag}_traffic.json
python3 tools/pmc_summary.py $out/pmc_FETCH_SIZE > $out/${tag}_pmc_fetch_size.txt 2>/dev/null || true
python3 tools/pmc_summary.py $out/pmc_WRITE_SIZE > $out/${tag}_pmc_write_size.txt 2>/dev/null || true
rm -rf $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
cp $out/${tag}_traffic.json profiles/${tag}_traffic.json          # bench.py reads roofline.traffic from profiles/
echo "[collect] default bench"
python3 bench.py > $out/${tag}_bench_default.json 2> $out/${tag}_bench_default.err
echo "[collect] done"
