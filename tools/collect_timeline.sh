#!/bin/bash
# rocprofv3 --kernel-trace of a short bench run -> per-stream kernel sequence of one training step (tools/step_timeline.py):
#   tools/collect_timeline.sh <tag> [step]   -> gpurun_out/<tag>_step_timeline.txt (step: which optimizer step of the traced run, default 3)
set -e
tag=${1:-r03}
root=$(pwd); out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/tl -- python3 $root/bench.py --plan-only --steps 12 --warmup 4 > /dev/null 2> $out/${tag}_tl.err
cd $root
python3 tools/step_timeline.py $(find $out/tl -name "*kernel_trace.csv" | head -1) ${2:-3} > $out/${tag}_step_timeline.txt
rm -rf $out/tl
