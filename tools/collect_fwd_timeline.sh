#!/bin/bash
# rocprofv3 --kernel-trace of the validation loop (tools/eval_loop.py) -> per-stream kernel sequence of one eval step:
#   tools/collect_fwd_timeline.sh <tag>     -> gpurun_out/<tag>_fwd_timeline.txt
set -e
tag=${1:-r04}
root=$(pwd); out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/tlf -- python3 $root/tools/eval_loop.py 12 > /dev/null 2> $out/${tag}_tlf.err
cd $root
python3 tools/step_timeline.py $(find $out/tlf -name "*kernel_trace.csv" | head -1) 3 --mark ranking_metrics_kernel > $out/${tag}_fwd_timeline.txt
rm -rf $out/tlf
