#!/bin/bash
# Same-box A/B of the whole training step between the working tree and another git revision:
#   (here)   tools/ab_tree.sh prepare <rev>      -> exports <rev> to build/ab_<rev>/ and builds its library
#   (on GPU) tools/ab_tree.sh run <rev> [bench flags]
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
mode=$1; rev=$2; shift 2 || true
d=$root/build/ab_$rev
if [ "$mode" = prepare ]; then
  rm -rf $d && mkdir -p $d
  git -C $root archive $rev | tar -x -C $d
  make -C $d/reactranker_amd/csrc -j4 > /dev/null 2>&1
  rm -f $d/reactranker_amd/csrc/*.o
  echo "prepared $d"
else
  for rep in 1 2; do
    for t in $d $root; do
      echo -n "$(basename $t)  "
      (cd $t && python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-profile --no-fwd-only --no-epoch --no-presets --no-f32-path "$@" 2>/dev/null) | \
        python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'ms/step', d.get('step_ms', {}).get('median'))"
    done
  done
fi
