#!/bin/bash
# Same-box A/B of the whole training step for several builds of the library (box-to-box variance is
# larger than most kernel-level effects): tools/ab_step.sh name1 name2 ...  (build/variants/lib_<name>.so)
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for v in "$@"; do
    echo -n "$v  "
    RR_LIB_PATH=build/variants/lib_$v.so python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-profile --no-fwd-only --no-epoch --no-presets --no-f32-path 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'ms/step', d['value'], 'q/s')"
  done
done
