#!/usr/bin/env python3
"""One line per variant of tools/ab_env.py result files: python tools/show_ab.py FILE.json [...]"""
import json
import sys

for f in sys.argv[1:]:
    d = json.load(open(f))
    for k, v in d.items():
        print(f, k, v["ms_per_step_median"], "%+.2f %%" % v["vs_base_pct"], v["all"])
