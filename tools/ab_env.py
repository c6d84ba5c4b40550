#!/usr/bin/env python3
"""Same-process, same-box A/B of the training step under environment knobs the library reads per call (e.g.
RR_NO_ROWDOT, RR_NO_TWO_STAGE_READOUT, RR_NO_TRAIN_PACK): one model, one step pool, the variants timed in alternation.
Usage: python tools/ab_env.py [--config mle64] [--steps 30] [--rounds 4] VAR1[,VAR2...] [VAR3 ...]
Each positional argument is one variant = the comma-separated knobs set to 1; "base" (always included) sets none."""
import argparse
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from reactranker_amd import functions as Fn  # noqa: E402
from reactranker_amd import loss as RL  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="*")
    ap.add_argument("--config", default="mle64")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--rounds", type=int, default=4)
    a = ap.parse_args()
    ap_torch = bool(os.environ.get("RR_AB_TORCH_ADAM"))
    args = argparse.Namespace(pad_width=4, foreach_adam=False, torch_fused_adam=ap_torch, steps=a.steps)
    cfg = dict(bench.PRESETS[a.config])
    torch.cuda.set_device(0)
    R = bench.Runner(a.config, cfg, args, 0, 1, 0, torch.device("cuda", 0), 6)
    variants = [("base", [])] + [(v, v.split(",")) for v in a.variants]
    knobs = sorted({k for _, ks in variants for k in ks})
    res = {name: [] for name, _ in variants}
    for rnd in range(a.rounds + 1):
        for name, ks in variants:
            for k in knobs:
                os.environ.pop(k, None)
            for k in ks:
                os.environ[k] = "1"
            # (Python-side switches: plan flags.)  Three exact bf16 terms are the default arithmetic (round 5); a variant that
            # names RR_F16X2 runs the opt-in two-f16-term form
            Fn.SplitGemm.f16 = "RR_F16X2" in ks
            RL.FusedStep.enabled = "NOFUSEDLOSS" not in ks   # loss + d loss / d score in one launch (loss.FusedStep)
            Fn.SideStream.enabled = "NOSIDE" not in ks
            Fn.AuxStream.enabled = "NOAUX" not in ks
            Fn.AuxStream.backward = "AUXBWD" in ks         # reactant-encoder backward on the aux stream (RR_PLAN_AUX_BACKWARD)
            for i in range(4):
                R.train_step(R.pool[i % len(R.pool)])
            secs, per, _ = R.timed(lambda i: R.pool[i % len(R.pool)], a.steps)
            if rnd > 0:                                   # round 0 warms allocator / clocks
                res[name].append(secs / a.steps * 1e3)
    out = {name: dict(ms_per_step_median=round(float(np.median(v)), 4), all=[round(x, 4) for x in v]) for name, v in res.items()}
    base = out["base"]["ms_per_step_median"]
    for name in out:
        out[name]["vs_base_pct"] = round(100.0 * (out[name]["ms_per_step_median"] / base - 1.0), 2)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
