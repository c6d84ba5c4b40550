#!/usr/bin/env python3
"""Per-workgroup start / end stamps of the persistent split GEMMs INSIDE a training step: how staggered do the 256 workgroups of a
launch start when the chip is shared with the weight-gradient stream, and how much earlier would the launch end if every workgroup
ended at the same time (max(end) - mean(end)) instead of carrying an equal static share from whenever it got its CU?
Needs a library built with -DRR_TRACE from a copy of csrc/ with tools/experiments/r05_trace_insitu.patch applied to linear.hip (one
trace slot per persistent launch, the k-loop count per workgroup):
  cp -r reactranker_amd/csrc build/trace_src && patch build/trace_src/linear.hip tools/experiments/r05_trace_insitu.patch
  (cd build/trace_src && make CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DRR_TRACE" && cp libreactranker_hip.so ../lib_trace.so)
Usage: RR_LIB_PATH=build/lib_trace.so python tools/trace_insitu.py [--config mle64] [--noside]
Round-5 output: profiles/r05_trace_insitu_persistent_gemm.txt."""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from reactranker_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="mle64")
    ap.add_argument("--noside", action="store_true")
    a = ap.parse_args()
    args = argparse.Namespace(pad_width=4, foreach_adam=False, torch_fused_adam=False, steps=4)
    cfg = dict(bench.PRESETS[a.config])
    torch.cuda.set_device(0)
    R = bench.Runner(a.config, cfg, args, 0, 1, 0, torch.device("cuda", 0), 6)
    if a.noside:
        from reactranker_amd import functions as Fn
        Fn.SideStream.enabled = Fn.AuxStream.enabled = False
    L = C.CDLL(_lib.LIB_PATH)
    L.rr_debug_set_trace.argtypes = [C.c_void_p]
    L.rr_debug_trace_rows.restype = C.c_longlong
    buf = torch.zeros(64 * 2048 + 32768, dtype=torch.int64, device="cuda")
    for i in range(6):
        R.train_step(R.pool[i % len(R.pool)])
    torch.cuda.synchronize()
    assert L.rr_debug_set_trace(C.c_void_p(buf.data_ptr())) == 0
    L.rr_debug_trace_reset()
    R.train_step(R.pool[0])
    torch.cuda.synchronize()
    n = L.rr_debug_trace_reset()
    t = buf.cpu().numpy()[: 64 * 2048].reshape(64, 256, 8)
    print(f"{n} persistent launches in one step ({a.config}{', one stream' if a.noside else ''}); times in us, per launch over its 256 workgroups")
    print("slot   rows    span | start spread p50  p90  max | busy p50  p90  max | end spread p50 max | max(end) - mean(end) | blocks per workgroup min mean max")
    tot_span = tot_gain = 0.0
    for s in range(min(n, 64)):
        w = t[s]
        ok = w[:, 0] > 0
        st, en = w[ok, 0] / 100.0, w[ok, 3] / 100.0
        t0 = st.min()
        busy = en - st
        span = en.max() - t0
        ss = st - t0
        es = en.max() - en
        gain = en.max() - en.mean()
        tot_span += span
        tot_gain += gain
        print(f"{s:4d} {L.rr_debug_trace_rows(s):7d} {span:7.1f} | {np.median(ss):12.1f} {np.percentile(ss, 90):5.1f} {ss.max():5.1f} |"
              f" {np.median(busy):7.1f} {np.percentile(busy, 90):5.1f} {busy.max():5.1f} | {np.median(es):9.1f} {es.max():5.1f} | {gain:8.1f} |"
              f" {int(w[ok, 7].min())} {w[ok, 7].mean():.2f} {int(w[ok, 7].max())}")
    print(f"sum of spans {tot_span:.0f} us; sum of (max end - mean end) {tot_gain:.0f} us")


if __name__ == "__main__":
    main()
