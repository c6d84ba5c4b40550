"""A training step issued from C++ through the C-ABI alone (examples/cxx_host/train_step.cpp: shard file -> one upload
-> rr_reaction_forward -> rr_listmle_fwd/bwd -> rr_reaction_backward) gives the same loss and the same gradients as the
Python modules on the same weights, packed step and dropout stream - the boundary is usable without Python or torch."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest
import torch

from reactranker_amd import featurization, shards, synth
from reactranker_amd import loss as RL
from oracle import ref_cpu as O
from tests.test_gpu_model import make_model

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "cxx_host", "train_step")


def _build():
    r = subprocess.run(["make", "-C", os.path.dirname(EXE)], capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(EXE):
        pytest.skip("cannot build the C++ host example here: " + r.stderr[-300:])


@pytest.mark.parametrize("arith", ["f16x2", "bf16x3"])
@pytest.mark.parametrize("p,depth,ddepth", [(0.15, 3, 3), (0.0, 3, 3), (0.2, 1, 2)])
def test_cxx_host_step_matches_the_python_modules(tmp_path, p, depth, ddepth, arith):
    """both hosts issue the same step plans: in either GEMM arithmetic (the C++ host's default is three exact bf16 terms, plan flags 0;
    RR_CXX_PLAN_FLAGS=32 selects the opt-in two-f16-term form) the results must agree to the last bit of what the JSON carries"""
    _build()
    from reactranker_amd import functions as Fn
    old_f16 = Fn.SplitGemm.f16
    Fn.SplitGemm.f16 = arith == "f16x2"
    try:
        _cxx_case(tmp_path, p, depth, ddepth, arith)
    finally:
        Fn.SplitGemm.f16 = old_f16


def _cxx_case(tmp_path, p, depth, ddepth, arith):
    H, F = 64, 1
    cfg = dict(hidden_size=H, mpnn_depth=depth, mpnn_diff_depth=ddepth, ffn_depth=3, use_bias=True, task_num=1,
               ffn_last_layer="with_softplus", task_type=None, add_features_dim=F)
    w = synth.seeded_weights(O.model_shapes(H, depth, ddepth, 3, 1, F, True), 21)
    model = make_model(cfg, w, dropout=p)
    model = model.train() if p > 0 else model.eval()
    qb = synth.make_queries(77, 4, [6, 3, 8, 5], atoms_lo=5, atoms_hi=12)
    rb, pb = featurization.BatchMolGraph(qb.r_specs, K=4), featurization.BatchMolGraph(qb.p_specs, K=4)
    shard = str(tmp_path / "one.rrshard")
    with shards.ShardWriter(shard) as wtr:
        wtr.add_step(rb, pb, qb.scope, qb.targets, qb.add_features)
    wfile = str(tmp_path / "weights.bin")
    with open(wfile, "wb") as f:
        f.write(struct.pack("<7i", H, depth, ddepth, 3, F, 1, model.ffn.head()))
        for q in model.flat_params():
            if q is not None:
                f.write(q.detach().cpu().numpy().astype(np.float32).tobytes())
    seed = 123456789
    model.dropout_seed = seed
    model.zero_grad()
    out = model(rb, pb, gpu=0, add_features=qb.add_features)
    l = RL.MLEloss()(out, qb.scope, torch.tensor(qb.targets), 0)
    l.sum().backward()
    want_sums = [float(q.grad.double().sum()) for q in model.flat_params() if q is not None]
    env = dict(os.environ, RR_CXX_PLAN_FLAGS="32" if arith == "f16x2" else "0")       # RR_PLAN_F16X2_GEMM
    r = subprocess.run([EXE, shard, "0", wfile, repr(p), str(seed)], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stdout[-500:] + r.stderr[-1500:]
    got = json.loads(r.stdout.strip().splitlines()[-1])
    assert got["M"] == pb.n_mols
    assert got["mode"] == (1 if p == 0 else (2 if depth >= 2 else 0))      # dedup / shared prefix / plain
    lv = float(l.detach().sum())
    assert abs(got["loss"] - lv) <= 1e-6 * (1 + abs(lv))
    assert abs(got["out_sum"] - float(out.detach().double().sum())) <= 1e-9 * (1 + abs(got["out_sum"]))
    assert len(got["grad_sums"]) == len(want_sums)
    for a, b in zip(got["grad_sums"], want_sums):
        assert abs(a - b) <= 1e-7 * (1e-3 + abs(b)), (a, b)
