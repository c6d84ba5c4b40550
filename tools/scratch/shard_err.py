import sys, os, torch, numpy as np
sys.path.insert(0, os.getcwd())
from tests.test_gpu_model import FULL_STEPS, _loss_of, make_model
from reactranker_amd import featurization, synth
from oracle import ref_cpu as O
cfg, Q, Cn, kind, n_spot = FULL_STEPS["cfg5_evidential_h600_d6_64x64"]
H, d = cfg["hidden_size"], cfg["mpnn_depth"]
w = synth.seeded_weights(O.model_shapes(H, d, d, 3, cfg["task_num"], 1, True), 77)
model = make_model(cfg, w).eval()
qb = synth.make_queries(123, Q, Cn)
rb, pb = featurization.BatchMolGraph(qb.r_specs, K=4), featurization.BatchMolGraph(qb.p_specs, K=4)
scope, targets = qb.scope, torch.tensor(qb.targets)
out1 = model(rb, pb, 0, qb.add_features); l1 = _loss_of(kind, out1, scope, targets)
model.zero_grad(); l1.sum().backward()
g1 = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
acc = {k: torch.zeros_like(v, dtype=torch.float64) for k, v in g1.items()}
for nsh in (2, 4):
    acc = {k: torch.zeros_like(v, dtype=torch.float64) for k, v in g1.items()}
    hq = Q // nsh
    for lo in range(0, Q, hq):
        sl = slice(lo * Cn, (lo + hq) * Cn)
        rs, ps = featurization.BatchMolGraph(qb.r_specs[sl], K=4), featurization.BatchMolGraph(qb.p_specs[sl], K=4)
        o = model(rs, ps, 0, qb.add_features[sl]); l = _loss_of(kind, o, scope[lo:lo + hq], targets[sl])
        model.zero_grad(); l.sum().backward()
        for k, p in model.named_parameters():
            if p.grad is not None: acc[k] += p.grad.double() / nsh
    print("shards", nsh)
    for k in g1:
        s = float(g1[k].abs().max())
        e = float((acc[k] - g1[k].double()).abs().max())
        print(f"  {k:32s} max|g| {s:10.3e} err {e:10.3e} rel {e/max(s,1e-30):9.2e}  |g|2 {float(g1[k].norm()):9.3e}")
