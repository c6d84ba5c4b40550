"""Train the same model on the same synthetic steps with the three-bf16-term GEMM path and with the exact-f32 MFMA path
(same initial weights, same dropout streams, Adam + NoamLR as bench.py) and print the two loss trajectories side by side.
Usage (GPU box): python tools/split_training_check.py [steps=300]"""
import sys, os, copy
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from reactranker_amd import featurization, synth, loss as RL, functions as Fn
from reactranker_amd.base_model import build_model
from reactranker_amd.train_utils import build_optimizer, build_lr_scheduler

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda", 0)
torch.manual_seed(0)
base = build_model(hidden_size=300, mpnn_depth=3, mpnn_diff_depth=3, ffn_depth=3, use_bias=True, dropout=0.1, task_num=1,
                   ffn_last_layer="with_softplus", task_type=None, add_features_dim=1).to(dev)
pool = []
for i in range(12):
    qb = synth.make_queries(1000 + i, 16, 32)
    pool.append(dict(r=featurization.BatchMolGraph(qb.r_specs, K=4), p=featurization.BatchMolGraph(qb.p_specs, K=4),
                     add=qb.add_features, scope=qb.scope, t=torch.tensor(qb.targets)))
crit = RL.MLEloss()
curves = {}
for name, split in (("split", True), ("f32", False)):
    Fn.SplitGemm.enabled = split
    model = copy.deepcopy(base).train()
    opt = build_optimizer(model, fused=True)
    sched = build_lr_scheduler(opt, warmup_epochs=2, total_epochs=25, train_data_size=100000, batch_size=16, init_lr=1e-4,
                               max_lr=1e-3, final_lr=1e-4)
    losses = []
    for s in range(steps):
        b = pool[s % len(pool)]
        model.dropout_seed = 7919 * s + 1
        out = model(b["r"], b["p"], gpu=0, add_features=b["add"])
        l = crit(out, b["scope"], b["t"], 0).sum() / len(b["scope"])
        opt.zero_grad(set_to_none=True)
        l.backward()
        opt.step()
        sched.step()
        losses.append(float(l.detach()))
    curves[name] = np.array(losses)
Fn.SplitGemm.enabled = True
a, b = curves["split"], curves["f32"]
print(f"{'step':>5s} {'loss split':>12s} {'loss f32':>12s} {'|diff|':>10s}")
for s in list(range(0, 10)) + list(range(10, steps, max(1, steps // 15))) + [steps - 1]:
    print(f"{s:5d} {a[s]:12.6f} {b[s]:12.6f} {abs(a[s] - b[s]):10.2e}")
w = min(50, steps // 3)
print(f"mean loss of the last {w} steps: split {a[-w:].mean():.6f}  f32 {b[-w:].mean():.6f}   first-step difference {abs(a[0] - b[0]):.2e}")
