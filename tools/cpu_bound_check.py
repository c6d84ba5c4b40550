"""Is the training step launch-bound?  Times the host-side enqueue of K steps (no sync) against the
wall time including the final sync.  Usage: python tools/cpu_bound_check.py [steps]"""
import sys, os, time, torch
sys.path.insert(0, os.getcwd())
import bench
from types import SimpleNamespace
from reactranker_amd import loss as RL
from reactranker_amd.base_model import build_model
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
args = SimpleNamespace(queries_per_step=64, cands=64, pool=6, hidden=300, depth=3, dropout=0.1, pad_width=4)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = build_model(hidden_size=300, mpnn_depth=3, mpnn_diff_depth=3, ffn_depth=3, use_bias=True, dropout=0.1, task_num=1,
                    ffn_last_layer="with_softplus", add_features_dim=1).to(dev)
model.train()
opt = torch.optim.Adam(model.parameters(), lr=1e-4)
mle = RL.MLEloss()
pool, _, _ = bench.build_pool(args, 0, dev)
def step(i):
    b = pool[i % len(pool)]
    out = model(b["r"], b["p"], gpu=0, add_features=b["add"])
    loss = mle(out, b["scope"], b["targets"], 0)
    opt.zero_grad(set_to_none=True)
    loss.sum().backward()
    opt.step()
for i in range(5): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps): step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3 * (t1 - t0) / steps:.2f} ms/step, total {1e3 * (t2 - t0) / steps:.2f} ms/step, drain after last enqueue {1e3 * (t2 - t1):.2f} ms")
