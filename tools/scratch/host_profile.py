import sys, os, time, torch, cProfile, pstats
from types import SimpleNamespace
sys.path.insert(0, os.getcwd())
import bench
args = SimpleNamespace(pad_width=4, foreach_adam=False)
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
R = bench.Runner("mle64", bench.PRESETS["mle64"], args, 0, 1, 0, dev, 6)
for i in range(5): R.train_step(R.pool[i % 6])
torch.cuda.synchronize()
# phase timing (host side only)
import collections
T = collections.OrderedDict()
def tick(name, t0): T[name] = T.get(name, 0.0) + time.perf_counter() - t0
N = 30
for i in range(N):
    b = R.pool[i % 6]
    t0 = time.perf_counter(); out = R.model(b["r"], b["p"], gpu=0, add_features=b["add"]); tick("model.forward", t0)
    t0 = time.perf_counter(); loss = R.loss(out, b); tick("loss", t0)
    t0 = time.perf_counter(); R.opt.zero_grad(set_to_none=True); tick("zero_grad", t0)
    t0 = time.perf_counter(); loss.sum().backward(); tick("backward", t0)
    t0 = time.perf_counter(); R.bucket.allreduce(1.0); R.sched.step(); tick("bucket+sched", t0)
    t0 = time.perf_counter(); R.opt.step(); tick("opt.step", t0)
torch.cuda.synchronize()
for k, v in T.items(): print(f"{k:16s} {v / N * 1e3:7.3f} ms/step")
pr = cProfile.Profile(); pr.enable()
for i in range(10): R.train_step(R.pool[i % 6])
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
