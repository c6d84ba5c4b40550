import sys, os, time, torch, ctypes as C
from types import SimpleNamespace
sys.path.insert(0, os.getcwd())
import bench
from reactranker_amd import functions as Fn, _lib
args = SimpleNamespace(pad_width=4, foreach_adam=False)
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
R = bench.Runner("mle64", bench.PRESETS["mle64"], args, 0, 1, 0, dev, 6)
for i in range(5): R.train_step(R.pool[i % 6])
torch.cuda.synchronize()
L = _lib.lib()
acc = {}
def wrap(name):
    f = getattr(L, name)
    def g(*a):
        t0 = time.perf_counter(); r = f(*a); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0; return r
    return g
class Proxy:
    def __getattr__(self, n):
        return wrap(n) if n in ("rr_reaction_forward", "rr_reaction_backward", "rr_reaction_workspace_bytes") else getattr(L, n)
Fn.lib = lambda: Proxy()
te = torch.empty
def timed_empty(*a, **k):
    t0 = time.perf_counter(); r = te(*a, **k); acc["torch.empty"] = acc.get("torch.empty", 0.0) + time.perf_counter() - t0; return r
Fn.torch.empty = timed_empty
N = 30
t0 = time.perf_counter()
for i in range(N): R.train_step(R.pool[i % 6])
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"enqueue {1e3*(t1-t0)/N:.3f} ms/step")
for k, v in acc.items(): print(f"  {k:32s} {v / N * 1e3:7.3f} ms/step")
