import sys, os, time, torch
from types import SimpleNamespace
sys.path.insert(0, os.getcwd())
import bench
from reactranker_amd import functions as Fn
args = SimpleNamespace(pad_width=4, foreach_adam=False)
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
R = bench.Runner("mle64", bench.PRESETS["mle64"], args, 0, 1, 0, dev, 6)
def run(n=30):
    for i in range(5): R.train_step(R.pool[i % 6])
    secs, per, _ = R.timed(lambda i: R.pool[i % 6], n)
    return secs / n * 1e3
for rep in range(2):
    Fn.StepPlan.enabled = True; Fn.AuxStream.backward = False
    a = run()
    Fn.StepPlan.enabled = False; Fn.AuxStream.backward = False
    b = run()
    Fn.StepPlan.enabled = False; Fn.AuxStream.backward = True
    c = run()
    print(f"plan {a:.3f}  per-op {b:.3f}  per-op + aux backward {c:.3f} ms/step")
