"""Is the training step launch-bound?  Times the host-side enqueue of K steps (no sync) against the
wall time including the final sync.  Usage: python tools/cpu_bound_check.py [steps] [preset]"""
import sys, os, time, torch
from types import SimpleNamespace
sys.path.insert(0, os.getcwd())
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
preset = sys.argv[2] if len(sys.argv) > 2 else "mle64"
args = SimpleNamespace(pad_width=4, foreach_adam=False)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
R = bench.Runner(preset, bench.PRESETS[preset], args, 0, 1, 0, dev, 6)
for i in range(5):
    R.train_step(R.pool[i % 6])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    R.train_step(R.pool[i % 6])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
# the same with an EMPTY queue before every step: the host's own cost of issuing one step (the figure above also
# contains the time the host spends blocked in the runtime once it is several steps ahead of the GPU)
alone = []
for i in range(steps):
    torch.cuda.synchronize()
    a = time.perf_counter()
    R.train_step(R.pool[i % 6])
    alone.append(time.perf_counter() - a)
torch.cuda.synchronize()
alone.sort()
print(f"{preset}: host cost of issuing one step into an empty queue: median {1e3 * alone[len(alone) // 2]:.2f} ms, "
      f"min {1e3 * alone[0]:.2f} ms")
print(f"{preset}: enqueue {1e3 * (t1 - t0) / steps:.2f} ms/step, total {1e3 * (t2 - t0) / steps:.2f} ms/step, "
      f"drain after last enqueue {1e3 * (t2 - t1):.2f} ms")
