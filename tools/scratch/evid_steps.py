import sys, os, time, torch
sys.path.insert(0, os.getcwd())
import bench as B
import argparse
args = argparse.Namespace(pad_width=4, foreach_adam=False)
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
R = B.Runner("evidential600", B.PRESETS["evidential600"], args, 0, 1, 0, dev, 2)
for rep in range(2):
    ts = []
    for i in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        R.train_step(R.pool[i % 2]); torch.cuda.synchronize()
        ts.append(round((time.perf_counter() - t0) * 1e3, 1))
    print("per-step ms (sync each):", ts, "reserved GB", round(torch.cuda.memory_reserved() / 1e9, 2), "alloc retries", torch.cuda.memory_stats().get("num_alloc_retries"))
secs, per, _ = R.timed(lambda i: R.pool[i % 2], 8)
print("timed:", [round(x, 1) for x in per])
secs, per, _ = R.timed(lambda i: R.pool[i % 2], 8)
print("timed:", [round(x, 1) for x in per])
