import sys, os, time, subprocess, threading, torch
from types import SimpleNamespace
sys.path.insert(0, os.getcwd())
import bench
args = SimpleNamespace(pad_width=4, foreach_adam=False)
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
def smi(flags):
    try:
        r = subprocess.run(["rocm-smi"] + flags, capture_output=True, text=True, timeout=20)
        return [l.strip() for l in r.stdout.splitlines() if any(k in l for k in ("Power", "power", "sclk", "Max", "Cap"))]
    except Exception as e:
        return [f"smi failed: {e}"]
print("caps:", smi(["--showmaxpower"]))
R = bench.Runner("mle64", bench.PRESETS["mle64"], args, 0, 1, 0, dev, 6)
for i in range(5): R.train_step(R.pool[i % 6])
torch.cuda.synchronize()
stop = False
def loop():
    i = 0
    while not stop:
        for _ in range(20):
            R.train_step(R.pool[i % 6]); i += 1
        torch.cuda.synchronize()
t = threading.Thread(target=loop); t.start()
time.sleep(2.0)
for k in range(4):
    print("training step loop:", smi(["--showpower"]))
    time.sleep(0.7)
stop = True; t.join()
# forward-only loop
R.model.eval()
stop = False
def loop2():
    i = 0
    while not stop:
        with torch.no_grad():
            for _ in range(20):
                b = R.pool[i % 6]; R.loss(R.model(b["r"], b["p"], gpu=0, add_features=b["add"]), b); i += 1
        torch.cuda.synchronize()
t = threading.Thread(target=loop2); t.start()
time.sleep(2.0)
print("eval forward loop:", smi(["--showpower"]))
stop = True; t.join()
