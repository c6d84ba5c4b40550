import sys, os, time, subprocess, threading, torch
sys.path.insert(0, os.getcwd())
from reactranker_amd import functions as Fn
nB, H = 138881, 300
dev = "cuda"
msg = torch.randn(nB, H, device=dev); W = Fn.LinW(torch.randn(H, H, device=dev) / 17, None); out = torch.empty(nB, H, device=dev)
dy = torch.randn(nB, H, device=dev); dw = torch.empty(H, H, device=dev); db = torch.empty(H, device=dev)
def smi():
    try:
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True, timeout=20)
        return "\n".join(l for l in r.stdout.splitlines() if any(k in l for k in ("Power", "sclk", "mclk", "junction", "Temperature (Sensor junction)")))[:1200]
    except Exception as e:
        return f"smi failed: {e}"
print("idle:\n" + smi())
for name, fn in (("linear", lambda: Fn.linear(nB, H, W.pk(H), w_packed=True, a1=msg, k1=H, out=out)),
                 ("wgrad", lambda: Fn.wgrad(nB, H, dy, dw, dbias=db, x1=msg, k1=H))):
    stop = False
    def loop():
        while not stop:
            for _ in range(50): fn()
            torch.cuda.synchronize()
    t = threading.Thread(target=loop); t.start()
    time.sleep(3.0)
    print(f"under {name} load:\n" + smi())
    stop = True; t.join()
