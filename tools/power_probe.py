"""Is a kernel at the chip's power limit?  Runs the plain split GEMM (139k x 300 x 300), the W_h weight gradient and the
bond gather back to back for a few seconds each, once on random operands and once on zero-filled ones, and samples socket
power and shader clock (rocm-smi) while they run.  A kernel that is limited by issue slots or memory runs as fast on
zeros as on random data and leaves the clock up; one at the power limit speeds up on zeros (fewer toggling bits) and
shows the clock held down under random data.
    python tools/power_probe.py            (RR_LIB_PATH selects a library build)"""
import json, os, subprocess, sys, threading, time, statistics, torch
sys.path.insert(0, os.getcwd())
from reactranker_amd import functions as Fn

dev = "cuda"
H, M = 300, 138881
SECONDS = float(os.environ.get("RR_PROBE_SECONDS", "3"))


def _sysfs_sclk():
    """Current shader clock from amdgpu's sysfs table (the starred line of pp_dpm_sclk), MHz."""
    import glob
    import re
    for f in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")):
        try:
            for ln in open(f):
                if "*" in ln:
                    m = re.search(r"(\d+)\s*[Mm][Hh]z", ln)
                    if m:
                        return float(m.group(1))
        except OSError:
            pass
    return None


_SMI_KEYS_SHOWN = False


def smi():
    """(watts, sclk MHz) from one rocm-smi call (+ sysfs for the clock); None where the field is missing.  Round 3 looked
    for a key starting with 'sclk clock level' and found none on this image (sclk nan): any key that mentions sclk and a
    value in MHz counts now, and the keys seen are printed once."""
    global _SMI_KEYS_SHOWN
    import re
    w = mhz = None
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout
        d = json.loads(out)
        card = d[sorted(d)[0]]
        if not _SMI_KEYS_SHOWN:
            _SMI_KEYS_SHOWN = True
            print("rocm-smi fields:", {k: v for k, v in card.items() if "clk" in k.lower() or "ower" in k}, flush=True)
        w = next((float(v) for k, v in card.items() if "ower" in k and "(W)" in k), None)
        for k, v in card.items():
            if "sclk" in k.lower():
                m = re.search(r"(\d+(?:\.\d+)?)\s*[Mm][Hh]z", str(v))
                if m:
                    mhz = float(m.group(1))
                    break
    except Exception:                                                     # noqa: BLE001
        pass
    if mhz is None:
        mhz = _sysfs_sclk()
    return w, mhz


def run(name, fn):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    samples, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            samples.append(smi())
            time.sleep(0.2)
    th = threading.Thread(target=poll)
    th.start()
    n, t0 = 0, time.time()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < SECONDS:
        for _ in range(50):
            fn()
        n += 50
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop.set()
    th.join()
    us = e0.elapsed_time(e1) / n * 1e3
    ws = [w for w, _ in samples[2:] if w is not None]
    cs = [c for _, c in samples[2:] if c is not None]
    print(f"{name:46s} {us:8.1f} us   power {statistics.median(ws) if ws else float('nan'):7.0f} W   sclk {statistics.median(cs) if cs else float('nan'):6.0f} MHz   ({len(ws)} samples)", flush=True)
    return us


print("idle", smi(), flush=True)
torch.manual_seed(0)
W = torch.randn(H, H, device=dev) / 17
L = Fn.LinW(W, torch.zeros(H, device=dev))
wt = L.pk_t(0, H)
wz = Fn.LinW(torch.zeros(H, H, device=dev), torch.zeros(H, device=dev)).pk_t(0, H)
out = torch.empty(M, H, device=dev)
for label, x, w in (("random", torch.randn(M, H, device=dev), wt), ("zeros", torch.zeros(M, H, device=dev), wz)):
    run(f"split GEMM plain dX, {label} operands", lambda: Fn.linear(M, H, w, w_packed=True, a1=x, k1=H, out=out))
# random activations against zero weights and vice versa: which operand's toggling costs
xr = torch.randn(M, H, device=dev)
run("split GEMM plain dX, random x / zero w", lambda: Fn.linear(M, H, wz, w_packed=True, a1=xr, k1=H, out=out))
xz = torch.zeros(M, H, device=dev)
run("split GEMM plain dX, zero x / random w", lambda: Fn.linear(M, H, wt, w_packed=True, a1=xz, k1=H, out=out))
for label, mk in (("random", torch.randn), ("zeros", torch.zeros)):
    x, dz = mk(M, H, device=dev), mk(M, H, device=dev)
    dw, db = torch.zeros(H, H, device=dev), torch.zeros(H, device=dev)
    run(f"split weight gradient (K 300), {label} operands", lambda: Fn.wgrad(M, H, dz, dw, dbias=db, x1=x, k1=H))
base = torch.arange(M, device=dev)
idx = ((base // 34)[:, None] * 34 + torch.randint(0, 34, (M, 4), device=dev)).clamp(max=M - 1).to(torch.int32)
for label, mk in (("random", torch.randn), ("zeros", torch.zeros)):
    src = mk(M, H, device=dev)
    run(f"gather_sum K=4, {label} rows", lambda: Fn.gather_sum(src, idx, H))
print("idle", smi(), flush=True)
