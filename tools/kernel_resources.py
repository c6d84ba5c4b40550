"""Registers / spills / scratch per kernel from a `hipcc -Rpass-analysis=kernel-resource-usage` log.
Usage: python tools/kernel_resources.py LOG [substring ...]"""
import re, subprocess, sys
rows, cur = [], None
for line in open(sys.argv[1]):
    m = re.search(r"remark: +Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark: +([A-Za-z ]+?)(?: \[bytes/\w+\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.split("\n")
for r, n in zip(rows, names):
    n = n.replace("(anonymous namespace)::", "").split("(")[0]
    if all(k in n for k in sys.argv[2:]):
        print(f"{n:62s} vgpr {r.get('VGPRs', -1):4d} agpr {r.get('AGPRs', -1):3d} spill {r.get('VGPRs Spill', -1):3d} scratch {r.get('ScratchSize', -1):4d} occ {r.get('Occupancy', -1)} lds {r.get('LDS Size', -1)}")
