"""Static per-basic-block instruction mix of one kernel in a hipcc -S listing (development aid).
usage: python tools/asm_blocks.py file.s <substring of the kernel symbol> [min_instrs]"""
import re
import sys

path, key = sys.argv[1], sys.argv[2]
thresh = int(sys.argv[3]) if len(sys.argv) > 3 else 20
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if key in l and l.rstrip().endswith(key and l.split(":")[0] + ": ; @" + l.split(":")[0]) or (key in l and re.match(r"^_Z\S+:", l)))
blocks, cur = [], ["entry", {}]
tot = {}
for l in lines[start + 1:]:
    t = l.strip()
    if not t or t.startswith(";"):
        continue
    if re.match(r"^\.LBB\d+_\d+:", t):
        blocks.append(cur)
        cur = [t.split(":")[0], {}]
        continue
    op = t.split()[0]
    if op == "s_endpgm":
        break
    kind = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else
            "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
    cur[1][kind] = cur[1].get(kind, 0) + 1
    tot[kind] = tot.get(kind, 0) + 1
blocks.append(cur)
for name, c in blocks:
    if sum(c.values()) >= thresh:
        print(f"{name:12s}", " ".join(f"{k}={v}" for k, v in sorted(c.items())))
print("TOTAL", tot)
for l in lines[start:]:
    if any(k in l for k in ("NumVgprs", "NumAgprs", "ScratchSize", "Occupancy", "LDSByteSize")) and l.strip().startswith(";"):
        print(l.strip())
    if l.strip().startswith("; -- End function") or "TotalNumVgprs" in l:
        if "TotalNumVgprs" in l:
            continue
