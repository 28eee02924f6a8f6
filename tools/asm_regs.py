"""Register / scratch / occupancy summary of the kernels in a hipcc -S listing whose symbol contains a substring.
usage: python tools/asm_regs.py file.s <substring>"""
import re
import sys
L = open(sys.argv[1]).read().splitlines()
for i, l in enumerate(L):
    m = re.match(r"^(_Z\S*" + re.escape(sys.argv[2]) + r"\S*):", l)
    if not m:
        continue
    out = []
    for t in L[i:]:
        g = re.search(r"; (NumVgprs|NumAgprs|ScratchSize|Occupancy): (\d+)", t)
        if g:
            out.append(g.group(1) + "=" + g.group(2))
        if "Occupancy" in t:
            break
    print(m.group(1)[:90], " ".join(out))
