"""Packed-f32 swizzle scan of the gfx950 code objects inside libvadx.so (no GPU needed).

DESIGN.md section 4e: a compiler-formed `v_pk_add_f32` whose second source was read with a CROSS swizzle (`op_sel:[0,1] op_sel_hi:[1,0]`: the low
result takes the high half of the register pair and the high result the low half) returned wrong sums in silero_encode_h2_kernel at four waves
per SIMD.  This tool lists every packed float32 VALU instruction (`v_pk_add_f32`, `v_pk_mul_f32`, `v_pk_fma_f32`) of every kernel and
classifies each source's swizzle:
    plain      op_sel 0 / op_sel_hi 1  (low result <- low half, high result <- high half): the default
    broadcast  both results read the same half
    cross      op_sel 1 / op_sel_hi 0  (the halves swapped) -- the form that failed
usage: python tools/pk_scan.py [libvadx.so] [--list]      -> per kernel: packed f32 ops, of them broadcast, cross
tests/test_cabi_cpu.py::test_no_cross_swizzled_packed_f32 runs `scan()` and fails on any cross instance in a product kernel."""
from __future__ import annotations

import os
import re
import struct
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
PK_OPS = ("v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32")


def code_objects(lib):
    """The gfx950 code objects of every clang offload bundle inside `lib` (one per translation unit)."""
    data = open(lib, "rb").read()
    out = []
    for m in re.finditer(MAGIC, data):
        base = m.start()
        (n,) = struct.unpack_from("<Q", data, base + 24)
        pos = base + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, pos)
            triple = data[pos + 24: pos + 24 + tl].decode()
            pos += 24 + tl
            if "gfx950" in triple and size:
                out.append(data[base + off: base + off + size])
    return out


def _swizzles(text, nsrc):
    """per source: 'plain' | 'broadcast' | 'cross' from the op_sel / op_sel_hi modifiers of one instruction line"""
    def bits(name, default):
        m = re.search(name + r":\[([01,]+)\]", text)
        v = [int(x) for x in m.group(1).split(",")] if m else []
        return v + [default] * (nsrc - len(v))
    lo, hi = bits("op_sel", 0), bits("op_sel_hi", 1)
    kinds = []
    for a, b in zip(lo[:nsrc], hi[:nsrc]):
        kinds.append("plain" if (a, b) == (0, 1) else "cross" if (a, b) == (1, 0) else "broadcast")
    return kinds


def scan(lib):
    """{kernel symbol: {"pk": n, "broadcast": n, "cross": n, "cross_lines": [...]}} over all gfx950 code objects of `lib`."""
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        for k, co in enumerate(code_objects(lib)):
            path = os.path.join(tmp, f"co{k}.o")
            with open(path, "wb") as fh:
                fh.write(co)
            asm = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", path], check=True, capture_output=True, text=True).stdout
            cur = None
            for line in asm.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    cur = res.setdefault(m.group(1), {"pk": 0, "broadcast": 0, "cross": 0, "cross_lines": []})
                    continue
                t = line.strip()
                op = t.split()[0] if t else ""
                if cur is None or op not in PK_OPS:
                    continue
                # a constant source (literal / inline constant / SGPR) has no halves to swap: only VGPR pairs count
                srcs = [s.strip() for s in t.split(None, 1)[1].split(",")]
                nsrc = 3 if op == "v_pk_fma_f32" else 2
                regs = [s for s in srcs[1:] if not s.startswith(("op_sel", "neg_"))][:nsrc]
                kinds = _swizzles(t, nsrc)
                kinds = [kd if r.startswith(("v[", "a[")) else "plain" for kd, r in zip(kinds, regs)]
                cur["pk"] += 1
                if "cross" in kinds:
                    cur["cross"] += 1
                    cur["cross_lines"].append(t)
                elif "broadcast" in kinds:
                    cur["broadcast"] += 1
    return res


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = args[0] if args else os.path.join(here, "voice-activity-detection-vad-onnx_amd", "libvadx.so")
    r = scan(lib)
    tot = [0, 0, 0]
    for name, c in sorted(r.items(), key=lambda kv: -kv[1]["cross"]):
        if c["pk"]:
            print(f"{name[:110]:110s} pk_f32={c['pk']:5d} broadcast={c['broadcast']:5d} cross={c['cross']:5d}")
            if "--list" in sys.argv:
                for l in c["cross_lines"]:
                    print("      ", l)
        tot = [tot[0] + c["pk"], tot[1] + c["broadcast"], tot[2] + c["cross"]]
    print(f"TOTAL kernels={len(r)} pk_f32={tot[0]} broadcast={tot[1]} cross={tot[2]}")
