"""Fused-CFB what-if timing (development aid): build full-library variants with -DCFB_EXP=mask (csrc/dfsmn_cfb.hip) and time
cfb_front / cfb_back alone on 3584 tiles.   python tools/exp_cfb.py build 0 1 2 ... ;  (GPU box) python tools/exp_cfb.py run 0 1 2 ..."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "voice-activity-detection-vad-onnx_amd")
EXP = os.path.join(PKG, "_exp")
sys.path.insert(0, ROOT)

CHILD = r"""
import os, sys, torch, ctypes as C
sys.path.insert(0, __ROOT__)
import vadx
from vadx import dfsmn, weights, _lib
net = dfsmn.Iccrn(weights.dfsmn_synthetic(1234))
chunks, frames = 512, 101
tiles = chunks * dfsmn.ft_tiles(frames)
def ms(fn, reps=3):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return min(a.elapsed_time(b) for a, b in ev)
out = {}
for name, cin in (("cfb_e2", 20), ("cfb_d2", 40)):
    x = dfsmn.FT(torch, net.device, chunks, frames, cin, 160); x.data.normal_(0.2, 0.8)
    o = dfsmn.FT(torch, net.device, chunks, frames, 20, 160)
    a, b = (x.view(), None) if cin == 20 else (x.view(0, 20), x.view(20, 20))
    sc, _ = net.cfb(name, a, b, o.view(), chunks, frames)
    s0 = net.stats(a, b, 160, tiles)
    cw, _k = net._cfb_tables(name)
    st = _lib.stream_ptr()
    L = _lib.lib()
    y1, li, hf, s1, sl = sc["y1"], sc["li"], sc["hf"], sc["stats1"], sc["stats_li"]
    part = net.new_part(tiles)
    front = lambda: _lib.check(L.vadx_dfsmn_cfb_front(C.byref(cw), C.byref(a), None if b is None else C.byref(b), s0.data_ptr(), y1.data.data_ptr(), s1.data_ptr(), li.data.data_ptr(), sl.data_ptr(), tiles, st))
    back = lambda: _lib.check(L.vadx_dfsmn_cfb_back(C.byref(cw), hf.data.data_ptr(), li.data.data_ptr(), y1.data.data_ptr(), s1.data_ptr(), C.byref(o.view()), part.data_ptr(), tiles, st))
    out[f"front{cin}"] = ms(front)
    if cin == 20: out["back"] = ms(back)
print("EXP", os.path.basename(os.environ["VADX_LIBRARY"]), f"{tiles} tiles: ms", {k: round(v, 3) for k, v in out.items()})
h = C.CDLL(os.environ["VADX_LIBRARY"])
if hasattr(h, "vadx_cfb_debug_cycles"):
    buf = (C.c_ulonglong * 32)()
    h.vadx_cfb_debug_cycles(buf, 1)
    back(); torch.cuda.synchronize()
    h.vadx_cfb_debug_cycles(buf, 1)
    names = {8: "back P: wait operands", 9: "back P: Linear + product + LDS", 10: "back P: copies + next loads", 11: "back P: barrier wait",
             12: "back D: prologue", 13: "back D: LDS reads + MFMAs", 14: "back D: barrier wait", 15: "back D: epilogue"}
    tot_p, tot_d = sum(buf[k] for k in (8, 9, 10, 11)), sum(buf[k] for k in (12, 13, 14, 15))
    for k, n in names.items():
        print("   %-36s %6.2f %% of its wave" % (n, 100.0 * buf[k] / max(1, tot_p if k < 12 else tot_d)))
    print("   cycles per tile: producer %.0f, DFT wave %.0f" % (tot_p / tiles, tot_d / tiles))
    h.vadx_cfb_debug_cycles(buf, 1)
    front(); torch.cuda.synchronize()
    h.vadx_cfb_debug_cycles(buf, 1)
    tot = sum(buf[16:21])
    if tot:
        for k, n in enumerate(("split front: convs of ten bins", "split front: barrier", "split front: DFT k-step", "split front: barrier", "split front: epilogue")):
            print("   %-36s %6.2f %% of wave 0" % (n, 100.0 * buf[16 + k] / tot))
        print("   cycles per tile: %.0f" % (tot / tiles))
"""

if __name__ == "__main__":
    from importlib import import_module
    import vadx  # noqa: F401
    build = import_module("vadx.build")
    extra = sys.argv[sys.argv.index("--") + 1:] if "--" in sys.argv else []
    ids = [int(a) for a in (sys.argv[2:sys.argv.index("--")] if "--" in sys.argv else sys.argv[2:])]
    os.makedirs(EXP, exist_ok=True)
    for n in ids:
        lib = os.path.join(EXP, f"libvadx_c{n}.so")
        if sys.argv[1] == "build":
            objs = [os.path.join(build.OBJ, s.replace(".hip", ".o")) for s in build.SOURCES if s != "dfsmn_cfb.hip"]
            obj = os.path.join(EXP, f"dfsmn_cfb_c{n}.o")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-DCFB_EXP={n}"] + extra
                                  + ["-c", os.path.join(PKG, "csrc", "dfsmn_cfb.hip"), "-o", obj])
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", obj] + objs + ["-o", lib])
            os.remove(obj)
            print("built", lib)
        else:
            r = subprocess.run([sys.executable, "-c", CHILD.replace('__ROOT__', repr(ROOT))], env=dict(os.environ, VADX_LIBRARY=lib), capture_output=True, text=True, timeout=300)
            line = [l for l in r.stdout.splitlines() if l.startswith(("EXP", "   "))]
            print("\n".join(line) if line else f"EXP {n} FAILED rc={r.returncode}\n{r.stderr[-800:]}", flush=True)
