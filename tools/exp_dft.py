"""dft_f what-if timing (development aid): library variants built with -DDFSMN_EXP=mask (8 one k-step instead of 40,
16 no copy-out, 32 no input request/park, 64 no barrier), one 960-window launch of each direction.
   python tools/exp_dft.py build 0 8 16 32 ;  (GPU box) python tools/exp_dft.py run 0 8 16 32"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "voice-activity-detection-vad-onnx_amd")
EXP = os.path.join(PKG, "_exp")
SRC = ["capi.hip", "silero.hip", "frontend.hip", "fsmn.hip", "firered.hip", "marblenet.hip", "dfsmn.hip", "ingest.hip"]

CHILD = r"""
import ctypes as C, os, sys, torch
sys.path.insert(0, %r)
import vadx
from vadx import _lib, dfsmn, weights
net = dfsmn.Iccrn(weights.dfsmn_synthetic(1234)) if hasattr(dfsmn, "Iccrn") else dfsmn.DfsmnEngine(weights.dfsmn_synthetic(1234)).net
t = torch
chunks, frames = 960, 101
tiles = chunks * dfsmn.ft_tiles(frames)
mk = lambda ch, bins: dfsmn.FT(t, net.device, chunks, frames, ch, bins, zero=False)
r, li, lo, ceps = mk(20, 160), mk(40, 81), mk(40, 81), mk(20, 160)
for x in (r, li, lo):
    x.data.normal_()
s2 = net.stats(r.view(), None, 160, tiles)
name = "cfb_e1"
L = _lib.lib()
def fwd():
    _lib.check(L.vadx_dfsmn_dft_f(0, C.byref(r.view()), None, C.byref(net._ln(s2, name + ".LN2")), net.tbl_fwd.data_ptr(),
                                  C.byref(li.view()), 20, tiles, None, _lib.stream_ptr()))
def inv():
    _lib.check(L.vadx_dfsmn_dft_f(1, C.byref(li.view()), C.byref(lo.view()), None, net.tbl_inv.data_ptr(),
                                  C.byref(ceps.view()), 20, tiles, None, _lib.stream_ptr()))
out = []
for f in (fwd, inv):
    f(); t.cuda.synchronize()
    a, b = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): f()
    b.record(); t.cuda.synchronize()
    out.append(a.elapsed_time(b) / 5)
print("EXP", os.path.basename(os.environ["VADX_LIBRARY"]), "fwd %%.3f ms  inv %%.3f ms" %% tuple(out))
"""

if __name__ == "__main__":
    ids = [int(a) for a in sys.argv[2:]]
    os.makedirs(EXP, exist_ok=True)
    procs = []
    for n in ids:
        lib = os.path.join(EXP, f"libvadx_d{n}.so")
        if sys.argv[1] == "build":
            procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                                           f"-DDFSMN_EXP={n}"] + [os.path.join(PKG, "csrc", s) for s in SRC] + ["-o", lib]))
        else:
            r = subprocess.run([sys.executable, "-c", CHILD % ROOT], env=dict(os.environ, VADX_LIBRARY=lib),
                               capture_output=True, text=True, timeout=600)
            line = [l for l in r.stdout.splitlines() if l.startswith("EXP")]
            print(line[0] if line else f"EXP {n} FAILED rc={r.returncode}\n{r.stderr[-1500:]}", flush=True)
    for p in procs:
        assert p.wait() == 0
