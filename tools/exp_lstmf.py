"""lstm_f what-if timing (development aid): library variants built with -DDFSMN_EXP=mask (1 no output stores, 2 no input
loads, 4 no gate non-linearities), one 960-window launch of the CepsUnit LSTM (IN = 40, F = 81).
   python tools/exp_lstmf.py build 0 1 2 4 ;  (GPU box) python tools/exp_lstmf.py run 0 1 2 4"""
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
li, hf = mk(40, 81), mk(40, 81)
li.data.normal_()
sl = net.stats(li.view(), None, 81, tiles)
name = "cfb_e1"
def run():
    net.lstm_f(name + ".ceps_unit.ch_lstm_f", li.view(), net._ln(sl, name + ".ceps_unit.LN"), hf.view(), 81, tiles)
out = []
for f in (run,):
    f(); t.cuda.synchronize()
    a, b = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): f()
    b.record(); t.cuda.synchronize()
    out.append(a.elapsed_time(b) / 5)
print("EXP", os.path.basename(os.environ["VADX_LIBRARY"]), "lstm_f<40> %%.3f ms" %% tuple(out))
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
