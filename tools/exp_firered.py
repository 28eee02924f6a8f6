"""FireRed what-if timing (development aid): full-library variants with -DFR_EXP=mask, timed on config 5's net.
   python tools/exp_firered.py build 0 1 2 4 8 ;  (GPU box) python tools/exp_firered.py run 0 1 2 4 8"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "voice-activity-detection-vad-onnx_amd")
EXP = os.path.join(PKG, "_exp")
SRC = ["capi.hip", "silero.hip", "frontend.hip", "fsmn.hip", "firered.hip", "marblenet.hip", "dfsmn.hip", "ingest.hip"]

CHILD = r"""
import os, sys, torch, ctypes as C
sys.path.insert(0, %r)
import vadx
from vadx import firered, weights, _lib
eng = firered.FireRedEngine(weights.firered_synthetic(1234))
nwin = 20480
logmel = torch.randn((nwin, eng.T, 80), device="cuda") * 3 + 14
probs = torch.empty((nwin, eng.odim, eng.T), device="cuda")
def run():
    _lib.check(_lib.lib().vadx_firered_run(C.byref(eng.cfg), eng.packed.data_ptr(), logmel.data_ptr(), nwin, probs.data_ptr(), _lib.stream_ptr()))
run(); torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
for a, b in ev:
    a.record(); run(); b.record()
torch.cuda.synchronize()
print("EXP", os.path.basename(os.environ["VADX_LIBRARY"]), "firered net, 20480 windows: ms", ["%%.2f" %% a.elapsed_time(b) for a, b in ev],
      "cycles/window (fir, pointwise r>=1, total):", [round(float(v)) for v in probs[:, 0, :3].mean(dim=0)])
"""

if __name__ == "__main__":
    ids = [int(a) for a in sys.argv[2:]]
    os.makedirs(EXP, exist_ok=True)
    for n in ids:
        lib = os.path.join(EXP, f"libvadx_f{n}.so")
        if sys.argv[1] == "build":          # firered.hip rebuilt with -DFR_EXP=n, linked with the product's other objects
            sys.path.insert(0, ROOT)
            import vadx  # noqa: F401
            from vadx import build as vbuild
            vbuild.build(verbose=False)
            obj = os.path.join(EXP, f"firered_f{n}.o")
            subprocess.check_call(["/opt/rocm/bin/hipcc"] + vbuild.FLAGS + [f"-DFR_EXP={n}", "-c", os.path.join(PKG, "csrc", "firered.hip"), "-o", obj])
            objs = [obj if s == "firered.hip" else os.path.join(vbuild.OBJ, s.replace(".hip", ".o")) for s in vbuild.SOURCES]
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib])
            print("built", lib)
        else:
            r = subprocess.run([sys.executable, "-c", CHILD % ROOT], env=dict(os.environ, VADX_LIBRARY=lib),
                               capture_output=True, text=True, timeout=600)
            line = [l for l in r.stdout.splitlines() if l.startswith("EXP")]
            print(line[0] if line else f"EXP {n} FAILED rc={r.returncode}\n{r.stderr[-800:]}", flush=True)
