"""DFSMN what-if timing (development aid): build full-library variants with -DDFSMN_EXP=mask and time one
960-window sub-batch with each.   python tools/exp_dfsmn.py build 0 1 2 4 ;  (GPU box) python tools/exp_dfsmn.py run 0 1 2 4"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "voice-activity-detection-vad-onnx_amd")
EXP = os.path.join(PKG, "_exp")
SRC = ["capi.hip", "silero.hip", "frontend.hip", "fsmn.hip", "firered.hip", "marblenet.hip", "dfsmn.hip", "ingest.hip"]

CHILD = r"""
import os, sys, torch
sys.path.insert(0, %r)
import vadx
from vadx import dfsmn, weights
de = dfsmn.DfsmnEngine(weights.dfsmn_synthetic(1234), sub_batch=960)
lb, stride = de.grid()
W = 15
n = (W - 1) * stride + de.L
near = torch.from_numpy(weights.burst_clips(16, n, seed=11)).cuda().repeat(4, 1)
far = torch.from_numpy(weights.burst_clips(16, n, seed=12)).cuda().repeat(4, 1)
de.run(near, far, W, stride); torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
for a, b in ev:
    a.record(); de.run(near, far, W, stride); b.record()
torch.cuda.synchronize()
print("EXP", os.path.basename(os.environ["VADX_LIBRARY"]), "960 windows: ms", ["%%.1f" %% a.elapsed_time(b) for a, b in ev])
"""

if __name__ == "__main__":
    ids = [int(a) for a in sys.argv[2:]]
    os.makedirs(EXP, exist_ok=True)
    for n in ids:
        lib = os.path.join(EXP, f"libvadx_d{n}.so")
        if sys.argv[1] == "build":
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                                   f"-DDFSMN_EXP={n}"] + [os.path.join(PKG, "csrc", s) for s in SRC] + ["-o", lib])
            print("built", lib)
        else:
            r = subprocess.run([sys.executable, "-c", CHILD % ROOT], env=dict(os.environ, VADX_LIBRARY=lib),
                               capture_output=True, text=True, timeout=600)
            line = [l for l in r.stdout.splitlines() if l.startswith("EXP")]
            print(line[0] if line else f"EXP {n} FAILED rc={r.returncode}\n{r.stderr[-800:]}", flush=True)
