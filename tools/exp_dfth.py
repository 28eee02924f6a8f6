"""dft_f wave-role what-if (development aid): library variants with -DDFT_HEAVY=mask (which two of the four waves own three row
tiles); build here, run on the GPU box: python tools/exp_dfth.py build 3 5 9 6 ; python tools/exp_dfth.py run 3 5 9 6"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "voice-activity-detection-vad-onnx_amd")
EXP = os.path.join(PKG, "_exp")
SRC = ["capi.hip", "silero.hip", "frontend.hip", "fsmn.hip", "firered.hip", "marblenet.hip", "dfsmn.hip", "ingest.hip"]

if __name__ == "__main__":
    ids = [int(a) for a in sys.argv[2:]]
    os.makedirs(EXP, exist_ok=True)
    procs = []
    for n in ids:
        lib = os.path.join(EXP, f"libvadx_h{n}.so")
        if sys.argv[1] == "build":
            procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                                           f"-DDFT_HEAVY={n}"] + [os.path.join(PKG, "csrc", s) for s in SRC] + ["-o", lib]))
        else:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from exp_dft import CHILD
            r = subprocess.run([sys.executable, "-c", CHILD % ROOT], env=dict(os.environ, VADX_LIBRARY=lib),
                               capture_output=True, text=True, timeout=600)
            print(f"HEAVY {n}:", r.stdout.strip()[-300:] or r.stderr[-600:], flush=True)
    for p in procs:
        assert p.wait() == 0
