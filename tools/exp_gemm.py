"""gemm_rt what-if timing on the FSMN / FireRed nets (development aid): library variants with -DVADX_GEMM_EXP=mask
(1 activations not read from LDS, 2 weight fragments from one address).  build here, run on the GPU box:
   python tools/exp_gemm.py build 0 1 2 3 ;  python tools/exp_gemm.py run 0 1 2 3"""
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
import vadx, bench_models as bm
dev = torch.device("cuda", 0)
r = bm.fsmn_c3(torch, dev, 2, 0, clips=1024) if hasattr(bm.fsmn_c3, "__call__") else None
f = bm.firered_c5(torch, dev, 2, 0, clips=512)
print("EXP", os.path.basename(os.environ["VADX_LIBRARY"]), "fsmn net %%.2f ms, front-end %%.2f ms (1024 clips)  firered net %%.2f ms, front-end %%.2f ms (512 clips)" %% (r["kernel_ms"]["vadx_fsmn_clips"], r["kernel_ms"]["vadx_frontend_logmel"], f["kernel_ms"]["vadx_firered_run"], f["kernel_ms"]["vadx_frontend_logmel"]))
"""
if __name__ == "__main__":
    ids = [int(a) for a in sys.argv[2:]]
    os.makedirs(EXP, exist_ok=True)
    procs = []
    for n in ids:
        lib = os.path.join(EXP, f"libvadx_g{n}.so")
        if sys.argv[1] == "build":
            procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                                           f"-DVADX_GEMM_EXP={n}"] + [os.path.join(PKG, "csrc", s) for s in SRC] + ["-o", lib]))
        else:
            r = subprocess.run([sys.executable, "-c", CHILD % ROOT], env=dict(os.environ, VADX_LIBRARY=lib), capture_output=True, text=True, timeout=900)
            line = [l for l in r.stdout.splitlines() if l.startswith("EXP")]
            print(line[0] if line else f"EXP {n} FAILED rc={r.returncode}\n{r.stderr[-1200:]}", flush=True)
    for p in procs:
        assert p.wait() == 0
