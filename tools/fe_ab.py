"""A/B of the front-end kernel at the three model geometries: timing, and a checksum of the output bits (VADX_LIBRARY selects the
build; `python tools/fe_ab.py build-head` links libvadx_head.so from HEAD's frontend.hip + the current other objects)."""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "voice-activity-detection-vad-onnx_amd")
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "build-head":
    import vadx  # noqa: F401
    from vadx import build as vbuild
    vbuild.build(verbose=False)
    exp = os.path.join(PKG, "_exp")
    os.makedirs(exp, exist_ok=True)
    src = os.path.join(exp, "frontend_head.hip")
    open(src, "w").write(subprocess.check_output(["git", "show", "HEAD:voice-activity-detection-vad-onnx_amd/csrc/frontend.hip"], cwd=ROOT, text=True))
    obj = os.path.join(exp, "frontend_head.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + vbuild.FLAGS + ["-I", os.path.join(PKG, "csrc"), "-c", src, "-o", obj])
    objs = [obj if s == "frontend.hip" else os.path.join(vbuild.OBJ, s.replace(".hip", ".o")) for s in vbuild.SOURCES]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", os.path.join(exp, "libvadx_head.so")])
    print("built", os.path.join(exp, "libvadx_head.so"))
    sys.exit(0)
import torch  # noqa: E402
import vadx  # noqa: E402,F401
from vadx import frontend, weights  # noqa: E402
for preset, n, rep in (("marblenet", 89431, 128), ("fsmn", 16000, 1024), ("firered", 16000, 512)):
    fe = frontend.Frontend(preset, n)
    clips = torch.from_numpy(weights.burst_clips(64, n, seed=5)).cuda().repeat(rep, 1)[: (8192 if n > 20000 else 65536)]
    for _ in range(2):
        out = fe.logmel(clips)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    for a, b in ev:
        a.record(); fe.logmel(clips, out=out); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    digest = hashlib.sha256(out[:64].cpu().numpy().tobytes()).hexdigest()[:16]
    print("FE", os.path.basename(os.environ.get("VADX_LIBRARY", "libvadx.so")), preset, "kind", fe.fold, clips.shape[0], "x", n,
          "median ms %.3f min %.3f" % (ts[2], ts[0]), "sha", digest)
