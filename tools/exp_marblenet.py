"""MarbleNet fused-block cycle accounting (development aid): marblenet.hip rebuilt with -DMB_EXP=1 sums thread-0 clock64 deltas per section
of jasper_block2_kernel.   python tools/exp_marblenet.py build ;  (GPU box) python tools/exp_marblenet.py run"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "voice-activity-detection-vad-onnx_amd")
LIB = os.path.join(PKG, "_exp", "libvadx_mb1.so")
NAMES = ["stage input tile", "residual 1x1", "depthwise 0", "pointwise 0 (48 columns)", "depthwise 1", "pointwise 1", "add + ReLU + store",
         "barrier waits of wave 0"]

if sys.argv[1] == "build":
    sys.path.insert(0, ROOT)
    import vadx  # noqa: F401
    from vadx import build as vbuild
    vbuild.build(verbose=False)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    obj = os.path.join(os.path.dirname(LIB), "marblenet_exp.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + vbuild.FLAGS + ["-DMB_EXP=1"] + sys.argv[2:] + ["-c", os.path.join(PKG, "csrc", "marblenet.hip"), "-o", obj])
    objs = [obj if s == "marblenet.hip" else os.path.join(vbuild.OBJ, s.replace(".hip", ".o")) for s in vbuild.SOURCES]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB])
    print("built", LIB)
else:
    os.environ["VADX_LIBRARY"] = LIB
    sys.path.insert(0, ROOT)
    import torch
    import vadx  # noqa: F401
    from vadx import _lib, marblenet, weights
    import bench_models as bm
    eng = marblenet.MarbleNetEngine(weights.marblenet_synthetic(1234))
    audio = bm.synth_pcm16(torch, torch.device("cuda:0"), 2048, 89431, seed=1404)
    h = _lib.lib()
    h.vadx_marblenet_debug_cycles.argtypes = [C.c_void_p, C.c_int]
    eng.run(audio)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 16)()
    h.vadx_marblenet_debug_cycles(buf, 1)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); eng.run(audio); b.record(); torch.cuda.synchronize()
    h.vadx_marblenet_debug_cycles(buf, 0)
    tot = sum(buf[:8])
    print("2048 clips x 89431 samples: %.2f ms per pass (front-end included); the three fused blocks:" % a.elapsed_time(b))
    for nm, v in zip(NAMES, buf[:8]):
        print("   %-28s %6.2f %%" % (nm, 100.0 * v / tot))
