"""FSMN cycle accounting (development aid): library built with -DFS_EXP=1 sums thread-0 clock64 deltas per section.
   python tools/exp_fsmn.py build ;  (GPU box) python tools/exp_fsmn.py run"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "voice-activity-detection-vad-onnx_amd")
LIB = os.path.join(PKG, "_exp", "libvadx_fs1.so")
SRC = ["capi.hip", "silero.hip", "frontend.hip", "fsmn.hip", "firered.hip", "marblenet.hip", "dfsmn.hip", "ingest.hip"]
NAMES = ["stage", "in_linear1", "in_linear2", "cache load (x4)", "linear (x4)", "FIR + cache store (x4)", "affine (x4)", "out1 + out2", "softmax",
         "barrier waits of wave 0 (after every section)"]

if sys.argv[1] == "build":          # fsmn.hip rebuilt with -DFS_EXP=1, linked with the product's other objects
    sys.path.insert(0, ROOT)
    import vadx  # noqa: F401
    from vadx import build as vbuild
    vbuild.build(verbose=False)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    obj = os.path.join(os.path.dirname(LIB), "fsmn_exp.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + vbuild.FLAGS + ["-DFS_EXP=1"] + sys.argv[2:] + ["-c", os.path.join(PKG, "csrc", "fsmn.hip"), "-o", obj])
    objs = [obj if s == "fsmn.hip" else os.path.join(vbuild.OBJ, s.replace(".hip", ".o")) for s in vbuild.SOURCES]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB])
    print("built", LIB)
else:
    os.environ["VADX_LIBRARY"] = LIB
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import vadx  # noqa: F401
    from vadx import _lib, fsmn, weights
    eng = fsmn.FsmnEngine(weights.fsmn_synthetic(1234))
    lb, stride = eng.grid()
    from vadx import timestamps as ts
    base = weights.burst_clips(64, 160000, seed=5)
    noise = np.random.default_rng(1).standard_normal((64, 40000))
    rows = np.stack([fsmn.pad_to_window_grid(ts.normalize_to_int16(base[b].astype(np.float32)), 16000, stride, noise[b]) for b in range(64)])
    W = (rows.shape[1] - eng.L) // stride + 1
    clips = torch.from_numpy(rows).cuda().repeat(16, 1)                                       # 1024 clips x 10 s on the window grid
    h = _lib.lib()
    h.vadx_fsmn_debug_cycles.argtypes = [C.c_void_p, C.c_int]
    eng.flags(clips, W)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 16)()
    h.vadx_fsmn_debug_cycles(buf, 1)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); eng.flags(clips, W); b.record(); torch.cuda.synchronize()
    h.vadx_fsmn_debug_cycles(buf, 0)
    tot = sum(buf[:10])
    print("1024 clips x 10 s: %.1f ms" % a.elapsed_time(b))
    for n, v in zip(NAMES, buf[:10]):
        print("%-26s %6.2f %%" % (n, 100.0 * v / tot))
