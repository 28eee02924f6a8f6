"""Front-end cycle accounting (development aid): library built with -DFE_EXP=1 sums thread-0 clock64 deltas per phase.
   python tools/exp_frontend.py build <tag> [-DFE_WHATIF=n ...] ;  (GPU box) python tools/exp_frontend.py run <tag>"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "voice-activity-detection-vad-onnx_amd")
TAG = sys.argv[2] if len(sys.argv) > 2 else "fe1"
LIB = os.path.join(PKG, "_exp", "libvadx_%s.so" % TAG)
SRC = ["capi.hip", "silero.hip", "frontend.hip", "fsmn.hip", "firered.hip", "marblenet.hip", "dfsmn.hip", "ingest.hip"]
NAMES = ["staging: barrier wait", "DFT GEMM + power", "last bin + zero pad rows", "mel GEMM + log + store", "staging: own work (wave 0)"]

if sys.argv[1] == "build":          # frontend.hip rebuilt with -DFE_EXP=1 (+ extra -D flags), linked with the product's other objects
    sys.path.insert(0, ROOT)
    import vadx  # noqa: F401
    from vadx import build as vbuild
    vbuild.build(verbose=False)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    obj = os.path.join(os.path.dirname(LIB), "frontend_%s.o" % TAG)
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + vbuild.FLAGS + ["-DFE_EXP=1"] + sys.argv[3:] + ["-c", os.path.join(PKG, "csrc", "frontend.hip"), "-o", obj])
    objs = [obj if s == "frontend.hip" else os.path.join(vbuild.OBJ, s.replace(".hip", ".o")) for s in vbuild.SOURCES]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB])
    print("built", LIB)
else:
    os.environ["VADX_LIBRARY"] = LIB
    sys.path.insert(0, ROOT)
    import torch
    import vadx  # noqa: F401
    from vadx import _lib, frontend, weights
    h = _lib.lib()
    h.vadx_frontend_debug_cycles.argtypes = [C.c_void_p, C.c_int]
    for preset, n, fold in (("marblenet", 89431, 5), ("marblenet", 89431, 4), ("fsmn", 16000, 5), ("fsmn", 16000, 4),
                            ("firered", 16000, 5), ("firered", 16000, 4)):
        fe = frontend.Frontend(preset, n, fold=fold)
        clips = torch.from_numpy(weights.burst_clips(64, n, seed=5)).cuda().repeat(32 if n > 20000 else 256, 1)
        fe.logmel(clips); torch.cuda.synchronize()
        buf = (C.c_ulonglong * 8)()
        h.vadx_frontend_debug_cycles(buf, 1)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fe.logmel(clips); b.record(); torch.cuda.synchronize()
        h.vadx_frontend_debug_cycles(buf, 0)
        tot = sum(buf[:5])
        print("%s (%s): %d clips x %d samples, %.2f ms" % (preset, {False: "dense", True: "folded", 4: "bf16 x 3 split products", 5: "fp16 x 2 split products"}[fold], clips.shape[0], n, a.elapsed_time(b)))
        print("   sum of wave-0 clock deltas over workgroups / kernel time: %.1f M ticks per ms" % (tot / 1e6 / a.elapsed_time(b)))
        if buf[6]:
            print("   clock64 / wall_clock64 (100 MHz) over the workgroups: %.1f -> shader clock %.2f GHz" % (buf[5] / buf[6], buf[5] / buf[6] / 10.0))
        for nm, v in zip(NAMES, buf[:5]):
            print("   %-30s %6.2f %%" % (nm, 100.0 * v / tot))
