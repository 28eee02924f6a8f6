"""Variant libraries for what-if timing: one source rebuilt with extra -D flags, linked with the product's other objects.
   python tools/exp_lib.py <tag> <source.hip> [-DX=1 ...]   ->  voice-activity-detection-vad-onnx_amd/_exp/libvadx_<tag>.so
   (GPU box)  VADX_LIBRARY=<that file> python tools/time_model.py fsmn"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "voice-activity-detection-vad-onnx_amd")
sys.path.insert(0, ROOT)
import vadx  # noqa: F401,E402
from vadx import build as vbuild  # noqa: E402

tag, src, defs = sys.argv[1], sys.argv[2], sys.argv[3:]
vbuild.build(verbose=False)
out_dir = os.path.join(PKG, "_exp")
os.makedirs(out_dir, exist_ok=True)
obj = os.path.join(out_dir, f"{tag}_{src.replace('.hip', '.o')}")
subprocess.check_call(["/opt/rocm/bin/hipcc"] + vbuild.FLAGS + defs + ["-c", os.path.join(PKG, "csrc", src), "-o", obj])
objs = [obj if s == src else os.path.join(vbuild.OBJ, s.replace(".hip", ".o")) for s in vbuild.SOURCES]
lib = os.path.join(out_dir, f"libvadx_{tag}.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib])
print("built", lib)
