"""full-size batch-invariance debug of the fp16 x 2 Silero kernels: per-stage LDS checksums of identical clip groups (H2_DUMP build)"""
import os, sys, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import numpy as np, torch
import vadx
from vadx import silero, weights, _lib
eng = silero.SileroEngine(weights.silero_synthetic(1234))
base = weights.burst_clips(32, 160000, seed=99).astype(np.float32) * np.float32(0.000030517578)
big = torch.from_numpy(base).cuda().repeat(128, 1)
h = C.CDLL(os.environ["VADX_LIBRARY"])
dump = torch.zeros((313, 128, 2, 4), dtype=torch.int32, device="cuda")       # tile id = t * 256 + group; group = rep * 2 + parity
h.vadx_silero_h2_dump.argtypes = [C.c_void_p]
assert h.vadx_silero_h2_dump(dump.data_ptr()) == 0
silero.encoder_mode("h2")
eng.encode(big); torch.cuda.synchronize()
g = eng._ws[:313 * 256 * 8192 * 4].view(torch.float32).view(313, 128, 2, 8192)
bad_gx = (g != g[:, 0:1]).any(dim=-1)
print("gx tiles differing from rep-group 0:", int(bad_gx.sum()))
for st, name in enumerate(("samples in registers", "operand planes in LDS", "e / o values in registers", "split o terms in registers")):
    b = dump[..., st] != dump[:, 0:1, :, st]
    print(f"stage {st} ({name}): tiles whose checksum differs from rep-group 0: {int(b.sum())}; of those also wrong in gx: {int((b & bad_gx).sum())}")
