"""FireRed config 5 per-entry times with the library named by VADX_LIBRARY / the arithmetic VADX_GEMM:  python tools/time_firered.py [clips]"""
import sys

import torch

sys.path.insert(0, ".")
import vadx  # noqa: F401,E402
import bench_models as bm  # noqa: E402
from vadx import firered, weights  # noqa: E402

clips = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda:0")
eng = firered.FireRedEngine(weights.firered_synthetic(1234), device=dev)
audio = bm.synth_pcm16(torch, dev, clips, 160000, seed=1505)
run = lambda: eng.run(audio, 10)  # noqa: E731
ms = bm.device_ms(torch, run, 5)
split, _ = bm._trace(run)
out = eng.run(audio, 10)
out = out[0] if isinstance(out, tuple) else out
print(f"firered {clips} clips: {ms:.2f} ms; " + ", ".join(f"{k[5:]} {v:.2f}" for k, v in split.items()) + f"; sum(p) {float(out.double().sum()):.6f}; mode {eng.blobs.mode()}")
