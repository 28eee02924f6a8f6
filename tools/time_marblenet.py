"""MarbleNet config 4 per-entry times with the library named by VADX_LIBRARY:  python tools/time_marblenet.py [clips]"""
import sys

import torch

sys.path.insert(0, ".")
import vadx  # noqa: F401,E402
import bench_models as bm  # noqa: E402
from vadx import marblenet, weights  # noqa: E402

clips = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda:0")
eng = marblenet.MarbleNetEngine(weights.marblenet_synthetic(1234), device=dev)
audio = bm.synth_pcm16(torch, dev, clips, 89431, seed=1404)
run = lambda: eng.run(audio)  # noqa: E731
ms = bm.device_ms(torch, run, 5)
split, calls = bm._trace(run)
out = eng.run(audio)
print(f"marblenet {clips} clips: {ms:.2f} ms; " + ", ".join(f"{k[5:]} {v:.2f}" for k, v in split.items()) + f"; sum(p) {float((out[0] if isinstance(out, tuple) else out).double().sum()):.6f}")
