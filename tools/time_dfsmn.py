"""DFSMN config 5 per-entry times (the engine's default arithmetic, or VADX_GEMM):  python tools/time_dfsmn.py [clip pairs]"""
import sys

import torch

sys.path.insert(0, ".")
import vadx  # noqa: F401,E402
import bench_models as bm  # noqa: E402
from vadx import dfsmn, weights  # noqa: E402

clips = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0")
eng = dfsmn.DfsmnEngine(weights.dfsmn_synthetic(1234), device=dev, sub_batch=3072)
lb, stride = eng.grid()
W = -(-(160000 - eng.L) // stride) + 1
padded = (W - 1) * stride + eng.L
near = bm.synth_pcm16(torch, dev, clips, padded, seed=1606)
far = bm.synth_pcm16(torch, dev, clips, padded, seed=1607)
run = lambda: eng.run(near, far, W, stride)  # noqa: E731
ms = bm.device_ms(torch, run, 2)
split, calls = bm._trace(run)
out = eng.run(near, far, W, stride)
print(f"dfsmn {clips} pairs ({clips * W} windows): {ms:.1f} ms; " + ", ".join(f"{k[11:]} {v:.1f}" for k, v in sorted(split.items(), key=lambda kv: -kv[1])[:8])
      + f"; sum {float(out.double().sum()):.6f}; fallbacks {eng.range_fallbacks}")
