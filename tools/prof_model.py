"""Run one secondary model a few times (for `rocprofv3 --kernel-trace [--pmc ...] -- python3 tools/prof_model.py <model>`)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vadx  # noqa: E402,F401
from vadx import dfsmn, firered, fsmn, marblenet, silero, weights  # noqa: E402
from vadx import timestamps as ts  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "firered"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
if which == "silero":         # the headline shape: 4096 x 10 s, encoder + recurrent kernels of the mode VADX_SILERO_ENCODER selects
    eng = silero.SileroEngine(weights.silero_synthetic(1234))
    big = (torch.from_numpy(weights.burst_clips(64, 160000, seed=1)).cuda().repeat(64, 1).float() * 0.000030517578).contiguous()
    probs = torch.empty((4096, 313), dtype=torch.float32, device="cuda")
    fn = lambda: (eng.encode(big), eng.recur(4096, 313, probs))      # noqa: E731
elif which == "firered":
    eng = firered.FireRedEngine(weights.firered_synthetic(1234))
    big = torch.from_numpy(weights.burst_clips(32, 160000, seed=321)).cuda().repeat(16, 1)       # 512 clips
    fn = lambda: eng.run(big, 10)      # noqa: E731
elif which == "fsmn":
    eng = fsmn.FsmnEngine(weights.fsmn_synthetic(1234))
    lb, stride = eng.grid()
    base = weights.burst_clips(32, 160000, seed=123)
    rows = np.stack([fsmn.pad_to_window_grid(ts.normalize_to_int16(base[b].astype(np.float32)), 16000, stride, np.zeros(20000)) for b in range(32)])
    W = (rows.shape[1] - 16000) // stride + 1
    big = torch.from_numpy(rows).cuda().repeat(32, 1)                                              # 1024 clips
    fn = lambda: eng.flags(big, W)     # noqa: E731
elif which == "dfsmn":
    eng = dfsmn.DfsmnEngine(weights.dfsmn_synthetic(1234), sub_batch=3072)
    lb, stride = eng.grid()
    W = -(-(160000 - eng.L) // stride) + 1
    n = (W - 1) * stride + eng.L
    near = torch.from_numpy(weights.burst_clips(32, n, seed=61)).cuda().repeat(4, 1)              # 128 clip pairs = 1920 windows
    far = torch.from_numpy(weights.burst_clips(32, n, seed=62)).cuda().repeat(4, 1)
    fn = lambda: eng.run(near, far, W, stride)      # noqa: E731
elif which == "marblenet":
    eng = marblenet.MarbleNetEngine(weights.marblenet_synthetic(1234))
    big = torch.from_numpy(weights.burst_clips(64, 89431, seed=55)).cuda().repeat(32, 1)          # 2048 clips
    fn = lambda: eng.run(big)          # noqa: E731
else:
    raise SystemExit(f"unknown model {which!r}")
for _ in range(reps):
    fn()
torch.cuda.synchronize()
print("done")
