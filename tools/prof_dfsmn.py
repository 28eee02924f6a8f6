"""Kernel-level breakdown helper: run one DFSMN sub-batch (run under `rocprofv3 --kernel-trace --stats`)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vadx  # noqa: E402,F401
from vadx import dfsmn, weights  # noqa: E402

de = dfsmn.DfsmnEngine(weights.dfsmn_synthetic(1234), sub_batch=960)
lb, stride = de.grid()
W = 15
n = (W - 1) * stride + de.L
near = torch.from_numpy(weights.burst_clips(16, n, seed=11)).cuda().repeat(4, 1)
far = torch.from_numpy(weights.burst_clips(16, n, seed=12)).cuda().repeat(4, 1)
for _ in range(2):
    de.run(near, far, W, stride)
torch.cuda.synchronize()
print("done")
