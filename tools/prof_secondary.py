"""One BASELINE-size pass of a secondary workload for rocprofv3 (tools/profile_secondary.sh):
    python3 tools/prof_secondary.py <fsmn|marblenet|firered|dfsmn> [passes]
Builds the same inputs as bench_models.py (GPU-generated int16 burst clips, seeded synthetic weights), runs `passes`
(default 2) passes of the hot path and prints PASSES=<n> so the summariser can turn per-kernel sums into per-pass numbers."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vadx  # noqa: E402,F401
import bench_models as bm  # noqa: E402
from vadx import dfsmn, firered, fsmn, marblenet, weights  # noqa: E402

which = sys.argv[1]
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda", 0)
if which == "fsmn":
    eng = fsmn.FsmnEngine(weights.fsmn_synthetic(1234), device=dev)
    lb, stride = eng.grid()
    W = -(-(160000 - eng.L) // stride) + 1
    audio = bm.synth_pcm16(torch, dev, 4096, (W - 1) * stride + eng.L, seed=1303)
    fn = lambda: eng.flags(audio, W)                       # noqa: E731
elif which == "marblenet":
    eng = marblenet.MarbleNetEngine(weights.marblenet_synthetic(1234), device=dev)
    audio = bm.synth_pcm16(torch, dev, 8192, 89431, seed=1404)
    fn = lambda: eng.run(audio)                            # noqa: E731
elif which == "firered":
    eng = firered.FireRedEngine(weights.firered_synthetic(1234), device=dev)
    audio = bm.synth_pcm16(torch, dev, 2048, 160000, seed=1505)
    fn = lambda: eng.run(audio, 10)                        # noqa: E731
else:
    eng = dfsmn.DfsmnEngine(weights.dfsmn_synthetic(1234), device=dev, sub_batch=3072)
    lb, stride = eng.grid()
    W = -(-(160000 - eng.L) // stride) + 1
    n = (W - 1) * stride + eng.L
    clips = int(os.environ.get("VADX_PROF_DFSMN_PAIRS", "2048"))
    near, far = bm.synth_pcm16(torch, dev, clips, n, seed=1606), bm.synth_pcm16(torch, dev, clips, n, seed=1607)
    fn = lambda: eng.run(near, far, W, stride)             # noqa: E731
# clock ramp: the first kernel after an idle GPU runs below its sustained clock (the front-end, first launch of a pass, measured 11 - 14 %
# slower than in bench.py's warm loop).  ~100 ms of library GEMMs first -- not vadx kernels, so the summariser ignores them
_x = torch.randn(4096, 4096, device=dev)
for _ in range(60):
    _x = (_x @ _x) * 1e-4
torch.cuda.synchronize()
for _ in range(passes):
    fn()
torch.cuda.synchronize()
print(f"PASSES={passes}")
