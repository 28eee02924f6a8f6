"""Time one model's full-size secondary config with the library named by VADX_LIBRARY (what-if builds of tools/exp_lib.py) and print a
   checksum of its decisions:  python tools/time_model.py fsmn [clips]"""
import hashlib
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import vadx  # noqa: F401,E402
from vadx import _lib, weights  # noqa: E402
from vadx import timestamps as ts  # noqa: E402

model = sys.argv[1]
clips = int(sys.argv[2]) if len(sys.argv) > 2 else {"fsmn": 4096}[model]


def timed(run, reps=5):
    out = run()
    torch.cuda.synchronize()
    t = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = run()
        b.record()
        torch.cuda.synchronize()
        t.append(a.elapsed_time(b))
    return out, t


if model == "fsmn":
    from vadx import fsmn
    eng = fsmn.FsmnEngine(weights.fsmn_synthetic(1234))
    lb, stride = eng.grid()
    base = weights.burst_clips(64, 160000, seed=5)
    noise = np.random.default_rng(1).standard_normal((64, 40000))
    rows = np.stack([fsmn.pad_to_window_grid(ts.normalize_to_int16(base[b].astype(np.float32)), 16000, stride, noise[b]) for b in range(64)])
    W = (rows.shape[1] - eng.L) // stride + 1
    audio = torch.from_numpy(rows).cuda().repeat(clips // 64, 1)
    out, t = timed(lambda: eng.flags(audio, W))
    digest = hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:12]
else:
    raise SystemExit("model?")
print(f"{model} {clips} clips: " + " ".join(f"{x:.2f}" for x in t) + f" ms; decisions sha1 {digest}")
