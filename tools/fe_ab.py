"""A/B timing of the front-end kernel at the three model geometries (VADX_LIBRARY selects the build)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vadx
from vadx import frontend, weights
for preset, n, rep in (("marblenet", 89431, 128), ("fsmn", 16000, 1024), ("firered", 16000, 512)):
    fe = frontend.Frontend(preset, n)
    clips = torch.from_numpy(weights.burst_clips(64, n, seed=5)).cuda().repeat(rep // 1 if n > 20000 else rep, 1)[: (8192 if n > 20000 else 65536)]
    for _ in range(2): fe.logmel(clips)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    for a, b in ev:
        a.record(); fe.logmel(clips); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    print("FE", os.path.basename(os.environ.get("VADX_LIBRARY", "libvadx.so")), preset, clips.shape[0], "x", n, "median ms %.3f min %.3f" % (ts[2], ts[0]))
