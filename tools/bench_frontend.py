"""Front-end timing, dense vs folded DFT product (development aid):  python tools/bench_frontend.py [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import vadx  # noqa: E402,F401
from vadx import frontend, weights  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for preset, n, B in (("marblenet", 960000, 1024), ("fsmn", 16000, 16384), ("firered", 16000, 16384), ("marblenet", 89431, 2048)):
    clips = torch.from_numpy(weights.burst_clips(64, n, seed=5)).cuda().repeat(B // 64, 1)
    line = "%-10s %5d x %7d:" % (preset, B, n)
    for fold in (False, True, 4):
        fe = frontend.Frontend(preset, n, fold=fold)
        out = fe.logmel(clips)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fe.logmel(clips, out=out)
        b.record()
        torch.cuda.synchronize()
        line += "   %s %8.3f ms" % ({False: "dense", True: "fold ", 4: "split"}[fold], a.elapsed_time(b) / iters)
        del out
    print(line, flush=True)
