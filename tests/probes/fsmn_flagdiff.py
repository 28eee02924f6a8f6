"""FSMN silence flags of the three front-end products (dense, kind 2 = default, kind 3 = opt-in) against the CPU oracle on the first N
clips of bench config 3 (development aid; the oracle takes ~30 s per 1024 clips on 8 threads).   python tests/probes/fsmn_flagdiff.py"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import vadx
from vadx import fsmn, weights
from oracle import fsmn as ofs
from oracle import postproc as opp
T = lambda x: torch.from_numpy(np.ascontiguousarray(x))
N = 1024
import bench_models as bm
w = weights.fsmn_synthetic(1234)
eng = fsmn.FsmnEngine(w)
lb, stride = eng.grid()
rng = np.random.default_rng(7)
# noise-like clips (the bench's kind of audio): coloured noise with bursts, so that many frames sit near the decision thresholds
# the bench's own config-3 batch (first N clips): synthetic PCM already on the window grid
W = -(-(160000 - eng.L) // stride) + 1
padded = (W - 1) * stride + eng.L
clips = bm.synth_pcm16(torch, torch.device("cuda:0"), 4096, padded, seed=1303)[:N].contiguous()
rows = clips.cpu().numpy()
f2 = eng.flags(clips, W).cpu().numpy().astype(bool)
os.environ["VADX_FRONTEND_FOLD"] = "3"
eng3 = fsmn.FsmnEngine(w)
assert eng3.fe.fold == 3
f3 = eng3.flags(clips, W).cpu().numpy().astype(bool)
os.environ["VADX_FRONTEND_FOLD"] = "0"
eng0 = fsmn.FsmnEngine(w)
f0 = eng0.flags(clips, W).cpu().numpy().astype(bool)
fe = ofs.Frontend()
ow = {k: T(v) for k, v in w.items()}
t0 = time.time()
want = []
torch.set_num_threads(8)
for b in range(N):
    # the clip is already int16 on the grid: run the oracle's window loop on it as it is (no padding noise needed: length fits the grid)
    _, fl = ofs.run_clip(fe, ow, rows[b], np.zeros(1))
    want.append(np.array(fl, bool))
want = np.stack(want)
print("oracle %.0f s; flags per path %d (speech fraction %.2f)" % (time.time() - t0, want.size, 1 - want.mean()))
for name, f in (("dense", f0), ("kind 2 (default)", f2), ("kind 3 (opt-in)", f3)):
    print("%-18s differs from the oracle on %d flags; from the dense path on %d" % (name, int((f != want).sum()), int((f != f0).sum())))
