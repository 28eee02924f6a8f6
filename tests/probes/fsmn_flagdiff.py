"""FSMN silence flags of every (front-end product, dense-layer arithmetic) pair that was ever a default against the CPU oracle on the first N
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
from vadx import frontend
def flags_of(fold, arith):
    e = fsmn.FsmnEngine(w)
    e.blobs.arithmetic = arith
    if fold is not None:
        e.fe = frontend.Frontend("fsmn", e.L, fold=fold)
    return e.flags(clips, W).cpu().numpy().astype(bool), e.fe.fold
f5, k5 = flags_of(None, "h2")            # round 5 default: front-end kind 5 + fp16 x 2 dense layers
f4, k4 = flags_of(4, "split")            # round 4 default: kind 4 + bf16 x 3
f2, k2 = flags_of(True, "f32")           # round 3 default: folded f32 front-end (kind 2) + f32 MFMAs
f3, k3 = flags_of(3, "f32")              # opt-in kind 3
f0, k0 = flags_of(False, "f32")          # dense f32 front-end + f32 MFMAs
assert (k5, k4, k2, k3, k0) == (5, 4, 2, 3, 0)
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
for name, f in (("dense f32 + f32 MFMAs", f0), ("kind 2 + f32 MFMAs (r3 default)", f2), ("kind 3 + f32 MFMAs (opt-in)", f3),
                ("kind 4 + bf16 x 3 (r4 default)", f4), ("kind 5 + fp16 x 2 (r5 default)", f5)):
    print("%-34s differs from the oracle on %d flags; from the dense f32 path on %d" % (name, int((f != want).sum()), int((f != f0).sum())))
