"""Edge-case probe (development aid): tiny / boundary inputs through every engine vs the oracle."""
import sys, os, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import vadx
from vadx import silero, firered, fsmn, marblenet, weights
from oracle import silero as osil, firered as ofr, fsmn as ofs, marblenet as omb, postproc as opp

def T(x): return torch.from_numpy(np.ascontiguousarray(x))
def run(name, fn):
    try:
        print("OK  ", name, fn())
    except Exception as e:
        print("FAIL", name, type(e).__name__, str(e)[:300])
        traceback.print_exc(limit=2)

# ---- Silero
model = silero.load_silero_vad(onnx=True, path="synthetic:1234")
ow = {k: T(v) for k, v in weights.silero_synthetic(1234).items()}
for n in (0, 1, 100, 511, 512, 513, 1024, 16000):
    def f(n=n):
        a = (weights.burst_clips(1, max(n, 1), seed=n + 1)[0][:n].astype(np.float32) / 32768.0)
        got = silero.get_speech_timestamps(T(a), model, return_seconds=False)
        want = osil.get_speech_timestamps(T(a), osil.OnnxWrapperOracle(ow), return_seconds=False)
        assert got == want, (got, want)
        return n, got
    run(f"silero n={n}", f)
# ---- FireRed
wfr = weights.firered_synthetic(1234)
efr = firered.FireRedEngine(wfr)
wt = {k: (T(v) if isinstance(v, np.ndarray) else v) for k, v in wfr.items()}
fe = ofr.Frontend()
for n in (1, 399, 400, 401, 15999, 16000, 16001, 32000):
    def f(n=n):
        clip = weights.burst_clips(1, n, seed=n)[0]
        noise = np.random.default_rng(n).standard_normal((1, 20000))
        got = efr.detect(clip[None, :], pad_noise=noise)
        want, probs, dec = ofr.run_clip(fe, wt, clip, noise[0])
        assert len(got[0]) == len(want), (got, want)
        return n, got[0][:2], want[:2]
    run(f"firered n={n}", f)
ws = weights.firered_synthetic(1234, dict(weights.FIRERED_CFG, N2=0, S2=0))
es = firered.FireRedEngine(ws, 2560)
wts = {k: (T(v) if isinstance(v, np.ndarray) else v) for k, v in ws.items()}
for n in (1, 100, 399, 400, 2560, 2561, 2959, 2960, 5000):
    def f(n=n):
        clip = weights.burst_clips(1, n, seed=n + 7)[0]
        got, tr = es.stream_detect(clip[None, :], post=(5, 0.3, 5, 8, 2000, 20), return_probs=True)
        want, op = ofr.run_clip_stream(fe, wts, clip, post=(5, 0.3, 5, 8, 2000, 20))
        assert tr[0].shape == op.shape, (tr[0].shape, op.shape)
        if len(op): np.testing.assert_allclose(tr[0], op, atol=1e-4, rtol=0)
        return n, len(op), got[0][:2], want[:2]
    run(f"firered stream n={n}", f)
# ---- FSMN
wf = weights.fsmn_synthetic(1234)
ef = fsmn.FsmnEngine(wf)
owf = {k: T(v) for k, v in wf.items()}
for n in (1, 1000, 15999, 16000, 16001, 40000):
    def f(n=n):
        clip = weights.burst_clips(1, n, seed=n + 3)[0]
        noise = np.random.default_rng(n).standard_normal((1, 40000))
        got = ef.detect(clip[None, :], pad_noise=noise)
        a = opp.normalize_to_int16(clip.astype(np.float32))
        want, _ = ofs.run_clip(ofs.Frontend(), owf, a, noise[0])
        assert got[0] == want, (got[0], want)
        return n, got[0][:2]
    run(f"fsmn n={n}", f)
# ---- MarbleNet
wm = weights.marblenet_synthetic(1234)
em = marblenet.MarbleNetEngine(wm)
for n in (1, 300, 1000, 5000, 16000, 89431):
    def f(n=n):
        clip = weights.burst_clips(1, n, seed=n + 5)[0]
        got = em.detect(clip[None, :])
        return n, got[0][:2]
    run(f"marblenet n={n}", f)
