"""Whole-config decision records: the decisions of the DEFAULT arithmetic (fp16 x 2 split products, front-end kind 5)
against the float32-MFMA kernel set (dense float32 front-end) on the FULL BASELINE workloads -- the bench's own synthetic batches:

  config 2 (Silero, 4096 x 10 s)     clips whose segment tables differ
  config 3 (FSMN, 4096 x 10 s)       silence flags differing, of 4 485 120
  config 4 (MarbleNet, 8192 clips)   clips whose segment tables differ
  config 5 (FireRed, 2048 x 10 s)    clips whose segment tables differ
  config 5 (DFSMN, 2048 pairs)       silence flags differing

and, for every clip with a differing decision, the excuse: the two score tracks agree within the score tolerance (1e-4) and the FIRST
differing decision sits on a threshold -- a (smoothed) score within TOL = 2e-4 of it in one of the two arithmetics.  (Later differences in a
clip may be consequences of the first: the segmenters and the FSMN noise floor carry state.)  `unexcused` must be 0.

Product-side tooling: nothing here touches oracle/.  Used by bench_models.run_all (-> the detail file) and by one -m gpu test per model.
"""
from __future__ import annotations

import numpy as np

TOL = 2e-4          # a decision may differ only if a score is this close to its threshold (tests/conftest.py: chain_or_threshold)
SCORE_ATOL = 1e-4   # north_star's score tolerance: the two arithmetics' tracks must agree within it


def _smooth(p, ws):
    """the reference post-processors' moving average (float32 cumsum, expanding at the left edge: Inference_FireRed_ONNX.py:236-254)"""
    p = np.asarray(p, np.float32)
    n = p.shape[0]
    if ws <= 1 or n == 0:
        return p
    cs = np.empty(n + 1, np.float32)
    cs[0] = 0.0
    np.cumsum(p, out=cs[1:])
    sm = np.empty(n, np.float32)
    for i in range(min(ws - 1, n)):
        sm[i] = cs[i + 1] / np.float32(i + 1)
    if n >= ws:
        sm[ws - 1:] = (cs[ws:] - cs[:n - ws + 1]) * np.float32(1.0 / ws)
    return sm


def _near(track_a, track_b, thresholds, ws=1):
    """smallest distance of a (smoothed) score of either track to any of the thresholds"""
    best = np.inf
    for tr in (track_a, track_b):
        sm = _smooth(tr, ws).astype(np.float64)
        for th in thresholds:
            best = min(best, float(np.abs(sm - th).min()))
    return best


def _record(name, total, unit, differing, clips_differing, unexcused, max_dscore, extra=None):
    r = {"config": name, "compared": total, "unit": unit, "differing": int(differing), "clips_with_a_differing_decision": int(clips_differing),
         "unexcused": int(unexcused), "max_abs_score_difference": float(max_dscore), "tolerance_to_threshold": TOL,
         "pair": "default arithmetic (fp16 x 2 split products, front-end kind 5) against the float32-MFMA kernel set (dense float32 front-end)"}
    if extra:
        r.update(extra)
    return r


def silero_c2(torch, device, clips=4096, samples=160000, audio=None):
    import bench
    from vadx import silero, weights
    eng = silero.SileroEngine(weights.silero_synthetic(1234), device=device)
    if audio is None:
        pcm = bench.synth_batch(torch, device, clips, samples, seed=1234, pcm16=True)
    else:
        pcm = audio
    prm = dict(threshold=0.5, max_speech_duration_s=20, min_speech_duration_ms=250, min_silence_duration_ms=250)
    out = {}
    for mode in ("h2", "f32"):
        eng.arithmetic = mode
        probs = eng.clips_pcm16(pcm)
        segs, counts = eng.segments(probs, samples, **prm)
        out[mode] = (probs.cpu().numpy(), segs.cpu().numpy(), counts.cpu().numpy())
    pa, sa, ca = out["h2"]
    pb, sb, cb = out["f32"]
    dmax = float(np.abs(pa - pb).max())
    diff = [b for b in range(pa.shape[0]) if ca[b] != cb[b] or not np.array_equal(sa[b, :ca[b]], sb[b, :cb[b]])]
    unexc = sum(1 for b in diff if _near(pa[b], pb[b], (0.5, 0.35)) > TOL)
    return _record("C2 Silero", pa.shape[0], "clips (segment tables)", len(diff), len(diff), unexc + (dmax > SCORE_ATOL), dmax,
                   {"range_fallbacks": eng.range_fallbacks})


def _fsmn_first_difference(eng_a, eng_b, row, W, stride, L, loop):
    """Replays one clip window by window on both engines (the reference's loop state: FIR caches + noise floor) and returns the smallest
    distance to a gate threshold among the frames of the FIRST window whose gate outputs differ (None if no window differs)."""
    import torch
    T = eng_a.T
    thr = np.float32(loop["one_minus_speech_threshold"])
    noise = [np.float32(np.float32(loop["noise_init_dB"] + loop["snr_threshold"]) * np.float32(0.1))] * 2
    snr = np.float32(loop["snr_threshold"] * 0.1)
    caches = [[torch.zeros(1, 128, 19) for _ in range(4)] for _ in range(2)]
    for k in range(W):
        a = torch.from_numpy(row[k * stride:k * stride + L].copy()).reshape(1, -1)
        res = []
        for e, eng in enumerate((eng_a, eng_b)):
            score, caches[e], noisy, psil = eng.run(a, caches[e], np.array([thr], np.float32), np.array([noise[e]], np.float32), return_psil=True)
            res.append((score.cpu().numpy()[0], float(noisy.cpu().numpy()[0]), psil.cpu().numpy()[0].astype(np.float64)))
        (sa, na, pa), (sb, nb, pb) = res
        bad = np.flatnonzero(sa != sb)
        if len(bad):
            d = min(min(abs(2.0 * pa[i] - thr), abs(2.0 * pb[i] - thr)) for i in bad)          # score = 2 p (ratio 1) against the threshold
            return float(d), float(np.abs(pa - pb).max())
        for e, nz in enumerate((na, nb)):
            if nz > 0.0:
                noise[e] = np.float32(0.5) * ((noise[e] + np.float32(nz)) + snr)
    return None


def fsmn_c3(torch, device, clips=4096, max_replays=48):
    import bench_models as bm
    from vadx import frontend, fsmn, weights
    w = weights.fsmn_synthetic(1234)
    eng = fsmn.FsmnEngine(w, device=device)
    lb, stride = eng.grid()
    W = -(-(160000 - eng.L) // stride) + 1
    padded = (W - 1) * stride + eng.L
    audio = bm.synth_pcm16(torch, device, clips, padded, seed=1303)
    eng.blobs.arithmetic = "h2"
    fa = eng.flags(audio, W).cpu().numpy()
    ref = fsmn.FsmnEngine(w, device=device)
    ref.blobs.arithmetic = "f32"
    ref.fe = frontend.Frontend("fsmn", ref.L, device=device, fold=False)          # the dense float32 DFT product
    fb = ref.flags(audio, W).cpu().numpy()
    d = fa != fb
    rows_bad = np.flatnonzero(d.any(axis=1))
    loop = dict(one_minus_speech_threshold=1.0, noise_init_dB=30.0, snr_threshold=10.0)
    # every differing clip must be examined: clips beyond the replay budget count as unexcused
    unexc, worst, dscore = max(0, len(rows_bad) - max_replays), 0.0, 0.0
    for b in rows_bad[:max_replays]:
        r = _fsmn_first_difference(eng, ref, audio[b].cpu().numpy(), W, stride, eng.L, loop)
        if r is None:            # the per-window replay agrees: the difference came from the vote's own state -- cannot be excused by a score
            unexc += 1
            continue
        dist, dp = r
        worst, dscore = max(worst, dist), max(dscore, dp)
        if dist > 2 * TOL or dp > SCORE_ATOL:      # gate value = 2 p: twice the score tolerance
            unexc += 1
    return _record("C3 FSMN", int(fa.size), "silence flags", int(d.sum()), len(rows_bad), unexc, dscore,
                   {"clips_replayed_window_by_window": int(min(len(rows_bad), max_replays)), "largest_distance_to_threshold_at_a_first_difference": worst,
                    "range_fallbacks": eng.blobs.range_fallbacks})


def _post_records(name, tracks_a, tracks_b, segs_a, segs_b, thr, ws, extra=None):
    B = tracks_a.shape[0]
    dmax = float(np.abs(tracks_a - tracks_b).max())
    diff = [b for b in range(B) if segs_a[b] != segs_b[b]]
    unexc = sum(1 for b in diff if _near(tracks_a[b], tracks_b[b], (thr,), ws) > TOL)
    return _record(name, B, "clips (segment lists)", len(diff), len(diff), unexc + (dmax > SCORE_ATOL), dmax, extra)


def marblenet_c4(torch, device, clips=8192):
    import bench_models as bm
    from vadx import frontend, marblenet, weights
    n = 89431
    audio = bm.synth_pcm16(torch, device, clips, n, seed=1404).cpu().numpy()
    w = weights.marblenet_synthetic(1234)
    eng = marblenet.MarbleNetEngine(w, device=device)
    eng.arithmetic = "h2"                                                           # fused blocks' 1x1 convs on fp16 x 2 (the default)
    got_a, tr_a, _ = eng.detect(audio, return_probs=True)
    ref = marblenet.MarbleNetEngine(w, device=device)
    ref.arithmetic = "f32"
    ref._fe[n] = frontend.Frontend("marblenet", n, device=device, fold=False)      # dense float32 front-end product
    got_b, tr_b, _ = ref.detect(audio, return_probs=True)
    return _post_records("C4 MarbleNet", tr_a.cpu().numpy(), tr_b.cpu().numpy(), got_a, got_b, 0.5, 3,
                         {"frontend_kind": int(eng.frontend(n).fold), "encoder": eng.mode(), "range_fallbacks": eng.range_fallbacks})


def firered_c5(torch, device, clips=2048):
    import bench_models as bm
    from vadx import firered, frontend, weights
    n = 160000
    audio = bm.synth_pcm16(torch, device, clips, n, seed=1505).cpu().numpy()
    w = weights.firered_synthetic(1234)
    eng = firered.FireRedEngine(w, device=device)
    eng.blobs.arithmetic = "h2"
    got_a, tr_a, _ = eng.detect(audio, return_probs=True)
    ref = firered.FireRedEngine(w, device=device)
    ref.blobs.arithmetic = "f32"
    ref.fe = frontend.Frontend("firered", ref.L, device=device, fold=False)
    got_b, tr_b, _ = ref.detect(audio, return_probs=True)
    return _post_records("C5 FireRed", tr_a.cpu().numpy(), tr_b.cpu().numpy(), got_a, got_b, 0.4, 5, {"range_fallbacks": eng.blobs.range_fallbacks})


def dfsmn_c5(torch, device, clips=2048, sub_batch=1920):
    """(these kernels have float32 MFMAs and bf16 x 3 split products: the default maps to bf16 x 3)"""
    import bench_models as bm
    from vadx import _lib, dfsmn, weights
    eng = dfsmn.DfsmnEngine(weights.dfsmn_synthetic(1234), device=device, sub_batch=sub_batch)
    lb, stride = eng.grid()
    W = -(-(160000 - eng.L) // stride) + 1
    padded = (W - 1) * stride + eng.L
    near = bm.synth_pcm16(torch, device, clips, padded, seed=1606)
    far = bm.synth_pcm16(torch, device, clips, padded, seed=1607)
    res = {}
    prev = _lib.gemm_mode()
    try:
        for mode in ("h2", "f32"):
            _lib.gemm_mode(mode)
            vad = eng.run(near, far, W, stride)
            flags = torch.empty((clips, W * (eng.T_A - lb) + lb), dtype=torch.uint8, device=device)
            _lib.check(_lib.lib().vadx_dfsmn_vote(vad.data_ptr(), clips, W, eng.T_A, lb, 0.5, 0.5, flags.data_ptr(), _lib.stream_ptr()))
            res[mode] = (vad.reshape(clips, -1).cpu().numpy(), flags.cpu().numpy())
    finally:
        _lib.gemm_mode(prev)
    (va, fa), (vb, fb) = res["h2"], res["f32"]
    d = fa != fb
    rows_bad = np.flatnonzero(d.any(axis=1))
    dmax = float(np.abs(va - vb).max())
    unexc = sum(1 for b in rows_bad if _near(va[b], vb[b], (0.5,)) > TOL)
    return _record("C5 DFSMN", int(fa.size), "silence flags", int(d.sum()), len(rows_bad), unexc + (dmax > SCORE_ATOL), dmax)


def run_all(torch, device, log=lambda m: None, small=False):
    out = {}
    for name, fn, kw in (("silero_c2", silero_c2, dict(clips=64 if small else 4096)), ("fsmn_c3", fsmn_c3, dict(clips=64 if small else 4096)),
                         ("marblenet_c4", marblenet_c4, dict(clips=64 if small else 8192)), ("firered_c5", firered_c5, dict(clips=64 if small else 2048)),
                         ("dfsmn_c5", dfsmn_c5, dict(clips=4 if small else 2048, sub_batch=60 if small else 1920))):
        try:
            out[name] = fn(torch, device, **kw)
            log(f"decision record {name}: {out[name]['differing']} of {out[name]['compared']} {out[name]['unit']} differ, {out[name]['unexcused']} unexcused")
        except Exception as e:       # noqa: BLE001
            out[name] = {"error": f"{type(e).__name__}: {e}"}
            log(f"decision record {name} failed: {out[name]['error']}")
        torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    import json
    import sys
    import torch
    import vadx  # noqa: F401
    small = "--small" in sys.argv
    r = run_all(torch, torch.device("cuda", 0), log=lambda m: print(m, file=sys.stderr, flush=True), small=small)
    print(json.dumps(r, indent=1))
