"""GPU parity: fused front-end kernel (through the C ABI) vs the oracle's torch-CPU restatement of
STFT_Process + wrapper prep + mel + log, for the FSMN / MarbleNet / FireRed geometries."""
import numpy as np
import pytest
import torch

import vadx  # noqa: F401
from vadx import frontend, weights
from oracle import firered as ofr
from oracle import mel as omel
from oracle import stft as ostft

pytestmark = pytest.mark.gpu
# log-mel features: float32 GEMM-order differences only (same table bits, same prep bits)
FEAT_ATOL = 2e-4


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def oracle_logmel(preset, windows_i16):
    """windows_i16 [W,1,L] int16 -> [W,T,80]"""
    if preset == "fsmn":
        a = ostft.prep_fsmn(windows_i16)
        w = ostft.padded_window(400, 512, "hamming", "v1")
        c, s = ostft.dft_tables(512, w, "v1")
        fb = omel.melscale_fbanks(257, 20, 8000, 80, 16000, None, "htk").t().unsqueeze(0)
        re, im = ostft.stft(a, c, s, 160, True)
        return omel.log_mel(re, im, fb, 1e-5, "clamp").transpose(1, 2)
    if preset == "marblenet":
        a = ostft.prep_two_tap(windows_i16, 1.0 / 32768.0)
        w = ostft.padded_window(400, 512, "hann_sym", "v2")
        c, s = ostft.dft_tables(512, w, "v2")
        fb = omel.melscale_fbanks(257, 0, 8000, 80, 16000, "slaney", "slaney").t().unsqueeze(0)
        re, im = ostft.stft(a, c, s, 160, True)
        return omel.log_mel(re, im, fb, 1e-7, "add").transpose(1, 2)
    return ofr.log_mel(ofr.Frontend(), windows_i16).transpose(1, 2)


@pytest.mark.parametrize("preset,L,W,stride,B", [
    ("fsmn", 16000, 3, 11040, 3),
    ("fsmn", 4000, 1, 4000, 2),
    ("marblenet", 89431, 1, 89431, 2),
    ("marblenet", 16000, 2, 16000, 1),
    ("firered", 16000, 2, 16000, 3),
    ("firered", 2560, 1, 2560, 2),
])
@pytest.mark.parametrize("fold", [True, False, 4, 5])
def test_logmel_matches_oracle(preset, L, W, stride, B, fold):
    n = (W - 1) * stride + L
    clips = weights.burst_clips(B, n, seed=L + W + B)
    clips[0, : min(n, 3000)] = 0                         # exact digital silence -> exercises the log floor
    fe = frontend.Frontend(preset, L, fold=fold)
    assert (fe.fold != 0) == bool(fold) and (fold not in (4, 5) or fe.fold == fold)      # 4 / 5 = dense product on bf16 x 3 / fp16 x 2 split operands
    out = fe.logmel(clips, windows_per_clip=W, win_stride=stride).cpu().numpy()
    wins = np.stack([clips[b, w * stride:w * stride + L] for b in range(B) for w in range(W)])
    ref = oracle_logmel(preset, T(wins).unsqueeze(1)).numpy()
    assert out.shape == ref.shape == (B * W, fe.frames, 80)
    err = np.abs(out - ref)
    assert np.isfinite(out).all()
    # relative agreement of the mel energies themselves (before log) where they are above the floor
    big = ref > np.log(1e-3)
    if fold == 5:
        # fp16 x 2 operands: same mean error as the other products (1.1e-6 against 1.0e-6), but a band more than 70 dB below its frame's
        # strongest band -- where every product keeps only a few digits -- may land past 2e-4 (two of 72 128 values at 2.2e-4 / 2.7e-4 in the
        # fsmn case, tests/probes/fe_kind5.py -> profiles/r05_frontend_kind5_error.txt; kind 4's worst there is 1.8e-4): the 2e-4 bound holds
        # within 16 log units of the frame's peak band, 5e-4 above the floor elsewhere
        near = big & (ref > ref.max(axis=-1, keepdims=True) - 16.0)
        assert err[near].max() < FEAT_ATOL, err[near].max()
        assert err[big].max() < 5e-4, err[big].max()
        assert err[big].mean() < 2e-6, err[big].mean()
    else:
        assert err[big].max() < FEAT_ATOL, err[big].max()
    assert err.max() < 5e-3, err.max()                   # near-floor values: log amplifies round-off


def test_frontend_rejects_bad_geometry():
    fe = frontend.Frontend("fsmn", 16000)
    with pytest.raises(ValueError):
        fe.logmel(np.zeros((1, 1000), np.int16))         # window runs past the clip
    with pytest.raises(ValueError):
        fe.logmel(torch.zeros((1, 16000), dtype=torch.float32))


def test_logmel_large_batch_invariance():
    """FSMN geometry at BASELINE config-3 window count: identical windows give identical features
    wherever they sit in the batch."""
    fe = frontend.Frontend("fsmn", 16000)
    base = weights.burst_clips(4, 14 * 11040 + 16000, seed=8)
    big = torch.from_numpy(base).cuda().repeat(64, 1)   # 256 clips x 15 windows
    out = fe.logmel(big, windows_per_clip=15, win_stride=11040).view(64, 4, 15, 101, 80)
    assert torch.equal(out[0], out[63]) and torch.equal(out[0], out[31])
    assert bool(torch.isfinite(out).all())


# ------------------------------------------------------------------ in-graph linear resample (IN_SAMPLE_RATE != 16000)
@pytest.mark.parametrize("preset,rate,L,W", [
    ("firered", 8000, 8000, 2), ("firered", 48000, 48000, 1), ("firered", 44100, 44100, 2), ("firered", 22050, 12345, 1),
    ("marblenet", 8000, 8000, 2), ("marblenet", 48000, 48000, 1), ("marblenet", 32000, 20001, 1), ("marblenet", 11025, 30000, 2),
])
@pytest.mark.parametrize("fold", [True, False])
def test_resampled_logmel_matches_oracle(preset, rate, L, W, fold):
    """(both DFT products: the folded one is what sessions run, the dense one is the fallback)
    prep 6 (interpolate, then pre-emphasis: input rate above 16 kHz) and prep 7 (pre-emphasis, then interpolate: below) of
    the fused kernel vs torch's own F.interpolate in the oracle (Export_NVIDIA_MarbleNet_VAD.py:237-254,
    FireRedVAD/Export_FireRedVAD.py:431-449)."""
    from oracle import marblenet as omb
    B = 2
    clips = weights.burst_clips(B, W * L, seed=rate + L)
    fe = frontend.Frontend(preset, L, in_sample_rate=rate, fold=fold)
    assert (fe.fold != 0) == fold
    assert fe.cfg.prep == (6 if rate > 16000 else 7) and fe.window_len == int(np.floor(L * (1.0 / (rate / 16000.0))))
    out = fe.logmel(clips, windows_per_clip=W).cpu().numpy()
    wins = T(np.stack([clips[b, w * L:(w + 1) * L] for b in range(B) for w in range(W)])).unsqueeze(1)
    ref = (ofr.log_mel(ofr.Frontend(), wins, rate) if preset == "firered" else omb.log_mel(omb.Frontend(), wins, rate)).transpose(1, 2).numpy()
    assert out.shape == ref.shape == (B * W, fe.frames, 80)
    assert_features_close(out, ref)


def assert_features_close(out, ref):
    """Interpolated audio is band-limited: the upper mel bins of an up-sampled window hold 1e-6 of the frame's energy and
    their float32 DFT sums cancel to that level in BOTH implementations, so the log-domain bound applies to bins within
    e^-7 (1e-3) of their frame's strongest bin; every bin is bounded in the linear domain relative to that peak."""
    err = np.abs(out - ref)
    peak = ref.max(axis=-1, keepdims=True)
    big = ref > peak - 7.0
    assert err[big].max() < FEAT_ATOL, err[big].max()
    lin = np.abs(np.exp(out - peak) - np.exp(ref - peak))
    assert lin.max() < 2e-5, lin.max()


@pytest.mark.parametrize("rate", [8000, 48000, 32000])
def test_resampled_marblenet_frontend_matches_reference_fixture(golden, rate):
    """... and vs the reference wrapper's own features (tests/golden/resample.npz: NVIDIA_VAD_Optimized front half)."""
    g = golden("resample")
    a = g[f"marble_{rate}_audio"]
    fe = frontend.Frontend("marblenet", a.shape[-1], in_sample_rate=rate)
    out = fe.logmel(a.reshape(1, -1)).cpu().numpy()[0]
    want = g[f"marble_{rate}_logmel"][0].T
    assert out.shape == want.shape
    assert_features_close(out[None], want[None])


@pytest.mark.parametrize("rate", [8000, 48000, 44100, 22050])
def test_resampled_firered_session_matches_reference_fixture(golden, rate):
    """FireRedVAD_ONNX(in_sample_rate) end to end: frame scores within 1e-4 of the reference wrapper's."""
    from vadx import firered
    g = golden("resample")
    cfg = dict(zip(("idim", "R", "M", "H", "P", "N1", "S1", "N2", "S2", "odim"), (int(v) for v in g["firered_cfg"])))
    a = g[f"firered_{rate}_audio"]
    sess = firered.FireRedSession(weights.firered_synthetic(7, cfg), a.shape[-1], in_sample_rate=rate)
    probs = sess.run(None, {"audio": a})[0]
    assert probs.shape == g[f"firered_{rate}_probs"].shape
    np.testing.assert_allclose(probs, g[f"firered_{rate}_probs"], rtol=0, atol=1e-4)


def exact_logmel(preset, windows_i16):
    """The oracle's chain with the float32 prep and the float32 TABLE BITS of the reference, everything after them in double.
    Returns (log-mel [W,T,80], amplitude scale [W,T,80] = sqrt(strongest bin power of the frame x largest weight of the mel row):
    what the round-off of the frame's largest spectral line amounts to in each mel band)."""
    if preset == "fsmn":
        a = ostft.prep_fsmn(windows_i16).double()
        w = ostft.padded_window(400, 512, "hamming", "v1")
        c, s = ostft.dft_tables(512, w, "v1")
        fb = omel.melscale_fbanks(257, 20, 8000, 80, 16000, None, "htk").t().unsqueeze(0).double()
        re, im = ostft.stft(a, c.double(), s.double(), 160, True)
        scale = ((re * re + im * im).amax(dim=1).unsqueeze(-1) * fb[0].amax(dim=1)).sqrt()
        return omel.log_mel(re, im, fb, 1e-5, "clamp").transpose(1, 2), scale
    a = ostft.prep_two_tap(windows_i16, 1.0 / 32768.0).double()
    w = ostft.padded_window(400, 512, "hann_sym", "v2")
    c, s = ostft.dft_tables(512, w, "v2")
    fb = omel.melscale_fbanks(257, 0, 8000, 80, 16000, "slaney", "slaney").t().unsqueeze(0).double()
    re, im = ostft.stft(a, c.double(), s.double(), 160, True)
    scale = ((re * re + im * im).amax(dim=1).unsqueeze(-1) * fb[0].amax(dim=1)).sqrt()
    return omel.log_mel(re, im, fb, 1e-7, "add").transpose(1, 2), scale


@pytest.mark.parametrize("preset,L,kind", [("fsmn", 16000, 3), ("fsmn", 16000, 2), ("marblenet", 40000, 1), ("marblenet", 16000, 1),
                                           ("firered", 16000, 1), ("fsmn", 5280, 3), ("fsmn", 5280, 2), ("firered", 2560, 1), ("fsmn", 800, 3),
                                           ("fsmn", 16000, 4), ("marblenet", 40000, 4), ("marblenet", 16000, 4), ("firered", 16000, 4),
                                           ("fsmn", 5280, 4), ("firered", 2560, 4), ("fsmn", 800, 4),
                                           ("fsmn", 16000, 5), ("marblenet", 40000, 5), ("marblenet", 16000, 5), ("firered", 16000, 5),
                                           ("fsmn", 5280, 5), ("firered", 2560, 5), ("fsmn", 800, 5)])
def test_folded_dft_is_the_dense_product(preset, L, kind):
    """Table-level proof of the folded DFT product (mirror-paired taps about the window centre + f16 residual, csrc/frontend.hip
    "Folded DFT"): the same clips through the dense f32 product and the folded one, both against the double-precision evaluation
    of the SAME float32 table.  In amplitude (sqrt of the mel energy) relative to the frame's strongest spectral line -- the scale
    float32 round-off of a length-400 product lives on -- the folded product is within 3e-7 of the exact one and of the dense one,
    and its mean log-mel error equals the dense product's (both are float32 accumulation orders of one sum).  On the log scale the
    rare worst case sits on bands 60+ dB below the frame's peak, where either order keeps only a few digits.
    kind 3 (opt-in time x frequency fold, FSMN only) is held to its own, documented bounds: a weak bin inherits round-off relative to
    its STRONG mirror bin, so the bound relative to the frame's strongest line is twice the dense product's, bands within 26 dB of
    the frame's peak are as exact as the dense product's, and the mean log-mel error stays within 2.5 x.
    kinds 4 / 5 (dense product on bf16 x 3 exactly split operands, csrc/split3.h / on fp16 x 2 split operands, csrc/split2.h: the default)
    are held to the bounds of kinds 1 / 2: no larger error than 1.25 x the dense f32-MFMA product's against the double evaluation of the
    same table."""
    B = 6
    clips = weights.burst_clips(B, L, seed=L + kind)
    clips[0, : min(L, 3000)] = 0
    clips[1] = (clips[1].astype(np.int32) // 64).astype(np.int16)            # a quiet clip: a few LSBs
    rng = np.random.default_rng(L)
    clips[2] = rng.integers(-32768, 32767, size=L, dtype=np.int16)            # full-scale white noise
    clips[3] = (32000 * np.sin(2 * np.pi * 1000.37 / 16000 * np.arange(L))).astype(np.int16)      # one loud tone: 60 dB of leakage range
    fd = frontend.Frontend(preset, L, fold=False)
    ff = frontend.Frontend(preset, L, fold=kind)
    assert fd.fold == 0 and ff.fold == kind and frontend.Frontend(preset, L, fold=True).fold == (2 if preset == "fsmn" else 1)
    d = fd.logmel(clips).cpu().numpy().astype(np.float64)
    f = ff.logmel(clips).cpu().numpy().astype(np.float64)
    assert d.shape == f.shape and np.isfinite(f).all()
    if preset != "firered":
        ex, scale = (t.numpy() for t in exact_logmel(preset, T(clips).unsqueeze(1)))
        scale = np.maximum(scale, 1e-30)
    else:
        ex = d

    def amp(z):
        return np.exp(0.5 * z)
    floor = np.log({"fsmn": 1e-5, "marblenet": 1e-7, "firered": 1e-7}[preset]) + 2.0
    big = (ex > floor) & (ex > ex.max(axis=-1, keepdims=True) - 18.0)      # log scale: bands within 78 dB of the frame's strongest one
    assert np.abs(f - d)[big].max() < (5e-3 if kind == 3 else 1e-3), np.abs(f - d)[big].max()
    if preset != "firered":
        # 3e-7 of the frame's strongest line, plus the float32 resolution of the stored log-mel itself (values up to 30: 2 ulp = 4e-6
        # on the log scale = 2e-6 relative on the amplitude)
        tol = 3e-7 * scale + 2e-6 * amp(ex)
        tolf = (6e-7 if kind == 3 else 3e-7) * scale + 2e-6 * amp(ex)
        assert (np.abs(amp(f) - amp(d)) <= tolf).all(), (np.abs(amp(f) - amp(d)) / tolf).max()
        assert (np.abs(amp(f) - amp(ex)) <= tolf).all(), (np.abs(amp(f) - amp(ex)) / tolf).max()
        assert (np.abs(amp(d) - amp(ex)) <= tol).all()
        ed, ef = np.abs(d - ex)[big], np.abs(f - ex)[big]
        assert ef.mean() <= max((2.5 if kind == 3 else 1.25) * ed.mean(), 1e-7), (ef.mean(), ed.mean())
        if kind == 3:
            near = big & (ex > ex.max(axis=-1, keepdims=True) - 6.0)
            assert np.abs(f - ex)[near].max() < 2e-5, np.abs(f - ex)[near].max()
        print("fold", preset, L, "worst err / tol: fold %.2f dense %.2f; log err mean fold %.2e dense %.2e, max fold %.2e dense %.2e" % (
            (np.abs(amp(f) - amp(ex)) / tol).max(), (np.abs(amp(d) - amp(ex)) / tol).max(), ef.mean(), ed.mean(), ef.max(), ed.max()))


def test_split_frontend_is_the_default_and_can_be_turned_off(monkeypatch):
    """Default = kind 5 (dense product on fp16 x 2 split operands) where the geometry has it; VADX_FRONTEND_FOLD selects the others."""
    monkeypatch.delenv("VADX_FRONTEND_FOLD", raising=False)
    assert frontend.Frontend("fsmn", 16000).fold == 5 and frontend.Frontend("marblenet", 16000).fold == 5 and frontend.Frontend("firered", 16000).fold == 5
    assert frontend.Frontend("marblenet", 48000, in_sample_rate=48000).fold == 1      # the in-graph resampling preps are not staged by kinds 4 / 5: folded f32 product
    monkeypatch.setenv("VADX_FRONTEND_FOLD", "4")           # round 4's default: the same product on bf16 x 3 split operands
    assert frontend.Frontend("fsmn", 16000).fold == 4 and frontend.Frontend("marblenet", 48000, in_sample_rate=48000).fold == 1
    monkeypatch.setenv("VADX_FRONTEND_FOLD", "1")           # round 3's default: the folded f32 product the table admits
    assert frontend.Frontend("fsmn", 16000).fold == 2 and frontend.Frontend("marblenet", 16000).fold == 1
    monkeypatch.setenv("VADX_FRONTEND_FOLD", "0")
    assert frontend.Frontend("fsmn", 16000).fold == 0
    monkeypatch.setenv("VADX_FRONTEND_FOLD", "3")           # opt-in kind 3: taken where the table admits it, the default kind elsewhere
    assert frontend.Frontend("fsmn", 16000).fold == 3 and frontend.Frontend("marblenet", 16000).fold == 1


@pytest.mark.parametrize("n_fft,win,hop,window,variant,center,want_default,want_f32fold", [
    (512, 320, 160, "hann_sym", "v2", True, 5, 1),          # two full passes; hop 160: split-product dense kernel (fp16 x 2) by default
    (512, 512, 128, "hann_sym", "v2", True, 1, 1),          # four passes, window = n_fft; hop 128: folded f32 product
    (256, 200, 80, "hamming", "v1", True, None, None),      # periodic window, small transform (129 bins: last-bin tile), hop 80
    (512, 400, 192, "hamming", "v1", True, None, None),     # hop 192: three passes of 192 + 16
    (400, 400, 96, "povey", "v2", False, "refused", None),  # snip-edges, 201 bins, hop 96: FIVE passes -- outside the kernels (include/vadx.h: at most four)
    (1024, 640, 160, "hann_sym", "v2", True, 0, 0),         # 513 bins / 640 taps: beyond the folded and split kernels -> dense f32 kernel
])
def test_folded_product_on_other_geometries(n_fft, win, hop, window, variant, center, want_default, want_f32fold, monkeypatch):
    """The C ABI takes any geometry with hop % 16 == 0, hop <= 320 and at most four hops per window: wherever a faster product applies
    (kind 5 split-product dense at hop 160, else the fold vadx_frontend_fold_kind admits) it must agree with the dense f32 kernel
    (same table bits), where none does the dense kernel runs, and a geometry outside the kernels is REFUSED loudly, never skipped."""
    preset = dict(n_fft=n_fft, win=win, hop=hop, window=window, variant=variant, center=center, prep=1,
                  k=(-0.97 / 32768.0, 1.0 / 32768.0), mel=("torchaudio", 0, 8000, "slaney", "slaney"), log_mode=1, log_floor=1e-7)
    L = 20000
    clips = weights.burst_clips(4, L, seed=n_fft + hop)
    clips[0, :2000] = 0
    monkeypatch.delenv("VADX_FRONTEND_FOLD", raising=False)
    if want_default == "refused":
        with pytest.raises(ValueError, match="not supported by the HIP kernel"):
            frontend.Frontend(preset, L, fold=False)
        return
    fd = frontend.Frontend(preset, L, fold=False)
    ff = frontend.Frontend(preset, L)
    d = fd.logmel(clips).cpu().numpy().astype(np.float64)
    f = ff.logmel(clips).cpu().numpy().astype(np.float64)
    assert np.isfinite(f).all() and d.shape == f.shape
    if want_default is not None:
        assert ff.fold == want_default, ff.fold
    monkeypatch.setenv("VADX_FRONTEND_FOLD", "1")           # the folded f32 product, where the table admits one
    f1 = frontend.Frontend(preset, L)
    if want_f32fold is not None:
        assert f1.fold == want_f32fold, f1.fold
    g = f1.logmel(clips).cpu().numpy().astype(np.float64)
    big1 = d > d.max(axis=-1, keepdims=True) - 18.0
    assert np.abs(g - d)[big1].max() < 1e-3 and np.abs(g - d)[big1].mean() < 5e-6
    print("geometry", n_fft, win, hop, window, "-> fold", ff.fold)       # (a pair region whose padding would leave X2 has no plan: dense kernel)
    big = d > d.max(axis=-1, keepdims=True) - 18.0
    assert np.abs(f - d)[big].max() < 1e-3, np.abs(f - d)[big].max()
    assert np.abs(f - d)[big].mean() < 5e-6, np.abs(f - d)[big].mean()
