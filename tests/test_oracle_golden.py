"""CPU: the oracle restatement vs fixtures produced by RUNNING the reference (tests/golden/make_golden.py)."""
import hashlib

import numpy as np
import pytest
import torch

import vadx  # noqa: F401
from vadx import weights
from oracle import firered as ofr
from oracle import fsmn as ofs
from oracle import mel as omel
from oracle import postproc as opp
from oracle import silero as osil
from oracle import stft as ostft


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def _sha(t):
    return hashlib.sha256(np.ascontiguousarray(t.numpy()).tobytes()).hexdigest()


# ------------------------------------------------------------------ a1-a3: tables are BIT-identical
@pytest.mark.parametrize("tag,n_fft,win,hop,wtype,variant,center", [
    ("fsmn", 512, 400, 160, "hamming", "v1", True),
    ("dfsmn_b", 319, 319, 160, "hamming", "v1b", True),
    ("dfsmn_a", 1024, 640, 320, "hamming", "v1b", True),
    ("marble", 512, 400, 160, "hann_sym", "v2", True),
    ("firered", 400, 400, 160, "povey", "v2", False),
])
def test_stft_tables_and_transform(golden, tag, n_fft, win, hop, wtype, variant, center):
    g = golden("stft")
    w = ostft.padded_window(win, n_fft, wtype, variant)
    cos_k, sin_k = ostft.dft_tables(n_fft, w, variant)
    if variant == "v2":
        assert _sha(torch.cat([cos_k, sin_k], 0)) == str(g[f"{tag}_kernel_sha256"])
    else:
        assert _sha(cos_k) == str(g[f"{tag}_cos_sha256"])
        assert _sha(sin_k) == str(g[f"{tag}_sin_sha256"])
    re, im = ostft.stft(T(g["x"]), cos_k, sin_k, hop, center_pad=center)
    # same table, same conv1d => agreement to float32 round-off of the accumulation order
    np.testing.assert_allclose(re.numpy(), g[f"{tag}_re"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(im.numpy(), g[f"{tag}_im"], rtol=0, atol=2e-5)


# ------------------------------------------------------------------ a10, a22
def test_host_helpers(golden):
    g = golden("host")
    for i in range(int(g["n_cases"])):
        fd = [0.01, 0.02][i % 2]
        raw = opp.vad_to_timestamps(list(g[f"flags_{i}"]), fd)
        assert np.array_equal(np.array(raw, dtype=np.float64).reshape(-1, 2), g[f"raw_{i}"])
        proc = opp.process_timestamps(raw, 0.3, [0.2, 0.25][i % 2])
        assert np.array_equal(np.array(proc, dtype=np.float64).reshape(-1, 2), g[f"proc_{i}"])
    for v, s in zip(g["fmt_in"], g["fmt_out"]):
        assert opp.format_time(float(v)) == str(s)
    assert np.array_equal(opp.normalize_to_int16(g["norm_in"]), g["norm_out"])


def test_reference_checked_in_timestamps(golden):
    """The only expected output the reference ships (DFSMN near+far): pins formatter + int() truncation."""
    g = golden("dfsmn_golden_txt")
    idx, sec = g["indices"], [str(s) for s in g["seconds"]]
    # every index is int(seconds * 16000) of a 20 ms-grid time; reconstruct the grid time and re-format
    for (a, b), line in zip(idx, sec):
        for val, txt in ((a, line.split(" --> ")[0]), (b, line.split(" --> ")[1])):
            # starts are i*0.02, ends are i*0.02+0.02 (vad_to_timestamps) -- e.g. 294*0.02+0.02 -> 94399
            ks = range(int(val // 320) - 1, int(val // 320) + 3)
            cand = [k * 0.02 for k in ks] + [k * 0.02 + 0.02 for k in ks]
            hits = [c for c in cand if int(c * 16000) == val]
            assert hits, (val, cand)
            assert opp.format_time(hits[0]) == txt
    assert opp.format_time(2.28) == "00:00:02.279"
    assert int((294 * 0.02 + 0.02) * 16000) == 94399


# ------------------------------------------------------------------ a16
def test_vadpostprocessor(golden):
    g = golden("vadpost")
    for c, cfg in enumerate(g["cfgs_f"]):
        pp = opp.VadPostprocessor(int(cfg[0]), float(cfg[1]), *[int(v) for v in cfg[2:7]])
        for i in range(int(g["n_cases"])):
            dec = pp.process(g[f"probs_{i}"])
            assert np.array_equal(dec, g[f"f{c}_dec_{i}"]), (c, i)
            wav = float(g[f"f{c}_wav_{i}"])
            seg = pp.decision_to_segment(dec, None if wav < 0 else wav)
            assert np.array_equal(np.array(seg, dtype=np.float64).reshape(-1, 2), g[f"f{c}_seg_{i}"]), (c, i)
    for c, cfg in enumerate(g["cfgs_m"]):
        pp = opp.VadPostprocessor(int(cfg[0]), float(cfg[1]), *[int(v) for v in cfg[2:7]],
                                  frame_shift_s=float(cfg[7]), frame_length_s=None)
        for i in range(int(g["n_cases"])):
            dec = pp.process(g[f"probs_{i}"])
            assert np.array_equal(dec, g[f"m{c}_dec_{i}"]), (c, i)
            wav = float(g[f"m{c}_wav_{i}"])
            seg = pp.decision_to_segment(dec, None if wav < 0 else wav)
            assert np.array_equal(np.array(seg, dtype=np.float64).reshape(-1, 2), g[f"m{c}_seg_{i}"]), (c, i)


# ------------------------------------------------------------------ a13, a11
def test_silero_segmenter(golden):
    g = golden("silero_host")
    for i in range(int(g["n_cases"])):
        kw = dict(eval(str(g["kwargs"][i])))
        res = opp.silero_segments([float(v) for v in g[f"probs_{i}"]], int(g[f"nsamp_{i}"]), **kw)
        got = np.array([[d["start"], d["end"]] for d in res], dtype=np.float64).reshape(-1, 2)
        assert np.array_equal(got, g[f"res_{i}"]), i


def test_silero_wrapper_state_and_context(golden):
    g = golden("silero_host")
    w = {k: T(v) for k, v in weights.silero_synthetic(1234).items()}
    m = osil.OnnxWrapperOracle(w)
    probs = m.audio_forward(T(g["wrap_audio"]), 16000).numpy()
    np.testing.assert_allclose(probs, g["wrap_probs"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(m._state.numpy(), g["wrap_final_state"], rtol=0, atol=1e-6)
    assert np.array_equal(m._context.numpy(), g["wrap_final_context"])
    # what the reference wrapper actually fed the session: [B,576] = 64 context + 512 new samples
    audio = g["wrap_audio"]
    pad = (-audio.shape[1]) % 512
    ap = np.pad(audio, ((0, 0), (0, pad)))
    for k, x in enumerate(g["wrap_inputs"]):
        ctx = np.zeros((2, 64), np.float32) if k == 0 else ap[:, 512 * k - 64:512 * k]
        assert np.array_equal(x, np.concatenate([ctx, ap[:, 512 * k:512 * k + 512]], axis=1))
    with pytest.raises(ValueError):
        m(torch.zeros(1, 100), 16000)
    with pytest.raises(ValueError):
        m(torch.zeros(1, 512), 44100)


# ------------------------------------------------------------------ a4-a8
@pytest.mark.parametrize("seed", [1234, 7])
def test_fsmn_forward(golden, seed):
    g = golden("fsmn_forward")
    fe = ofs.Frontend()
    w = {k: T(v) for k, v in weights.fsmn_synthetic(seed).items()}
    clip = g[f"s{seed}_clip"]
    caches = [torch.zeros(1, 128, 19, 1) for _ in range(4)]
    for k in range(4):
        a = T(clip[k * 11040:k * 11040 + 16000].copy()).reshape(1, 1, -1)
        noise = g[f"s{seed}_noise_in_{k}"]
        score, caches, noisy, raw, db = ofs.forward(fe, w, a, caches, np.array([1.0], np.float32), noise, return_raw=True)
        want = g[f"s{seed}_score_{k}"]
        # uint8 gate: identical except where the float score/dB sits within round-off of a threshold
        diff = np.flatnonzero(score[0].numpy() != want)
        for i in diff:
            assert abs(float(raw[0, i]) - 1.0) < 1e-4 or abs(float(db[0, i]) - float(noise[0])) < 1e-4
        assert len(diff) <= 1
        for ci in range(4):
            np.testing.assert_allclose(caches[ci][0, :, :, 0].numpy(), g[f"s{seed}_cache{ci}_{k}"], rtol=0, atol=2e-4)
        if np.isnan(g[f"s{seed}_noisy_{k}"]):
            assert np.isnan(noisy.numpy()[0])
        elif len(diff) == 0:
            np.testing.assert_allclose(noisy.numpy()[0], g[f"s{seed}_noisy_{k}"], rtol=0, atol=1e-4)


def test_fsmn_hostloop(golden):
    g = golden("fsmn_hostloop")
    for c in range(int(g["n_cases"])):
        scores, noisy = g[f"scores_{c}"], g[f"noisy_{c}"]
        silence, saved = True, []
        noise = np.array([40.0], np.float32) * np.float32(0.1)
        for k in range(scores.shape[0]):
            flags, silence = opp.lookahead_vote(scores[k], 71, 30, 0.5, 0.5, silence)
            saved += flags
            if noisy[k] > 0.0:
                noise = 0.5 * (noise + noisy[k] + 1.0)
        flags, silence = opp.tail_flags_fsmn(scores[-1], 71, 101, silence)
        saved += flags
        assert np.array_equal(np.array(saved, bool), g[f"saved_{c}"]), c
        np.testing.assert_array_equal(np.asarray(noise, np.float32), g[f"noise_final_{c}"])


# ------------------------------------------------------------------ a21 + FireRed front-end
@pytest.mark.parametrize("seed", [1234, 7, 9])
def test_firered_forward(golden, seed):
    g = golden("firered_forward")
    cfgs = {1234: weights.FIRERED_CFG, 7: dict(weights.FIRERED_CFG, R=3, M=2, H=64, P=32, N1=8, S1=2, N2=4, S2=3),
            9: dict(weights.FIRERED_CFG, R=2, M=1, H=48, P=24, N1=5, S1=1, N2=0, S2=0, odim=3)}
    fe = ofr.Frontend()
    if seed == 1234:
        assert np.array_equal(fe.fbank[0].numpy(), g["kaldi_fbank"])
    w = {k: (T(v) if isinstance(v, np.ndarray) else v) for k, v in weights.firered_synthetic(seed, cfgs[seed]).items()}
    probs = ofr.forward(fe, w, T(g[f"s{seed}_audio"])).numpy()
    # the reference's own PyTorch-vs-ORT bar for this graph is rtol=atol=1e-5 (Export_FireRedVAD.py:1560)
    np.testing.assert_allclose(probs, g[f"s{seed}_probs"], rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------ f1: FireRed Stream-VAD (cache-carrying chunks)
STREAM_CFGS = {1234: dict(weights.FIRERED_CFG, N2=0, S2=0),
               7: dict(weights.FIRERED_CFG, R=3, M=2, H=64, P=32, N1=8, S1=2, N2=0, S2=0)}


@pytest.mark.parametrize("seed", [1234, 7])
def test_firered_stream_forward(golden, seed):
    g = golden("firered_stream")
    fe = ofr.Frontend()
    cfg = STREAM_CFGS[seed]
    w = {k: (T(v) if isinstance(v, np.ndarray) else v) for k, v in weights.firered_synthetic(seed, cfg).items()}
    clip = g[f"s{seed}_clip"]
    # first chunk: probabilities and the updated caches
    caches0 = torch.zeros(cfg["R"], 1, cfg["P"], (cfg["N1"] - 1) * cfg["S1"])
    pr, c1 = ofr.forward_stream(fe, w, T(clip[:2560]).reshape(1, 1, -1), caches0)
    np.testing.assert_allclose(pr[0, 0].numpy(), g[f"s{seed}_probs"][:14], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(c1.numpy(), g[f"s{seed}_caches_first"], rtol=1e-5, atol=1e-5)
    # whole ragged clip through the chunk loop (short last chunk zero-padded to 400 samples)
    _, allp = ofr.run_clip_stream(fe, w, clip)
    ref = g[f"s{seed}_probs"][:ofr.valid_frame_count(len(clip))]
    assert allp.shape == ref.shape
    np.testing.assert_allclose(allp, ref, rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_stream_vadpostprocessor(golden, impl):
    """Bit-exact segment boundaries for the streaming state machine, whole-track and fed 14 frames at a time
    (host code on both sides: the product class runs without a GPU)."""
    from vadx import vadpost
    cls = ofr.StreamVadPostprocessor if impl == "oracle" else vadpost.StreamVadPostprocessor
    g = golden("firered_stream")
    for c, cfg in enumerate(g["post_cfgs"]):
        args = (int(cfg[0]), float(cfg[1]), *[int(v) for v in cfg[2:]])
        for i in range(int(g["post_n"])):
            p = g[f"post_probs_{i}"]
            seg = np.array(cls(*args).process_batch(p.copy()), dtype=np.float64).reshape(-1, 2)
            assert np.array_equal(seg, g[f"post{c}_seg_{i}"]), (c, i)
            pp = cls(*args)
            pieces = [np.array(pp.process_batch(p[k:k + 14].copy()), dtype=np.float64).reshape(-1, 2)
                      for k in range(0, len(p), 14)]
            got = np.concatenate(pieces) if pieces else np.zeros((0, 2))
            assert np.array_equal(got, g[f"post{c}_chunked_{i}"]), (c, i)
            pp.reset()
            again = np.array(pp.process_batch(p.copy()), dtype=np.float64).reshape(-1, 2)
            assert np.array_equal(again, g[f"post{c}_seg_{i}"]), (c, i)


def test_melscale_fbanks_properties():
    """torchaudio is un-vendored (parity unpinned): check the published algorithm's invariants."""
    for scale, norm, fmin in (("htk", None, 20.0), ("slaney", "slaney", 0.0)):
        fb = omel.melscale_fbanks(257, fmin, 8000, 80, 16000, norm, scale)
        assert fb.shape == (257, 80) and fb.dtype == torch.float32
        assert float(fb.min()) >= 0.0
        peaks = fb.argmax(0)
        assert bool((peaks[1:] >= peaks[:-1]).all())
        if norm is None:
            assert float(fb.max()) <= 1.0 + 1e-6


# ------------------------------------------------------------------ a14 (in-tree BN folding)
def test_marblenet_bn_fold(golden):
    from oracle import marblenet as omb
    g = golden("marblenet_fold")
    for i in range(int(g["n_cases"])):
        has_b = bool(g[f"has_bias_{i}"])
        args = [g[f"{k}_{i}"] for k in ("gamma", "beta", "mean", "var")]
        w2, b2 = omb.fold_bn(T(g[f"w_{i}"]), T(g[f"b_{i}"]) if has_b else None, *[T(a) for a in args], float(g[f"eps_{i}"]))
        assert np.array_equal(w2.numpy(), g[f"fw_{i}"]) and np.array_equal(b2.numpy(), g[f"fb_{i}"])
        w3, b3 = weights.fold_bn(g[f"w_{i}"], g[f"b_{i}"] if has_b else None, *args, float(g[f"eps_{i}"]))      # product (numpy)
        np.testing.assert_allclose(w3, g[f"fw_{i}"], rtol=3e-7, atol=0)
        np.testing.assert_allclose(b3, g[f"fb_{i}"], rtol=3e-6, atol=2e-7)


# ------------------------------------------------------------------ a18-a20 (DFSMN near+far, ICCRN, UniDeepFsmn)
@pytest.mark.parametrize("seed", [1234, 7])
def test_dfsmn_forward(golden, seed):
    from oracle import dfsmn as od
    g = golden("dfsmn_forward")
    w = {k: T(v) for k, v in weights.dfsmn_synthetic(seed).items()}
    w["mask.shift"] = w["mask.shift"] + torch.log(torch.tensor(32768 ** 2, dtype=torch.float32))   # wrapper __init__ :291
    vad, aec = od.forward(od.Frontend(), w, T(g[f"s{seed}_near"]), T(g[f"s{seed}_far"]), weights.DFSMN_MASK["layers"])
    np.testing.assert_allclose(aec.numpy()[0, 0], g[f"s{seed}_aec"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(vad.numpy(), g[f"s{seed}_vad"], rtol=0, atol=2e-5)


def test_dfsmn_near_only_forward(golden):
    """The near-end-only export (DFSMN/only_near_end_audio): same graph with the far end replaced by two baked
    white-noise tensors, which the fixture carries as that model's constants."""
    from oracle import dfsmn as od
    g = golden("dfsmn_near_only")
    w = {k: T(v) for k, v in weights.dfsmn_synthetic(1234).items()}
    w["mask.shift"] = w["mask.shift"] + torch.log(torch.tensor(32768 ** 2, dtype=torch.float32))
    consts = (T(g["pow_far"].astype(np.float32)), T(g["far_comp"].astype(np.float32)))
    vad, _ = od.forward(od.Frontend(), w, T(g["near"]), None, weights.DFSMN_MASK["layers"], consts)
    np.testing.assert_allclose(vad.numpy(), g["vad"], rtol=0, atol=2e-5)


def test_dfsmn_hostloop(golden):
    from oracle import dfsmn as od
    g = golden("dfsmn_hostloop")
    for c in range(int(g["n_cases"])):
        scores = g[f"scores_{c}"]
        silence, saved = True, []
        for k in range(scores.shape[0]):
            flags, silence = opp.lookahead_vote(scores[k], 51 - 15, 15, 0.5, 0.5, silence, thresholds=(0.5, 0.5))
            saved += flags
        flags, silence = od.tail_flags(scores[-1], 51 - 15, 51, silence)
        saved += flags
        assert np.array_equal(np.array(saved, bool), g[f"saved_{c}"]), c


# ------------------------------------------------------------------ f3: audio ingest (pydub = audioop tomono + ratecv)
@pytest.mark.parametrize("ch,in_rate,out_rate,n", [(2, 48000, 16000, 10001), (1, 44100, 16000, 5000), (1, 8000, 16000, 777),
                                                   (2, 22050, 16000, 3001), (1, 16000, 16000, 100), (1, 48000, 16000, 1)])
def test_ingest_oracle_matches_audioop(ch, in_rate, out_rate, n):
    """The restatement is pinned against the stdlib audioop itself (what pydub calls in every reference driver)."""
    import audioop
    from oracle import ingest as oing
    x = np.random.default_rng(n).integers(-32768, 32768, n * ch).astype(np.int16)
    data = x.tobytes()
    if ch == 2:
        data = audioop.tomono(data, 2, 0.5, 0.5)
    if in_rate != out_rate:
        data, _ = audioop.ratecv(data, 2, 1, in_rate, out_rate, None)
    assert np.array_equal(oing.ingest(x, ch, in_rate, out_rate), np.frombuffer(data, dtype=np.int16))


# ------------------------------------------------------------------ round-2 fixtures
@pytest.mark.parametrize("tag", ["r05", "r20"])
def test_fsmn_speech_2_noise_ratio_branches(golden, tag):
    """SPEECH_2_NOISE_RATIO < 1 (score + 1) and > 1 (score + score ** ratio): FSMN/Export_FSMN_VAD.py:87-92."""
    g = golden("fsmn_extra")
    fe = ofs.Frontend()
    w = {k: T(v) for k, v in weights.fsmn_synthetic(1234).items()}
    ratio = float(g[f"{tag}_ratio"])
    caches = [torch.zeros(1, 128, 19, 1) for _ in range(4)]
    for k in range(2):
        a = T(g["clip"][k * 11040:k * 11040 + 16000].copy()).reshape(1, 1, -1)
        thr = np.array([g[f"{tag}_thr_{k}"]], np.float32)
        score, caches, noisy, raw, db = ofs.forward(fe, w, a, caches, thr, np.array([4.0], np.float32), ratio, return_raw=True)
        want = g[f"{tag}_score_{k}"]
        diff = np.flatnonzero(score[0].numpy() != want)
        for i in diff:
            assert abs(float(raw[0, i]) - float(thr[0])) < 1e-4 or abs(float(db[0, i]) - 4.0) < 1e-4
        assert len(diff) <= 1
        if len(diff) == 0:
            np.testing.assert_allclose(noisy.numpy()[0], g[f"{tag}_noisy_{k}"], rtol=0, atol=1e-4)
        assert 0 < int(want.sum()) < 101              # the gate really toggles inside the window


def test_fsmn_hostloop_look_backward_zero(golden):
    """LOOK_BACKWARD = 0: slide_range = score_len (taken BEFORE look_backward is bumped to 1), stride = L - 160, so every
    window contributes all 101 flags and the tail loop is empty (Inference_FSMN_VAD_ONNX.py:79-86, 188-234)."""
    g = golden("fsmn_extra")
    for c in range(int(g["lb0_n_cases"])):
        scores = g[f"lb0_scores_{c}"]
        silence, saved = True, []
        for k in range(scores.shape[0]):
            flags, silence = opp.lookahead_vote(scores[k], 101, 1, 0.5, 0.5, silence)
            saved += flags
        flags, silence = opp.tail_flags_fsmn(scores[-1], 101, 101, silence)
        saved += flags
        assert np.array_equal(np.array(saved, bool), g[f"lb0_saved_{c}"]), c
        assert len(saved) == 101 * scores.shape[0]


def test_normalise_audio(golden):
    """Oracle AND product copy of normalise_audio (host code, no GPU) against the reference function."""
    from vadx import timestamps as ts
    g = golden("host_extra")
    for i in range(int(g["n_cases"])):
        for fn in (opp.normalise_audio, ts.normalise_audio):
            got = fn(g[f"in_{i}"].copy())
            assert got.dtype == g[f"out_{i}"].dtype and np.array_equal(got, g[f"out_{i}"]), (i, fn.__module__)
    for fn in (opp.normalise_audio, ts.normalise_audio):
        assert np.array_equal(fn(g["target_4096_in"].copy(), 4096.0), g["target_4096_out"])


def test_marblenet_static_window_hostloop(golden):
    """Static-shape MarbleNet export: window grid, noise padding, per-window valid frames, concatenation -- the reference's
    module-level loop (Inference_NVIDIA_MarbleNet_VAD_ONNX.py:130-147, 369-388) run on the oracle network."""
    from oracle import marblenet as omb
    g = golden("marblenet_hostloop")
    w = {k: T(v) for k, v in weights.marblenet_synthetic(1234).items()}
    fe = omb.Frontend()
    for c in range(int(g["n_cases"])):
        noise = np.random.default_rng(int(g[f"noise_seed_{c}"])).standard_normal(40000)
        L = int(g[f"window_{c}"])
        padded, _ = opp.pad_to_window_grid(g[f"clip_{c}"], L, L, noise)
        assert np.array_equal(padded, g[f"aligned_{c}"])
        assert list(g[f"calls_{c}"]) == [L] * (len(padded) // L)
        _, probs, _ = omb.run_clip(fe, w, g[f"clip_{c}"], window=L, pad_noise=noise)
        assert probs.shape == g[f"probs_{c}"].shape
        np.testing.assert_allclose(probs, g[f"probs_{c}"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("tag,n_fft,win,wtype,variant", [
    ("fsmn", 512, 400, "hamming", "v1"), ("dfsmn_b", 319, 319, "hamming", "v1b"), ("dfsmn_a", 1024, 640, "hamming", "v1b"),
    ("marble", 512, 400, "hann_sym", "v2"), ("firered", 400, 400, "povey", "v2")])
def test_product_dft_tables_are_the_reference_tables(golden, tag, n_fft, win, wtype, variant):
    """The PRODUCT's host-side table builder (vadx.tables, fed to csrc/frontend.hip as data) hashes to the reference's
    STFT_Process buffers -- the GPU front-end tolerance (2e-4 on log-mel) could not tell a reduced-angle table from the
    reference's unreduced-float32-angle one, this can."""
    from vadx import tables
    g = golden("stft")
    cos_t, sin_t = tables.windowed_dft(n_fft, tables.analysis_window(wtype, win, n_fft, variant), variant)
    if variant == "v2":
        assert _sha(torch.cat([cos_t, sin_t], 0)) == str(g[f"{tag}_kernel_sha256"])
    else:
        assert _sha(cos_t) == str(g[f"{tag}_cos_sha256"]) and _sha(sin_t) == str(g[f"{tag}_sin_sha256"])


def test_product_kaldi_mel_bank_is_the_reference_bank(golden):
    from vadx import tables
    g = golden("firered_forward")
    fb = tables.as_np(tables.mel_filters_kaldi(400, 80, 16000, 20.0, 0.0))
    assert fb.shape == g["kaldi_fbank"].shape and np.array_equal(fb, g["kaldi_fbank"])


@pytest.mark.parametrize("rate", [8000, 48000, 44100, 22050])
def test_firered_in_graph_resample(golden, rate):
    """FireRedVAD_ONNX built with IN_SAMPLE_RATE != 16000 (Export_FireRedVAD.py:431-449): interpolate before the pre-emphasis
    for higher rates, after it for lower ones."""
    g = golden("resample")
    cfg = dict(zip(("idim", "R", "M", "H", "P", "N1", "S1", "N2", "S2", "odim"), (int(v) for v in g["firered_cfg"])))
    w = {k: (T(v) if isinstance(v, np.ndarray) else v) for k, v in weights.firered_synthetic(7, cfg).items()}
    probs = ofr.forward(ofr.Frontend(), w, T(g[f"firered_{rate}_audio"]), rate)
    assert probs.shape == g[f"firered_{rate}_probs"].shape
    np.testing.assert_allclose(probs.numpy(), g[f"firered_{rate}_probs"], rtol=0, atol=2e-6)


@pytest.mark.parametrize("rate", [8000, 48000, 32000])
def test_marblenet_in_graph_resample_frontend(golden, rate):
    from oracle import marblenet as omb
    g = golden("resample")
    lm = omb.log_mel(omb.Frontend(), T(g[f"marble_{rate}_audio"]), rate)
    assert lm.shape == g[f"marble_{rate}_logmel"].shape
    np.testing.assert_allclose(lm.numpy(), g[f"marble_{rate}_logmel"], rtol=0, atol=2e-5)


def test_silero_segmenter_8k(golden):
    """sampling_rate = 8000: 256-sample windows (utils_vad.py:345) through the same state machine."""
    g = golden("silero_8k")
    for i in range(int(g["n_cases"])):
        kw = dict(eval(str(g["kwargs"][i])))
        res = opp.silero_segments([float(v) for v in g[f"probs_{i}"]], int(g[f"nsamp_{i}"]), **kw)
        got = np.array([[d["start"], d["end"]] for d in res], dtype=np.float64).reshape(-1, 2)
        assert np.array_equal(got, g[f"res_{i}"]), i


def test_hostloops_with_asymmetric_thresholds(golden):
    """Round 3: SPEAKING_SCORE != SILENCE_SCORE (0.7 / 0.3, 0.3 / 0.7, 0.9 / 0.1, 0.6 / 0.6, 0.2 / 0.4) through both look-ahead loops --
    the reference's own loops on replayed scores (FSMN/Inference_FSMN_VAD_ONNX.py:188-234, DFSMN/.../Inference_DFSMN_VAD_ONNX.py:231-273)."""
    from oracle import dfsmn as od
    g = golden("hostloop_thresholds")
    for c, (spk, sil) in enumerate(g["pairs"]):
        spk, sil = float(spk), float(sil)
        scores = g[f"fsmn_scores_{c}"]
        silence, saved = True, []
        for k in range(scores.shape[0]):
            flags, silence = opp.lookahead_vote(scores[k], 71, 30, spk, sil, silence)
            saved += flags
        flags, silence = opp.tail_flags_fsmn(scores[-1], 71, 101, silence)
        saved += flags
        assert np.array_equal(np.array(saved, bool), g[f"fsmn_saved_{c}"]), ("fsmn", c)
        scores = g[f"dfsmn_scores_{c}"]
        silence, saved = True, []
        for k in range(scores.shape[0]):
            flags, silence = opp.lookahead_vote(scores[k], 51 - 15, 15, spk, sil, silence, thresholds=(spk, sil))
            saved += flags
        flags, silence = od.tail_flags(scores[-1], 51 - 15, 51, silence, spk, sil)
        saved += flags
        assert np.array_equal(np.array(saved, bool), g[f"dfsmn_saved_{c}"]), ("dfsmn", c)
