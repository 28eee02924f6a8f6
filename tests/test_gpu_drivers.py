"""GPU: BASELINE config 1 plumbing -- the reference's own sample file through the script-level
drop-ins (file in -> the two timestamp text files out), checked against the oracle."""
import os

import numpy as np
import pytest
from conftest import chain_or_threshold
import torch

import vadx  # noqa: F401
from vadx import audio_io, drivers, silero, weights
from oracle import fsmn as ofs
from oracle import postproc as opp
from oracle import silero as osil

pytestmark = pytest.mark.gpu
WAV = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vad_sample.wav")


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def test_silero_on_vad_sample(tmp_path):
    sec, idx = str(tmp_path / "timestamps_second.txt"), str(tmp_path / "timestamps_indices.txt")
    model = silero.load_silero_vad(onnx=True, use_cpu=True, path="synthetic:1234")
    lines = []
    got = drivers.inference_silero(WAV, model, sec, idx, echo=lines.append)
    # oracle: the reference script's arithmetic step by step (Silero/Inference_Silero_VAD_ONNX.py:83-120)
    audio = audio_io.load_wav(WAV).astype(np.float32) * np.float32(0.000030517578)
    assert audio.shape == (89431,)
    ow = {k: T(v) for k, v in weights.silero_synthetic(1234).items()}
    res = osil.get_speech_timestamps(T(audio), osil.OnnxWrapperOracle(ow), threshold=0.5, max_speech_duration_s=20,
                                     min_speech_duration_ms=250, min_silence_duration_ms=250, return_seconds=True)
    want = opp.process_timestamps([(d["start"], d["end"]) for d in res], 0.3, 0.25)
    assert got == want
    sec_lines, idx_lines = opp.timestamp_lines(want, 16000)
    assert open(sec).read() == "".join(sec_lines) and open(idx).read() == "".join(idx_lines)
    assert any("Timestamps in Second" in str(l) for l in lines)


def test_fsmn_on_vad_sample(tmp_path):
    from vadx import fsmn
    sec, idx = str(tmp_path / "s.txt"), str(tmp_path / "i.txt")
    noise = np.random.default_rng(3).standard_normal((1, 20000))
    eng = fsmn.FsmnEngine(weights.fsmn_synthetic(1234))
    got = drivers.inference_fsmn(WAV, eng, sec, idx, pad_noise=noise, echo=lambda *_: None)
    a = opp.normalize_to_int16(audio_io.load_wav(WAV).astype(np.float32))
    ow = {k: T(v) for k, v in weights.fsmn_synthetic(1234).items()}
    want, _ = ofs.run_clip(ofs.Frontend(), ow, a, noise[0])
    assert got == want
    assert open(idx).read() == "".join(opp.timestamp_lines(want, 16000)[1])


def test_firered_aed_on_vad_sample():
    """RUN_AED drop-in: three event tracks, one device postprocessor per event, ratios of frames over threshold."""
    from vadx import firered
    from oracle import firered as ofr
    cfg = dict(weights.FIRERED_CFG, odim=3)
    wts = weights.firered_synthetic(1234, cfg)
    noise = np.random.default_rng(5).standard_normal((1, 20000))
    lines = []
    ts, ratio = drivers.inference_firered_aed(WAV, firered.FireRedEngine(wts), pad_noise=noise, echo=lines.append)
    wt = {k: (T(v) if isinstance(v, np.ndarray) else v) for k, v in wts.items()}
    a = audio_io.load_wav(WAV)
    ots, oratio, oprobs = ofr.run_clip_aed(ofr.Frontend(), wt, a, noise[0])
    assert list(ts) == ["speech", "singing", "music"] == list(oratio)
    thr = {"speech": 0.4, "singing": 0.5, "music": 0.5}
    for i, ev in enumerate(ts):
        near = int(np.sum(np.abs(oprobs[i] - thr[ev]) < 2e-4))      # frames that may legitimately flip
        assert abs(ratio[ev] - oratio[ev]) <= near / oprobs.shape[1] + 1e-3
        if near == 0:
            assert ts[ev] == ots[ev], ev
    assert any("AED Results" in str(l) for l in lines)
    with pytest.raises(ValueError):
        firered.FireRedEngine(weights.firered_synthetic(1234)).detect_events(a[None, :])


def test_firered_stream_on_vad_sample():
    from vadx import firered
    from oracle import firered as ofr
    cfg = dict(weights.FIRERED_CFG, N2=0, S2=0)
    wts = weights.firered_synthetic(1234, cfg)
    eng = firered.FireRedEngine(wts, firered.STREAM_CHUNK_SAMPLES)
    got = drivers.inference_firered_stream(WAV, eng, STREAM_VAD_THRESHOLD=0.3, echo=lambda *_: None)
    wt = {k: (T(v) if isinstance(v, np.ndarray) else v) for k, v in wts.items()}
    a = audio_io.load_wav(WAV)
    want, oprobs = ofr.run_clip_stream(ofr.Frontend(), wt, a, post=(5, 0.3, 5, 8, 2000, 20))
    _, track = eng.stream_detect(a[None, :], post=(5, 0.3, 5, 8, 2000, 20), return_probs=True)
    np.testing.assert_allclose(track[0], oprobs, rtol=0, atol=1e-4)
    csum = np.concatenate([[0.0], np.cumsum(oprobs, dtype=np.float64)])
    k = np.arange(1, len(oprobs) + 1)
    smooth = (csum[k] - csum[np.maximum(k - 5, 0)]) / np.minimum(k, 5)
    if np.min(np.abs(smooth - 0.3)) > 1e-3:
        assert got == want


@pytest.mark.parametrize("ch,in_rate,out_rate,n", [(2, 48000, 16000, 100001), (1, 44100, 16000, 50000), (1, 8000, 16000, 7777),
                                                   (2, 22050, 16000, 30001), (1, 16000, 16000, 1000)])
def test_device_ingest_matches_audioop(ch, in_rate, out_rate, n):
    """Stereo down-mix + resampling on the GPU, bit-exact against the stdlib audioop (= pydub's arithmetic), batched."""
    import audioop
    x = np.random.default_rng(n).integers(-32768, 32768, (3, n * ch)).astype(np.int16)
    got = audio_io.ingest_device(x, ch, in_rate, out_rate).cpu().numpy()
    for b in range(3):
        data = x[b].tobytes()
        if ch == 2:
            data = audioop.tomono(data, 2, 0.5, 0.5)
        if in_rate != out_rate:
            data, _ = audioop.ratecv(data, 2, 1, in_rate, out_rate, None)
        assert np.array_equal(got[b], np.frombuffer(data, dtype=np.int16)), b


def test_device_ingest_on_vad_sample():
    raw, nch, rate = audio_io.read_wav_raw(WAV)
    assert (nch, rate) == (2, 48000)
    got = audio_io.ingest_device(raw, nch, rate)[0].cpu().numpy()
    assert np.array_equal(got, audio_io.load_wav(WAV)) and got.shape == (89431,)


def test_dfsmn_on_reference_example_pair(tmp_path):
    """The reference's own near-end / far-end example recordings (DFSMN/near_and_far_end_audio/examples) through the
    script-level drop-in, against the oracle's restatement of the same driver."""
    from vadx import dfsmn
    from oracle import dfsmn as od
    gold = os.path.join(os.path.dirname(WAV))
    near_p, far_p = os.path.join(gold, "dfsmn_nearend_mic.wav"), os.path.join(gold, "dfsmn_farend_speech.wav")
    wts = weights.dfsmn_synthetic(1234)
    nz1, nz2 = np.random.default_rng(8).standard_normal((1, 20000)), np.random.default_rng(9).standard_normal((1, 20000))
    sec, idx = str(tmp_path / "s.txt"), str(tmp_path / "i.txt")
    got = drivers.inference_dfsmn(near_p, far_p, dfsmn.DfsmnEngine(wts), sec, idx, pad_noise_near=nz1, pad_noise_far=nz2, echo=lambda *_: None)
    w = {k: T(v) for k, v in wts.items()}
    w["mask.shift"] = w["mask.shift"] + torch.log(torch.tensor(32768 ** 2, dtype=torch.float32))
    a, f = audio_io.load_wav(near_p).astype(np.float32), audio_io.load_wav(far_p).astype(np.float32)
    assert a.shape == (159999,) and f.shape == (159999,)
    want, _ = od.run_clip(od.Frontend(), w, a, f, nz1[0], nz2[0], weights.DFSMN_MASK["layers"])
    assert got == want
    assert open(idx).read() == "".join(opp.timestamp_lines(want, 16000)[1])


def test_firered_and_marblenet_on_vad_sample(tmp_path):
    """The remaining two script-level drop-ins on the reference's sample file: probabilities within 1e-4 of the oracle
    drivers, identical segments whenever no frame sits within tolerance of a decision threshold, files written."""
    from vadx import firered, marblenet
    from oracle import firered as ofr
    from oracle import marblenet as omb
    a = audio_io.load_wav(WAV)
    # FireRed VAD
    wfr = weights.firered_synthetic(1234)
    noise = np.random.default_rng(11).standard_normal((1, 20000))
    sec, idx = str(tmp_path / "fs.txt"), str(tmp_path / "fi.txt")
    efr = firered.FireRedEngine(wfr)
    got = drivers.inference_firered(WAV, efr, sec, idx, pad_noise=noise, echo=lambda *_: None)
    wt = {k: (T(v) if isinstance(v, np.ndarray) else v) for k, v in wfr.items()}
    want, oprobs, odec = ofr.run_clip(ofr.Frontend(), wt, a, noise[0])
    _, track, dec = efr.detect(a[None, :], pad_noise=noise, return_probs=True)
    np.testing.assert_allclose(track[0].cpu().numpy(), oprobs, rtol=0, atol=1e-4)
    # unconditional: device decisions == oracle post-processor on the device scores; files == oracle segments of those
    opost = opp.VadPostprocessor(5, 0.4, 20, 2000, 20, 5, 0)
    d2 = opost.process(track[0].cpu().numpy())
    assert np.array_equal(dec[0].cpu().numpy(), d2)
    seg2 = opost.decision_to_segment(d2, len(a) / 16000)
    assert got == seg2 and open(idx).read() == "".join(opp.timestamp_lines(seg2, 16000)[1])
    assert open(sec).read() == "".join(opp.timestamp_lines(seg2, 16000)[0])
    chain_or_threshold(opost, track[0].cpu().numpy(), oprobs, d2, np.asarray(odec), got, want)      # whole chain, or a frame ON the threshold
    # MarbleNet (dynamic axis: the whole clip is one window)
    wm = weights.marblenet_synthetic(1234)
    em = marblenet.MarbleNetEngine(wm)
    sec2, idx2 = str(tmp_path / "ms.txt"), str(tmp_path / "mi.txt")
    got_m = drivers.inference_marblenet(WAV, em, sec2, idx2, echo=lambda *_: None)
    ow = {k: T(v) for k, v in wm.items()}
    want_m, p_m, dec_m = omb.run_clip(omb.Frontend(), ow, a)
    _, track_m, dec_g = em.detect(a[None, :], return_probs=True)
    np.testing.assert_allclose(track_m[0].cpu().numpy(), p_m, rtol=0, atol=1e-4)
    opost = opp.VadPostprocessor(3, 0.5, 10, 1000, 10, 3, 0, frame_shift_s=0.02, frame_length_s=None)
    d2 = opost.process(track_m[0].cpu().numpy())
    assert np.array_equal(dec_g[0].cpu().numpy(), d2)
    seg2 = opost.decision_to_segment(d2, len(a) / 16000)
    assert got_m == seg2 and open(idx2).read() == "".join(opp.timestamp_lines(seg2, 16000)[1])
    chain_or_threshold(opost, track_m[0].cpu().numpy(), p_m, d2, dec_m, got_m, want_m)


def test_dfsmn_near_only_on_vad_sample(tmp_path):
    from vadx import dfsmn
    from oracle import dfsmn as od
    wts = weights.dfsmn_synthetic(1234)
    pf, fc = weights.dfsmn_near_only_constants(1234)
    eng = dfsmn.DfsmnEngine(wts)
    eng.set_near_only_constants(pf, fc)
    nz = np.random.default_rng(12).standard_normal((1, 20000))
    sec, idx = str(tmp_path / "s.txt"), str(tmp_path / "i.txt")
    got = drivers.inference_dfsmn_near_only(WAV, eng, sec, idx, pad_noise=nz, echo=lambda *_: None)
    w = {k: T(v) for k, v in wts.items()}
    w["mask.shift"] = w["mask.shift"] + torch.log(torch.tensor(32768 ** 2, dtype=torch.float32))
    a = audio_io.load_wav(WAV).astype(np.float32)
    want, _ = od.run_clip(od.Frontend(), w, a, None, nz[0], None, weights.DFSMN_MASK["layers"], near_only=(T(pf), T(fc)))
    assert got == want
    assert open(idx).read() == "".join(opp.timestamp_lines(want, 16000)[1])


def test_file_lists_run_as_length_groups_and_match_single_runs(tmp_path):
    """A LIST of files: equal-length files share one device batch (drivers._grouped), results come back in input order and
    equal the one-file-at-a-time runs; weights given as the explicit "synthetic:<seed>" string resolve through checkpoints."""
    import wave
    from vadx import firered, fsmn, marblenet
    rng = np.random.default_rng(77)
    paths, lens = [], [40000, 25000, 40000, 16000]
    for k, n in enumerate(lens):
        pcm = weights.burst_clips(1, n, seed=500 + k)[0]
        pth = str(tmp_path / f"c{k}.wav")
        with wave.open(pth, "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(pcm.tobytes())
        paths.append(pth)
    noise = rng.standard_normal((len(paths), 20000))
    quiet = lambda *_: None                                                     # noqa: E731
    for run, eng in ((drivers.inference_fsmn, fsmn.FsmnEngine("synthetic:1234")),
                     (drivers.inference_firered, firered.FireRedEngine("synthetic:1234")),
                     (drivers.inference_marblenet, marblenet.MarbleNetEngine("synthetic:1234"))):
        kw = {} if run is drivers.inference_marblenet else {"pad_noise": noise}
        many = run(paths, eng, str(tmp_path / "s.txt"), str(tmp_path / "i.txt"), echo=quiet, **kw)
        assert len(many) == len(paths)
        for k, pth in enumerate(paths):
            kw1 = {} if run is drivers.inference_marblenet else {"pad_noise": noise[k:k + 1]}
            one = run(pth, eng, str(tmp_path / "s1.txt"), str(tmp_path / "i1.txt"), echo=quiet, **kw1)
            assert many[k] == one, (run.__name__, k)
    with pytest.raises(ValueError, match="no weights given"):
        drivers.inference_fsmn(paths[0], None, echo=quiet)


@pytest.mark.parametrize("pinned", [True, False])
def test_host_feed_is_bitwise_the_resident_path(pinned):
    """SURVEY 8e for the int16 engines: the batch stays in (pinned) HOST memory and crosses PCIe chunk by chunk on a copy stream
    while the previous chunk's launches run (vadx.feed.HostPcmFeed) -- FSMN flags, MarbleNet / FireRed scores and DFSMN window
    scores equal the resident batch's bit for bit, also with a ragged last chunk and when the feed object is reused."""
    from vadx import dfsmn, feed, firered, fsmn, marblenet
    prep = (lambda a: feed.pin(a)) if pinned else (lambda a: torch.from_numpy(a))
    # FSMN: 3 windows per clip on the engine's grid, 7 clips in chunks of 3 (3 + 3 + 1)
    eng = fsmn.FsmnEngine("synthetic:1234")
    lb, stride = eng.grid()
    W = 3
    pcm = weights.burst_clips(7, (W - 1) * stride + eng.L, seed=31)
    f = feed.HostPcmFeed(eng.device, pcm.shape[1], 3)
    for _ in range(2):
        assert torch.equal(eng.flags_from_host(prep(pcm), W, feed=f), eng.flags(torch.from_numpy(pcm).cuda(), W))
    # MarbleNet: one dynamic-axis window per clip
    mb = marblenet.MarbleNetEngine("synthetic:1234")
    pcm = weights.burst_clips(5, 24000, seed=32)
    got, want = mb.run_from_host(prep(pcm), chunk_clips=2), mb.run(torch.from_numpy(pcm).cuda())
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]) and got[2] == want[2]
    # FireRed: two one-second windows per clip
    fr = firered.FireRedEngine("synthetic:1234")
    pcm = weights.burst_clips(5, 32000, seed=33)
    assert torch.equal(fr.run_from_host(prep(pcm), 2, chunk_clips=4), fr.run(torch.from_numpy(pcm).cuda(), 2))
    # DFSMN near + far: two parallel streams per chunk
    df = dfsmn.DfsmnEngine("synthetic:1234")
    near, far = weights.burst_clips(3, 16001, seed=34), weights.burst_clips(3, 16001, seed=35)
    assert torch.equal(df.run_from_host(prep(near), prep(far), chunk_clips=2), df.run(torch.from_numpy(near).cuda(), torch.from_numpy(far).cuda()))
    with pytest.raises(ValueError):
        f.map([torch.from_numpy(pcm).cuda()], lambda a: a)


def test_fp16_io_variants_are_the_f32_results_rounded_once(tmp_path):
    """I/O-compatible fp16 variants (float32 arithmetic inside): MarbleNet / DFSMN sessions with io_dtype="float16" return the float32
    session's scores rounded once to float16 with float16 metadata (NVIDIA_.../Optimize_ONNX.py:37-44, DFSMN/.../Optimize_ONNX.py:40-44:
    keep_io_types=False); the Silero driver with use_fp16 feeds float16-quantised samples (Inference_Silero_VAD_ONNX.py:16, :83)."""
    from vadx import dfsmn, marblenet
    a = weights.burst_clips(1, 20000, seed=41).reshape(1, 1, -1)
    m32, m16 = marblenet.MarbleNetSession("synthetic:1234"), marblenet.MarbleNetSession("synthetic:1234", io_dtype="float16")
    r32, r16 = m32.run(None, {"audio": a}), m16.run(None, {"audio": a})
    assert r16[0].dtype == np.float16 and r16[1].dtype == np.float16 and r16[2].dtype == np.int32
    assert np.array_equal(r16[1], r32[1].astype(np.float16)) and np.array_equal(r16[2], r32[2])
    assert m16.get_outputs()[0].type == "tensor(float16)" and m32.get_outputs()[0].type == "tensor(float)"
    near, far = weights.burst_clips(1, 16001, seed=42).reshape(1, 1, -1), weights.burst_clips(1, 16001, seed=43).reshape(1, 1, -1)
    d32, d16 = dfsmn.DfsmnSession("synthetic:1234"), dfsmn.DfsmnSession("synthetic:1234", io_dtype="float16")
    v32 = d32.run(None, {"near_end_audio": near, "far_end_audio": far})[0]
    v16 = d16.run(None, {"near_end_audio": near, "far_end_audio": far})[0]
    assert v16.dtype == np.float16 and np.array_equal(v16, v32.astype(np.float16)) and d16.get_outputs()[0].type == "tensor(float16)"
    # Silero: the driver's float16 feed == running the float32 path on the quantised samples
    model = silero.load_silero_vad(onnx=True, use_cpu=True, path="synthetic:1234")
    got = drivers.inference_silero(WAV, model, str(tmp_path / "s.txt"), str(tmp_path / "i.txt"), use_fp16=True, echo=lambda *_: None)
    q = (audio_io.load_wav(WAV).astype(np.float16) * np.float16(0.000030517578)).astype(np.float32)
    res = silero.get_speech_timestamps_batch(q[None, :], model, threshold=0.5, max_speech_duration_s=20, min_speech_duration_ms=250,
                                             min_silence_duration_ms=250, return_seconds=True)[0]
    from vadx import timestamps
    assert got == timestamps.process_timestamps([(d["start"], d["end"]) for d in res], 0.3, 0.25)
