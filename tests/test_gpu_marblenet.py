"""GPU parity: MarbleNet sub-block kernels + classifier + device post-processing vs the oracle
(encoder restated from the published NeMo config: parity unpinned, see oracle/marblenet.py)."""
import numpy as np
import pytest
from conftest import chain_or_threshold
import torch

import vadx  # noqa: F401
from vadx import marblenet, weights
from oracle import marblenet as omb
from oracle import postproc as opp

pytestmark = pytest.mark.gpu
ATOL = 1e-4


@pytest.fixture(autouse=True, params=["f32", "h2"])
def gemm(request):
    """Every test of this file runs on both arithmetics of the fused blocks' 1x1 convs: float32 MFMAs and fp16 x 2 split products (the default)."""
    from vadx import _lib
    prev = _lib.gemm_mode(request.param)
    yield request.param
    _lib.gemm_mode(prev)


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


@pytest.mark.parametrize("seed,L", [(1234, 16000), (7, 48000), (1234, 89431), (7, 4000)])
def test_session_matches_oracle(seed, L):
    """The reference's own validation recipe: randint int16 at several lengths (Export_...:392-402)."""
    sess = marblenet.MarbleNetSession(weights.marblenet_synthetic(seed))
    assert isinstance(sess._inputs_meta[0].shape[-1], str)           # dynamic axis -> whole-clip windows
    rng = np.random.default_rng(1234)
    audio = rng.integers(-32768, 32767, (1, 1, L)).astype(np.int16)
    audio[0, 0, L // 3: L // 2] //= 300
    sil, act, slen = sess.run(None, {"audio": audio})
    ow = {k: T(v) for k, v in weights.marblenet_synthetic(seed).items()}
    osil, oact, olen = omb.forward(omb.Frontend(), ow, T(audio))
    assert act.shape == tuple(oact.shape) and sil.shape == tuple(osil.shape)
    assert slen.dtype == np.int32 and int(slen[0]) == int(olen)
    np.testing.assert_allclose(act, oact.numpy(), rtol=0, atol=ATOL)
    np.testing.assert_allclose(sil, osil.numpy(), rtol=0, atol=ATOL)
    np.testing.assert_allclose(act + sil, 1.0, rtol=0, atol=1e-5)


@pytest.mark.parametrize("n,window", [(89431, None), (160000, None), (40000, 16000), (9000, 16000), (50001, 24000)])
def test_whole_clip_segments(n, window):
    """Dynamic-axis (whole clip = one window) and static-window exports: device scores within 1e-4 of the oracle driver's,
    device decisions == the oracle post-processor run on the device scores, segments == the oracle's decision_to_segment of
    those decisions -- all unconditional; the end-to-end list equals the oracle's whenever its decisions do."""
    seed, B = 1234, 3
    post = (3, 0.5, 10, 1000, 10, 3, 0)
    eng = marblenet.MarbleNetEngine(weights.marblenet_synthetic(seed))
    ow = {k: T(v) for k, v in weights.marblenet_synthetic(seed).items()}
    fe = omb.Frontend()
    clips = weights.burst_clips(B, n, seed=n)
    noise = np.random.default_rng(2).standard_normal((B, 40000))
    got, track, dec = eng.detect(clips, window_len=window, pad_noise=noise, return_probs=True)
    full_chain = 0
    for b in range(B):
        want_seg, want_p, want_dec = omb.run_clip(fe, ow, clips[b], window=window, pad_noise=noise[b])
        assert track[b].shape[0] == want_p.shape[0]
        np.testing.assert_allclose(track[b].cpu().numpy(), want_p, rtol=0, atol=ATOL)
        opost = opp.VadPostprocessor(*post, frame_shift_s=0.02, frame_length_s=None)
        d2 = opost.process(track[b].cpu().numpy())
        assert np.array_equal(dec[b].cpu().numpy(), d2)
        assert got[b] == opost.decision_to_segment(d2, n / 16000)
        full_chain += chain_or_threshold(opost, track[b].cpu().numpy(), want_p, d2, want_dec, got[b], want_seg)
    assert full_chain >= 1          # a clip may sit on a threshold (checked by chain_or_threshold), not all of them


def test_ten_minute_clip_in_one_dynamic_window():
    """The dynamic-axis export takes a whole recording as ONE window, up to 3600 s (Inference_NVIDIA_MarbleNet_VAD_ONNX.py:130-135): a
    606 s clip (9.7 M samples -> 60 626 STFT frames -> 30 313 scores) through engine.detect against the oracle driver -- scores
    within 1e-4, decisions (max-speech splits at 1000 frames fire dozens of times) and segments as in test_whole_clip_segments."""
    n, post = 9_700_000, (3, 0.5, 10, 1000, 10, 3, 0)
    eng = marblenet.MarbleNetEngine(weights.marblenet_synthetic(1234))
    ow = {k: T(v) for k, v in weights.marblenet_synthetic(1234).items()}
    clip = weights.burst_clips(1, n, seed=606)
    got, track, dec = eng.detect(clip, window_len=None, return_probs=True)
    want_seg, want_p, want_dec = omb.run_clip(omb.Frontend(), ow, clip[0], window=None, pad_noise=None)
    assert tuple(track.shape) == (1, want_p.shape[0]) and want_p.shape[0] > 30000
    np.testing.assert_allclose(track[0].cpu().numpy(), want_p, rtol=0, atol=ATOL)
    opost = opp.VadPostprocessor(*post, frame_shift_s=0.02, frame_length_s=None)
    d2 = opost.process(track[0].cpu().numpy())
    assert np.array_equal(dec[0].cpu().numpy(), d2)
    assert got[0] == opost.decision_to_segment(d2, n / 16000) and len(got[0]) > 20
    chain_or_threshold(opost, track[0].cpu().numpy(), want_p, d2, want_dec, got[0], want_seg)      # whole chain, or a frame ON the threshold


def test_full_size_config4_properties():
    """BASELINE config 4 shape on one GPU: B=8192 clips of 89,431 samples (one window each)."""
    import time
    eng = marblenet.MarbleNetEngine(weights.marblenet_synthetic(1234))
    base = weights.burst_clips(64, 89431, seed=55)
    big = torch.from_numpy(base).cuda().repeat(128, 1)             # 8192 clips
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s0, s1, slen = eng.run(big)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert s1.shape == (8192, 280) and slen == 279
    v = s1.view(128, 64, -1)
    assert torch.equal(v[0], v[77]) and torch.equal(v[0], v[127])
    assert bool(torch.isfinite(s1).all()) and float((s0 + s1 - 1).abs().max()) < 1e-5
    ow = {k: T(v2) for k, v2 in weights.marblenet_synthetic(1234).items()}
    _, oact, _ = omb.forward(omb.Frontend(), ow, T(base[9]).reshape(1, 1, -1))
    np.testing.assert_allclose(s1[9].cpu().numpy(), oact[0, :, 0].numpy(), rtol=0, atol=ATOL)
    print(f"MarbleNet config-4 pass: {dt * 1e3:.1f} ms for 8192 x 5.59 s ({8192 * 89431 / 512 / dt / 1e6:.1f} M 512-hop frames/s)")


_ORACLE_C4 = {}


def test_config4_scores_and_segments_against_the_oracle(gemm):
    """BASELINE config 4 on the bench's OWN batch (bench_models.synth_pcm16, seed 1404: 8192 unique clips of 89 431 samples, one dynamic-axis
    window each): 64 clips spread over the batch against oracle.marblenet.run_clip on the same int16 samples -- scores within 1e-4, device
    decisions == the oracle post-processor on the device scores, segment lists equal unless a smoothed frame sits on the threshold."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench_models as bm
    post = (3, 0.5, 10, 1000, 10, 3, 0)
    w = weights.marblenet_synthetic(1234)
    eng = marblenet.MarbleNetEngine(w)
    audio = bm.synth_pcm16(torch, torch.device("cuda:0"), 8192, 89431, seed=1404).cpu().numpy()
    got, track, dec = eng.detect(audio, window_len=None, return_probs=True)
    assert eng.mode() == ("h2" if gemm == "h2" else "f32") and eng.range_fallbacks == 0
    idx = [int(k * 128 + (k * 11) % 128) for k in range(64)]
    if "ref" not in _ORACLE_C4:
        fe = omb.Frontend()
        ow = {k: T(v) for k, v in w.items()}
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        _ORACLE_C4["ref"] = [omb.run_clip(fe, ow, audio[b], window=None, pad_noise=None) for b in idx]
    full = 0
    for b, (want_seg, want_p, want_dec) in zip(idx, _ORACLE_C4["ref"]):
        tr = track[b].cpu().numpy()
        assert tr.shape[0] == want_p.shape[0]
        np.testing.assert_allclose(tr, want_p, rtol=0, atol=ATOL)
        opost = opp.VadPostprocessor(*post, frame_shift_s=0.02, frame_length_s=None)
        d2 = opost.process(tr)
        assert np.array_equal(dec[b].cpu().numpy(), d2)
        full += chain_or_threshold(opost, tr, want_p, d2, want_dec, got[b], want_seg)
    assert full >= len(idx) - 2, full


def test_fused_blocks_equal_the_per_sub_block_launches(gemm):
    """The fused residual-block / tail kernels against the ten per-sub-block launches they replace (same FIR order per element;
    the fused blocks sum each 1x1 conv's K in two halves, the per-sub-block launches in one run: scores agree to float32
    round-off), including a clip shorter than one tile and a ragged last tile.  On "h2" the fused launches run fp16 x 2 split products
    and the per-sub-block launches float32 MFMAs: the bound then also covers the two arithmetics' rounding (measured 5.1e-6 at worst)."""
    atol = 5e-6 if gemm == "f32" else 2e-5
    w = weights.marblenet_synthetic(7)
    eng = marblenet.MarbleNetEngine(w)
    assert eng.fused
    for n in (89431, 16000, 3000, 800):
        clips = T(weights.burst_clips(3, n, seed=n)).cuda()
        s0, s1, slen = eng.run(clips)
        eng.fused = False
        r0, r1, rlen = eng.run(clips)
        eng.fused = True
        assert slen == rlen and s1.shape == r1.shape
        np.testing.assert_allclose(s1.cpu().numpy(), r1.cpu().numpy(), rtol=0, atol=atol)
        np.testing.assert_allclose(s0.cpu().numpy(), r0.cpu().numpy(), rtol=0, atol=atol)


def test_fp16_range_protocol(gemm):
    """fp16 x 2 has no float32 exponent range: with the prologue's 1x1 weights scaled so that block 1's input exceeds 65504 the fused block
    raises the range flag, the engine recomputes the batch on float32 MFMAs and returns exactly their scores; weights that cannot be
    packed as fp16 keep the engine on float32 from the start."""
    if gemm != "h2":
        pytest.skip("the range protocol belongs to the fp16 x 2 arithmetic")
    w = dict(weights.marblenet_synthetic(7))
    w["b0r0_pw"] = np.asarray(w["b0r0_pw"]) * 3e4
    clips = T(weights.burst_clips(3, 16000, seed=5)).cuda()
    eng = marblenet.MarbleNetEngine(w)
    assert eng.mode() == "h2"
    s0, s1, _ = eng.run(clips)
    assert eng.range_fallbacks == 1 and int(eng._flag[0].item()) == 0
    ref = marblenet.MarbleNetEngine(w)
    ref.arithmetic = "f32"
    r0, r1, _ = ref.run(clips)
    assert torch.equal(s1, r1) and torch.equal(s0, r0) and ref.range_fallbacks == 0
    ok = marblenet.MarbleNetEngine(weights.marblenet_synthetic(7))
    ok.run(clips)
    assert ok.mode() == "h2" and ok.range_fallbacks == 0                    # in-range audio and weights never flag
    w2 = dict(weights.marblenet_synthetic(7))
    pw = np.array(w2["b1r0_pw"], dtype=np.float32)
    pw.flat[0] = 1e9                                                        # (BatchNorm folding scales it, still far outside fp16)
    w2["b1r0_pw"] = pw
    big = marblenet.MarbleNetEngine(w2)
    assert not big.h2_ok and big.mode() == "f32"


def test_whole_config_decision_record(gemm):
    """BASELINE config 4 at full size (8192 clips): segment lists of the default set (front-end kind 5, fused blocks on fp16 x 2) against the
    float32 set (dense float32 front-end product, float32 MFMAs)."""
    if gemm != "h2":
        pytest.skip("one run: the record sets both engines' arithmetic itself")
    import decision_records
    r = decision_records.marblenet_c4(torch, torch.device("cuda", 0))
    print(r)
    assert r["compared"] == 8192 and r["unexcused"] == 0, r
