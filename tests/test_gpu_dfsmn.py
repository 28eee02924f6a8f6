"""GPU parity: DFSMN near+far building blocks and full path vs the oracle (itself pinned against the
reference's ICCRN / wrapper classes by tests/golden/dfsmn_forward.npz)."""
import numpy as np
import pytest
import torch

import vadx  # noqa: F401
from vadx import _lib, dfsmn, weights
from oracle import dfsmn as od

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["f32", "split", "h2"])
def gemm(request):
    """Every test of this file runs on both arithmetics of the kernels that have both (the CepsUnit's frequency-axis LSTM): exact-f32
    MFMAs and bf16 x 3 split products."""
    prev = _lib.gemm_mode(request.param)
    yield request.param
    _lib.gemm_mode(prev)


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


@pytest.fixture(scope="module")
def wts():
    w = weights.dfsmn_synthetic(1234)
    ow = {k[len("iccrn."):]: T(v) for k, v in w.items() if k.startswith("iccrn.")}
    return w, ow


@pytest.mark.parametrize("name,cin,frames,chunks", [("cfb_e1", 20, 101, 2), ("cfb_d4", 40, 37, 1)])
def test_cfb_block(wts, name, cin, frames, chunks):
    w, ow = wts
    net = dfsmn.Iccrn(w)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(chunks, cin, 160, frames, generator=g) * 0.7
    tb = od.IccrnTables(200).ceps
    want = torch.cat([od.cfb(x[n:n + 1], ow, name, tb) for n in range(chunks)], 0)
    xin = dfsmn.to_ft(torch, x, net.device)
    out = dfsmn.FT(torch, net.device, chunks, frames, 20, 160)
    if cin == 20:
        net.cfb(name, xin.view(), None, out.view(), chunks, frames)
    else:
        net.cfb(name, xin.view(0, 20), xin.view(20, 20), out.view(), chunks, frames)
    got = dfsmn.from_ft(out, chunks).cpu()
    err = (got - want).abs().max().item()
    assert err < 2e-4 * max(1.0, want.abs().max().item()), err


@pytest.mark.parametrize("name,cin,frames,chunks", [("cfb_e2", 20, 101, 3), ("cfb_d2", 40, 53, 2), ("cfb_d5", 20, 16, 300)])
def test_fused_cfb_equals_the_unfused_chain(wts, name, cin, frames, chunks):
    """cfb_front -> lstm_f -> cfb_back (LayerNorm 1 / 2 commuted behind the (3,1) conv / the DFT, every intermediate on chip) against
    the six-launch chain of building-block kernels with every intermediate in HBM: same block, re-associated sums only; the
    partial statistics cfb_back emits merge to what a separate pass over its output measures; more tiles than workgroups
    (300 > 256 CUs) exercise the persistent loop."""
    w, _ = wts
    net = dfsmn.Iccrn(w)
    tiles = chunks * dfsmn.ft_tiles(frames)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(chunks, cin, 160, frames, generator=g) * 0.9 + 0.3
    xin = dfsmn.to_ft(torch, x, net.device)
    a, b = (xin.view(), None) if cin == 20 else (xin.view(0, 20), xin.view(20, 20))
    out_f = dfsmn.FT(torch, net.device, chunks, frames, 40, 160)          # written into channel slice [20, 40) of a 40-channel tensor
    out_u = dfsmn.FT(torch, net.device, chunks, frames, 20, 160)
    _, part_f = net.cfb(name, a, b, out_f.view(20, 20), chunks, frames)
    _, part_u = net.cfb_unfused(name, a, b, out_u.view(), chunks, frames)
    got, want = dfsmn.from_ft(out_f, chunks)[:, 20:].cpu(), dfsmn.from_ft(out_u, chunks).cpu()
    assert torch.isfinite(got).all()
    scale = max(1.0, want.abs().max().item())
    assert (got - want).abs().max().item() < 2e-5 * scale
    sf, su = net.merged_stats(part_f, None, tiles), net.stats(out_f.view(20, 20), None, 160, tiles)
    valid = torch.arange(dfsmn.ft_tiles(frames) * 16, device=sf.device).view(-1, 16) < frames      # padding frames hold zeros / garbage
    valid = valid.repeat(chunks, 1)
    torch.testing.assert_close(sf[valid], su[valid], rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("name,cin,frames,chunks", [("cfb_e2", 20, 101, 3), ("cfb_d5", 20, 16, 300)])
def test_opt_in_split_second_half_equals_the_unfused_chain(wts, monkeypatch, gemm, name, cin, frames, chunks):
    """cfb_back on split products (VADX_CFB_BACK=split; opt-in: measured no faster than the f32-MFMA kernel, this half waits on its
    64-byte rows, not on the matrix pipe): inverse DFT with the 81 + 79 parts ordered as five k-steps, bin 80 riding in the slot of the
    non-existent im of bin 0 -- same block as the six-launch chain."""
    _lib.gemm_mode("split")                                   # (the opt-in kernel exists on the split arithmetic only; the fixture restores the mode)
    w, _ = wts
    net = dfsmn.Iccrn(w)
    net.cfb_back_split = True                                 # what VADX_CFB_BACK=split sets at construction
    g = torch.Generator().manual_seed(13)
    x = torch.randn(chunks, cin, 160, frames, generator=g) * 0.9 + 0.3
    xin = dfsmn.to_ft(torch, x, net.device)
    out_f = dfsmn.FT(torch, net.device, chunks, frames, 20, 160)
    out_u = dfsmn.FT(torch, net.device, chunks, frames, 20, 160)
    net.cfb(name, xin.view(), None, out_f.view(), chunks, frames)
    net.cfb_back_split = False
    net.cfb_unfused(name, xin.view(), None, out_u.view(), chunks, frames)
    got, want = dfsmn.from_ft(out_f, chunks).cpu(), dfsmn.from_ft(out_u, chunks).cpu()
    assert torch.isfinite(got).all()
    assert (got - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("chunks,frames", [(3, 101), (4200, 16)])       # 21 tiles: two workgroups per tile; 4200: one
def test_fused_layernorm_statistics_match_the_separate_pass(wts, chunks, frames):
    """The partial statistics pw_conv emits while writing a tensor, merged (one tensor and a channel concatenation of
    two), equal what frame_stats computes by re-reading it -- also with a large mean (cancellation) -- and torch's."""
    w, _ = wts
    net = dfsmn.Iccrn(w)
    tiles = chunks * dfsmn.ft_tiles(frames)
    torch.manual_seed(chunks)
    xin = dfsmn.FT(torch, net.device, chunks, frames, 24, 160)
    xin.data.normal_(25.0, 0.7)                                                        # mean >> std
    y = dfsmn.FT(torch, net.device, chunks, frames, 40, 160)
    p0, p1 = net.new_part(tiles), net.new_part(tiles)
    for half, part in ((0, p0), (20, p1)):
        net.pw(0, xin.view(0, 20) if half == 0 else xin.view(4, 20), xin.view(20, 4) if half == 0 else xin.view(0, 4), None,
               "in_conv.weight", "in_conv.bias", y.view(half, 20), 160, 20, tiles=tiles, part0=part)
    for got, ref in ((net.merged_stats(p0, None, tiles), net.stats(y.view(0, 20), None, 160, tiles)),
                     (net.merged_stats(p0, p1, tiles), net.stats(y.view(0, 20), y.view(20, 20), 160, tiles))):
        torch.testing.assert_close(got, ref, rtol=2e-5, atol=1e-6)
    yt = dfsmn.from_ft(y, chunks)[:2].cpu().double()                                   # [2, 40, 160, frames]
    mean = yt.mean(dim=(1, 2)); inv = 1.0 / (yt.std(dim=(1, 2), unbiased=True) + 1e-6)
    got = net.merged_stats(p0, p1, tiles).cpu().double().view(chunks, -1, 2)[:2, :frames]
    torch.testing.assert_close(got[..., 0], mean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(got[..., 1], inv, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("seed", [1234, 7])
def test_session_matches_reference_fixture(golden, seed):
    """Whole graph, two streams in -> vad_results, against what the REFERENCE classes produced."""
    g = golden("dfsmn_forward")
    sess = dfsmn.DfsmnSession(weights.dfsmn_synthetic(seed))
    assert [m.name for m in sess.get_inputs()] == ["near_end_audio", "far_end_audio"]
    vad, aec = sess.engine.run(g[f"s{seed}_near"].reshape(1, -1), g[f"s{seed}_far"].reshape(1, -1), return_aec=True)
    a = aec.cpu().numpy()[0]
    assert np.abs(a - g[f"s{seed}_aec"]).max() < 2e-4 * max(1.0, np.abs(g[f"s{seed}_aec"]).max())
    np.testing.assert_allclose(vad.cpu().numpy()[0], g[f"s{seed}_vad"], rtol=0, atol=1e-4)
    out = sess.run(None, {"near_end_audio": g[f"s{seed}_near"], "far_end_audio": g[f"s{seed}_far"]})[0]
    assert out.shape == (51,) and out.dtype == np.float32


def test_whole_clip_pair_segments():
    seed = 1234
    eng = dfsmn.DfsmnEngine(weights.dfsmn_synthetic(seed))
    w = {k: T(v) for k, v in weights.dfsmn_synthetic(seed).items()}
    w["mask.shift"] = w["mask.shift"] + torch.log(torch.tensor(32768 ** 2, dtype=torch.float32))
    fe = od.Frontend()
    B, n = 2, 40000
    near, far = weights.burst_clips(B, n, seed=3), weights.burst_clips(B, n + 500, seed=4)
    nz1, nz2 = np.random.default_rng(5).standard_normal((B, 20000)), np.random.default_rng(6).standard_normal((B, 20000))
    got = eng.detect(near, far, nz1, nz2)
    for b in range(B):
        want, _ = od.run_clip(fe, w, near[b], far[b], nz1[b], nz2[b], weights.DFSMN_MASK["layers"])
        assert got[b] == want


def test_sub_batching_and_batch_position_do_not_change_scores():
    """A clip pair's window scores must not depend on its neighbours, its position in the batch or how the engine cuts
    the batch into sub-batches (the only batch-size dependence is the summation order of the fused LayerNorm statistics:
    two workgroups per tile below 4096 tiles -- last-bit differences, bounded here)."""
    w = weights.dfsmn_synthetic(1234)
    big, small = dfsmn.DfsmnEngine(w, sub_batch=960), dfsmn.DfsmnEngine(w, sub_batch=7)
    lb, stride = big.grid()
    W = 3
    n = (W - 1) * stride + big.L
    near = torch.from_numpy(weights.burst_clips(5, n, seed=21)).cuda()
    far = torch.from_numpy(weights.burst_clips(5, n, seed=22)).cuda()
    ref = big.run(near, far, W, stride)                                   # [15, 51] in one sub-batch
    cut = small.run(near, far, W, stride)                                 # 2 clips (6 windows) per sub-batch
    assert torch.equal(ref.view(5, W, -1)[:4], cut.view(5, W, -1)[:4])    # equal tile counts per launch shape -> bitwise
    torch.testing.assert_close(cut, ref, rtol=0, atol=2e-6)
    alone = big.run(near[3:4], far[3:4], W, stride)
    torch.testing.assert_close(alone, ref.view(5, W, -1)[3], rtol=0, atol=2e-6)
    rep = big.run(near.repeat(2, 1), far.repeat(2, 1), W, stride).view(2, 5, W, -1)
    assert torch.equal(rep[0], rep[1])                                    # position in the batch: bitwise


def test_time_lstm_fp16_range_protocol(gemm):
    """The two-layer time LSTM's fp16 x 2 form splits its weights and inputs itself and raises the engine's range flag when one lies outside
    the fp16 range: the engine then sweeps the batch again with that kernel on float32 MFMAs and returns exactly those scores."""
    if gemm != "h2":
        pytest.skip("the range protocol belongs to the fp16 x 2 arithmetic")
    w = dict(weights.dfsmn_synthetic(1234))
    key = "iccrn.ch_lstm.lstm2.weight_ih_l0"
    big = np.array(w[key], dtype=np.float32)
    big[3, 2] = 1.0e6
    w[key] = big
    eng, ref = dfsmn.DfsmnEngine(w, sub_batch=960), dfsmn.DfsmnEngine(w, sub_batch=960)
    ref.iccrn.lstm_t_h2 = False
    lb, stride = eng.grid()
    W = 2
    n = (W - 1) * stride + eng.L
    near = torch.from_numpy(weights.burst_clips(3, n, seed=31)).cuda()
    far = torch.from_numpy(weights.burst_clips(3, n, seed=32)).cuda()
    got, want = eng.run(near, far, W, stride), ref.run(near, far, W, stride)
    assert eng.range_fallbacks == 1 and ref.range_fallbacks == 0 and int(eng.iccrn.range_flag[0].item()) == 0
    assert torch.equal(got.nan_to_num(), want.nan_to_num())
    ok = dfsmn.DfsmnEngine(weights.dfsmn_synthetic(1234), sub_batch=960)
    ok.run(near, far, W, stride)
    assert ok.range_fallbacks == 0                                        # in-range weights and audio never flag


def test_near_only_session_matches_reference_fixture(golden):
    """DFSMN/only_near_end_audio: one stream in, the far end replaced by the export's baked white-noise tensors
    (carried by the fixture), against what the reference wrapper produced."""
    g = golden("dfsmn_near_only")
    consts = (g["pow_far"].astype(np.float32), g["far_comp"].astype(np.float32))
    sess = dfsmn.DfsmnSession(weights.dfsmn_synthetic(1234), near_only=consts)
    assert [m.name for m in sess.get_inputs()] == ["audio"]
    out = sess.run(None, {"audio": g["near"]})[0]
    assert out.shape == (51,)
    np.testing.assert_allclose(out, g["vad"], rtol=0, atol=1e-4)
    # batched windows give the same answer, and the two-stream path of the same engine is untouched
    three = np.repeat(g["near"].reshape(1, -1), 3, axis=0)
    v3 = sess.engine.run(three, None).cpu().numpy()
    assert np.array_equal(v3[0], v3[2]) and np.array_equal(v3[0], out)
    with pytest.raises(ValueError):
        dfsmn.DfsmnEngine(weights.dfsmn_synthetic(1234)).run(three, None)


def test_full_size_config5_properties():
    """BASELINE config 5 (DFSMN half) at FULL size: 2048 clip pairs x 10 s = 30 720 windows of 16 001 samples in one call
    (32 sub-batches of 960 windows).  Bitwise batch-position invariance, finite scores in [0, 1], and agreement with the
    oracle on the windows of one clip pair."""
    import time
    from oracle import dfsmn as od
    w = weights.dfsmn_synthetic(1234)
    eng = dfsmn.DfsmnEngine(w, sub_batch=960)
    lb, stride = eng.grid()
    W = 15
    n = (W - 1) * stride + eng.L
    base_n, base_f = weights.burst_clips(16, n, seed=61), weights.burst_clips(16, n, seed=62)
    near = torch.from_numpy(base_n).cuda().repeat(128, 1)            # 2048 pairs, pair i == pair i % 16
    far = torch.from_numpy(base_f).cuda().repeat(128, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    vad = eng.run(near, far, W, stride)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert vad.shape == (2048 * W, 51) and bool(torch.isfinite(vad).all())
    assert float(vad.min()) >= 0.0 and float(vad.max()) <= 1.0
    v = vad.view(128, 16, W, 51)
    assert torch.equal(v[0], v[1]) and torch.equal(v[0], v[77]) and torch.equal(v[0], v[127])
    ow = {k: torch.from_numpy(np.ascontiguousarray(x)) for k, x in w.items()}
    ow["mask.shift"] = ow["mask.shift"] + torch.log(torch.tensor(32768.0 ** 2))
    fe = od.Frontend()
    for k in (0, 7, 14):
        a = torch.from_numpy(base_n[5, k * stride:k * stride + eng.L].copy()).reshape(1, 1, -1)
        f = torch.from_numpy(base_f[5, k * stride:k * stride + eng.L].copy()).reshape(1, 1, -1)
        want, _ = od.forward(fe, ow, a, f, weights.DFSMN_MASK["layers"])
        np.testing.assert_allclose(v[3, 5, k].cpu().numpy(), want.numpy(), rtol=0, atol=1e-4)
    print(f"DFSMN config-5 pass: {dt:.2f} s for 2048 x 10 s pairs ({2048 * 313 / dt / 1e3:.0f} k 512-hop frames/s)")


def test_vote_kernel_with_asymmetric_thresholds_replays_the_reference_loop(golden):
    """vadx_dfsmn_vote on the float scores the REFERENCE loop was run on, SPEAKING_SCORE != SILENCE_SCORE (0.7 / 0.3, 0.3 / 0.7, 0.9 / 0.1,
    0.6 / 0.6, 0.2 / 0.4; scores sitting exactly on float32(threshold) included): the `saved` flags bit for bit
    (DFSMN/near_and_far_end_audio/Inference_DFSMN_VAD_ONNX.py:231-273)."""
    import ctypes as C
    from vadx import _lib
    g = golden("hostloop_thresholds")
    for c, (spk, sil) in enumerate(g["pairs"]):
        scores = torch.from_numpy(g[f"dfsmn_scores_{c}"]).cuda().contiguous()         # [W, 51]
        W, Tn, lb = scores.shape[0], scores.shape[1], 15
        rep = scores.unsqueeze(0).repeat(3, 1, 1).contiguous()                          # three identical clips: per-clip independence
        flags = torch.empty((3, W * (Tn - lb) + lb), dtype=torch.uint8, device="cuda")
        _lib.check(_lib.lib().vadx_dfsmn_vote(rep.data_ptr(), 3, W, Tn, lb, float(spk), float(sil), flags.data_ptr(), _lib.stream_ptr()))
        got = flags.cpu().numpy().astype(bool)
        for b in range(3):
            assert np.array_equal(got[b], g[f"dfsmn_saved_{c}"]), (c, b)


def test_whole_clip_pair_segments_with_asymmetric_thresholds():
    eng = dfsmn.DfsmnEngine(weights.dfsmn_synthetic(1234))
    w = {k: T(v) for k, v in weights.dfsmn_synthetic(1234).items()}
    w["mask.shift"] = w["mask.shift"] + torch.log(torch.tensor(32768 ** 2, dtype=torch.float32))
    fe = od.Frontend()
    near, far = weights.burst_clips(1, 30000, seed=13), weights.burst_clips(1, 30000, seed=14)
    nz1, nz2 = np.random.default_rng(15).standard_normal((1, 20000)), np.random.default_rng(16).standard_normal((1, 20000))
    # thresholds placed inside the synthetic net's score range so that both branches fire
    vad = eng.run(near[:, :16001], far[:, :16001]).cpu().numpy()[0]
    hi, lo = float(np.quantile(vad, 0.7)), float(np.quantile(vad, 0.3))
    got = eng.detect(near, far, nz1, nz2, speaking_score=hi, silence_score=lo)
    want, _ = od.run_clip(fe, w, near[0], far[0], nz1[0], nz2[0], weights.DFSMN_MASK["layers"], speaking=hi, silence_score=lo)
    assert got[0] == want


def test_packed_frame_layout_is_bitwise_the_unpacked_one(wts):
    """The ICCRN on the packed layout (windows 104 frames apart, 6.5 tiles each: a tile can hold the tail of one window and the head
    of the next) against windows on 7 tiles of their own: every per-frame kernel is independent per column and the time-axis LSTMs
    walk a window's frames by (chunk * stride + t), so the two spectra are equal bit for bit -- odd and even window counts
    (the last tile half empty or not), and more tiles than workgroups."""
    w, _ = wts
    net = dfsmn.Iccrn(w)
    for chunks in (1, 3, 40):
        g = torch.Generator().manual_seed(chunks)
        x = torch.randn(chunks, 4, 160, 101, generator=g) * 0.5
        x4 = dfsmn.to_ft(torch, x, net.device)
        y_p = dfsmn.from_ft(net.forward(x4, chunks, 101, pack=True), chunks)
        y_u = dfsmn.from_ft(net.forward(x4, chunks, 101, pack=False), chunks)
        assert torch.isfinite(y_u).all() and torch.equal(y_p, y_u), chunks


def test_whole_config_decision_record():
    """BASELINE config 5 (DFSMN half) at full size: silence flags of the default arithmetic (bf16 x 3 in these kernels) against float32 MFMAs."""
    import decision_records
    r = decision_records.dfsmn_c5(torch, torch.device("cuda", 0))
    print(r)
    assert r["unexcused"] == 0, r
