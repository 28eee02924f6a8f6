"""Weight containers for the VAD nets + seeded synthetic initialisers.

No trained checkpoint ships with the reference (every script points at
/home/DakeQQ/Downloads/..., e.g. FSMN/Export_FSMN_VAD.py:14-15), so kernels are validated and
benchmarked on SEEDED SYNTHETIC weights of the real architecture; a real checkpoint drops in
through the same dict keys (`load_*` helpers take a state-dict-like mapping of numpy arrays).

All tensors are float32 numpy arrays on the host; `Session` objects upload them once.
"""
from __future__ import annotations

import numpy as np


def _rng(seed, tag):
    # independent stream per tensor so that adding a tensor never shifts the others
    return np.random.default_rng([seed, sum(ord(c) * (i + 1) for i, c in enumerate(tag)) & 0x7FFFFFFF])


def _normal(seed, tag, shape, std):
    return (_rng(seed, tag).standard_normal(shape) * std).astype(np.float32)


# --------------------------------------------------------------------------- Silero v5 (16 kHz)
SILERO_ENC = ((129, 128, 1), (128, 64, 2), (64, 64, 2), (64, 128, 1))   # (c_in, c_out, stride), k=3, pad=1


def silero_stft_basis():
    """[258,256] hann-windowed real DFT basis, filter 256 / hop 128 (published silero-vad v5
    `forward_basis_buffer`): rows 0..128 = cos, rows 129..257 = -sin... sign of the imaginary
    part does not matter to |.|; we use +imag(fft(eye)) like the published STFT module."""
    n = 256
    k = np.arange(129, dtype=np.float64)[:, None]
    t = np.arange(n, dtype=np.float64)[None, :]
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n, dtype=np.float64) / n)     # periodic hann
    ang = 2.0 * np.pi * k * t / n
    basis = np.concatenate([np.cos(ang), -np.sin(ang)], axis=0) * win[None, :]
    return basis.astype(np.float32)


def silero_synthetic(seed=1234):
    """Seeded weights of the silero-vad v5 architecture (see oracle/silero.py header)."""
    w = {"stft_basis": silero_stft_basis()}
    for i, (ci, co, _s) in enumerate(SILERO_ENC):
        w[f"enc{i}_w"] = _normal(seed, f"enc{i}_w", (co, ci, 3), 1.6 / np.sqrt(3 * ci))
        w[f"enc{i}_b"] = _normal(seed, f"enc{i}_b", (co,), 0.05)
    # the first conv sees |STFT| of +-1-scaled audio (speech-level bursts ~0.1 rms -> |X| ~ 1)
    w["enc0_w"] *= np.float32(2.0)
    h = 128
    w["lstm_w_ih"] = _normal(seed, "lstm_w_ih", (4 * h, h), 1.0 / np.sqrt(h))
    w["lstm_w_hh"] = _normal(seed, "lstm_w_hh", (4 * h, h), 1.0 / np.sqrt(h))
    w["lstm_b_ih"] = _normal(seed, "lstm_b_ih", (4 * h,), 0.1)
    w["lstm_b_hh"] = _normal(seed, "lstm_b_hh", (4 * h,), 0.1)
    # decoder: positive taps + negative bias so louder input -> higher speech probability,
    # which makes the synthetic burst clips cross both the 0.5 and 0.35 thresholds.
    w["dec_w"] = np.abs(_normal(seed, "dec_w", (h,), 0.25)).astype(np.float32)
    w["dec_b"] = np.array([-2.0], dtype=np.float32)
    return w


def silero_check(w):
    """Shape/dtype validation of a Silero weight dict (raises ValueError)."""
    want = {"stft_basis": (258, 256), "lstm_w_ih": (512, 128), "lstm_w_hh": (512, 128),
            "lstm_b_ih": (512,), "lstm_b_hh": (512,), "dec_w": (128,), "dec_b": (1,)}
    for i, (ci, co, _s) in enumerate(SILERO_ENC):
        want[f"enc{i}_w"] = (co, ci, 3)
        want[f"enc{i}_b"] = (co,)
    for k, shp in want.items():
        if k not in w:
            raise ValueError(f"silero weights: missing '{k}'")
        a = np.asarray(w[k])
        if tuple(a.shape) != shp or a.dtype != np.float32:
            raise ValueError(f"silero weights: '{k}' must be float32{shp}, got {a.dtype}{a.shape}")
    return True


# --------------------------------------------------------------------------- synthetic audio
def burst_clips(batch, num_samples, seed=1234, loud=3000.0, quiet=30.0, sample_rate=16000):
    """int16 [batch, num_samples]: 0.5-2 s segments alternating N(0,loud) / N(0,quiet) so every
    branch of the decision state machines fires (SURVEY §8d synthetic-input recipe)."""
    rng = np.random.default_rng(seed)
    out = np.empty((batch, num_samples), dtype=np.int16)
    for b in range(batch):
        pos = 0
        is_loud = bool(rng.integers(0, 2))
        while pos < num_samples:
            seg = int(rng.uniform(0.5, 2.0) * sample_rate)
            seg = min(seg, num_samples - pos)
            sigma = loud if is_loud else quiet
            x = rng.standard_normal(seg) * sigma
            out[b, pos:pos + seg] = np.clip(x, -32768, 32767).astype(np.int16)
            pos += seg
            is_loud = not is_loud
    return out


# --------------------------------------------------------------------------- FSMN-VAD (FunASR)
# Layer widths of speech_fsmn_vad_zh-cn-16k-common (external FunASR config; the cache shape
# [1,128,19,1] and input 400 are pinned in-tree: FSMN/Export_FSMN_VAD.py:116, :82-86).
FSMN_DIMS = dict(input_dim=400, input_affine_dim=140, fsmn_layers=4, linear_dim=250, proj_dim=128,
                 lorder=20, output_affine_dim=140, output_dim=248)


_FSMN_SIL_BIAS = {1234: 7.2, 7: 8.4}


def fsmn_synthetic(seed=1234, dims=None):
    d = dict(FSMN_DIMS if dims is None else dims)
    D, A, L, P, K, A2, O = (d["input_dim"], d["input_affine_dim"], d["linear_dim"], d["proj_dim"],
                            d["lorder"], d["output_affine_dim"], d["output_dim"])
    w = {
        "in1_w": _normal(seed, "in1_w", (A, D), 1.0 / np.sqrt(D)), "in1_b": _normal(seed, "in1_b", (A,), 0.05),
        "in2_w": _normal(seed, "in2_w", (L, A), 1.4 / np.sqrt(A)), "in2_b": _normal(seed, "in2_b", (L,), 0.05),
        "out1_w": _normal(seed, "out1_w", (A2, L), 1.0 / np.sqrt(L)), "out1_b": _normal(seed, "out1_b", (A2,), 0.05),
        "out2_w": _normal(seed, "out2_w", (O, A2), 2.0 / np.sqrt(A2)), "out2_b": _normal(seed, "out2_b", (O,), 0.05),
    }
    for l in range(d["fsmn_layers"]):
        w[f"l{l}_lin_w"] = _normal(seed, f"l{l}_lin_w", (P, L), 1.0 / np.sqrt(L))
        w[f"l{l}_fir_w"] = _normal(seed, f"l{l}_fir_w", (P, K), 0.6 / np.sqrt(K))
        w[f"l{l}_aff_w"] = _normal(seed, f"l{l}_aff_w", (L, P), 1.4 / np.sqrt(P))
        w[f"l{l}_aff_b"] = _normal(seed, f"l{l}_aff_b", (L,), 0.05)
    # class 0 = silence: bias it so P(silence) hovers around 0.5 and the score gate toggles
    # (offsets found offline for the seeds the tests/bench use; other seeds get the generic one)
    w["out2_b"][0] += np.float32(5.5 + _FSMN_SIL_BIAS.get(seed, 7.0))
    # CMVN of the LFR'd log-mel (log of int16-scale power ~ 10..25): centre and scale it
    w["cmvn_means"] = (-(14.0 + _rng(seed, "cmvn_means").standard_normal(D))).astype(np.float32)
    w["cmvn_vars"] = (0.25 + 0.02 * _rng(seed, "cmvn_vars").standard_normal(D)).astype(np.float32)
    return w


# --------------------------------------------------------------------------- FireRedVAD DFSMN
# R/M/H/P/N1/S1/N2/S2/odim come from the checkpoint's `args` (FireRedVAD/Export_FireRedVAD.py:336-337);
# these are the placeholders SURVEY Appendix B uses.
FIRERED_CFG = dict(idim=80, R=8, M=1, H=256, P=128, N1=20, S1=1, N2=20, S2=1, odim=1)


_FIRERED_OUT_CALIB = {1234: (1.21, -3.97), 7: (-7.55, -9.13)}


def firered_synthetic(seed=1234, cfg=None):
    c = dict(FIRERED_CFG if cfg is None else cfg)
    D, R, M, H, P = c["idim"], c["R"], c["M"], c["H"], c["P"]
    w = {"cfg": c}
    # CMVN is folded into fc1 by the reference loader (:350-360): log-mel of int16-scale audio
    # sits around 10..25, so the synthetic fc1 carries a centring bias.
    w["fc1_w"] = _normal(seed, "fc1_w", (H, D), 0.3 / np.sqrt(D))
    w["fc1_b"] = (-14.0 * w["fc1_w"].sum(axis=1) + _normal(seed, "fc1_b", (H,), 0.05)).astype(np.float32)
    w["fc2_w"] = _normal(seed, "fc2_w", (P, H), 1.4 / np.sqrt(H))
    w["fc2_b"] = _normal(seed, "fc2_b", (P,), 0.05)
    for r in range(R):
        w[f"fsmn{r}_lb"] = _normal(seed, f"fsmn{r}_lb", (P, c["N1"]), 0.5 / np.sqrt(c["N1"]))
        if c["N2"] > 0:
            w[f"fsmn{r}_la"] = _normal(seed, f"fsmn{r}_la", (P, c["N2"]), 0.5 / np.sqrt(c["N2"]))
        if r > 0:
            w[f"blk{r}_fc1_w"] = _normal(seed, f"blk{r}_fc1_w", (H, P), 1.0 / np.sqrt(P))
            w[f"blk{r}_fc1_b"] = _normal(seed, f"blk{r}_fc1_b", (H,), 0.05)
            w[f"blk{r}_fc2_w"] = _normal(seed, f"blk{r}_fc2_w", (P, H), 0.5 / np.sqrt(H))
    for m in range(M):
        w[f"dnn{m}_w"] = _normal(seed, f"dnn{m}_w", (H, P if m == 0 else H), 1.0 / np.sqrt(P if m == 0 else H))
        w[f"dnn{m}_b"] = _normal(seed, f"dnn{m}_b", (H,), 0.05)
    w["out_w"] = _normal(seed, "out_w", (c["odim"], H), 1.0 / np.sqrt(H))
    w["out_b"] = _normal(seed, "out_b", (c["odim"],), 0.05)
    # output affine (scale, shift) found offline so loud bursts sit near logit +2 and quiet
    # stretches near -2 for the seeds the tests/bench use (random nets have a random polarity)
    s, t = _FIRERED_OUT_CALIB.get(seed, (1.0, 0.0))
    w["out_w"] = (w["out_w"] * np.float32(s)).astype(np.float32)
    w["out_b"] = (w["out_b"] * np.float32(s) + np.float32(t)).astype(np.float32)
    return w


# --------------------------------------------------------------------------- MarbleNet (NeMo 3x2x64, 20 ms)
MARBLENET_BLOCKS = (  # (filters, repeat, kernel, stride, dilation, residual, separable)
    (128, 1, 11, 2, 1, False, True), (64, 2, 13, 1, 1, True, True), (64, 2, 15, 1, 1, True, True),
    (64, 2, 17, 1, 1, True, True), (128, 1, 29, 1, 2, False, True), (128, 1, 1, 1, 1, False, False))
MARBLENET_BN_EPS = 1e-3


def marblenet_synthetic(seed=1234):
    """Unfolded (conv + BatchNorm statistics) weights of the published MarbleNet 3x2x64 layout."""
    w = {}
    cin = 80

    def bn(prefix, c):
        w[prefix + "_gamma"] = (1.0 + 0.1 * _rng(seed, prefix + "g").standard_normal(c)).astype(np.float32)
        w[prefix + "_beta"] = _normal(seed, prefix + "b", (c,), 0.1)
        w[prefix + "_mean"] = _normal(seed, prefix + "m", (c,), 0.2)
        w[prefix + "_var"] = (0.5 + _rng(seed, prefix + "v").uniform(0, 1, c)).astype(np.float32)

    for bi, (filt, rep, k, _s, _d, residual, sep) in enumerate(MARBLENET_BLOCKS):
        block_cin = cin
        for r in range(rep):
            p = f"b{bi}r{r}"
            if sep:
                w[p + "_dw"] = _normal(seed, p + "_dw", (cin, k), 1.0 / np.sqrt(k))
            w[p + "_pw"] = _normal(seed, p + "_pw", (filt, cin), 1.0 / np.sqrt(cin))
            bn(p, filt)
            cin = filt
        if residual:
            w[f"b{bi}res_pw"] = _normal(seed, f"b{bi}res_pw", (filt, block_cin), 0.7 / np.sqrt(block_cin))
            bn(f"b{bi}res", filt)
    # log-mel of 1/32768-scaled audio sits around -12..-3: centre the first depthwise/pointwise pair
    w["b0r0_mean"] = (w["b0r0_mean"] + (-8.0 * (w["b0r0_pw"] * w["b0r0_dw"].sum(axis=1)[None, :]).sum(axis=1))).astype(np.float32)
    w["dec_w"] = _normal(seed, f"dec_w_mb{_MARBLENET_DEC_TAG.get(seed, 0)}", (2, 128), 1.0 / np.sqrt(128))
    w["dec_b"] = _normal(seed, "dec_b_mb", (2,), 0.05)
    s, t = _MARBLENET_DEC_CALIB.get(seed, (1.0, 0.0))
    w["dec_w"][1] = w["dec_w"][0] + (w["dec_w"][1] - w["dec_w"][0]) * np.float32(s)
    w["dec_b"][1] = w["dec_b"][0] + (w["dec_b"][1] - w["dec_b"][0]) * np.float32(s) + np.float32(t)
    return w


# decoder draw + output affine found offline so that loud bursts / quiet stretches straddle 0.5
_MARBLENET_DEC_TAG = {1234: 16, 7: 0}
_MARBLENET_DEC_CALIB = {1234: (5.6, -0.26), 7: (5.9, 9.9)}


def fold_bn(conv_w, conv_b, gamma, beta, mean, var, eps):
    """BatchNorm (eval) folded into the preceding conv: W' = W*g/sqrt(v+eps), b' = (b-mean)*g/sqrt(v+eps)+beta
    (Export_NVIDIA_MarbleNet_VAD.py:58-105).  float32 numpy, same op order as the reference."""
    scale = (gamma * (np.float32(1.0) / np.sqrt(var + np.float32(eps)))).astype(np.float32)
    new_w = (conv_w * scale.reshape((-1,) + (1,) * (conv_w.ndim - 1))).astype(np.float32)
    base = (conv_b - mean) if conv_b is not None else (-mean)
    return new_w, (base * scale + beta).astype(np.float32)


# --------------------------------------------------------------------------- DFSMN near+far (SDAEC ICCRN + mask-net)
DFSMN_MASK = dict(hidden=128, fsmn_hidden=256, layers=4, lorder=20)     # modelscope container dims are external: stand-in


def _lstm_keys(prefix, in_dim, hid, layers, bi):
    out = {}
    for l in range(layers):
        d_in = in_dim if l == 0 else hid * (2 if bi else 1)
        for suf in ([""] + (["_reverse"] if bi else [])):
            out[f"{prefix}weight_ih_l{l}{suf}"] = (4 * hid, d_in)
            out[f"{prefix}weight_hh_l{l}{suf}"] = (4 * hid, hid)
            out[f"{prefix}bias_ih_l{l}{suf}"] = (4 * hid,)
            out[f"{prefix}bias_hh_l{l}{suf}"] = (4 * hid,)
    return out


def dfsmn_shapes(ch=20, mask=None):
    """state_dict names -> shapes of AlphaPredictor ('alpha.'), ICCRN NET ('iccrn.') and the mask-net
    stand-in ('mask.') (DFSMN/near_and_far_end_audio/Export_DFSMN_VAD.py:65-284)."""
    m = dict(DFSMN_MASK if mask is None else mask)
    s = {"alpha.linear1.weight": (1, 2), "alpha.linear1.bias": (1,), "alpha.linear2.weight": (1, 10), "alpha.linear2.bias": (1,)}
    ic = {}
    ic.update(_lstm_keys("in_ch_lstm.lstm2.", 4, ch, 1, True))
    ic.update({"in_ch_lstm.linear.weight": (ch, 2 * ch), "in_ch_lstm.linear.bias": (ch,),
               "in_conv.weight": (ch, 4 + ch, 1, 1), "in_conv.bias": (ch,)})
    for name, cin in (("cfb_e1", ch), ("cfb_e2", ch), ("cfb_e3", ch), ("cfb_e4", ch), ("cfb_e5", ch), ("cfb_d5", ch),
                      ("cfb_d4", 2 * ch), ("cfb_d3", 2 * ch), ("cfb_d2", 2 * ch), ("cfb_d1", 2 * ch)):
        ic.update({f"{name}.conv_gate.weight": (ch, cin, 1, 1), f"{name}.conv_gate.bias": (ch,),
                   f"{name}.conv_input.weight": (ch, cin, 1, 1), f"{name}.conv_input.bias": (ch,),
                   f"{name}.conv.weight": (ch, ch, 3, 1), f"{name}.conv.bias": (ch,),
                   f"{name}.LN0.w": (1, cin, 160, 1), f"{name}.LN0.b": (1, cin, 160, 1),
                   f"{name}.LN1.w": (1, ch, 160, 1), f"{name}.LN1.b": (1, ch, 160, 1),
                   f"{name}.LN2.w": (1, ch, 160, 1), f"{name}.LN2.b": (1, ch, 160, 1),
                   f"{name}.ceps_unit.LN.w": (1, 2 * ch, 81, 1), f"{name}.ceps_unit.LN.b": (1, 2 * ch, 81, 1),
                   f"{name}.ceps_unit.ch_lstm_f.linear.weight": (2 * ch, 2 * ch), f"{name}.ceps_unit.ch_lstm_f.linear.bias": (2 * ch,)})
        ic.update(_lstm_keys(f"{name}.ceps_unit.ch_lstm_f.lstm2.", 2 * ch, ch, 1, True))
    ic.update({"ln.w": (1, ch, 160, 1), "ln.b": (1, ch, 160, 1)})
    ic.update(_lstm_keys("ch_lstm.lstm2.", ch, 2 * ch, 2, False))
    ic.update({"ch_lstm.linear.weight": (ch, 2 * ch), "ch_lstm.linear.bias": (ch,)})
    ic.update(_lstm_keys("out_ch_lstm.lstm2.", 2 * ch, ch, 1, False))
    ic.update({"out_ch_lstm.linear.weight": (2 * ch, ch), "out_ch_lstm.linear.bias": (2 * ch,),
               "out_conv.weight": (2, 3 * ch, 1, 1), "out_conv.bias": (2,)})
    s.update({"iccrn." + k: v for k, v in ic.items()})
    H, H2 = m["hidden"], m["fsmn_hidden"]
    s.update({"mask.shift": (240,), "mask.scale": (240,), "mask.linear1.weight": (H, 240), "mask.linear1.bias": (H,),
              "mask.linear3.weight": (1, H), "mask.linear3.bias": (1,)})
    for i in range(m["layers"]):
        s.update({f"mask.deepfsmn.{i}.linear.weight": (H2, H), f"mask.deepfsmn.{i}.linear.bias": (H2,),
                  f"mask.deepfsmn.{i}.project.weight": (H, H2), f"mask.deepfsmn.{i}.conv1.weight": (H, 1, m["lorder"], 1)})
    return s


def dfsmn_synthetic(seed=1234, mask=None):
    w = {}
    for k, shp in dfsmn_shapes(mask=mask).items():
        if k.endswith(".w"):                                     # LayerNorm gain
            w[k] = (1.0 + 0.1 * _rng(seed, k).standard_normal(shp)).astype(np.float32)
        elif k.endswith(".b") and ".LN" in k or k.endswith("ln.b"):
            w[k] = (_rng(seed, k).uniform(0, 1, shp) * 1e-4).astype(np.float32)
        elif "bias" in k:
            w[k] = _normal(seed, k, shp, 0.05)
        elif k == "mask.shift":
            w[k] = (13.0 + 0.5 * _rng(seed, k).standard_normal(shp)).astype(np.float32)   # + log(32768^2) is added at load
        elif k == "mask.scale":
            w[k] = (0.3 + 0.02 * _rng(seed, k).standard_normal(shp)).astype(np.float32)
        else:
            fan_in = int(np.prod(shp[1:])) if len(shp) > 1 else shp[0]
            w[k] = _normal(seed, k, shp, 1.0 / np.sqrt(max(fan_in, 1)))
    w["alpha.linear1.weight"] = np.array([[0.6, -0.3]], np.float32)
    w["alpha.linear2.weight"] = np.full((1, 10), 0.1, np.float32)
    s, t = _DFSMN_OUT_CALIB.get(seed, (1.0, 0.0))
    w["mask.linear3.weight"] = (w["mask.linear3.weight"] * np.float32(s)).astype(np.float32)
    w["mask.linear3.bias"] = (w["mask.linear3.bias"] * np.float32(s) + np.float32(t)).astype(np.float32)
    return w


_DFSMN_OUT_CALIB = {}


def dfsmn_near_only_constants(seed=1234, frames=200, k=10, n_bins=160):
    """Stand-in for the two white-noise tensors the near-end-only DFSMN export bakes into its graph
    (DFSMN/only_near_end_audio/Export_DFSMN_VAD.py:309-310): float16 randn, squared for the power term.
    -> (pow_far [n_bins, frames, k], far_comp [2, n_bins, frames]) as float32 arrays holding float16 values."""
    rng = _rng(seed, "dfsmn_near_only")
    pow_far = np.square(rng.standard_normal((n_bins, frames, k)).astype(np.float16)).astype(np.float16)
    far_comp = rng.standard_normal((2, n_bins, frames)).astype(np.float16)
    return pow_far.astype(np.float32), far_comp.astype(np.float32)
