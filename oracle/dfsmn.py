"""Oracle: DFSMN near+far path (SURVEY §8 rows a18, a19, a20 + the DFSMN half of a9).

TEST INFRASTRUCTURE -- CPU restatement in torch float32 (batch 1 like the reference; the ICCRN's
views hard-code batch 1).  Weight dict keys are the reference modules' state_dict names so a real
SDAEC `ICCRN.ckpt` / `alpha.ckpt` drops in unchanged.

In tree and pinned by fixtures (tests/golden/dfsmn_forward.npz): AlphaPredictor, ICCRN `NET`
(CFB, CepsUnit, LayerNorm, CH_LSTM_F, CH_LSTM_T), the `DFSMN_VAD` wrapper, `UniDeepFsmn.compute1`.
NOT in tree: the modelscope mask-net container (`linear1 -> relu -> deepfsmn -> linear3`, dims and
`preprocessor.feature.shift/scale` external) -- a stand-in of that shape is used.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import mel as omel
from . import postproc
from . import stft as ostft

NFFT_B, HOP_B = 319, 160
NFFT_A, WIN_A, HOP_A = 1024, 640, 320
ALPHA_K = 10
F_BINS = NFFT_B // 2 + 1          # 160
CEPS_N = F_BINS                   # CepsUnit n_fft = NFFT_B // 2 + 1 = 160
CEPS_F = CEPS_N // 2 + 1          # 81


def _lstm(x, w, prefix, in_dim, hid, layers=1, bi=False):
    """torch.nn.LSTM(batch_first=True) forward with weights from dict (state_dict names)."""
    m = torch.nn.LSTM(in_dim, hid, num_layers=layers, batch_first=True, bidirectional=bi)
    sd = {k[len(prefix):]: v for k, v in w.items() if k.startswith(prefix)}
    m.load_state_dict(sd, strict=True)
    m.eval()
    with torch.no_grad():
        return m(x)[0]


def layer_norm(x, w, p):
    """ref: LayerNorm.forward, Export_DFSMN_VAD.py:163-167 (unbiased std over (C,F), eps outside)."""
    mean = x.mean([1, 2], keepdim=True)
    std = x.std([1, 2], keepdim=True)
    return (x - mean) / (std + 1e-6) * w[p + ".w"] + w[p + ".b"]


def ch_lstm_f(x, w, p, in_ch, feat, out_ch, f):
    """bi-LSTM across the frequency axis. ref: CH_LSTM_F :270-284."""
    y = x.permute(0, 3, 2, 1).contiguous().view(-1, f, in_ch)
    y = _lstm(y, w, p + ".lstm2.", in_ch, feat, 1, True)
    y = F.linear(y, w[p + ".linear.weight"], w[p + ".linear.bias"])
    return y.view(1, -1, f, out_ch).permute(0, 3, 2, 1).contiguous()


def ch_lstm_t(x, w, p, in_ch, feat, out_ch, layers, f=F_BINS):
    """LSTM across time, one sequence per bin. ref: CH_LSTM_T :252-267."""
    y = x.permute(0, 2, 3, 1).contiguous().view(f, -1, in_ch)
    y = _lstm(y, w, p + ".lstm2.", in_ch, feat, layers, False)
    y = F.linear(y, w[p + ".linear.weight"], w[p + ".linear.bias"])
    return y.view(1, f, -1, out_ch).permute(0, 3, 1, 2).contiguous()


class CepsTables:
    """Buffers of CepsUnit.__init__ (:104-130): rectangular-window real DFT of length 160 along
    the frequency axis and its pinv-based inverse."""

    def __init__(self):
        n, half = CEPS_N, CEPS_N // 2
        t = torch.arange(n, dtype=torch.float32).unsqueeze(0)
        f = torch.arange(half + 1, dtype=torch.float32).unsqueeze(1)
        omega = 2 * torch.pi * f * t / n
        self.cos_k = torch.cos(omega).unsqueeze(1)
        self.sin_k = (-torch.sin(omega)).unsqueeze(1)
        fb = torch.fft.fft(torch.eye(n, dtype=torch.float32))
        basis = torch.vstack([torch.real(fb[:half + 1]), torch.imag(fb[:half + 1])]).float()
        self.inv_basis = torch.linalg.pinv(basis).T.unsqueeze(1)          # [162,1,160] (window = ones)


def ceps_unit(x0, w, p, ch, tb):
    """ref: CepsUnit.forward :132-154."""
    xr = x0.permute(0, 1, 3, 2).contiguous().view(-1, 1, CEPS_N)
    re = F.conv1d(xr, tb.cos_k, stride=CEPS_N).view(1, ch, -1, CEPS_F).permute(0, 1, 3, 2).contiguous()
    im = F.conv1d(xr, tb.sin_k, stride=CEPS_N).view(1, ch, -1, CEPS_F).permute(0, 1, 3, 2).contiguous()
    li = torch.cat([re, im], 1)
    lo = ch_lstm_f(layer_norm(li, w, p + ".LN"), w, p + ".ch_lstm_f", ch * 2, ch, ch * 2, CEPS_F)
    pr, pi = lo[:, :ch], lo[:, ch:]
    o_re = pr * re - pi * im
    o_im = pr * im + pi * re
    a = o_re.permute(0, 1, 3, 2).contiguous().view(-1, CEPS_F, 1)
    b = o_im.permute(0, 1, 3, 2).contiguous().view(-1, CEPS_F, 1)
    inv = F.conv_transpose1d(torch.cat((a, b), dim=1), tb.inv_basis, stride=CEPS_N)
    return inv.view(1, ch, -1, CEPS_N).permute(0, 1, 3, 2).contiguous()


def cfb(x, w, p, tb, ch=20):
    """ref: CFB.forward :87-93."""
    g = torch.sigmoid(F.conv2d(layer_norm(x, w, p + ".LN0"), w[p + ".conv_gate.weight"], w[p + ".conv_gate.bias"]))
    xi = F.conv2d(x, w[p + ".conv_input.weight"], w[p + ".conv_input.bias"])
    gx = g * xi
    y = F.conv2d(layer_norm(gx, w, p + ".LN1"), w[p + ".conv.weight"], w[p + ".conv.bias"], padding=(1, 0))
    return y + ceps_unit(layer_norm(xi - gx, w, p + ".LN2"), w, p + ".ceps_unit", ch, tb)


class IccrnTables:
    """ISTFT buffers of NET.__init__ (:183-207)."""

    def __init__(self, max_frames=200):
        n, hop, half = NFFT_B, HOP_B, NFFT_B // 2
        self.window = torch.hamming_window(n)
        fe = torch.fft.fft(torch.eye(n, dtype=torch.float32))
        fb = torch.vstack([torch.real(fe[:half + 1]), torch.imag(fe[:half + 1])]).float()
        pinv_t = torch.linalg.pinv((fb * n) / hop).T
        self.inverse_basis = pinv_t.unsqueeze(1) * self.window.view(1, 1, -1)        # [320,1,319]
        out_len = (max_frames - 1) * hop + n
        ws = torch.zeros(out_len, dtype=torch.float32)
        wsq = self.window ** 2
        for i in range(max_frames):
            s = i * hop
            L = min(n, out_len - s)
            if L <= 0:
                break
            ws[s:s + L] += wsq[:L]
        self.window_sum_inv = n / (ws * hop + 1e-6)
        self.ceps = CepsTables()


def iccrn(x, w, tb, ch=20):
    """[1,4,160,T] -> (aec waveform [1,1,L], L). ref: NET.forward :226-249 (+ istft :220-224)."""
    c = tb.ceps
    e0 = ch_lstm_f(x, w, "in_ch_lstm", 4, ch, ch, F_BINS)
    e0 = F.conv2d(torch.cat([e0, x], 1), w["in_conv.weight"], w["in_conv.bias"])
    e1 = cfb(e0, w, "cfb_e1", c)
    e2 = cfb(e1, w, "cfb_e2", c)
    e3 = cfb(e2, w, "cfb_e3", c)
    e4 = cfb(e3, w, "cfb_e4", c)
    e5 = cfb(e4, w, "cfb_e5", c)
    lo = ch_lstm_t(layer_norm(e5, w, "ln"), w, "ch_lstm", ch, ch * 2, ch, 2)
    d5 = cfb(e5 * lo, w, "cfb_d5", c)
    d4 = cfb(torch.cat([e4, d5], 1), w, "cfb_d4", c)
    d3 = cfb(torch.cat([e3, d4], 1), w, "cfb_d3", c)
    d2 = cfb(torch.cat([e2, d3], 1), w, "cfb_d2", c)
    d1 = cfb(torch.cat([e1, d2], 1), w, "cfb_d1", c)
    d0 = ch_lstm_t(torch.cat([e0, d1], 1), w, "out_ch_lstm", 2 * ch, ch, 2 * ch, 1)
    out = F.conv2d(torch.cat([d0, d1], 1), w["out_conv.weight"], w["out_conv.bias"])
    half = NFFT_B // 2
    inv = F.conv_transpose1d(out.reshape(1, 2 * F_BINS, -1), tb.inverse_basis, stride=HOP_B)
    e = inv.size(-1) - half
    return inv[..., half:e] * tb.window_sum_inv[half:e], e - half


def uni_deep_fsmn(x, w, p, lorder=20):
    """x [1,T,H]. ref: UniDeepFsmn.compute1, uni_deep_fsmn.py:311-329 (dilation 1, skip_connect)."""
    h = F.relu(F.linear(x, w[p + ".linear.weight"], w[p + ".linear.bias"]))
    pr = F.linear(h, w[p + ".project.weight"]).transpose(1, 2).unsqueeze(-1)          # [1,H,T,1]
    y = torch.cat([torch.zeros(1, pr.shape[1], lorder - 1, 1), pr], dim=-2)
    out = F.conv2d(y, w[p + ".conv1.weight"], groups=pr.shape[1]) + pr
    return x + out.transpose(1, 2).squeeze(-1)


class Frontend:
    def __init__(self, max_frames=200):
        wb = ostft.padded_window(NFFT_B, NFFT_B, "hamming", "v1b")
        self.cos_b, self.sin_b = ostft.dft_tables(NFFT_B, wb, "v1b")
        wa = ostft.padded_window(WIN_A, NFFT_A, "hamming", "v1b")
        self.cos_a, self.sin_a = ostft.dft_tables(NFFT_A, wa, "v1b")
        self.fbank = omel.melscale_fbanks(NFFT_A // 2 + 1, 20, 8000, 80, 16000, None, "htk").t().unsqueeze(0)
        self.tb = IccrnTables(max_frames)
        self.frame_starts = torch.arange(max_frames).unsqueeze(1) + torch.arange(ALPHA_K).unsqueeze(0)


def forward(fe, w, near_i16, far_i16, n_fsmn, near_only=None):
    """session.run equivalent: two int16 [1,1,L] -> vad_results f32 [T_A].
    ref: DFSMN_VAD.forward, DFSMN/near_and_far_end_audio/Export_DFSMN_VAD.py:317-354.  far_i16 None = the near-end-only
    model (DFSMN/only_near_end_audio/Export_DFSMN_VAD.py:318-350): near_only = (pow_far [160,T',10], far_comp [2,160,T'])
    are its baked white-noise tensors.  Returns (vad, aec waveform) for staged tests."""
    inv = float(1.0 / 32768.0)
    near = near_i16.float() * inv
    near = near - torch.mean(near)
    nre, nim = ostft.stft(near, fe.cos_b, fe.sin_b, HOP_B, True)
    mix = torch.cat([nre, nim], dim=0).unsqueeze(0)                      # [1,2,160,T]
    T = mix.shape[-1]
    pad = torch.zeros((1, 2, F_BINS, ALPHA_K - 1))
    idx = fe.frame_starts[:T]
    mu = torch.cat([pad, mix], dim=-1)[..., idx]                         # [1,2,160,T,10]
    pow_mix = (mu * mu).sum(dim=1, keepdim=True)
    if far_i16 is None:
        pow_far = near_only[0][:, :T, :].float().reshape(1, 1, F_BINS, T, ALPHA_K)
        farc = near_only[1][:, :, :T].float().reshape(1, 1, 2, F_BINS, T)[0]          # [1,2,160,T]
    else:
        far = far_i16.float() * inv
        far = far - torch.mean(far)
        fre, fim = ostft.stft(far, fe.cos_b, fe.sin_b, HOP_B, True)
        farc = torch.cat([fre, fim], dim=0).unsqueeze(0)
        fu = torch.cat([pad, farc], dim=-1)[..., idx]
        pow_far = (fu * fu).sum(dim=1, keepdim=True)
    ci = torch.stack([pow_far, pow_mix], dim=-1).unsqueeze(dim=1)
    alpha = F.linear(torch.sum(ci, dim=2, keepdim=True), w["alpha.linear1.weight"], w["alpha.linear1.bias"]).squeeze(dim=-1)
    alpha = F.linear(alpha, w["alpha.linear2.weight"], w["alpha.linear2.bias"]).squeeze(dim=-1)
    farc = farc * torch.abs(alpha)
    iw = {k[len("iccrn."):]: v for k, v in w.items() if k.startswith("iccrn.")}
    aec, min_len = iccrn(torch.cat([mix, farc.squeeze(1)], dim=1), iw, fe.tb)
    near = near[..., :min_len]
    pe = torch.tensor(0.97, dtype=torch.float32)
    near = torch.cat([near[:, :, :1], near[:, :, 1:] - pe * near[:, :, :-1]], dim=-1)
    aec_pe = torch.cat([aec[:, :, :1], aec[:, :, 1:] - pe * aec[:, :, :-1]], dim=-1)
    echo = near - float(1.15) * aec_pe
    feats = []
    for sig in (near, aec_pe, echo):
        re, im = ostft.stft(sig, fe.cos_a, fe.sin_a, HOP_A, True)
        feats.append(torch.matmul(fe.fbank, re * re + im * im))
    feat = torch.cat(feats, dim=1).transpose(1, 2).clamp(1e-6).log()
    feat = (feat + w["mask.shift"].view(1, 1, -1)) * w["mask.scale"].view(1, 1, -1)
    x = F.relu(F.linear(feat, w["mask.linear1.weight"], w["mask.linear1.bias"]))
    for i in range(n_fsmn):
        x = uni_deep_fsmn(x, w, f"mask.deepfsmn.{i}")
    vad = torch.sigmoid(F.linear(x, w["mask.linear3.weight"], w["mask.linear3.bias"])).squeeze()
    return vad, aec


def tail_flags(score, start, stop, silence, hi=0.5, lo=0.5):
    """Tail rule of the DFSMN loop. ref: DFSMN/.../Inference_DFSMN_VAD_ONNX.py:262-273."""
    flags = []
    hi, lo = np.float32(hi), np.float32(lo)       # float32 comparisons, as NumPy 2 evaluates `float32_score >= python_float`
    for i in range(start, stop):
        if silence:
            silence = not (np.float32(score[i]) >= hi)
        else:
            silence = bool(np.float32(score[i]) <= lo)
        flags.append(silence)
    return flags, silence


def run_clip(fe, w, near_1d, far_1d, pad_noise_near, pad_noise_far, n_fsmn, L=16001, look_backward_s=0.3, near_only=None,
             speaking=0.5, silence_score=0.5):
    """Whole-clip driver. ref: Inference_DFSMN_VAD_ONNX.py:124-163 (prep), :221-278 (loop); far_1d None = the
    near-end-only driver (DFSMN/only_near_end_audio/Inference_DFSMN_VAD_ONNX.py:120-145, same loop)."""
    n = len(near_1d) if far_1d is None else min(len(near_1d), len(far_1d))
    near = postproc.normalize_to_int16(np.asarray(near_1d[:n], dtype=np.float32))
    frame = 320
    lb = int(look_backward_s * 16000 // frame)
    stride = L - (lb + 1) * frame
    near, _ = postproc.pad_to_window_grid(near, L, stride, pad_noise_near)
    far = None
    if far_1d is not None:
        far = postproc.normalize_to_int16(np.asarray(far_1d[:n], dtype=np.float32))
        far, _ = postproc.pad_to_window_grid(far, L, stride, pad_noise_far)
    silence, saved, s, vad = True, [], 0, None
    while s + L <= near.shape[0]:
        a = torch.from_numpy(near[s:s + L].copy()).reshape(1, 1, -1)
        b = None if far is None else torch.from_numpy(far[s:s + L].copy()).reshape(1, 1, -1)
        vad = forward(fe, w, a, b, n_fsmn, near_only)[0].numpy()
        flags, silence = postproc.lookahead_vote(vad, len(vad) - lb, lb, speaking, silence_score, silence, thresholds=(speaking, silence_score))
        saved += flags
        s += stride
    flags, silence = tail_flags(vad, len(vad) - lb, len(vad), silence, speaking, silence_score)
    saved += flags
    ts = postproc.vad_to_timestamps(saved, frame / 16000)
    return postproc.process_timestamps(ts, 0.3, 0.2), saved
