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
def test_logmel_matches_oracle(preset, L, W, stride, B):
    n = (W - 1) * stride + L
    clips = weights.burst_clips(B, n, seed=L + W + B)
    clips[0, : min(n, 3000)] = 0                         # exact digital silence -> exercises the log floor
    fe = frontend.Frontend(preset, L)
    out = fe.logmel(clips, windows_per_clip=W, win_stride=stride).cpu().numpy()
    wins = np.stack([clips[b, w * stride:w * stride + L] for b in range(B) for w in range(W)])
    ref = oracle_logmel(preset, T(wins).unsqueeze(1)).numpy()
    assert out.shape == ref.shape == (B * W, fe.frames, 80)
    err = np.abs(out - ref)
    assert np.isfinite(out).all()
    # relative agreement of the mel energies themselves (before log) where they are above the floor
    big = ref > np.log(1e-3)
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
