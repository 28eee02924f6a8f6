"""The C ABI used from plain C (tests/c/cabi_silero.c, compiled with gcc against include/vadx.h + libvadx.so): no
Python or torch in the process.  Its output must equal the Python host path's bit for bit."""
import os
import subprocess

import numpy as np
import pytest
import torch

import vadx  # noqa: F401
from vadx import _lib, silero, weights

from test_cabi_cpu import build_c_client

pytestmark = pytest.mark.gpu
ORDER = (["stft_basis"] + [f"enc{i}_w" for i in range(4)] + [f"enc{i}_b" for i in range(4)] +
         ["lstm_w_ih", "lstm_w_hh", "lstm_b_ih", "lstm_b_hh", "dec_w", "dec_b"])


def test_plain_c_client_matches_python_host_path(tmp_path):
    exe = build_c_client(tmp_path)
    w = weights.silero_synthetic(1234)
    B, N, CAP = 21, 40000, 32
    audio = weights.burst_clips(B, N, seed=77).astype(np.float32) * np.float32(0.000030517578)
    np.concatenate([np.ascontiguousarray(w[k], dtype=np.float32).ravel() for k in ORDER]).tofile(tmp_path / "w.bin")
    audio.tofile(tmp_path / "a.bin")
    r = subprocess.run([exe, str(tmp_path / "w.bin"), str(tmp_path / "a.bin"), str(B), str(N), str(tmp_path / "o.bin")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    T = (N + 511) // 512
    raw = (tmp_path / "o.bin").read_bytes()
    probs = np.frombuffer(raw, dtype=np.float32, count=B * T).reshape(B, T)
    counts = np.frombuffer(raw, dtype=np.int32, count=B, offset=4 * B * T)
    segs = np.frombuffer(raw, dtype=np.int64, count=B * CAP * 2, offset=4 * B * T + 4 * B).reshape(B, CAP, 2)

    eng = silero.SileroEngine(w)
    want = eng.clips(torch.from_numpy(audio).cuda())
    assert np.array_equal(probs, want.cpu().numpy())
    s2, c2 = eng.segments(want, torch.full((B,), N, dtype=torch.int64, device="cuda"), cap=CAP,
                          min_silence_duration_ms=100)
    assert np.array_equal(counts, c2.cpu().numpy())
    for b in range(B):
        assert np.array_equal(segs[b, :counts[b]], s2[b, :counts[b]].cpu().numpy())
    assert counts.max() > 0


def _run_client(tmp_path, exe, w, audio, *extra):
    B, N = audio.shape
    np.concatenate([np.ascontiguousarray(w[k], dtype=np.float32).ravel() for k in ORDER]).tofile(tmp_path / "w.bin")
    audio.tofile(tmp_path / "a.bin")
    r = subprocess.run([exe, str(tmp_path / "w.bin"), str(tmp_path / "a.bin"), str(B), str(N), str(tmp_path / "o.bin"), *extra],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    T = (N + 511) // 512
    probs = np.frombuffer((tmp_path / "o.bin").read_bytes(), dtype=np.float32, count=B * T).reshape(B, T)
    return probs, r.stdout, r.stderr


def test_plain_c_client_runs_the_range_protocol(tmp_path):
    """the C ABI's default arithmetic (AUTO = F16X2) needs the caller's cooperation -- the example client now gives it.
    Audio at 100000 x full scale drives an activation beyond 65504: the client reads vadx_silero_range_flag, recomputes on BF16X3 and ends
    with exactly the scores the Python engine's guarded path returns; a client that SKIPS the protocol reads NaN for the flagged clips, never
    plausible numbers (the flagged workgroups poison their gx)."""
    exe = build_c_client(tmp_path)
    w = weights.silero_synthetic(1234)
    B, N = 37, 6000
    audio = weights.burst_clips(B, N, seed=78).astype(np.float32) * np.float32(0.000030517578)
    audio[16:32] *= np.float32(1.0 / 100000.0)        # the second clip group stays inside the fp16 range after the x 100000 gain
    probs, out, err = _run_client(tmp_path, exe, w, audio, "100000")
    assert "range fallbacks 1" in out and "recomputed on VADX_ARITH_BF16X3" in err and "NaN scores 0" in out
    eng = silero.SileroEngine(w)
    prev = silero.encoder_mode("h2")
    n0 = eng.range_fallbacks
    want = eng.clips(torch.from_numpy(audio * np.float32(100000.0)).cuda()).cpu().numpy()
    assert eng.range_fallbacks == n0 + 1
    assert np.array_equal(probs, want)
    bad, out2, _ = _run_client(tmp_path, exe, w, audio, "100000", "unchecked")
    assert "range fallbacks 0" in out2
    # a flagged workgroup (two tiles) hands the recurrent kernel NaN: the loud clips end in NaN, NaN never turns back into a number, and every
    # FINITE score the unchecked caller reads is a correct one (an unflagged window on an unpoisoned state)
    assert np.isnan(bad[:16, -1]).all() and np.isnan(bad[32:, -1]).all()
    isn = np.isnan(bad)
    assert (isn[:, 1:] >= isn[:, :-1]).all()
    silero.encoder_mode("split")
    exact = eng.clips(torch.from_numpy(audio * np.float32(100000.0)).cuda()).cpu().numpy()
    silero.encoder_mode(prev)
    assert np.abs(np.where(isn, 0.0, bad - exact)).max() <= 1e-4
    quiet, out3, _ = _run_client(tmp_path, exe, w, audio)
    assert "range fallbacks 0" in out3 and "NaN scores 0" in out3 and np.isfinite(quiet).all()
