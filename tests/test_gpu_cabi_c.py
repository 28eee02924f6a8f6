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
