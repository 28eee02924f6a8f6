"""GPU: BASELINE config 1 plumbing -- the reference's own sample file through the script-level
drop-ins (file in -> the two timestamp text files out), checked against the oracle."""
import os

import numpy as np
import pytest
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
