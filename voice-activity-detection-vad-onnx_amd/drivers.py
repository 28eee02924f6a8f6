"""Script-level drop-ins for the reference's `Inference_*_VAD_ONNX.py` programs: raw audio file(s) in,
`timestamps_second.txt` / `timestamps_indices.txt` out, same constants (as keyword arguments instead
of module-level globals), same stdout summary.  Each function also accepts a LIST of files: equal-length files run as one
device batch per length group (`_grouped`; the Silero driver pads ragged clips into one batch instead); the printed RTF is
over the total audio duration.  `engine` / `model` may be an engine object, a weight dict, a checkpoint path, or the explicit
opt-in "synthetic:<seed>" (vadx.checkpoints.resolve) -- never an implicit default."""
from __future__ import annotations

import time

import numpy as np

from . import audio_io, timestamps


def _as_list(x):
    return list(x) if isinstance(x, (list, tuple)) else [x]


def _grouped(clips, noises, run):
    """Run `run(stacked_clips [G, n], stacked_noise or None) -> list of G results` once per group of equal-length clips
    (one device batch each) and hand the results back in input order."""
    by_len = {}
    for k, c in enumerate(clips):
        by_len.setdefault(len(c), []).append(k)
    out = [None] * len(clips)
    for idx in by_len.values():
        for k, r in zip(idx, run(np.stack([clips[k] for k in idx]), _stack_noise(noises, idx))):
            out[k] = r
    return out


def _stack_noise(noises, idx):
    """Per-clip pad-noise rows of one length group as a matrix: rows may be ragged (each clip ran alone in the reference), so
    they are trimmed to the group's shortest row -- the engines take `[B, >= pad]` and use the first `pad` samples."""
    if noises is None:
        return None
    rows = [np.asarray(noises[k]).reshape(-1) for k in idx]
    n = min(len(r) for r in rows)
    return np.stack([r[:n] for r in rows])


def _finish(all_ts, sample_rate, save_second, save_indices, single, elapsed, echo):
    if single:
        timestamps.write_timestamp_files(all_ts[0], sample_rate, save_second, save_indices, echo)
    else:
        for k, ts in enumerate(all_ts):
            timestamps.write_timestamp_files(ts, sample_rate, f"{save_second}.{k}", f"{save_indices}.{k}", lambda *_: None)
    echo(f"\nVAD Process Complete.\n\nTime Cost: {elapsed:.3f} Seconds")
    return all_ts[0] if single else all_ts


def inference_silero(test_vad_audio="./vad_sample.wav", model=None, save_timestamps_second="./timestamps_second.txt",
                     save_timestamps_indices="./timestamps_indices.txt", ACTIVATE_THRESHOLD=0.5, FUSION_THRESHOLD=0.3,
                     MIN_SPEECH_DURATION=0.25, MAX_SPEECH_DURATION=20, MIN_SILENCE_DURATION=250, SAMPLE_RATE=16000,
                     use_fp16=False, echo=print):
    """Silero/Inference_Silero_VAD_ONNX.py:80-120.  use_fp16 (:16, :83): the samples are quantised to float16 and scaled there, as
    the reference feeds its fp16-optimised model -- I/O-compatible: the network arithmetic here stays float32."""
    from . import silero
    files = _as_list(test_vad_audio)
    model = silero.load_silero_vad(onnx=True, use_cpu=True, path=model) if (model is None or isinstance(model, (str, dict))) else model
    if use_fp16:            # np.array(samples, dtype=float16) * 0.000030517578: NumPy keeps float16 (the constant is 2^-15: exact, subnormals round)
        clips = [(audio_io.load_wav(f, SAMPLE_RATE).astype(np.float16) * np.float16(0.000030517578)).astype(np.float32) for f in files]
    else:
        clips = [audio_io.load_wav(f, SAMPLE_RATE).astype(np.float32) * np.float32(0.000030517578) for f in files]
    echo("\nStart to run the VAD process.")
    t0 = time.time()
    n = max(len(c) for c in clips)
    batch = np.zeros((len(clips), n), np.float32)
    for k, c in enumerate(clips):
        batch[k, :len(c)] = c
    res = silero.get_speech_timestamps_batch(batch, model, lengths=[len(c) for c in clips], threshold=ACTIVATE_THRESHOLD,
                                             max_speech_duration_s=MAX_SPEECH_DURATION,
                                             min_speech_duration_ms=int(MIN_SPEECH_DURATION * 1000),
                                             min_silence_duration_ms=MIN_SILENCE_DURATION, return_seconds=True)
    elapsed = time.time() - t0
    echo(f"\nVAD Complete. Time Cost: {elapsed:.3f} seconds.")
    all_ts = [timestamps.process_timestamps([(d["start"], d["end"]) for d in r], FUSION_THRESHOLD, MIN_SPEECH_DURATION) for r in res]
    return _finish(all_ts, SAMPLE_RATE, save_timestamps_second, save_timestamps_indices, not isinstance(test_vad_audio, (list, tuple)), elapsed, echo)


def inference_fsmn(test_vad_audio="./vad_sample.wav", engine=None, save_timestamps_second="./timestamps_second.txt",
                   save_timestamps_indices="./timestamps_indices.txt", FUSION_THRESHOLD=0.3, MIN_SPEECH_DURATION=0.2,
                   SPEAKING_SCORE=0.5, SILENCE_SCORE=0.5, LOOK_BACKWARD=0.3, SNR_THRESHOLD=10.0,
                   BACKGROUND_NOISE_dB_INIT=30.0, ONE_MINUS_SPEECH_THRESHOLD=1.0, pad_noise=None, echo=print):
    """FSMN/Inference_FSMN_VAD_ONNX.py:66-260."""
    from . import fsmn
    files = _as_list(test_vad_audio)
    engine = fsmn.FsmnEngine(engine) if (engine is None or isinstance(engine, (str, dict))) else engine
    clips = [audio_io.load_wav(f, 16000) for f in files]
    echo("\nRunning the FSMN_VAD by ONNX Runtime.")
    t0 = time.time()
    all_ts = _grouped(clips, pad_noise, lambda batch, nz: engine.detect(
        batch.astype(np.float32), pad_noise=nz, fusion_threshold=FUSION_THRESHOLD, min_speech_duration=MIN_SPEECH_DURATION,
        look_backward_s=LOOK_BACKWARD, speaking_score=SPEAKING_SCORE, silence_score=SILENCE_SCORE,
        snr_threshold=SNR_THRESHOLD, noise_init_dB=BACKGROUND_NOISE_dB_INIT,
        one_minus_speech_threshold=ONE_MINUS_SPEECH_THRESHOLD))
    elapsed = time.time() - t0
    return _finish(all_ts, 16000, save_timestamps_second, save_timestamps_indices, not isinstance(test_vad_audio, (list, tuple)), elapsed, echo)


def inference_firered(test_vad_audio="./vad_sample.wav", engine=None, save_timestamps_second="./timestamps_second.txt",
                      save_timestamps_indices="./timestamps_indices.txt", NORMALIZE_AUDIO=False, SMOOTH_WINDOW_SIZE=5,
                      SPEAKING_SCORE=0.4, MIN_SPEECH_FRAME=20, MAX_SPEECH_FRAME=2000, MIN_SILENCE_FRAME=20,
                      MERGE_SILENCE_FRAME=5, EXTEND_SPEECH_FRAME=0, pad_noise=None, echo=print):
    """FireRedVAD/Inference_FireRed_ONNX.py:523-613 (RUN_VAD)."""
    from . import firered
    files = _as_list(test_vad_audio)
    engine = firered.FireRedEngine(engine) if (engine is None or isinstance(engine, (str, dict))) else engine
    clips = [audio_io.load_wav(f, 16000) for f in files]
    if NORMALIZE_AUDIO:
        clips = [timestamps.normalise_audio(c) for c in clips]
    echo("\nRunning the FireRedVAD by ONNX Runtime.")
    t0 = time.time()
    post = (SMOOTH_WINDOW_SIZE, SPEAKING_SCORE, MIN_SPEECH_FRAME, MAX_SPEECH_FRAME, MIN_SILENCE_FRAME,
            MERGE_SILENCE_FRAME, EXTEND_SPEECH_FRAME)
    all_ts = _grouped(clips, pad_noise, lambda batch, nz: engine.detect(batch, pad_noise=nz, post=post))
    elapsed = time.time() - t0
    echo(f"RTF: {elapsed / (sum(len(c) for c in clips) / 16000):.4f}")
    return _finish(all_ts, 16000, save_timestamps_second, save_timestamps_indices, not isinstance(test_vad_audio, (list, tuple)), elapsed, echo)


def inference_firered_aed(test_aed_audio="./vad_sample.wav", engine=None, NORMALIZE_AUDIO=False, SMOOTH_WINDOW_SIZE=5,
                          SPEAKING_SCORE=0.4, SINGING_THRESHOLD=0.5, MUSIC_THRESHOLD=0.5, MIN_EVENT_FRAME=20,
                          MAX_EVENT_FRAME=2000, MIN_SILENCE_FRAME=20, MERGE_SILENCE_FRAME=5, EXTEND_SPEECH_FRAME=0,
                          pad_noise=None, echo=print):
    """FireRedVAD/Inference_FireRed_ONNX.py:620-742 (RUN_AED): -> (event2timestamps, event2ratio) per file."""
    from . import firered
    files = _as_list(test_aed_audio)
    engine = firered.FireRedEngine(engine) if (engine is None or isinstance(engine, (str, dict))) else engine      # None raises (checkpoints.resolve)
    clips = [audio_io.load_wav(f, 16000) for f in files]
    if NORMALIZE_AUDIO:
        clips = [timestamps.normalise_audio(c) for c in clips]
    echo("\nRunning the FireRedAED by ONNX Runtime.")
    t0 = time.time()
    post = (SMOOTH_WINDOW_SIZE, MIN_EVENT_FRAME, MAX_EVENT_FRAME, MIN_SILENCE_FRAME, MERGE_SILENCE_FRAME, EXTEND_SPEECH_FRAME)
    results = []
    for c, nz in zip(clips, pad_noise if pad_noise is not None else [None] * len(clips)):
        results += engine.detect_events(c[None, :], pad_noise=None if nz is None else nz[None, :],
                                        thresholds=(SPEAKING_SCORE, SINGING_THRESHOLD, MUSIC_THRESHOLD), post=post)
    elapsed = time.time() - t0
    for (ts, ratio), c in zip(results, clips):
        echo(f"\nAED Results:\n  Audio duration: {len(c) / 16000:.3f}s\n  Event ratios: {ratio}")
        for event, seg in ts.items():
            echo(f"\n  [{event}] segments ({len(seg)}):")
            for a, b in seg:
                echo(f"    {timestamps.format_time(a)} --> {timestamps.format_time(b)}")
    echo(f"\nAED Process Complete.\n\nTime Cost: {elapsed:.3f} Seconds\nRTF: {elapsed / (len(clips[0]) / 16000):.4f}")
    return results[0] if not isinstance(test_aed_audio, (list, tuple)) else results


def inference_firered_stream(test_vad_audio="./vad_sample.wav", engine=None, NORMALIZE_AUDIO=False, SMOOTH_WINDOW_SIZE=5,
                             STREAM_VAD_THRESHOLD=0.4, PAD_START_FRAME=5, MIN_SPEECH_FRAME_STREAM=8,
                             MAX_SPEECH_FRAME_STREAM=2000, MIN_SILENCE_FRAME_STREAM=20, STREAM_CHUNK_SAMPLES=2560, echo=print):
    """FireRedVAD/Inference_FireRed_ONNX.py:744-840 (RUN_STREAM_VAD): -> [(start_s, end_s)] per file."""
    from . import firered
    files = _as_list(test_vad_audio)
    engine = firered.FireRedEngine(engine, STREAM_CHUNK_SAMPLES) if (engine is None or isinstance(engine, (str, dict))) else engine
    clips = [audio_io.load_wav(f, 16000) for f in files]
    if NORMALIZE_AUDIO:
        clips = [timestamps.normalise_audio(c) for c in clips]
    echo("\nRunning the FireRedStreamVAD by ONNX Runtime.")
    t0 = time.time()
    post = (SMOOTH_WINDOW_SIZE, STREAM_VAD_THRESHOLD, PAD_START_FRAME, MIN_SPEECH_FRAME_STREAM, MAX_SPEECH_FRAME_STREAM,
            MIN_SILENCE_FRAME_STREAM)
    results = []
    for c in clips:
        results += engine.stream_detect(c[None, :], chunk=STREAM_CHUNK_SAMPLES, post=post)
    elapsed = time.time() - t0
    for seg, c in zip(results, clips):
        echo(f"\nStream-VAD Results:\n  Audio duration: {len(c) / 16000:.3f}s\n  Segments detected: {len(seg)}\n\n  Timestamps in Second:")
        for a, b in seg:
            echo(f"    {timestamps.format_time(a)} --> {timestamps.format_time(b)}")
    echo(f"\nStream-VAD Process Complete.\n\nTime Cost: {elapsed:.3f} Seconds\nRTF: {elapsed / (len(clips[0]) / 16000):.4f}")
    return results[0] if not isinstance(test_vad_audio, (list, tuple)) else results


def inference_marblenet(test_vad_audio="./vad_sample.wav", engine=None, save_timestamps_second="./timestamps_second.txt",
                        save_timestamps_indices="./timestamps_indices.txt", NORMALIZE_AUDIO=False, SMOOTH_WINDOW_SIZE=3,
                        SPEAKING_SCORE=0.5, MIN_SPEECH_FRAME=10, MAX_SPEECH_FRAME=1000, MIN_SILENCE_FRAME=10,
                        MERGE_SILENCE_FRAME=3, EXTEND_SPEECH_FRAME=0, INPUT_AUDIO_LENGTH=None, pad_noise=None, echo=print):
    """NVIDIA_.../Inference_NVIDIA_MarbleNet_VAD_ONNX.py:120-422 (dynamic axis: one window per clip)."""
    from . import marblenet
    files = _as_list(test_vad_audio)
    engine = marblenet.MarbleNetEngine(engine) if (engine is None or isinstance(engine, (str, dict))) else engine
    clips = [audio_io.load_wav(f, 16000) for f in files]
    if NORMALIZE_AUDIO:
        clips = [timestamps.normalise_audio(c) for c in clips]
    echo("\nRunning the NVIDIA_VAD by ONNX Runtime.")
    t0 = time.time()
    post = (SMOOTH_WINDOW_SIZE, SPEAKING_SCORE, MIN_SPEECH_FRAME, MAX_SPEECH_FRAME, MIN_SILENCE_FRAME,
            MERGE_SILENCE_FRAME, EXTEND_SPEECH_FRAME)
    all_ts = _grouped(clips, pad_noise, lambda batch, nz: engine.detect(batch, window_len=INPUT_AUDIO_LENGTH, pad_noise=nz, post=post))
    elapsed = time.time() - t0
    return _finish(all_ts, 16000, save_timestamps_second, save_timestamps_indices, not isinstance(test_vad_audio, (list, tuple)), elapsed, echo)


def inference_dfsmn(test_near_end_audio="./examples/nearend_mic.wav", test_far_end_audio="./examples/farend_speech.wav", engine=None,
                    save_timestamps_second="./timestamps_second.txt", save_timestamps_indices="./timestamps_indices.txt",
                    SPEAKING_SCORE=0.5, SILENCE_SCORE=0.5, FUSION_THRESHOLD=0.3, MIN_SPEECH_DURATION=0.2,
                    pad_noise_near=None, pad_noise_far=None, echo=print):
    """DFSMN/near_and_far_end_audio/Inference_DFSMN_VAD_ONNX.py:121-300: near-end mic + far-end reference -> AEC ->
    mask-net VAD -> look-ahead vote -> timestamps (both text files).  Lists of paths run as a batch of clip pairs."""
    from . import dfsmn
    nears, fars = _as_list(test_near_end_audio), _as_list(test_far_end_audio)
    engine = dfsmn.DfsmnEngine(engine) if (engine is None or isinstance(engine, (str, dict))) else engine
    echo(f"\nTest Input Near_End Audio: {test_near_end_audio}\nTest Input Far_End Audio: {test_far_end_audio}")
    pairs = []
    for pn, pf in zip(nears, fars):                 # files are read before the clock starts, like the reference script
        a, f = audio_io.load_wav(pn, 16000).astype(np.float32), audio_io.load_wav(pf, 16000).astype(np.float32)
        n = min(len(a), len(f))
        pairs.append(np.stack([a[:n], f[:n]]))
    t0 = time.time()
    by_len = {}
    for k, pr in enumerate(pairs):
        by_len.setdefault(pr.shape[1], []).append(k)
    all_ts = [None] * len(pairs)
    for idx in by_len.values():                     # one device batch per group of equal-length pairs
        near, far = np.stack([pairs[k][0] for k in idx]), np.stack([pairs[k][1] for k in idx])
        nzn, nzf = _stack_noise(pad_noise_near, idx), _stack_noise(pad_noise_far, idx)      # each side independently, either may be None
        res = engine.detect(near, far, nzn, nzf, fusion_threshold=FUSION_THRESHOLD, min_speech_duration=MIN_SPEECH_DURATION,
                            speaking_score=SPEAKING_SCORE, silence_score=SILENCE_SCORE)
        for k, r in zip(idx, res):
            all_ts[k] = r
    elapsed = time.time() - t0
    return _finish(all_ts, 16000, save_timestamps_second, save_timestamps_indices, not isinstance(test_near_end_audio, (list, tuple)), elapsed, echo)


def inference_dfsmn_near_only(test_vad_audio="./vad_sample.wav", engine=None, save_timestamps_second="./timestamps_second.txt",
                              save_timestamps_indices="./timestamps_indices.txt", SPEAKING_SCORE=0.5, SILENCE_SCORE=0.5,
                              FUSION_THRESHOLD=0.3, MIN_SPEECH_DURATION=0.2, pad_noise=None, echo=print):
    """DFSMN/only_near_end_audio/Inference_DFSMN_VAD_ONNX.py: one microphone file in, timestamps out.  `engine` must
    carry the export's baked white-noise constants (DfsmnEngine.set_near_only_constants)."""
    from . import dfsmn, weights
    files = _as_list(test_vad_audio)
    if engine is None or isinstance(engine, (str, dict)):
        raise ValueError("inference_dfsmn_near_only needs a DfsmnEngine carrying the export's baked white-noise constants "
                         "(DfsmnEngine.set_near_only_constants); weights.dfsmn_near_only_constants(seed) gives a stand-in")
    echo(f"\nTest Input Audio: {test_vad_audio}")
    all_ts = []
    t0 = time.time()
    for k, pth in enumerate(files):
        a = audio_io.load_wav(pth, 16000).astype(np.float32)
        nz = None if pad_noise is None else np.asarray(pad_noise)[k][None, :]
        all_ts += engine.detect(a[None, :], None, nz, None, fusion_threshold=FUSION_THRESHOLD, min_speech_duration=MIN_SPEECH_DURATION,
                                speaking_score=SPEAKING_SCORE, silence_score=SILENCE_SCORE)
    elapsed = time.time() - t0
    return _finish(all_ts, 16000, save_timestamps_second, save_timestamps_indices, not isinstance(test_vad_audio, (list, tuple)), elapsed, echo)
