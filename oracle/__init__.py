"""CPU oracle for the VAD hot path -- TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU (numpy + torch-CPU float32), the arithmetic of
DakeQQ/Voice-Activity-Detection-VAD-ONNX's raw-audio -> speech-timestamps path.
Every function cites the reference file:line it follows.

Rules (see DESIGN.md "Oracle"):
  * Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
    import this package.  The product package never imports it and has no CPU fallback.
  * Pinning status:
      - STFT variants, FSMN wrapper/encoder, FireRed wrapper/DetectModel, UniDeepFsmn,
        VadPostprocessor, vad_to_timestamps / process_timestamps / format_time,
        get_speech_timestamps: PINNED against the reference's own Python executed in the
        build container (tests/golden/make_golden.py -> tests/golden/*.npz).
      - Silero network (pip `silero_vad`, un-vendored, version unpinned),
        MarbleNet encoder/decoder (NeMo, un-vendored), torchaudio.melscale_fbanks
        (un-vendored): restated from the published algorithms -- PARITY UNPINNED for
        those three; the call sites around them are pinned.
"""
