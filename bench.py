#!/usr/bin/env python
"""bench.py -- BASELINE.json metric on BASELINE config[1]:
Silero-VAD f32, batch = 4096 synthetic 10 s @ 16 kHz clips per MI355X (weak scaling: every rank owns
its own 4096 clips, no data-path collective), raw audio resident in HBM -> speech-segment tables.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over the resident batch: encoder kernel (STFT conv + conv stack
+ W_ih, f32 MFMA) -> persistent LSTM kernel -> device segmenter.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CLIPS_PER_GPU = 4096
SAMPLES = 160000
WINDOW = 512
STEPS_PER_CLIP = (SAMPLES + WINDOW - 1) // WINDOW            # 313

# Algorithmic flops per 512-sample window as the reference network computes it on non-padding
# samples (DESIGN.md "Silero path"): MACs x 2
MAC_STFT = 4 * 258 * 256
MAC_CONV1 = 4 * 128 * 129 * 3
MAC_CONV2 = 2 * 64 * 128 * 3
MAC_CONV3 = 64 * 64 * 2
MAC_CONV4 = 128 * 64
MAC_IH = 512 * 128
MAC_HH = 512 * 128
FLOP_ENCODE = 2 * (MAC_STFT + MAC_CONV1 + MAC_CONV2 + MAC_CONV3 + MAC_CONV4 + MAC_IH)
FLOP_RECUR = 2 * MAC_HH
# What the encoder kernel actually issues per 16-window tile: v_mfma_f32_16x16x4 counts per phase
# (folded STFT 1024, conv1 2560, conv2 640, conv3 128, conv4 128, W_ih 1024), 2048 flop each.  Lower than the
# dense count because the DFT's time and frequency symmetries quarter the STFT contraction (DESIGN.md "Silero path").
MFMA_PER_TILE = 1024 + 2560 + 640 + 128 + 128 + 1024
FLOP_ENCODE_ISSUED = MFMA_PER_TILE * 2048 // 16
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 = f32 vector rate


def profiled_traffic():
    """HBM bytes per encoder launch from the newest committed rocprofv3 PMC summary (profiles/rNN*/SUMMARY.txt, written by
    tools/profile_bench.sh from separate --pmc FETCH_SIZE / WRITE_SIZE passes of this very command).  FETCH_SIZE is
    doubled (gfx950 counts a wide coalesced read at half its bytes, MI355X_MICROARCH.md "HBM"); both are KiB."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "SUMMARY.txt"))):
        fetch = write = None
        grid = -1
        for line in open(path):
            if "silero_encode_kernel" not in line or "grid=" not in line or "_SIZE" not in line:
                continue
            g = int(line.split("grid=")[1].split()[0])
            name, val = line.split()[-3], float(line.split()[-2])
            if name in ("FETCH_SIZE", "WRITE_SIZE") and g >= grid:
                grid = g
                if name == "FETCH_SIZE":
                    fetch = val
                else:
                    write = val
        if fetch is not None and write is not None:
            best = {"bytes": (2.0 * fetch + write) * 1024.0, "source": os.path.relpath(path, ROOT), "grid_threads": grid}
    return best


def synth_batch(torch, device, batch, samples, seed):
    """int16-quantised burst clips generated on the GPU (every clip unique): 0.5-2 s segments
    alternating N(0,3000) / N(0,30), then x 1/32768 as the reference feeds Silero
    (Silero/Inference_Silero_VAD_ONNX.py:83)."""
    g = torch.Generator(device=device).manual_seed(seed)
    out = torch.empty((batch, samples), dtype=torch.float32, device=device)
    chunk = 512
    for b0 in range(0, batch, chunk):
        nb = min(chunk, batch - b0)
        dur = (torch.rand((nb, 24), generator=g, device=device) * 1.5 + 0.5) * 16000.0
        edges = torch.cumsum(dur, dim=1)
        pos = torch.arange(samples, device=device, dtype=torch.float32).unsqueeze(0).expand(nb, -1).contiguous()
        seg = torch.searchsorted(edges, pos)
        first = torch.randint(0, 2, (nb, 1), generator=g, device=device)
        loud = ((seg + first) % 2) == 0
        sigma = torch.where(loud, torch.tensor(3000.0, device=device), torch.tensor(30.0, device=device))
        x = torch.randn((nb, samples), generator=g, device=device) * sigma
        x = torch.clamp(torch.round(x), -32768, 32767)
        out[b0:b0 + nb] = x * 0.000030517578
        del dur, edges, pos, seg, loud, sigma, x
    return out


def cpu_baseline(budget_s=10.0):
    """The oracle (torch-CPU restatement of the reference graph) driven as the reference drives ORT:
    batch 1, one call per 512-sample window, state carried -- timed on this host's cores.
    Batch-1 windows are tiny ops, so more threads is not faster: a short calibration picks the
    best intra-op thread count (the reference uses ORT's auto setting / physical cores)."""
    import torch
    from oracle import silero as osil
    import vadx  # noqa: F401
    from vadx import weights
    w = {k: torch.from_numpy(v) for k, v in weights.silero_synthetic(1234).items()}
    clips = weights.burst_clips(4, SAMPLES, seed=4321).astype(np.float32) * np.float32(0.000030517578)
    model = osil.OnnxWrapperOracle(w)

    def run(seconds):
        """windows/s over ~`seconds`, checked every window so a slow host cannot overrun"""
        model.reset_states()
        n, t0 = 0, time.perf_counter()
        with torch.no_grad():
            while True:
                a = clips[(n // STEPS_PER_CLIP) % 4]
                s = (n % STEPS_PER_CLIP) * WINDOW
                chunk = torch.from_numpy(a[s:s + WINDOW])
                if chunk.shape[0] < WINDOW:
                    chunk = torch.nn.functional.pad(chunk, (0, WINDOW - chunk.shape[0]))
                if n % STEPS_PER_CLIP == 0:
                    model.reset_states()
                model(chunk, 16000).item()
                n += 1
                el = time.perf_counter() - t0
                if el >= seconds:
                    return n / el, n, el

    ncpu = os.cpu_count() or 1
    best_thr, best_rate = 1, 0.0
    for thr in sorted({1, 2, 4, 8, min(16, ncpu)}):
        if thr > ncpu:
            continue
        torch.set_num_threads(thr)
        run(0.2)
        rate, _, _ = run(0.8)
        if rate > best_rate:
            best_thr, best_rate = thr, rate
    torch.set_num_threads(best_thr)
    rate, n, el = run(budget_s)
    # the same oracle batched over clips on all cores (NOT how the reference runs; shown for scale)
    torch.set_num_threads(min(ncpu, 64))
    bb = 64
    xb = torch.from_numpy(weights.burst_clips(bb, 16 * WINDOW, seed=99).astype(np.float32) * np.float32(0.000030517578))
    m2 = osil.OnnxWrapperOracle(w)
    with torch.no_grad():
        m2.audio_forward(xb[:, :2 * WINDOW], 16000)
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 4.0:
            m2.audio_forward(xb, 16000)
            reps += 1
        batched = reps * bb * 16 / (time.perf_counter() - t0)
    return {"value": rate, "unit": "frames/s", "cores": best_thr, "kind": "port",
            "sample": f"{n} windows of synthetic 10 s clips, batch 1, one call per 512-sample window, state carried "
                      f"(torch-CPU oracle stand-in for ORT-CPU; best of 1/2/4/8/16 intra-op threads on a "
                      f"{ncpu}-CPU host), {el:.1f} s",
            "batched_value": batched, "batched_note": f"same oracle, batch {bb}, {min(ncpu, 64)} threads (not the reference's mode)"}


_T0 = time.perf_counter()


def log(msg):
    print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--clips", type=int, default=CLIPS_PER_GPU, help="clips per GPU (default = BASELINE config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import vadx  # noqa: F401
    from vadx import _lib, shard, silero, weights

    rank, local_rank, world = shard.env_rank()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the vadx product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = shard.init("nccl", device)        # None for a single process; RCCL otherwise (barrier/timing only)

    log(f"rank {rank}/{world} on {torch.cuda.get_device_name(device)}; torch imported")
    B = args.clips
    eng = silero.SileroEngine(weights.silero_synthetic(1234), device=device)
    audio = synth_batch(torch, device, B, SAMPLES, seed=1234 + rank)          # resident in HBM
    torch.cuda.synchronize()
    log(f"{B} x {SAMPLES} synthetic clips resident ({audio.numel() * 4 / 2**30:.2f} GiB)")
    L = _lib.lib()
    T = STEPS_PER_CLIP
    probs = torch.empty((B, T), dtype=torch.float32, device=device)
    cap = 64
    segs = torch.empty((B, cap, 2), dtype=torch.int64, device=device)
    counts = torch.empty((B,), dtype=torch.int32, device=device)
    lens = torch.full((B,), SAMPLES, dtype=torch.int64, device=device)
    prm = silero.seg_params(threshold=0.5, max_speech_duration_s=20, min_speech_duration_ms=250,
                            min_silence_duration_ms=250)     # Silero/Inference_Silero_VAD_ONNX.py:88-96
    ws = eng._workspace(B, T)
    st = _lib.stream_ptr()

    def step(ev=None):
        if ev:
            ev[0].record()
        _lib.check(L.vadx_silero_encode(eng.packed.data_ptr(), audio.data_ptr(), B, SAMPLES, _lib.row_stride(audio),
                                        ws.data_ptr(), ws.numel(), st))
        if ev:
            ev[1].record()
        _lib.check(L.vadx_silero_recur(eng.packed.data_ptr(), ws.data_ptr(), ws.numel(), B, T, None,
                                       probs.data_ptr(), None, st))
        if ev:
            ev[2].record()
        _lib.check(L.vadx_silero_segments(probs.data_ptr(), B, T, lens.data_ptr(), C.byref(prm), segs.data_ptr(),
                                          counts.data_ptr(), cap, st))
        if ev:
            ev[3].record()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    log("warm-up done")

    def fence():
        shard.fence(dist, torch.cuda.synchronize)

    events = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(events[k])
    fence()
    elapsed = time.perf_counter() - t0
    elapsed = shard.max_over_ranks(dist, elapsed, device)

    log(f"timed region done: {elapsed / args.steps * 1e3:.2f} ms/step")
    enc_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in events]))
    rec_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in events]))
    seg_ms = float(np.mean([e[2].elapsed_time(e[3]) for e in events]))
    assert int(counts.max().item()) <= cap, "segment table overflow"
    assert bool(torch.isfinite(probs).all())
    n_seg = int(counts.sum().item())

    frames_per_step = world * B * T
    value = frames_per_step * args.steps / elapsed
    achieved = (B * T * FLOP_ENCODE_ISSUED) / (enc_ms * 1e-3) / 1e12

    rtf_b1 = None
    cpu = None
    if rank == 0:
        one = audio[:1].contiguous()
        eng.clips(one)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            eng.clips(one)
        torch.cuda.synchronize()
        rtf_b1 = (time.perf_counter() - t1) / 5 / (SAMPLES / 16000.0)
        log(f"rtf_batch1 = {rtf_b1:.5f}")
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline()
            log("cpu baseline done")

    if rank == 0:
        traffic = profiled_traffic() if (B, T) == (CLIPS_PER_GPU, STEPS_PER_CLIP) else None
        line = {
            "metric": "audio frames/sec/GPU (16 kHz, 512-sample hop); RTF at batch=1",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Silero-VAD f32, batch=4096 synthetic 10 s @16 kHz clips per GPU "
                                   "(STFT conv + conv1d stack + LSTM cell HIP, seeded synthetic weights)",
                       "clips_per_gpu": B, "samples_per_clip": SAMPLES, "frames_per_clip": T,
                       "parallelism": f"clip-sharded x{world}, no collective"},
            "per_gpu_value": value / world, "rtf_batch1": rtf_b1,
            "kernel_ms": {"silero_encode_kernel": enc_ms, "silero_lstm_kernel": rec_ms, "silero_segments_kernel": seg_ms},
            "segments_found": n_seg,
            # achieved = the flops the kernel's algorithm needs (MFMA-issued: folded DFT, no padding taps) / its time;
            # the reference's dense arithmetic would count FLOP_ENCODE per frame ("dense_equivalent").
            "roofline": {"bound": "mfma", "kernel": "silero_encode_kernel", "achieved": achieved,
                         "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_F32_MFMA_TFLOPS,
                         "traffic": traffic["bytes"] if traffic else None, "traffic_unit": "B/launch",
                         "traffic_source": traffic["source"] if traffic else None,
                         "algorithmic_bytes_per_launch": B * T * (2048 + 2048),
                         "flop_per_frame": FLOP_ENCODE_ISSUED, "frames_per_launch": B * T,
                         "dense_equivalent": {"flop_per_frame": FLOP_ENCODE,
                                              "achieved": achieved * FLOP_ENCODE / FLOP_ENCODE_ISSUED}},
            # the recurrent kernel is matrix-pipe work too (W_hh x h, 16 clips = one MFMA tile wide): its own fraction
            "roofline_recurrent": {"bound": "mfma", "kernel": "silero_lstm_kernel", "flop_per_frame": FLOP_RECUR,
                                   "achieved": B * T * FLOP_RECUR / (rec_ms * 1e-3) / 1e12, "peak": PEAK_F32_MFMA_TFLOPS,
                                   "unit": "TFLOP/s", "frac": B * T * FLOP_RECUR / (rec_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS},
            "hbm": {"algorithmic_bytes_per_frame": 2048 + 4,
                    "achieved_GBps_whole_step": B * T * 2052 / (elapsed / args.steps) / 1e9, "peak_GBps": 8000.0},
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
