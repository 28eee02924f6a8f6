#!/usr/bin/env python
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE'S OWN PYTHON in the build container.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Inputs are seeded, weights are the seeded synthetic ones from vadx.weights (pushed into the
reference nn.Modules through their state dicts), outputs are whatever the reference code computes.
Only numbers are stored (fixtures are data; no reference text).  The GPU box never runs this.
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import _refload as R                      # noqa: E402
import vadx                               # noqa: E402,F401
from vadx import weights                  # noqa: E402
from oracle import mel as omel            # noqa: E402  (torchaudio stand-in: melscale_fbanks is un-vendored)

R.install_stubs(omel.melscale_fbanks)
torch.set_num_threads(4)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  wrote {name}.npz ({os.path.getsize(path) / 1024:.1f} KiB)")


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


# ------------------------------------------------------------------------------------ STFT
def gen_stft():
    print("STFT variants")
    out = {}
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(1, 1, 2400, generator=g)
    out["x"] = x.numpy()
    v1 = R.load_module("FSMN/STFT_Process.py", "ref_stft_v1")
    v1b = R.load_module("DFSMN/near_and_far_end_audio/STFT_Process.py", "ref_stft_v1b")
    v2 = R.load_module("NVIDIA_Frame_VAD_Multilingual_MarbleNet/STFT_Process.py", "ref_stft_v2")
    v2f = R.load_module("FireRedVAD/STFT_Process.py", "ref_stft_v2f")

    def rec(tag, mod_out, kern):
        re, im = mod_out
        out[tag + "_re"] = re.numpy()
        out[tag + "_im"] = im.numpy()
        for k, v in kern.items():          # tables: a few rows verbatim + sha256 of all bytes
            a = np.ascontiguousarray(v.numpy())
            rows = sorted({0, 1, 2, 37, a.shape[0] // 2, a.shape[0] - 2, a.shape[0] - 1})
            out[f"{tag}_{k}_rowidx"] = np.array(rows)
            out[f"{tag}_{k}_rows"] = a[rows]
            out[f"{tag}_{k}_sha256"] = np.array(__import__("hashlib").sha256(a.tobytes()).hexdigest())
            out[f"{tag}_{k}_shape"] = np.array(a.shape)

    with torch.no_grad():
        m = v1.STFT_Process("stft_B", n_fft=512, win_length=400, hop_len=160, max_frames=0, window_type="hamming").eval()
        rec("fsmn", m(x, "constant"), {"cos": m.cos_kernel[:, 0, :], "sin": m.sin_kernel[:, 0, :]})
        m = v1b.STFT_Process("stft_B", n_fft=319, win_length=319, hop_len=160, max_frames=0, window_type="hamming", center_pad=True).eval()
        rec("dfsmn_b", m(x, "constant"), {"cos": m.cos_kernel[:, 0, :], "sin": m.sin_kernel[:, 0, :]})
        m = v1b.STFT_Process("stft_B", n_fft=1024, win_length=640, hop_len=320, max_frames=0, window_type="hamming", center_pad=True).eval()
        rec("dfsmn_a", m(x, "constant"), {"cos": m.cos_kernel[:, 0, :], "sin": m.sin_kernel[:, 0, :]})
        m = v2.STFT_Process("stft_B", n_fft=512, win_length=400, hop_len=160, max_frames=0, window_type="hann_sym", center_pad=True, pad_mode="constant").eval()
        rec("marble", m(x), {"kernel": m.stft_kernel[:, 0, :]})
        m = v2f.STFT_Process("stft_B", n_fft=400, win_length=400, hop_len=160, max_frames=0, window_type="povey", center_pad=False, pad_mode="constant").eval()
        rec("firered", m(x), {"kernel": m.stft_kernel[:, 0, :]})
    save("stft", **out)


# ------------------------------------------------------------------------------------ host helpers
def gen_host():
    print("host helpers (vad_to_timestamps / process_timestamps / format_time)")
    ns = {"timedelta": __import__("datetime").timedelta, "np": np}
    R.select_nodes("FSMN/Inference_FSMN_VAD_ONNX.py",
                   {"process_timestamps", "vad_to_timestamps", "format_time", "normalize_to_int16"}, ns)
    rng = np.random.default_rng(1234)
    flags, raw_ts, proc_ts, fmt_in, fmt_out = [], [], [], [], []
    for case in range(24):
        n = int(rng.integers(0, 400))
        p = rng.uniform(0.02, 0.5)
        f = np.zeros(n, dtype=bool)
        state = bool(rng.integers(0, 2))
        for i in range(n):
            if rng.uniform() < p * 0.2:
                state = not state
            f[i] = state
        fd = [0.01, 0.02][case % 2]
        ts = ns["vad_to_timestamps"](list(f), fd)
        pt = ns["process_timestamps"](list(ts), 0.3, [0.2, 0.25][case % 2])
        flags.append(f)
        raw_ts.append(np.array(ts, dtype=np.float64).reshape(-1, 2))
        proc_ts.append(np.array(pt, dtype=np.float64).reshape(-1, 2))
    for v in list(rng.uniform(0, 5000, 64)) + [2.28, 2.74, 0.0, 3599.9996, 5.9, 19.46, 0.5, 2.5, 5.599]:
        fmt_in.append(float(v))
        fmt_out.append(ns["format_time"](float(v)))
    aud = (rng.standard_normal(4000) * 1234.5).astype(np.float32)
    save("host", n_cases=np.array(len(flags)),
         **{f"flags_{i}": a for i, a in enumerate(flags)},
         **{f"raw_{i}": a for i, a in enumerate(raw_ts)},
         **{f"proc_{i}": a for i, a in enumerate(proc_ts)},
         fmt_in=np.array(fmt_in), fmt_out=np.array(fmt_out),
         norm_in=aud, norm_out=ns["normalize_to_int16"](aud))
    # the only checked-in expected output of the reference (DFSMN near+far on *1.wav): data fixture
    with open(os.path.join(R.REF, "DFSMN/near_and_far_end_audio/timestamps_indices.txt")) as fh:
        idx = np.array([[int(t) for t in ln.split("-->")] for ln in fh if ln.strip()], dtype=np.int64)
    with open(os.path.join(R.REF, "DFSMN/near_and_far_end_audio/timestamps_second.txt")) as fh:
        sec = np.array([ln.strip() for ln in fh if ln.strip()])
    save("dfsmn_golden_txt", indices=idx, seconds=sec)


# ------------------------------------------------------------------------------------ VadPostprocessor
def prob_tracks(rng):
    tracks = []
    for case in range(14):
        n = [0, 1, 3, 7, 98, 280, 980, 2500, 5000, 980, 980, 600, 333, 4100][case]
        if n == 0:
            tracks.append(np.zeros(0, np.float32))
            continue
        kind = case % 4
        if kind == 0:
            p = rng.uniform(0, 1, n)
        elif kind == 1:                                  # bursty
            p = np.zeros(n)
            pos, hot = 0, False
            while pos < n:
                seg = int(rng.integers(2, 120))
                p[pos:pos + seg] = rng.uniform(0.55, 1.0, min(seg, n - pos)) if hot else rng.uniform(0, 0.45, min(seg, n - pos))
                pos += seg
                hot = not hot
        elif kind == 2:                                  # all speech, longer than max_speech_frame
            p = rng.uniform(0.6, 1.0, n)
        else:
            p = np.clip(0.5 + 0.3 * np.sin(np.arange(n) / 9.0) + 0.15 * rng.standard_normal(n), 0, 1)
        tracks.append(p.astype(np.float32))
    return tracks


def gen_vadpost():
    print("VadPostprocessor (FireRed + MarbleNet flavours)")
    rng = np.random.default_rng(1234)
    tracks = prob_tracks(rng)
    out = {"n_cases": np.array(len(tracks))}
    ns_f = {"np": np}
    R.select_nodes("FireRedVAD/Inference_FireRed_ONNX.py", {"VadPostprocessor"}, ns_f,
                   consts={"_VAD_SILENCE", "_VAD_POSSIBLE_SPEECH", "_VAD_SPEECH", "_VAD_POSSIBLE_SILENCE",
                           "FRAME_SHIFT_MS", "FRAME_LENGTH_MS", "FRAME_SHIFT_S", "FRAME_LENGTH_S",
                           "_FRAME_SHIFT_F32", "_FRAME_LENGTH_F32"})
    ns_m = {"np": np}
    R.select_nodes("NVIDIA_Frame_VAD_Multilingual_MarbleNet/Inference_NVIDIA_MarbleNet_VAD_ONNX.py",
                   {"VadPostprocessor"}, ns_m,
                   consts={"_VAD_SILENCE", "_VAD_POSSIBLE_SPEECH", "_VAD_SPEECH", "_VAD_POSSIBLE_SILENCE"})
    cfgs_f = [(5, 0.4, 20, 2000, 20, 5, 0), (3, 0.5, 8, 300, 10, 4, 2), (1, 0.5, 0, 100, 0, 0, 0)]
    cfgs_m = [(3, 0.5, 10, 1000, 10, 3, 0, 0.02), (5, 0.45, 5, 200, 6, 2, 1, 0.02)]
    out["cfgs_f"] = np.array(cfgs_f, dtype=np.float64)
    out["cfgs_m"] = np.array(cfgs_m, dtype=np.float64)
    for i, p in enumerate(tracks):
        out[f"probs_{i}"] = p
        for c, cfg in enumerate(cfgs_f):
            pp = ns_f["VadPostprocessor"](*cfg)
            dec = pp.process(p.copy())
            wav = [None, len(p) * 0.01 + 0.013, len(p) * 0.01 + 0.5][c % 3] if len(p) else None
            seg = pp.decision_to_segment(dec, wav)
            out[f"f{c}_dec_{i}"] = np.asarray(dec, np.int8)
            out[f"f{c}_seg_{i}"] = np.array(seg, dtype=np.float64).reshape(-1, 2)
            out[f"f{c}_wav_{i}"] = np.array(-1.0 if wav is None else wav)
        for c, cfg in enumerate(cfgs_m):
            pp = ns_m["VadPostprocessor"](*cfg[:7], frame_shift_s=cfg[7])
            dec = pp.process(p.copy())
            wav = [len(p) * 0.02 - 0.007, None][c % 2] if len(p) else None
            seg = pp.decision_to_segment(dec, wav)
            out[f"m{c}_dec_{i}"] = np.asarray(dec, np.int8)
            out[f"m{c}_seg_{i}"] = np.array(seg, dtype=np.float64).reshape(-1, 2)
            out[f"m{c}_wav_{i}"] = np.array(-1.0 if wav is None else wav)
    save("vadpost", **out)


# ------------------------------------------------------------------------------------ Silero host side
def gen_silero_host():
    print("Silero get_speech_timestamps + OnnxWrapper (network replaced by a replay / the oracle net)")
    ns = {"torch": torch, "warnings": __import__("warnings"), "Callable": __import__("typing").Callable,
          "List": __import__("typing").List}
    R.select_nodes("Silero/modeling_modified/utils_vad.py", {"get_speech_timestamps", "OnnxWrapper"}, ns)

    class Replay:
        def __init__(self, probs):
            self.probs, self.i = probs, 0

        def reset_states(self):
            self.i = 0

        def __call__(self, chunk, sr):
            assert chunk.shape[-1] == 512
            v = self.probs[self.i]
            self.i += 1
            return torch.tensor([[v]], dtype=torch.float32)

    rng = np.random.default_rng(1234)
    out = {}
    cases = []
    for case in range(16):
        n_samples = int([160000, 89431, 512, 100, 700, 160000, 400000, 33000, 160000, 48000,
                         800000, 160000, 5120, 160000, 250000, 160000][case])
        n_win = (n_samples + 511) // 512
        kind = case % 4
        if kind == 0:
            p = rng.uniform(0, 1, n_win)
        elif kind == 1:
            p = np.zeros(n_win)
            pos, hot = 0, bool(case & 4)
            while pos < n_win:
                seg = int(rng.integers(3, 90))
                p[pos:pos + seg] = rng.uniform(0.5, 1.0, min(seg, n_win - pos)) if hot else rng.uniform(0, 0.4, min(seg, n_win - pos))
                pos += seg
                hot = not hot
        elif kind == 2:                                   # long speech with short dips -> max_speech split
            p = rng.uniform(0.55, 1.0, n_win)
            for _ in range(max(1, n_win // 60)):
                a = int(rng.integers(0, n_win))
                p[a:a + int(rng.integers(2, 7))] = rng.uniform(0.0, 0.3)
        else:
            p = np.clip(0.45 + 0.35 * np.sin(np.arange(n_win) / 7.0) + 0.1 * rng.standard_normal(n_win), 0, 1)
        p = p.astype(np.float32)
        kw = [dict(threshold=0.5, max_speech_duration_s=20, min_speech_duration_ms=250, min_silence_duration_ms=250, return_seconds=True),
              dict(threshold=0.5, max_speech_duration_s=6, min_speech_duration_ms=250, min_silence_duration_ms=100, return_seconds=False),
              dict(threshold=0.6, max_speech_duration_s=4, min_speech_duration_ms=100, min_silence_duration_ms=250, return_seconds=True,
                   use_max_poss_sil_at_max_speech=False),
              dict(threshold=0.5, return_seconds=False)][case % 4 if case < 12 else (case + 1) % 4]
        audio = torch.zeros(n_samples)
        res = ns["get_speech_timestamps"](audio, Replay([float(v) for v in p]), **kw)
        out[f"probs_{case}"] = p
        out[f"nsamp_{case}"] = np.array(n_samples)
        out[f"res_{case}"] = np.array([[d["start"], d["end"]] for d in res], dtype=np.float64).reshape(-1, 2)
        cases.append(repr(sorted(kw.items())))
    out["kwargs"] = np.array(cases)
    out["n_cases"] = np.array(len(cases))

    # OnnxWrapper state/context carry: onnxruntime stub whose session.run is the ORACLE network
    from oracle import silero as osil
    w = {k: T(v) for k, v in weights.silero_synthetic(1234).items()}
    calls = []

    class FakeSession:
        def run(self, _names, feeds):
            x, st = T(feeds["input"]), T(feeds["state"])
            assert int(feeds["sr"]) == 16000 and feeds["sr"].dtype == np.int64
            calls.append((feeds["input"].copy(), feeds["state"].copy()))
            o, s = osil.net_forward(w, x, st)
            return [o.numpy(), s.numpy()]

    wrapper = ns["OnnxWrapper"].__new__(ns["OnnxWrapper"])
    ns["np"] = np
    wrapper.session = FakeSession()
    wrapper.sample_rates = [16000]
    wrapper.reset_states()
    clip = weights.burst_clips(2, 8000, seed=11).astype(np.float32) * 0.000030517578
    probs = wrapper.audio_forward(torch.from_numpy(clip), 16000).numpy()
    out["wrap_audio"] = clip
    out["wrap_probs"] = probs
    out["wrap_inputs"] = np.stack([c[0] for c in calls])
    out["wrap_states"] = np.stack([c[1] for c in calls])
    out["wrap_final_state"] = wrapper._state.numpy()
    out["wrap_final_context"] = wrapper._context.numpy()
    save("silero_host", **out)


# ------------------------------------------------------------------------------------ FSMN
def gen_fsmn():
    print("FSMN_VAD wrapper + encoder + host loop")
    enc = R.load_module("FSMN/modeling_modified/encoder.py", "ref_fsmn_encoder")
    stft_mod = R.load_module("FSMN/STFT_Process.py", "ref_stft_v1")
    import torchaudio
    ns = {"torch": torch, "torchaudio": torchaudio, "math": __import__("math"), "np": np}
    R.select_nodes("FSMN/Export_FSMN_VAD.py", {"FSMN_VAD"}, ns)
    d = weights.FSMN_DIMS
    out = {}
    for seed in (1234, 7):
        w = weights.fsmn_synthetic(seed)
        net = enc.FSMN(d["input_dim"], d["input_affine_dim"], d["fsmn_layers"], d["linear_dim"], d["proj_dim"],
                       d["lorder"], 0, 1, 0, d["output_affine_dim"], d["output_dim"]).eval()
        sd = {"in_linear1.linear.weight": w["in1_w"], "in_linear1.linear.bias": w["in1_b"],
              "in_linear2.linear.weight": w["in2_w"], "in_linear2.linear.bias": w["in2_b"],
              "out_linear1.linear.weight": w["out1_w"], "out_linear1.linear.bias": w["out1_b"],
              "out_linear2.linear.weight": w["out2_w"], "out_linear2.linear.bias": w["out2_b"]}
        for l in range(4):
            sd[f"fsmn.{l}.linear.linear.weight"] = w[f"l{l}_lin_w"]
            sd[f"fsmn.{l}.fsmn_block.conv_left.weight"] = w[f"l{l}_fir_w"].reshape(128, 1, 20, 1)
            sd[f"fsmn.{l}.affine.linear.weight"] = w[f"l{l}_aff_w"]
            sd[f"fsmn.{l}.affine.linear.bias"] = w[f"l{l}_aff_b"]
        net.load_state_dict({k: T(v) for k, v in sd.items()}, strict=True)
        stft = stft_mod.STFT_Process(model_type="stft_B", n_fft=512, hop_len=160, win_length=400, max_frames=0, window_type="hamming").eval()
        L = 16000
        model = ns["FSMN_VAD"](net, stft, 512, L // 160 + 1, 80, 16000, 0.97, 5, 1, (L // 160 + 1), 1.0, L, 160,
                               T(w["cmvn_means"]).reshape(1, 1, -1), T(w["cmvn_vars"]).reshape(1, 1, -1))
        clip = weights.burst_clips(1, 3 * 11040 + 16000, seed=seed + 100)[0]
        peak = np.max(np.abs(clip.astype(np.float32)))
        clip = (clip.astype(np.float32) * float(32767.0 / peak)).astype(np.int16)
        caches = [torch.zeros(1, 128, 19, 1) for _ in range(4)]
        noise = np.array([4.0], dtype=np.float32)
        with torch.no_grad():
            for k in range(4):
                a = T(clip[k * 11040:k * 11040 + L].copy()).reshape(1, 1, -1)
                # raw encoder output (pre-threshold), for the float tolerance check
                aa = a.float()
                aa = aa - torch.mean(aa)
                aa = torch.cat([aa[:, :, :1], aa[:, :, 1:] - 0.97 * aa[:, :, :-1]], dim=-1)
                score, c0, c1, c2, c3, noisy = model(a, *caches, torch.tensor([1.0]), T(noise))
                out[f"s{seed}_score_{k}"] = score.numpy()
                out[f"s{seed}_noisy_{k}"] = np.array(noisy.numpy())
                out[f"s{seed}_noise_in_{k}"] = noise.copy()
                for ci, c in enumerate((c0, c1, c2, c3)):
                    out[f"s{seed}_cache{ci}_{k}"] = c.numpy()[0, :, :, 0]
                caches = [c0, c1, c2, c3]
                nd = noisy.numpy()
                if nd > 0.0:
                    noise = (0.5 * (noise + nd + 1.0)).astype(np.float32)
        out[f"s{seed}_clip"] = clip
    save("fsmn_forward", **out)

    # host loop: replay seeded uint8 score chunks through the reference's own module-level loop
    rng = np.random.default_rng(1234)
    hl = {}
    for case in range(6):
        n_chunks = [1, 2, 5, 15, 15, 3][case]
        scores = []
        for k in range(n_chunks):
            p = [0.5, 0.2, 0.8, 0.5, 0.35, 0.65][case]
            s = np.zeros(101, np.uint8)
            st = int(rng.integers(0, 2))
            for i in range(101):
                if rng.uniform() < 0.15:
                    st = 1 - st
                s[i] = st if rng.uniform() < 0.85 else int(rng.uniform() < p)
            scores.append(s)
        noisy_seq = [float(v) for v in rng.uniform(-0.5, 2.0, n_chunks)]

        class FakeSess:
            def __init__(self):
                self.k = 0

            def run(self, names, feeds):
                k = self.k
                self.k += 1
                z = np.zeros((1, 128, 19, 1), np.float32)
                return scores[k], z, z, z, z, np.float32(noisy_seq[k])

        aligned = (n_chunks - 1) * 11040 + 16000
        env = dict(np=np, time=__import__("time"), ort_session_A=FakeSess(), model_type="tensor(float)",
                   BACKGROUND_NOISE_dB_INIT=30.0, SNR_THRESHOLD=10.0, ONE_MINUS_SPEECH_THRESHOLD=1.0,
                   INPUT_AUDIO_LENGTH=16000, aligned_len=aligned, audio=np.zeros((1, 1, aligned), np.int16),
                   slide_range=71, look_backward=30, inv_look_backward=float(1.0 / 30), SPEAKING_SCORE=0.5,
                   SILENCE_SCORE=0.5, inv_audio_len=0.0, stride_step=11040, score_len=101,
                   print=lambda *a, **k: None)
        for i in range(7):
            env[f"in_name_A{i}"] = f"i{i}"
        for i in range(6):
            env[f"out_name_A{i}"] = f"o{i}"
        R.select_lines("FSMN/Inference_FSMN_VAD_ONNX.py", 156, 234, env)
        hl[f"scores_{case}"] = np.stack(scores)
        hl[f"noisy_{case}"] = np.array(noisy_seq, np.float32)
        hl[f"saved_{case}"] = np.array(env["saved"], dtype=bool)
        hl[f"noise_final_{case}"] = np.asarray(env["noise_average_dB"], np.float32)
    hl["n_cases"] = np.array(6)
    save("fsmn_hostloop", **hl)


# ------------------------------------------------------------------------------------ FireRed
def gen_firered():
    print("FireRedVAD_ONNX wrapper + DetectModel")
    stft_mod = R.load_module("FireRedVAD/STFT_Process.py", "ref_stft_v2f")
    ns = {"torch": torch, "math": __import__("math"), "np": np, "STFT_Process": stft_mod.STFT_Process}
    R.select_nodes("FireRedVAD/Export_FireRedVAD.py",
                   {"FSMN", "DFSMNBlock", "DFSMN", "DetectModel", "FireRedVAD_ONNX", "build_kaldi_mel_filterbank"}, ns)
    out = {}
    cfgs = {1234: weights.FIRERED_CFG, 7: dict(weights.FIRERED_CFG, R=3, M=2, H=64, P=32, N1=8, S1=2, N2=4, S2=3),
            9: dict(weights.FIRERED_CFG, R=2, M=1, H=48, P=24, N1=5, S1=1, N2=0, S2=0, odim=3)}
    for seed, cfg in cfgs.items():
        w = weights.firered_synthetic(seed, cfg)
        args = types.SimpleNamespace(**cfg)
        dm = ns["DetectModel"](args).eval()
        sd = {"dfsmn.fc1.0.weight": w["fc1_w"][:, :, None], "dfsmn.fc1.0.bias": w["fc1_b"],
              "dfsmn.fc2.0.weight": w["fc2_w"][:, :, None], "dfsmn.fc2.0.bias": w["fc2_b"],
              "dfsmn.fsmn1.lookback_filter.weight": w["fsmn0_lb"][:, None, :],
              "out.weight": w["out_w"][:, :, None], "out.bias": w["out_b"]}
        if cfg["N2"] > 0:
            sd["dfsmn.fsmn1.lookahead_filter.weight"] = w["fsmn0_la"][:, None, :]
        for r in range(1, cfg["R"]):
            p = f"dfsmn.fsmns.{r - 1}."
            sd[p + "fc1.0.weight"] = w[f"blk{r}_fc1_w"][:, :, None]
            sd[p + "fc1.0.bias"] = w[f"blk{r}_fc1_b"]
            sd[p + "fc2.weight"] = w[f"blk{r}_fc2_w"][:, :, None]
            sd[p + "fsmn.lookback_filter.weight"] = w[f"fsmn{r}_lb"][:, None, :]
            if cfg["N2"] > 0:
                sd[p + "fsmn.lookahead_filter.weight"] = w[f"fsmn{r}_la"][:, None, :]
        for m in range(cfg["M"]):
            sd[f"dfsmn.dnns.{2 * m}.weight"] = w[f"dnn{m}_w"][:, :, None]
            sd[f"dfsmn.dnns.{2 * m}.bias"] = w[f"dnn{m}_b"]
        dm.load_state_dict({k: T(v) for k, v in sd.items()}, strict=True)
        model = ns["FireRedVAD_ONNX"](dm, 400, 160, 400, 80, 16000, 0.97, "povey", 16000).eval()
        if seed == 1234:
            np.random.seed(1234)      # the reference's own validate_export input recipe (:1539-1543)
            a = np.random.randint(-8000, 8000, (1, 1, 16000)).astype(np.int16)
        else:
            a = weights.burst_clips(1, 16000, seed=seed)[0].reshape(1, 1, -1)
        with torch.no_grad():
            probs = model(T(a)).numpy()
        out[f"s{seed}_audio"] = a
        out[f"s{seed}_probs"] = probs
        if seed == 1234:
            out["kaldi_fbank"] = model.fbank_conv[:, :, 0].numpy()
    save("firered_forward", **out)


def gen_firered_stream():
    print("FireRedStreamVAD_ONNX wrapper + DetectModel_Streaming + StreamVadPostprocessor")
    stft_mod = R.load_module("FireRedVAD/STFT_Process.py", "ref_stft_v2s")
    ns = {"torch": torch, "math": __import__("math"), "np": np, "STFT_Process": stft_mod.STFT_Process}
    R.select_nodes("FireRedVAD/Export_FireRedVAD.py",
                   {"FSMN_Streaming", "DFSMNBlock_Streaming", "DFSMN_Streaming", "DetectModel_Streaming",
                    "FireRedStreamVAD_ONNX", "StreamVadPostprocessor", "build_kaldi_mel_filterbank"}, ns,
                   consts={"_VAD_SILENCE", "_VAD_POSSIBLE_SPEECH", "_VAD_SPEECH", "_VAD_POSSIBLE_SILENCE",
                           "FRAME_PER_SECONDS", "FRAME_SHIFT_MS", "FRAME_LENGTH_MS"})
    out = {}
    cfgs = {1234: dict(weights.FIRERED_CFG, N2=0, S2=0), 7: dict(weights.FIRERED_CFG, R=3, M=2, H=64, P=32, N1=8, S1=2, N2=0, S2=0)}
    for seed, cfg in cfgs.items():
        w = weights.firered_synthetic(seed, cfg)
        dm = ns["DetectModel_Streaming"](types.SimpleNamespace(**cfg)).eval()
        sd = {"dfsmn.fc1.0.weight": w["fc1_w"][:, :, None], "dfsmn.fc1.0.bias": w["fc1_b"],
              "dfsmn.fc2.0.weight": w["fc2_w"][:, :, None], "dfsmn.fc2.0.bias": w["fc2_b"],
              "dfsmn.fsmn1.lookback_filter.weight": w["fsmn0_lb"][:, None, :],
              "out.weight": w["out_w"][:, :, None], "out.bias": w["out_b"]}
        for r in range(1, cfg["R"]):
            p = f"dfsmn.fsmns.{r - 1}."
            sd[p + "fc1.0.weight"] = w[f"blk{r}_fc1_w"][:, :, None]
            sd[p + "fc1.0.bias"] = w[f"blk{r}_fc1_b"]
            sd[p + "fc2.weight"] = w[f"blk{r}_fc2_w"][:, :, None]
            sd[p + "fsmn.lookback_filter.weight"] = w[f"fsmn{r}_lb"][:, None, :]
        for m in range(cfg["M"]):
            sd[f"dfsmn.dnns.{2 * m}.weight"] = w[f"dnn{m}_w"][:, :, None]
            sd[f"dfsmn.dnns.{2 * m}.bias"] = w[f"dnn{m}_b"]
        dm.load_state_dict({k: T(v) for k, v in sd.items()}, strict=True)
        model = ns["FireRedStreamVAD_ONNX"](dm, 400, 160, 400, 80, 16000, 0.97, "povey", 16000).eval()
        # the chunk loop of the stream driver (Inference_FireRed_ONNX.py:788-808) on a ragged-length clip
        n = 2 * 16000 + 2560 * 3 + (300 if seed == 1234 else 1111)
        clip = weights.burst_clips(1, n, seed=seed)[0]
        caches = torch.zeros((cfg["R"], 1, cfg["P"], (cfg["N1"] - 1) * cfg["S1"]))
        probs, pos, first = [], 0, None
        with torch.no_grad():
            while pos < n:
                end = min(pos + 2560, n)
                chunk = clip[pos:end]
                if len(chunk) < 400:
                    chunk = np.pad(chunk, (0, 400 - len(chunk)), mode="constant")
                pr, caches = model(T(chunk.reshape(1, 1, -1).astype(np.int16)), caches)
                if first is None:
                    first = caches.numpy().copy()
                probs.append(pr[0, 0].numpy())
                pos = end
        out[f"s{seed}_clip"] = clip
        out[f"s{seed}_probs"] = np.concatenate(probs)
        out[f"s{seed}_caches_first"] = first
        out[f"s{seed}_caches_last"] = caches.numpy()
    # StreamVadPostprocessor on the shared probability tracks (+ max-speech split and chunked feeding)
    rng = np.random.default_rng(4321)
    tracks = prob_tracks(rng)
    pcfgs = [(5, 0.4, 5, 8, 2000, 20), (3, 0.5, 2, 4, 60, 6), (1, 0.5, 0, 1, 25, 1)]
    out["post_cfgs"] = np.array(pcfgs, dtype=np.float64)
    out["post_n"] = np.array(len(tracks))
    for i, p in enumerate(tracks):
        out[f"post_probs_{i}"] = p
        for c, cfg in enumerate(pcfgs):
            seg = ns["StreamVadPostprocessor"](*cfg).process_batch(p.copy())
            out[f"post{c}_seg_{i}"] = np.array(seg, dtype=np.float64).reshape(-1, 2)
            pp = ns["StreamVadPostprocessor"](*cfg)         # fed in 14-frame pieces: state carries over
            pieces = [np.array(pp.process_batch(p[k:k + 14].copy()), dtype=np.float64).reshape(-1, 2)
                      for k in range(0, len(p), 14)]
            out[f"post{c}_chunked_{i}"] = np.concatenate(pieces) if pieces else np.zeros((0, 2))
    save("firered_stream", **out)


def gen_firered_ckpt():
    print("FireRed checkpoint path: load_cmvn + DetectModel.from_pretrained (CMVN fused into fc1)")
    import tempfile
    ns = {"torch": torch, "math": __import__("math"), "np": np}
    R.select_nodes("FireRedVAD/Export_FireRedVAD.py", {"FSMN", "DFSMNBlock", "DFSMN", "DetectModel", "load_cmvn"}, ns)
    cfg = dict(weights.FIRERED_CFG, R=3, M=2, H=64, P=32, N1=8, S1=1, N2=4, S2=1, odim=3)
    w = weights.firered_synthetic(21, cfg)
    rng = np.random.default_rng(21)
    dim, count = cfg["idim"], 123456.0
    mean = rng.normal(12.0, 3.0, dim)
    var = rng.uniform(0.5, 9.0, dim)
    stats = np.zeros((2, dim + 1), dtype=np.float64)
    stats[0, :dim], stats[0, dim] = mean * count, count
    stats[1, :dim] = (var + mean * mean) * count
    # kaldiio is not installed: the reference only uses it to fetch the statistics matrix
    sys.modules["kaldiio"].load_mat = lambda _path: stats.copy()
    means, inv_std = ns["load_cmvn"]("unused")
    state = checkpoint_state(w, cfg)
    d = tempfile.mkdtemp()
    torch.save({"args": types.SimpleNamespace(**cfg), "model_state_dict": state}, os.path.join(d, "model.pth.tar"))
    model = ns["DetectModel"].from_pretrained(d, means, inv_std)
    sd = model.state_dict()
    save("firered_ckpt", stats=stats, means=means.numpy(), inv_std=inv_std.numpy(),
         fc1_w=sd["dfsmn.fc1.0.weight"][:, :, 0].numpy(), fc1_b=sd["dfsmn.fc1.0.bias"].numpy(),
         blk1_fc2_w=sd["dfsmn.fsmns.0.fc2.weight"][:, :, 0].numpy(), fsmn2_la=sd["dfsmn.fsmns.1.fsmn.lookahead_filter.weight"][:, 0, :].numpy(),
         dnn1_w=sd["dfsmn.dnns.2.weight"][:, :, 0].numpy(), out_w=sd["out.weight"][:, :, 0].numpy())


def checkpoint_state(w, cfg):
    """A FireRed `model_state_dict` as the training code saves it (Linear weights 2-D, FIR filters [P,1,N])."""
    sd = {"dfsmn.fc1.0.weight": w["fc1_w"], "dfsmn.fc1.0.bias": w["fc1_b"], "dfsmn.fc2.0.weight": w["fc2_w"],
          "dfsmn.fc2.0.bias": w["fc2_b"], "dfsmn.fsmn1.lookback_filter.weight": w["fsmn0_lb"][:, None, :],
          "out.weight": w["out_w"], "out.bias": w["out_b"]}
    if cfg["N2"] > 0:
        sd["dfsmn.fsmn1.lookahead_filter.weight"] = w["fsmn0_la"][:, None, :]
    for r in range(1, cfg["R"]):
        p = f"dfsmn.fsmns.{r - 1}."
        sd[p + "fc1.0.weight"], sd[p + "fc1.0.bias"] = w[f"blk{r}_fc1_w"], w[f"blk{r}_fc1_b"]
        sd[p + "fc2.weight"] = w[f"blk{r}_fc2_w"]
        sd[p + "fsmn.lookback_filter.weight"] = w[f"fsmn{r}_lb"][:, None, :]
        if cfg["N2"] > 0:
            sd[p + "fsmn.lookahead_filter.weight"] = w[f"fsmn{r}_la"][:, None, :]
    for m in range(cfg["M"]):
        sd[f"dfsmn.dnns.{2 * m}.weight"], sd[f"dfsmn.dnns.{2 * m}.bias"] = w[f"dnn{m}_w"], w[f"dnn{m}_b"]
    return {k: T(v) for k, v in sd.items()}


# ------------------------------------------------------------------------------------ DFSMN near+far
def gen_dfsmn():
    print("DFSMN_VAD wrapper + ICCRN + UniDeepFsmn (near+far)")
    import types as _t
    import torchaudio
    stft_mod = R.load_module("DFSMN/near_and_far_end_audio/STFT_Process.py", "ref_stft_v1b")
    # fake parent package so uni_deep_fsmn's relative import resolves
    pkg = _t.ModuleType("aecpkg")
    pkg.__path__ = []
    sys.modules["aecpkg"] = pkg
    lb = _t.ModuleType("aecpkg.layer_base")

    class LayerBase(torch.nn.Module):
        pass
    lb.LayerBase = LayerBase
    lb.expect_token_number = lb.expect_kaldi_matrix = lb.to_kaldi_matrix = lambda *a, **k: None
    sys.modules["aecpkg.layer_base"] = lb
    spec = __import__("importlib.util").util.spec_from_file_location(
        "aecpkg.uni_deep_fsmn", os.path.join(R.REF, "DFSMN/near_and_far_end_audio/modeling_modified/uni_deep_fsmn.py"))
    udf = __import__("importlib.util").util.module_from_spec(spec)
    sys.modules["aecpkg.uni_deep_fsmn"] = udf
    spec.loader.exec_module(udf)
    ns = {"torch": torch, "torchaudio": torchaudio, "np": np}
    R.select_nodes("DFSMN/near_and_far_end_audio/Export_DFSMN_VAD.py",
                   {"AlphaPredictor", "CFB", "CepsUnit", "LayerNorm", "NET", "CH_LSTM_T", "CH_LSTM_F", "DFSMN_VAD"}, ns,
                   consts={"NFFT_B", "WINDOW_LENGTH_B", "HOP_LENGTH_B", "ALPHA_K"})
    out = {}
    for seed in (1234, 7):
        w = weights.dfsmn_synthetic(seed)
        m = weights.DFSMN_MASK
        torch.manual_seed(0)
        net = ns["NET"](max_frames=200)
        missing = net.load_state_dict({k[len("iccrn."):]: T(v) for k, v in w.items() if k.startswith("iccrn.")}, strict=False)
        assert not missing.unexpected_keys and all(("kernel" in k or "basis" in k or "window_sum" in k) for k in missing.missing_keys), missing
        net = net.float().eval()
        alpha = ns["AlphaPredictor"](10)
        alpha.load_state_dict({k[len("alpha."):]: T(v) for k, v in w.items() if k.startswith("alpha.")}, strict=True)
        alpha = alpha.float().eval()

        class Mask(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.linear1 = torch.nn.Linear(240, m["hidden"])
                self.relu = torch.nn.ReLU()
                self.deepfsmn = torch.nn.Sequential(*[udf.UniDeepFsmn(m["hidden"], m["hidden"], m["lorder"], m["fsmn_hidden"])
                                                       for _ in range(m["layers"])])
                self.linear3 = torch.nn.Linear(m["hidden"], 1)
        mask = Mask()
        mask.load_state_dict({k[len("mask."):]: T(v) for k, v in w.items() if k.startswith("mask.") and k not in ("mask.shift", "mask.scale")}, strict=True)
        holder = _t.SimpleNamespace(model=mask.eval(), preprocessor=_t.SimpleNamespace(
            feature=_t.SimpleNamespace(shift=T(w["mask.shift"]), scale=T(w["mask.scale"]))))
        mk = lambda n, h, wl: stft_mod.STFT_Process(model_type="stft_B", n_fft=n, hop_len=h, win_length=wl, max_frames=0, window_type="hamming").eval()   # noqa: E731
        model = ns["DFSMN_VAD"](holder, net, alpha, mk(1024, 320, 640), mk(640, 320, 640), mk(319, 160, 319),
                                1024, 319, 10, 200, 0.97, 16000, 80)
        cap = {}
        net.register_forward_hook(lambda mod, inp, res: cap.update(x=inp[0].detach().clone(), y=res[0].detach().clone()))
        near = weights.burst_clips(1, 16001, seed=seed + 1)[0].reshape(1, 1, -1)
        far = weights.burst_clips(1, 16001, seed=seed + 2)[0].reshape(1, 1, -1)
        with torch.no_grad():
            vad = model(T(near), T(far))
        out[f"s{seed}_near"], out[f"s{seed}_far"] = near, far
        out[f"s{seed}_vad"] = vad.numpy()
        out[f"s{seed}_aec"] = cap["y"].numpy()[0, 0]
        out[f"s{seed}_iccrn_in_ds"] = cap["x"].numpy()[0, :, ::8, ::5]          # subsampled ICCRN input (alpha-scaled far)
    save("dfsmn_forward", **out)

    # host loop of the DFSMN driver on replayed float scores
    rng = np.random.default_rng(1234)
    hl = {}
    for case in range(5):
        n_chunks = [1, 2, 6, 15, 3][case]
        scores = [np.clip(0.5 + 0.4 * np.sin(np.arange(51) / (2.0 + case) + k) + 0.2 * rng.standard_normal(51), 0, 1).astype(np.float32)
                  for k in range(n_chunks)]
        if case == 4:
            scores[1][:] = 0.5                                         # exactly on both thresholds

        class FakeSess:
            def __init__(self):
                self.k = 0

            def run(self, names, feeds):
                self.k += 1
                return [scores[self.k - 1]]
        L, stride = 16001, 16001 - 16 * 320
        aligned = (n_chunks - 1) * stride + L
        env = dict(np=np, time=__import__("time"), ort_session_A=FakeSess(), out_name_A0="o", in_name_A0="a", in_name_A1="b",
                   near_end_audio=np.zeros((1, 1, aligned), np.int16), far_end_audio=np.zeros((1, 1, aligned), np.int16),
                   INPUT_AUDIO_LENGTH=L, aligned_len=aligned, look_backward=15, stride_step=stride, SPEAKING_SCORE=0.5,
                   SILENCE_SCORE=0.5, inv_audio_len=0.0, print=lambda *a, **k: None)
        R.select_lines("DFSMN/near_and_far_end_audio/Inference_DFSMN_VAD_ONNX.py", 221, 273, env)
        hl[f"scores_{case}"] = np.stack(scores)
        hl[f"saved_{case}"] = np.array(env["saved"], dtype=bool)
    hl["n_cases"] = np.array(5)
    save("dfsmn_hostloop", **hl)


def gen_dfsmn_near_only():
    print("DFSMN_VAD wrapper, near-end-only variant (baked white-noise far end)")
    import types as _t
    import torchaudio
    D = "DFSMN/only_near_end_audio/"
    stft_mod = R.load_module(D + "STFT_Process.py", "ref_stft_v1c")
    pkg = _t.ModuleType("aecpkg2")
    pkg.__path__ = []
    sys.modules["aecpkg2"] = pkg
    lb = _t.ModuleType("aecpkg2.layer_base")

    class LayerBase(torch.nn.Module):
        pass
    lb.LayerBase = LayerBase
    lb.expect_token_number = lb.expect_kaldi_matrix = lb.to_kaldi_matrix = lambda *a, **k: None
    sys.modules["aecpkg2.layer_base"] = lb
    spec = __import__("importlib.util").util.spec_from_file_location(
        "aecpkg2.uni_deep_fsmn", os.path.join(R.REF, D + "modeling_modified/uni_deep_fsmn.py"))
    udf = __import__("importlib.util").util.module_from_spec(spec)
    sys.modules["aecpkg2.uni_deep_fsmn"] = udf
    spec.loader.exec_module(udf)
    ns = {"torch": torch, "torchaudio": torchaudio, "np": np}
    R.select_nodes(D + "Export_DFSMN_VAD.py",
                   {"AlphaPredictor", "CFB", "CepsUnit", "LayerNorm", "NET", "CH_LSTM_T", "CH_LSTM_F", "DFSMN_VAD"}, ns,
                   consts={"NFFT_B", "WINDOW_LENGTH_B", "HOP_LENGTH_B", "ALPHA_K"})
    seed = 1234
    w = weights.dfsmn_synthetic(seed)
    m = weights.DFSMN_MASK
    torch.manual_seed(0)
    net = ns["NET"](max_frames=200)
    net.load_state_dict({k[len("iccrn."):]: T(v) for k, v in w.items() if k.startswith("iccrn.")}, strict=False)
    net = net.float().eval()
    alpha = ns["AlphaPredictor"](10)
    alpha.load_state_dict({k[len("alpha."):]: T(v) for k, v in w.items() if k.startswith("alpha.")}, strict=True)
    alpha = alpha.float().eval()

    class Mask(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.linear1 = torch.nn.Linear(240, m["hidden"])
            self.relu = torch.nn.ReLU()
            self.deepfsmn = torch.nn.Sequential(*[udf.UniDeepFsmn(m["hidden"], m["hidden"], m["lorder"], m["fsmn_hidden"])
                                                   for _ in range(m["layers"])])
            self.linear3 = torch.nn.Linear(m["hidden"], 1)
    mask = Mask()
    mask.load_state_dict({k[len("mask."):]: T(v) for k, v in w.items() if k.startswith("mask.") and k not in ("mask.shift", "mask.scale")}, strict=True)
    holder = _t.SimpleNamespace(model=mask.eval(), preprocessor=_t.SimpleNamespace(
        feature=_t.SimpleNamespace(shift=T(w["mask.shift"]), scale=T(w["mask.scale"]))))
    mk = lambda n, h, wl: stft_mod.STFT_Process(model_type="stft_B", n_fft=n, hop_len=h, win_length=wl, max_frames=0, window_type="hamming").eval()   # noqa: E731
    torch.manual_seed(4321)              # the export draws its white-noise constants here (Export_DFSMN_VAD.py:309-310)
    model = ns["DFSMN_VAD"](holder, net, alpha, mk(1024, 320, 640), mk(640, 320, 640), mk(319, 160, 319),
                            1024, 319, 10, 200, 0.97, 16000, 80)
    near = weights.burst_clips(1, 16001, seed=seed + 1)[0].reshape(1, 1, -1)
    with torch.no_grad():
        # the wrapper calls .unsqueeze on a shape entry: it only runs under tracing (as in torch.onnx.export), where
        # sizes are tensors -- so the module is traced on the input and the traced graph evaluated
        traced = torch.jit.trace(model, (T(near),), check_trace=False)
        vad = traced(T(near))
    save("dfsmn_near_only", near=near, vad=vad.numpy(),
         pow_far=model.pow_far_white_noise[0, 0, :, :101].numpy(), far_comp=model.far_comp_white_noise[0, 0, :, :, :101].numpy())


# ------------------------------------------------------------------------------------ MarbleNet BN fold
def gen_marblenet_fold():
    print("MarbleNet fold_bn_into_conv1d")
    ns = {"torch": torch}
    R.select_nodes("NVIDIA_Frame_VAD_Multilingual_MarbleNet/Export_NVIDIA_MarbleNet_VAD.py", {"fold_bn_into_conv1d"}, ns)
    out = {}
    torch.manual_seed(1234)
    cases = [(80, 128, 1, False, 1, 1e-3), (64, 64, 1, True, 1, 1e-5), (64, 64, 13, False, 64, 1e-3), (128, 2, 3, True, 1, 1e-3)]
    for i, (ci, co, k, bias, groups, eps) in enumerate(cases):
        conv = torch.nn.Conv1d(ci, co, k, groups=groups, bias=bias).eval()
        bn = torch.nn.BatchNorm1d(co, eps=eps).eval()
        with torch.no_grad():
            bn.weight.copy_(1 + 0.1 * torch.randn(co))
            bn.bias.copy_(0.1 * torch.randn(co))
            bn.running_mean.copy_(0.2 * torch.randn(co))
            bn.running_var.copy_(0.5 + torch.rand(co))
        f = ns["fold_bn_into_conv1d"](conv, bn)
        out[f"w_{i}"] = conv.weight.detach().numpy()
        out[f"b_{i}"] = conv.bias.detach().numpy() if bias else np.zeros(0, np.float32)
        out[f"has_bias_{i}"] = np.array(bias)
        out[f"gamma_{i}"], out[f"beta_{i}"] = bn.weight.detach().numpy(), bn.bias.detach().numpy()
        out[f"mean_{i}"], out[f"var_{i}"], out[f"eps_{i}"] = bn.running_mean.numpy(), bn.running_var.numpy(), np.array(eps)
        out[f"fw_{i}"], out[f"fb_{i}"] = f.weight.detach().numpy(), f.bias.detach().numpy()
    out["n_cases"] = np.array(len(cases))
    save("marblenet_fold", **out)



# ------------------------------------------------------------------------------------ round-2 additions
def _fsmn_reference_model(seed, ratio):
    enc = R.load_module("FSMN/modeling_modified/encoder.py", "ref_fsmn_encoder")
    stft_mod = R.load_module("FSMN/STFT_Process.py", "ref_stft_v1")
    import torchaudio
    ns = {"torch": torch, "torchaudio": torchaudio, "math": __import__("math"), "np": np}
    R.select_nodes("FSMN/Export_FSMN_VAD.py", {"FSMN_VAD"}, ns)
    d = weights.FSMN_DIMS
    w = weights.fsmn_synthetic(seed)
    net = enc.FSMN(d["input_dim"], d["input_affine_dim"], d["fsmn_layers"], d["linear_dim"], d["proj_dim"],
                   d["lorder"], 0, 1, 0, d["output_affine_dim"], d["output_dim"]).eval()
    sd = {"in_linear1.linear.weight": w["in1_w"], "in_linear1.linear.bias": w["in1_b"],
          "in_linear2.linear.weight": w["in2_w"], "in_linear2.linear.bias": w["in2_b"],
          "out_linear1.linear.weight": w["out1_w"], "out_linear1.linear.bias": w["out1_b"],
          "out_linear2.linear.weight": w["out2_w"], "out_linear2.linear.bias": w["out2_b"]}
    for l in range(4):
        sd[f"fsmn.{l}.linear.linear.weight"] = w[f"l{l}_lin_w"]
        sd[f"fsmn.{l}.fsmn_block.conv_left.weight"] = w[f"l{l}_fir_w"].reshape(128, 1, 20, 1)
        sd[f"fsmn.{l}.affine.linear.weight"] = w[f"l{l}_aff_w"]
        sd[f"fsmn.{l}.affine.linear.bias"] = w[f"l{l}_aff_b"]
    net.load_state_dict({k: T(v) for k, v in sd.items()}, strict=True)
    stft = stft_mod.STFT_Process(model_type="stft_B", n_fft=512, hop_len=160, win_length=400, max_frames=0, window_type="hamming").eval()
    L = 16000
    return ns["FSMN_VAD"](net, stft, 512, L // 160 + 1, 80, 16000, 0.97, 5, 1, (L // 160 + 1), ratio, L, 160,
                          T(w["cmvn_means"]).reshape(1, 1, -1), T(w["cmvn_vars"]).reshape(1, 1, -1))


def gen_fsmn_extra():
    """SPEECH_2_NOISE_RATIO != 1 branches of the FSMN graph (Export_FSMN_VAD.py:87-92) and the LOOK_BACKWARD = 0 edge of
    the host loop (Inference_FSMN_VAD_ONNX.py:79-86: slide_range is taken before look_backward is bumped to 1)."""
    print("FSMN extras: speech_2_noise_ratio 0.5 / 2.0, look_backward 0")
    out = {}
    clip = weights.burst_clips(1, 16000 + 11040, seed=1357)[0]
    clip = (clip.astype(np.float32) * float(32767.0 / np.max(np.abs(clip.astype(np.float32))))).astype(np.int16)
    out["clip"] = clip
    for tag, ratio, thrs in (("r05", 0.5, (1.5, 1.45)), ("r20", 2.0, (1.0, 0.7))):
        model = _fsmn_reference_model(1234, ratio)
        caches = [torch.zeros(1, 128, 19, 1) for _ in range(4)]
        with torch.no_grad():
            for k in range(2):
                a = T(clip[k * 11040:k * 11040 + 16000].copy()).reshape(1, 1, -1)
                score, c0, c1, c2, c3, noisy = model(a, *caches, torch.tensor([thrs[k]]), torch.tensor([4.0]))
                out[f"{tag}_score_{k}"] = score.numpy()
                out[f"{tag}_noisy_{k}"] = np.array(noisy.numpy())
                out[f"{tag}_thr_{k}"] = np.array(thrs[k], np.float32)
                caches = [c0, c1, c2, c3]
        out[f"{tag}_ratio"] = np.array(ratio)
    # host loop with LOOK_BACKWARD = 0.0: look_backward 0 -> stride = L - 160, slide_range = score_len, vote length 1, empty tail
    rng = np.random.default_rng(4321)
    for case in range(3):
        n_chunks = [1, 3, 6][case]
        scores = []
        for k in range(n_chunks):
            s = np.zeros(101, np.uint8)
            st = int(rng.integers(0, 2))
            for i in range(101):
                if rng.uniform() < 0.2:
                    st = 1 - st
                s[i] = st
            scores.append(s)
        noisy_seq = [float(v) for v in rng.uniform(-0.5, 2.0, n_chunks)]

        class FakeSess:
            def __init__(self):
                self.k = 0

            def run(self, names, feeds):
                k = self.k
                self.k += 1
                z = np.zeros((1, 128, 19, 1), np.float32)
                return scores[k], z, z, z, z, np.float32(noisy_seq[k])

        stride = 16000 - 160
        aligned = (n_chunks - 1) * stride + 16000
        # the reference's own derivation (:79-86) with LOOK_BACKWARD = 0.0
        env = dict(np=np, LOOK_BACKWARD=0.0, SAMPLE_RATE=16000, OUTPUT_FRAME_LENGTH=160, INPUT_AUDIO_LENGTH=16000, score_len=101)
        R.select_lines("FSMN/Inference_FSMN_VAD_ONNX.py", 79, 86, env)
        assert (env["look_backward"], env["stride_step"], env["slide_range"]) == (1, stride, 101)
        env.update(time=__import__("time"), ort_session_A=FakeSess(), model_type="tensor(float)",
                   BACKGROUND_NOISE_dB_INIT=30.0, SNR_THRESHOLD=10.0, ONE_MINUS_SPEECH_THRESHOLD=1.0,
                   aligned_len=aligned, audio=np.zeros((1, 1, aligned), np.int16), SPEAKING_SCORE=0.5,
                   SILENCE_SCORE=0.5, inv_audio_len=0.0, print=lambda *a, **k: None)
        for i in range(7):
            env[f"in_name_A{i}"] = f"i{i}"
        for i in range(6):
            env[f"out_name_A{i}"] = f"o{i}"
        R.select_lines("FSMN/Inference_FSMN_VAD_ONNX.py", 156, 234, env)
        out[f"lb0_scores_{case}"] = np.stack(scores)
        out[f"lb0_noisy_{case}"] = np.array(noisy_seq, np.float32)
        out[f"lb0_saved_{case}"] = np.array(env["saved"], dtype=bool)
    out["lb0_n_cases"] = np.array(3)
    out["lb0_stride"] = np.array(16000 - 160)
    save("fsmn_extra", **out)


def gen_hostloop_thresholds():
    """Round 3: the asymmetric SPEAKING_SCORE / SILENCE_SCORE branches of both look-ahead host loops (every earlier fixture used the
    defaults 0.5 / 0.5): FSMN/Inference_FSMN_VAD_ONNX.py:21-23,188-215 on replayed uint8 scores and
    DFSMN/near_and_far_end_audio/Inference_DFSMN_VAD_ONNX.py:22-27,231-258 on replayed float scores, the reference's own
    module-level loops executed with the constants overridden."""
    rng = np.random.default_rng(4321)
    out = {}
    pairs = [(0.7, 0.3), (0.3, 0.7), (0.9, 0.1), (0.6, 0.6), (0.2, 0.4)]
    out["pairs"] = np.array(pairs, np.float64)
    for case, (spk, sil) in enumerate(pairs):
        # ---- FSMN: 6 chunks of uint8 scores with persistent runs
        n_chunks = 6
        scores = []
        st = int(rng.integers(0, 2))
        for k in range(n_chunks):
            s_ = np.zeros(101, np.uint8)
            for i in range(101):
                if rng.uniform() < 0.08:
                    st = 1 - st
                s_[i] = st if rng.uniform() < 0.75 else int(rng.uniform() < 0.5)
            scores.append(s_)
        noisy_seq = [float(v) for v in rng.uniform(-0.5, 2.0, n_chunks)]

        class FakeSess:
            def __init__(self):
                self.k = 0

            def run(self, names, feeds):
                k = self.k
                self.k += 1
                z = np.zeros((1, 128, 19, 1), np.float32)
                return scores[k], z, z, z, z, np.float32(noisy_seq[k])

        aligned = (n_chunks - 1) * 11040 + 16000
        env = dict(np=np, time=__import__("time"), ort_session_A=FakeSess(), model_type="tensor(float)",
                   BACKGROUND_NOISE_dB_INIT=30.0, SNR_THRESHOLD=10.0, ONE_MINUS_SPEECH_THRESHOLD=1.0,
                   INPUT_AUDIO_LENGTH=16000, aligned_len=aligned, audio=np.zeros((1, 1, aligned), np.int16),
                   slide_range=71, look_backward=30, inv_look_backward=float(1.0 / 30), SPEAKING_SCORE=spk,
                   SILENCE_SCORE=sil, inv_audio_len=0.0, stride_step=11040, score_len=101,
                   print=lambda *a, **k: None)
        for i in range(7):
            env[f"in_name_A{i}"] = f"i{i}"
        for i in range(6):
            env[f"out_name_A{i}"] = f"o{i}"
        R.select_lines("FSMN/Inference_FSMN_VAD_ONNX.py", 156, 234, env)
        out[f"fsmn_scores_{case}"] = np.stack(scores)
        out[f"fsmn_noisy_{case}"] = np.array(noisy_seq, np.float32)
        out[f"fsmn_saved_{case}"] = np.array(env["saved"], dtype=bool)
        # ---- DFSMN: 6 chunks of float scores swinging through both thresholds
        fsc = [np.clip(0.5 + 0.45 * np.sin(np.arange(51) / (1.5 + case) + 0.7 * k) + 0.15 * rng.standard_normal(51), 0, 1).astype(np.float32)
               for k in range(n_chunks)]
        fsc[2][10:20] = np.float32(spk)                                  # exactly on the thresholds
        fsc[3][5:15] = np.float32(sil)

        class FakeSessD:
            def __init__(self):
                self.k = 0

            def run(self, names, feeds):
                self.k += 1
                return [fsc[self.k - 1]]
        L, stride = 16001, 16001 - 16 * 320
        aligned = (n_chunks - 1) * stride + L
        env = dict(np=np, time=__import__("time"), ort_session_A=FakeSessD(), out_name_A0="o", in_name_A0="a", in_name_A1="b",
                   near_end_audio=np.zeros((1, 1, aligned), np.int16), far_end_audio=np.zeros((1, 1, aligned), np.int16),
                   INPUT_AUDIO_LENGTH=L, aligned_len=aligned, look_backward=15, stride_step=stride, SPEAKING_SCORE=spk,
                   SILENCE_SCORE=sil, inv_audio_len=0.0, print=lambda *a, **k: None)
        R.select_lines("DFSMN/near_and_far_end_audio/Inference_DFSMN_VAD_ONNX.py", 221, 273, env)
        out[f"dfsmn_scores_{case}"] = np.stack(fsc)
        out[f"dfsmn_saved_{case}"] = np.array(env["saved"], dtype=bool)
    save("hostloop_thresholds", **out)


def gen_host_extra():
    """normalise_audio (RMS -> 8192 with clipping, Inference_NVIDIA_MarbleNet_VAD_ONNX.py:110-118; FireRed carries the same
    function) -- the optional NORMALIZE_AUDIO path of the MarbleNet / FireRed drivers."""
    print("host extras: normalise_audio")
    ns = {"np": np}
    R.select_nodes("NVIDIA_Frame_VAD_Multilingual_MarbleNet/Inference_NVIDIA_MarbleNet_VAD_ONNX.py", {"normalise_audio"}, ns)
    ns2 = {"np": np}
    R.select_nodes("FireRedVAD/Inference_FireRed_ONNX.py", {"normalise_audio"}, ns2)
    rng = np.random.default_rng(99)
    out = {}
    cases = [(rng.standard_normal(5000) * 300).astype(np.int16),            # quiet -> scaled up, no clipping
             (rng.standard_normal(5000) * 9000).astype(np.int16),           # loud -> scaled down
             np.concatenate([(rng.standard_normal(3000) * 20).astype(np.int16), np.array([32767, -32768, 30000] * 5, np.int16)]),  # clips
             np.zeros(100, np.int16),                                       # rms == 0 -> returned unchanged
             np.array([7], np.int16)]
    for i, a in enumerate(cases):
        got = ns["normalise_audio"](a.copy())
        got2 = ns2["normalise_audio"](a.copy())
        assert np.array_equal(got, got2) and got.dtype == got2.dtype
        out[f"in_{i}"] = a
        out[f"out_{i}"] = got
    out["target_4096_in"] = cases[0]
    out["target_4096_out"] = ns["normalise_audio"](cases[0].copy(), 4096.0)
    out["n_cases"] = np.array(len(cases))
    save("host_extra", **out)


def gen_marblenet_hostloop():
    """The MarbleNet driver's window grid + loop (Inference_NVIDIA_MarbleNet_VAD_ONNX.py:130-147, 369-388) in STATIC-window mode
    (integer in the input shape), run on a fake session whose `run` is the oracle network, np.random.normal replayed."""
    print("MarbleNet host loop (static window)")
    from oracle import marblenet as omb
    w = {k: T(v) for k, v in weights.marblenet_synthetic(1234).items()}
    fe = omb.Frontend()
    out = {}
    for case, (n, L) in enumerate([(40000, 16000), (16000, 16000), (9000, 16000), (50001, 24000)]):
        clip = weights.burst_clips(1, n, seed=200 + case)[0]
        noise = np.random.default_rng(300 + case).standard_normal(40000)
        calls = []

        class FakeSess:
            class _M:
                shape = [1, 1, L]
            _inputs_meta = [_M()]

            def run(self, names, feeds):
                a = feeds["audio"]
                calls.append(a.shape[-1])
                sil, act, slen = omb.forward(fe, w, T(np.ascontiguousarray(a)))
                return sil.numpy(), act.numpy(), np.array([int(slen)], np.int32)

        class FakeRandom:
            @staticmethod
            def normal(loc=0.0, scale=1.0, size=None):
                k = int(np.prod(size))
                return noise[:k].reshape(size)

        fake_np = types.SimpleNamespace(**{k: getattr(np, k) for k in ("ceil", "sqrt", "mean", "concatenate", "float32", "zeros")})
        fake_np.random = FakeRandom
        env = dict(np=fake_np, ort_session_A=FakeSess(), IN_SAMPLE_RATE=16000, audio=clip.reshape(1, 1, -1).copy(), audio_len=n,
                   time=__import__("time"), in_name_A0="audio", out_name_A0="score_silence", out_name_A1="score_active",
                   out_name_A2="signal_len", print=lambda *a, **k: None)
        R.select_lines("NVIDIA_Frame_VAD_Multilingual_MarbleNet/Inference_NVIDIA_MarbleNet_VAD_ONNX.py", 130, 147, env)
        R.select_lines("NVIDIA_Frame_VAD_Multilingual_MarbleNet/Inference_NVIDIA_MarbleNet_VAD_ONNX.py", 369, 388, env)
        out[f"clip_{case}"] = clip
        out[f"noise_seed_{case}"] = np.array(300 + case)         # noise = default_rng(seed).standard_normal(40000)
        out[f"window_{case}"] = np.array(L)
        out[f"aligned_{case}"] = env["audio"][0, 0]
        out[f"probs_{case}"] = np.asarray(env["all_vad_probs"], np.float32)
        out[f"calls_{case}"] = np.array(calls)
    out["n_cases"] = np.array(4)
    save("marblenet_hostloop", **out)



def gen_resample():
    """In-graph linear resample of exports built with IN_SAMPLE_RATE != 16000 (F.interpolate before / after the pre-emphasis):
    the FireRedVAD_ONNX wrapper end to end, and the front half of NVIDIA_VAD_Optimized (audio -> log-mel) captured at the
    encoder's input (the NeMo encoder itself is not in the reference tree: a pass-through stand-in receives the features)."""
    print("in-graph resample (FireRed wrapper, MarbleNet front half)")
    out = {}
    stft_mod = R.load_module("FireRedVAD/STFT_Process.py", "ref_stft_v2f")
    ns = {"torch": torch, "math": __import__("math"), "np": np, "STFT_Process": stft_mod.STFT_Process}
    R.select_nodes("FireRedVAD/Export_FireRedVAD.py",
                   {"FSMN", "DFSMNBlock", "DFSMN", "DetectModel", "FireRedVAD_ONNX", "build_kaldi_mel_filterbank"}, ns)
    cfg = dict(weights.FIRERED_CFG, R=3, M=1, H=64, P=32, N1=8, S1=1, N2=4, S2=1)
    w = weights.firered_synthetic(7, cfg)
    dm = ns["DetectModel"](types.SimpleNamespace(**cfg)).eval()
    sd = {"dfsmn.fc1.0.weight": w["fc1_w"][:, :, None], "dfsmn.fc1.0.bias": w["fc1_b"],
          "dfsmn.fc2.0.weight": w["fc2_w"][:, :, None], "dfsmn.fc2.0.bias": w["fc2_b"],
          "dfsmn.fsmn1.lookback_filter.weight": w["fsmn0_lb"][:, None, :], "dfsmn.fsmn1.lookahead_filter.weight": w["fsmn0_la"][:, None, :],
          "out.weight": w["out_w"][:, :, None], "out.bias": w["out_b"]}
    for r in range(1, cfg["R"]):
        p = f"dfsmn.fsmns.{r - 1}."
        sd[p + "fc1.0.weight"], sd[p + "fc1.0.bias"] = w[f"blk{r}_fc1_w"][:, :, None], w[f"blk{r}_fc1_b"]
        sd[p + "fc2.weight"] = w[f"blk{r}_fc2_w"][:, :, None]
        sd[p + "fsmn.lookback_filter.weight"] = w[f"fsmn{r}_lb"][:, None, :]
        sd[p + "fsmn.lookahead_filter.weight"] = w[f"fsmn{r}_la"][:, None, :]
    sd["dfsmn.dnns.0.weight"], sd["dfsmn.dnns.0.bias"] = w["dnn0_w"][:, :, None], w["dnn0_b"]
    dm.load_state_dict({k: T(v) for k, v in sd.items()}, strict=True)
    out["firered_cfg"] = np.array([cfg[k] for k in ("idim", "R", "M", "H", "P", "N1", "S1", "N2", "S2", "odim")])
    for rate, n in ((8000, 8000), (48000, 48000), (44100, 44100), (22050, 12345)):
        model = ns["FireRedVAD_ONNX"](dm, 400, 160, 400, 80, 16000, 0.97, "povey", rate).eval()
        a = weights.burst_clips(1, n, seed=rate).reshape(1, 1, -1)
        with torch.no_grad():
            out[f"firered_{rate}_probs"] = model(T(a)).numpy()
        out[f"firered_{rate}_audio"] = a

    # MarbleNet: the reference wrapper with a stand-in `nvidia_vad` whose encoder hands back the log-mel features it is given
    stft_v2 = R.load_module("NVIDIA_Frame_VAD_Multilingual_MarbleNet/STFT_Process.py", "ref_stft_v2")
    import torchaudio
    ns2 = {"torch": torch, "F": torch.nn.functional, "torchaudio": torchaudio, "np": np}
    R.select_nodes("NVIDIA_Frame_VAD_Multilingual_MarbleNet/Export_NVIDIA_MarbleNet_VAD.py",
                   {"NVIDIA_VAD_Optimized", "fold_encoder_batchnorms", "fold_bn_into_conv1d"}, ns2)

    class Enc(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.encoder = torch.nn.ModuleList()
            self.seen = None

        def forward(self, packed):
            feats, length = packed
            self.seen = feats[0]
            return feats[0], length

    class Dec(torch.nn.Module):
        def forward(self, x):
            return x[..., :2]

    class Stand(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.encoder, self.decoder = Enc(), Dec()

    for rate, n in ((8000, 8000), (48000, 48000), (32000, 20001)):
        stft = stft_v2.STFT_Process("stft_B", n_fft=512, win_length=400, hop_len=160, max_frames=0, window_type="hann_sym",
                                    center_pad=True, pad_mode="constant").eval()
        net = Stand()
        model = ns2["NVIDIA_VAD_Optimized"](net, stft, 512, 80, 16000, 0.97, rate).eval()
        a = weights.burst_clips(1, n, seed=rate + 1).reshape(1, 1, -1)
        with torch.no_grad():
            model(T(a))
        out[f"marble_{rate}_audio"] = a
        out[f"marble_{rate}_logmel"] = net.encoder.seen.numpy()
    save("resample", **out)



def gen_silero_8k():
    """get_speech_timestamps at sampling_rate = 8000 (256-sample windows, utils_vad.py:345) on replayed probabilities, and the
    OnnxWrapper's 8 kHz plumbing (256 + 32-sample context, :99,:115): what the session is FED, with a stand-in network."""
    print("Silero 8 kHz branch (segmenter + wrapper plumbing)")
    ns = {"torch": torch, "warnings": __import__("warnings"), "Callable": __import__("typing").Callable,
          "List": __import__("typing").List, "np": np}
    R.select_nodes("Silero/modeling_modified/utils_vad.py", {"get_speech_timestamps", "OnnxWrapper"}, ns)

    class Replay:
        def __init__(self, probs):
            self.probs, self.i = probs, 0

        def reset_states(self):
            self.i = 0

        def __call__(self, chunk, sr):
            assert chunk.shape[-1] == 256 and sr == 8000
            v = self.probs[self.i]
            self.i += 1
            return torch.tensor([[v]], dtype=torch.float32)

    rng = np.random.default_rng(808)
    out, cases = {}, []
    for case in range(8):
        n_samples = int([80000, 44715, 256, 100, 300, 200000, 80000, 16000][case])
        n_win = (n_samples + 255) // 256
        kind = case % 4
        if kind == 0:
            p = rng.uniform(0, 1, n_win)
        elif kind == 1:
            p = np.zeros(n_win)
            pos, hot = 0, bool(case & 4)
            while pos < n_win:
                seg = int(rng.integers(3, 90))
                p[pos:pos + seg] = rng.uniform(0.5, 1.0, min(seg, n_win - pos)) if hot else rng.uniform(0, 0.4, min(seg, n_win - pos))
                pos += seg
                hot = not hot
        elif kind == 2:
            p = rng.uniform(0.55, 1.0, n_win)
            for _ in range(max(1, n_win // 60)):
                a = int(rng.integers(0, n_win))
                p[a:a + int(rng.integers(2, 7))] = rng.uniform(0.0, 0.3)
        else:
            p = np.clip(0.45 + 0.35 * np.sin(np.arange(n_win) / 7.0) + 0.1 * rng.standard_normal(n_win), 0, 1)
        p = p.astype(np.float32)
        kw = [dict(threshold=0.5, sampling_rate=8000, max_speech_duration_s=20, min_speech_duration_ms=250, min_silence_duration_ms=250, return_seconds=True),
              dict(threshold=0.5, sampling_rate=8000, max_speech_duration_s=6, min_speech_duration_ms=250, min_silence_duration_ms=100, return_seconds=False),
              dict(threshold=0.6, sampling_rate=8000, max_speech_duration_s=4, min_speech_duration_ms=100, min_silence_duration_ms=250, return_seconds=True,
                   use_max_poss_sil_at_max_speech=False),
              dict(threshold=0.5, sampling_rate=8000, return_seconds=False)][case % 4]
        res = ns["get_speech_timestamps"](torch.zeros(n_samples), Replay([float(v) for v in p]), **kw)
        out[f"probs_{case}"] = p
        out[f"nsamp_{case}"] = np.array(n_samples)
        out[f"res_{case}"] = np.array([[d["start"], d["end"]] for d in res], dtype=np.float64).reshape(-1, 2)
        cases.append(repr(sorted(kw.items())))
    out["kwargs"] = np.array(cases)
    out["n_cases"] = np.array(len(cases))

    fed = []

    class FakeSession:
        def run(self, _names, feeds):
            assert int(feeds["sr"]) == 8000 and feeds["input"].shape[1] == 256 + 32
            fed.append(feeds["input"].copy())
            return [np.full((feeds["input"].shape[0], 1), 0.25, np.float32), feeds["state"] + np.float32(1.0)]

    wrapper = ns["OnnxWrapper"].__new__(ns["OnnxWrapper"])
    wrapper.session = FakeSession()
    wrapper.sample_rates = [8000, 16000]
    wrapper.reset_states()
    clip = (weights.burst_clips(2, 1000, seed=12).astype(np.float32) * 0.000030517578)
    probs = wrapper.audio_forward(torch.from_numpy(clip), 8000).numpy()
    out["wrap_audio"], out["wrap_probs"], out["wrap_inputs"] = clip, probs, np.stack(fed)
    out["wrap_final_state"], out["wrap_final_context"] = wrapper._state.numpy(), wrapper._context.numpy()
    save("silero_8k", **out)


if __name__ == "__main__":
    which = set(sys.argv[1:])
    gens = dict(stft=gen_stft, host=gen_host, vadpost=gen_vadpost, silero_host=gen_silero_host,
                fsmn=gen_fsmn, firered=gen_firered, firered_stream=gen_firered_stream, firered_ckpt=gen_firered_ckpt, marblenet_fold=gen_marblenet_fold, dfsmn=gen_dfsmn, dfsmn_near_only=gen_dfsmn_near_only,
                fsmn_extra=gen_fsmn_extra, host_extra=gen_host_extra, marblenet_hostloop=gen_marblenet_hostloop,
                resample=gen_resample, silero_8k=gen_silero_8k, hostloop_thresholds=gen_hostloop_thresholds)
    for name, fn in gens.items():
        if not which or name in which:
            fn()
