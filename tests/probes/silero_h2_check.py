"""first GPU check of the fp16 x 2 Silero kernels: gx against the float64 oracle for all three kernel sets, step/clips vs oracle, timing"""
import sys, time
import os; R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch, ctypes as C
import vadx
from vadx import _lib, silero, weights
from oracle import silero as osil
from test_gpu_silero import gx_of, T
eng = silero.SileroEngine(weights.silero_synthetic(1234))
n = 5120
rng = np.random.default_rng(11)
clips = weights.burst_clips(27, n, seed=21).astype(np.float32)
clips[0] = 0
clips[1] = rng.integers(-1, 2, n)
clips[2] = rng.integers(-32768, 32768, n)
clips[3] = np.round(30000 * np.sin(2 * np.pi * 1000.0 / 16000 * np.arange(n)))
clips[4] = np.round(300 * np.sin(2 * np.pi * 3999.0 / 16000 * np.arange(n))) + rng.integers(-2, 3, n)
x = torch.from_numpy(clips * np.float32(0.000030517578))
w64 = {k: T(v).double() for k, v in weights.silero_synthetic(1234).items()}
xp = torch.cat([torch.zeros(27, 64), x], dim=1).double()
ref = torch.stack([osil.input_projection(w64, xp[:, 512 * t:512 * t + 576]) for t in range(n // 512)])
for mode in ("f32", "split", "h2"):
    silero.encoder_mode(mode)
    e = (gx_of(eng, x.cuda()).double().cpu() - ref).abs()
    print(mode, "gx err max %.3e mean %.3e  (scale %.3g)" % (float(e.max()), float(e.mean()), float(ref.abs().max())), " per-clip max:", ["%.1e" % float(e[:, c].max()) for c in range(6)])
flag = C.c_uint32(0); am = C.c_float(0)
flag.value, am.value = eng.range_flag()
print("range flag", flag.value, am.value)
# step + clips vs oracle
ow = {k: T(v) for k, v in weights.silero_synthetic(1234).items()}
for mode in ("split", "h2"):
    silero.encoder_mode(mode)
    rng = np.random.default_rng(5)
    xb = (rng.standard_normal((37, 576)) * rng.uniform(0.001, 0.3, (37, 1))).astype(np.float32)
    st = (rng.standard_normal((2, 37, 128)) * 0.5).astype(np.float32)
    out, stn = eng.step(xb, st)
    o_ref, s_ref = osil.net_forward(ow, T(xb), T(st))
    print(mode, "step err", float((out.cpu() - o_ref).abs().max()), float((stn.cpu() - s_ref).abs().max()))
    a = (weights.burst_clips(21, 16000, seed=3).astype(np.float32) * np.float32(0.000030517578))
    p = eng.clips(torch.from_numpy(a).cuda())
    pr = osil.OnnxWrapperOracle(ow).audio_forward(T(a), 16000)
    print(mode, "clips err", float((p.cpu() - pr).abs().max()))
# overflow flag: huge audio
silero.encoder_mode("h2")
big = torch.from_numpy((rng.standard_normal((16, 5120)) * 3000).astype(np.float32)).cuda()
eng.encode(big); torch.cuda.synchronize()
flag.value, am.value = eng.range_flag()
print("range flag after huge audio", flag.value, am.value)
flag.value, am.value = eng.range_flag()
print("range flag after reset", flag.value, am.value)
# timing at the bench shape
B, N = 4096, 160000
pcm = torch.from_numpy(weights.burst_clips(64, N, seed=1).astype(np.int16)).cuda().repeat(B // 64, 1).contiguous()
for mode in ("split", "h2"):
    silero.encoder_mode(mode)
    ws = None
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.encode_pcm16(pcm)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        pr = eng.recur(B, (N + 511) // 512, torch.empty((B, (N + 511) // 512), dtype=torch.float32, device="cuda"))
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print(mode, "encode %.3f ms  recur %.3f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
flag.value, am.value = eng.range_flag()
print("range flag after bench", flag.value, am.value)
