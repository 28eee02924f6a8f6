"""What does overlapping batch k's recurrent + segment kernels with batch k+1's encoder buy?  (two streams, two workspaces)
   python tools/pipe_probe.py [clips] [steps]"""
import ctypes as C
import sys
import time

import torch

sys.path.insert(0, ".")
import vadx  # noqa: F401,E402
from vadx import _lib, silero, weights  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
N = 160000
T = (N + 511) // 512
dev = torch.device("cuda:0")
eng = silero.SileroEngine(weights.silero_synthetic(1234), device=dev)
g = torch.Generator(device=dev).manual_seed(1)
audio = (torch.rand((B, N), device=dev, generator=g) - 0.5) * 0.2
L = _lib.lib()
need = L.vadx_silero_workspace_bytes(B, T)
ws = [torch.empty(need, dtype=torch.uint8, device=dev) for _ in range(2)]
probs = [torch.empty((B, T), dtype=torch.float32, device=dev) for _ in range(2)]
segs = torch.empty((B, 64, 2), dtype=torch.int64, device=dev)
counts = torch.empty((B,), dtype=torch.int32, device=dev)
lens = torch.full((B,), N, dtype=torch.int64, device=dev)
prm = silero.seg_params(threshold=0.5, max_speech_duration_s=20, min_speech_duration_ms=250, min_silence_duration_ms=250)
cfg = eng.cfg()
sa = torch.cuda.current_stream()
sb = torch.cuda.Stream()
pa, pb = C.c_void_p(sa.cuda_stream), C.c_void_p(sb.cuda_stream)


def enc(k, st):
    _lib.check(L.vadx_silero_encode(eng.packed.data_ptr(), audio.data_ptr(), B, N, _lib.row_stride(audio), ws[k & 1].data_ptr(), need, st, cfg))


def rec(k, st):
    _lib.check(L.vadx_silero_recur(eng.packed.data_ptr(), ws[k & 1].data_ptr(), need, B, T, None, probs[k & 1].data_ptr(), None, st, cfg))
    _lib.check(L.vadx_silero_segments(probs[k & 1].data_ptr(), B, T, lens.data_ptr(), C.byref(prm), segs.data_ptr(), counts.data_ptr(), 64, st))


def serial(n):
    for k in range(n):
        enc(k, pa)
        rec(k, pa)


def piped(n):
    encoded = [torch.cuda.Event() for _ in range(2)]
    freed = [torch.cuda.Event() for _ in range(2)]
    for k in range(n):
        if k >= 2:
            sa.wait_event(freed[k & 1])
        enc(k, pa)
        encoded[k & 1].record(sa)
        sb.wait_event(encoded[k & 1])
        rec(k, pb)
        freed[k & 1].record(sb)


for name, fn in (("serial", serial), ("piped", piped), ("serial", serial), ("piped", piped)):
    fn(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(K)
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / K * 1e3:.3f} ms/step", flush=True)
serial(1)
torch.cuda.synchronize()
ref = probs[0].clone()
piped(4)
torch.cuda.synchronize()
print("identical:", bool(torch.equal(ref, probs[0]) and torch.equal(ref, probs[1])), eng.range_flag())
