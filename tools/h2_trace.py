"""Barrier timeline of silero_encode_h2_kernel (development aid; csrc/silero_h2.hip built with -DH2_TRACE=1).
   build : python tools/h2_trace.py build           -> _exp/libvadx_h2trace.so   (silero sources only)
   run   : python tools/h2_trace.py run             (GPU box) config-2 launch; sixteen workgroups spread over the grid record lane 0's shader
           clock of every wave right before / after each of the 23 barriers of a tile pair.  Prints, averaged over the traced workgroups:
           per phase (= the code between two barriers) the time of the FASTEST, the mean and the SLOWEST wave, and the time the phase's
           barrier held the waves on average -- i.e. how much of a workgroup's life is work and how much is waiting for its slowest wave."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "voice-activity-detection-vad-onnx_amd")
LIB = os.path.join(PKG, "_exp", "libvadx_h2trace.so")
PHASES = ["stage X (global loads -> LDS)", "samples -> registers", "e / o operand planes", "STFT GEMM + magnitudes", "|X| planes + scratch",
          "bin 64 slot", "conv1 GEMM", "conv1 store", "conv2 GEMM + exchange", "conv2 finish + store"]
TAIL = ["conv3 GEMM + exchange", "conv3 finish + store", "conv4", "W_ih + gx store"]

if sys.argv[1] == "build":
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-shared", "-DH2_TRACE=1",
           "-DVADX_SILERO_ENCODER_DEFAULT=2"] + sys.argv[2:] + [os.path.join(PKG, "csrc", s) for s in ("capi.hip", "silero.hip", "silero_split.hip", "silero_h2.hip")] + ["-o", LIB]
    subprocess.check_call(cmd)
    print("built", LIB)
    sys.exit(0)

sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import vadx  # noqa: E402,F401
from vadx import _lib, weights  # noqa: E402
import bench  # noqa: E402

h = C.CDLL(LIB)
B, T, N = 4096, 313, 160000
w = weights.silero_synthetic(1234)
h.vadx_silero_packed_floats.restype = C.c_size_t
h.vadx_silero_workspace_bytes.restype = C.c_size_t
h.vadx_silero_workspace_bytes.argtypes = [C.c_int, C.c_int]
hw = _lib.SileroWeightsHost()
keep = []


def ptr(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    keep.append(a)
    return a.ctypes.data


hw.stft_basis = ptr(w["stft_basis"])
for k in range(4):
    hw.enc_w[k] = ptr(w[f"enc{k}_w"])
    hw.enc_b[k] = ptr(w[f"enc{k}_b"])
hw.lstm_w_ih, hw.lstm_w_hh, hw.lstm_b_ih, hw.lstm_b_hh = (ptr(w[k]) for k in ("lstm_w_ih", "lstm_w_hh", "lstm_b_ih", "lstm_b_hh"))
hw.dec_w, hw.dec_b = ptr(w["dec_w"]), ptr(w["dec_b"])
packed = np.zeros(h.vadx_silero_packed_floats(), np.float32)
assert h.vadx_silero_pack_host(C.byref(hw), C.c_void_p(packed.ctypes.data)) == 0
pk = torch.from_numpy(packed).cuda()
audio = bench.synth_batch(torch, torch.device("cuda:0"), B, N, 1234)
nws = h.vadx_silero_workspace_bytes(B, T)
ws = torch.empty(nws, dtype=torch.uint8, device="cuda")
h.vadx_silero_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_longlong, C.c_longlong, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def enc():
    assert h.vadx_silero_encode(pk.data_ptr(), audio.data_ptr(), B, N, audio.stride(0), ws.data_ptr(), nws, st, None) == 0


for _ in range(3):
    enc()
torch.cuda.synchronize()
buf = (C.c_ulonglong * (16 * 8 * 128))()
h.vadx_silero_h2_trace(buf, 1)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); enc(); b.record()
torch.cuda.synchronize()
h.vadx_silero_h2_trace(buf, 0)
tr = np.frombuffer(buf, dtype=np.uint64).reshape(16, 8, 128).astype(np.int64)
print("launch %.3f ms; cycles are shader clocks of lane 0 of each wave" % a.elapsed_time(b))
ok = [g for g in range(16) if tr[g, 0, 127] > tr[g, 0, 126] > 0]
print("traced workgroups:", len(ok))
rows = []          # (name, fastest, mean, slowest, barrier hold mean) per phase, averaged over workgroups
tot_work, tot_wait = 0.0, 0.0


def phase(name, start_marks, arrive_idx, release_idx):
    global tot_work, tot_wait
    fa, me, sl, hold = [], [], [], []
    for g in ok:
        start = start_marks(g)                       # [8] per-wave start of the phase
        arr = tr[g, :, arrive_idx]
        rel = tr[g, :, release_idx]
        d = arr - start
        fa.append(d.min()); me.append(d.mean()); sl.append(d.max()); hold.append((rel - arr).mean())
    rows.append((name, np.mean(fa), np.mean(me), np.mean(sl), np.mean(hold)))
    tot_work += np.mean(me); tot_wait += np.mean(hold)


for sub in range(2):
    off = 32 * sub
    for k, name in enumerate(PHASES):
        if k == 0:
            start = (lambda g, off=off, sub=sub: tr[g, :, 126] if sub == 0 else tr[g, :, 32 * 0 + 2 * 9 + 1])
        else:
            start = (lambda g, off=off, k=k: tr[g, :, off + 2 * (k - 1) + 1])
        phase("tile %d: %s" % (sub, name), start, off + 2 * k, off + 2 * k + 1)
for k, name in enumerate(TAIL[:3]):
    start = (lambda g, k=k: tr[g, :, 32 + 2 * 9 + 1] if k == 0 else tr[g, :, 64 + 2 * (k - 1) + 1])
    phase("pair: " + name, start, 64 + 2 * k, 64 + 2 * k + 1)
# the last phase ends at mark 127 (no barrier)
fa, me, sl = [], [], []
for g in ok:
    d = tr[g, :, 127] - tr[g, :, 64 + 2 * 2 + 1]
    fa.append(d.min()); me.append(d.mean()); sl.append(d.max())
rows.append(("pair: " + TAIL[3], np.mean(fa), np.mean(me), np.mean(sl), 0.0))
tot_work += np.mean(me)
life = np.mean([tr[g, :, 127].max() - tr[g, :, 126].min() for g in ok])
print("%-44s %9s %9s %9s %12s" % ("phase (cycles)", "fastest", "mean", "slowest", "barrier hold"))
for r in rows:
    print("%-44s %9.0f %9.0f %9.0f %12.0f" % r)
print("workgroup life %.0f cycles for two tiles; sum of mean phase work %.0f (%.0f %%), sum of mean barrier holds %.0f (%.0f %%)" %
      (life, tot_work, 100 * tot_work / life, tot_wait, 100 * tot_wait / life))
