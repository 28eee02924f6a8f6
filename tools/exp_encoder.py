"""Encoder what-if timing (development aid, not part of the product or the tests).

build : python tools/exp_encoder.py build 0 1 2 ...   -> voice-activity-detection-vad-onnx_amd/_exp/libvadx_expN.so
        (silero.hip compiled with -DVADX_EXP=N, N a BIT MASK of switches (bit 1 no global loads, 2 no barriers, 3 no conv2-4,
        4 no STFT pass, 5 no conv1, 6 no W_ih, 7 no Nyquist bin, 8 no gx stores; LSTM kernel: 9 no gate non-linearities, 10 no per-step barrier, 11 no gx loads, 12 no probs reduce); the variants skip or alter parts of silero_encode_kernel, so their
        RESULTS ARE WRONG on purpose -- only the kernel time is of interest)
run   : python tools/exp_encoder.py run 0 1 2 ...     (on the GPU box) -> one line per variant with the mean
        silero_encode_kernel time at the bench shape.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "voice-activity-detection-vad-onnx_amd")
EXP = os.path.join(PKG, "_exp")
SRC = ["capi.hip", "silero.hip", "silero_split.hip", "silero_h2.hip"]
if os.environ.get("EXP_FULL") == "1":     # every source: a library the ctypes binding (vadx._lib, VADX_LIBRARY=...) can load
    SRC += ["frontend.hip", "fsmn.hip", "firered.hip", "marblenet.hip", "dfsmn.hip", "dfsmn_cfb.hip", "ingest.hip"]
MODE = int(os.environ.get("EXP_SPLIT", "2"))             # EXP_SPLIT=0: the exact-f32 encoder, 1: bf16 x 3 (silero_split.hip), 2: fp16 x 2 (silero_h2.hip)
SPLIT = MODE == 1
DEFS = os.environ.get("EXP_DEFS", "").split()            # extra -D switches, e.g. EXP_DEFS="-DVADX_H2_NSUB=1"; EXP_TAG names the library
TAG = os.environ.get("EXP_TAG", "")


def build(ids):
    os.makedirs(EXP, exist_ok=True)
    for n in ids:
        out = os.path.join(EXP, f"libvadx_exp{TAG}{n}.so")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-shared", f"-DVADX_EXP={n}",
               f"-DVADX_SILERO_ENCODER_DEFAULT={MODE}"] + DEFS
        cmd += [os.path.join(PKG, "csrc", s) for s in SRC] + ["-o", out]
        subprocess.check_call(cmd)
        print("built", out)


CHILD = r"""
import ctypes as C, os, sys, torch
sys.path.insert(0, %r)
import vadx
from vadx import _lib, silero, weights
import bench
h = C.CDLL(os.environ["VADX_LIBRARY"])
B, T, N = 4096, 313, 160000
w = weights.silero_synthetic(1234)
# minimal binding: only the symbols this library has
h.vadx_silero_packed_floats.restype = C.c_size_t
h.vadx_silero_workspace_bytes.restype = C.c_size_t
h.vadx_silero_workspace_bytes.argtypes = [C.c_int, C.c_int]
h.vadx_last_error.restype = C.c_char_p
import numpy as np
hw = _lib.SileroWeightsHost()
keep = []
def ptr(a):
    a = np.ascontiguousarray(a, dtype=np.float32); keep.append(a); return a.ctypes.data
hw.stft_basis = ptr(w["stft_basis"])
for k in range(4):
    hw.enc_w[k] = ptr(w[f"enc{k}_w"]); hw.enc_b[k] = ptr(w[f"enc{k}_b"])
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
    rc = h.vadx_silero_encode(pk.data_ptr(), audio.data_ptr(), B, N, audio.stride(0), ws.data_ptr(), nws, st, None)     # cfg NULL = the library's default (-DVADX_SILERO_ENCODER_DEFAULT)
    assert rc == 0, h.vadx_last_error()
h.vadx_silero_recur.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
probs = torch.empty((B, T), dtype=torch.float32, device="cuda")
def rec():
    rc = h.vadx_silero_recur(pk.data_ptr(), ws.data_ptr(), nws, B, T, None, probs.data_ptr(), None, st, None)
    assert rc == 0, h.vadx_last_error()
for _ in range(2): enc(); rec()
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
for a, b in ev:
    a.record(); enc(); b.record()
torch.cuda.synchronize()
ts = [a.elapsed_time(b) for a, b in ev]
for a, b in ev:
    a.record(); rec(); b.record()
torch.cuda.synchronize()
tr = [a.elapsed_time(b) for a, b in ev]
if hasattr(h, "vadx_silero_h2_debug_cycles") and %d == 2:
    buf = (C.c_ulonglong * 16)()
    h.vadx_silero_h2_debug_cycles(buf, 1)
    enc(); torch.cuda.synchronize()
    h.vadx_silero_h2_debug_cycles(buf, 0)
    if buf[15]:
        print("PHASES shader clock over the workgroups: %%.2f GHz (s_memtime / 100 MHz counter)" %% (buf[14] / buf[15] / 10.0))
    buf[14] = buf[15] = 0
    tot = sum(buf) or 1
    names = {0: "stage X", 1: "samples -> regs", 2: "e/o planes", 3: "STFT GEMM + |.|", 4: "|X| planes + bin 64", 5: "conv1", 6: "conv1 store", 7: "conv2", 8: "conv3", 9: "conv4", 10: "W_ih + gx store"}
    print("PHASES (share of wave-0 cycles, sum over workgroups):", ", ".join(f"{names.get(k, k)} {100.0 * v / tot:.1f}%%" for k, v in enumerate(buf) if v),
          "| cycles per tile: %%.0f" %% (tot / (B // 16 * T)))
elif hasattr(h, "vadx_silero_split_debug_cycles") and %d == 1:
    buf = (C.c_ulonglong * 16)()
    h.vadx_silero_split_debug_cycles(buf, 1)
    enc(); torch.cuda.synchronize()
    h.vadx_silero_split_debug_cycles(buf, 0)
    if buf[15]:
        print("PHASES shader clock over the workgroups: %%.2f GHz (s_memtime / 100 MHz counter)" %% (buf[14] / buf[15] / 10.0))
    buf[14] = buf[15] = 0
    tot = sum(buf) or 1
    names = {0: "stage X", 1: "STFT fold + bin 64", 2: "|.| planes", 3: "bin 64 slot", 4: "conv1", 5: "conv1 store", 6: "conv2", 7: "conv3", 8: "conv4", 9: "W_ih + gx store"}
    print("PHASES (share of wave-0 cycles, sum over workgroups):", ", ".join(f"{names.get(k, k)} {100.0 * v / tot:.1f}%%" for k, v in enumerate(buf) if v),
          "| cycles per tile: %%.0f" %% (tot / (B // 16 * T)))
elif hasattr(h, "vadx_silero_debug_cycles"):
    buf = (C.c_ulonglong * 16)()
    h.vadx_silero_debug_cycles(buf, 1)
    enc(); torch.cuda.synchronize()
    h.vadx_silero_debug_cycles(buf, 0)
    tot = sum(buf) or 1
    names = {0: "stage X", 1: "STFT fold + bin 64", 2: "|.| + transform (fp 0)", 3: "dense STFT", 4: "dense transform", 5: "transform (fp 1) / bin 64 row",
             6: "conv1 planes", 7: "A^T + store A1", 8: "conv2", 9: "conv3", 10: "conv4", 11: "X loads issued", 12: "X loads landed", 13: "X -> LDS", 15: "W_ih + gx store"}
    print("PHASES (share of wave-0 cycles, sum over workgroups):", ", ".join(f"{names.get(k, k)} {100.0 * v / tot:.1f}%%" for k, v in enumerate(buf) if v),
          "| cycles per tile: %%.0f" %% (tot / (B // 16 * T)))
print("EXP", os.path.basename(os.environ["VADX_LIBRARY"]), "encode ms mean %%.3f min %%.3f | recur ms mean %%.3f min %%.3f" %% (sum(ts) / len(ts), min(ts), sum(tr) / len(tr), min(tr)))
"""


def run(ids):
    for n in ids:
        env = dict(os.environ, VADX_LIBRARY=os.path.join(EXP, f"libvadx_exp{TAG}{n}.so"))
        r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, MODE, MODE)], env=env, capture_output=True, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("EXP")]
        for ph in [l for l in r.stdout.splitlines() if l.startswith("PHASES")]:
            print(ph, flush=True)
        print(line[0] if line else f"EXP {n} FAILED rc={r.returncode}\n{r.stderr[-800:]}", flush=True)


if __name__ == "__main__":
    ids = [int(a) for a in sys.argv[2:]]
    {"build": build, "run": run}[sys.argv[1]](ids)
