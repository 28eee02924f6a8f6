"""why do spans differ from single launches on the fp16 x 2 kernels?"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import numpy as np, torch
import vadx
from vadx import silero, weights, _lib
eng = silero.SileroEngine(weights.silero_synthetic(1234))
silero.encoder_mode("h2")
batch, n, span = 37, 20000, 16
a = torch.from_numpy(weights.burst_clips(batch, n, seed=batch + n).astype(np.float32) * np.float32(0.000030517578)).cuda()
p1 = eng.clips(a).clone(); p2 = eng.clips(a).clone()
print("run-to-run identical:", bool(torch.equal(p1, p2)))
steps = 40
L = _lib.lib()
# gx whole
eng.encode(a); torch.cuda.synchronize()
gw = eng._ws[:steps * 3 * 8192 * 4].view(torch.float32).view(steps, 3, 8192).clone()
# gx by spans
tile_bytes = L.vadx_silero_workspace_bytes(batch, 1)
for first in range(0, steps, span):
    ns = min(span, steps - first)
    ws = eng._workspace(batch, steps)
    _lib.check(L.vadx_silero_encode_span(eng.packed.data_ptr(), a.data_ptr(), batch, n, _lib.row_stride(a), first, ns, ws.data_ptr(), ns * tile_bytes, _lib.stream_ptr(), eng.cfg()))
    torch.cuda.synchronize()
    gs = ws[:ns * 3 * 8192 * 4].view(torch.float32).view(ns, 3, 8192)
    bad = (gs != gw[first:first + ns]).any(dim=-1)
    print("span", first, "gx tiles differing:", bad.nonzero().tolist()[:10], "max diff", float((gs - gw[first:first + ns]).abs().max()))
got = torch.full((batch, steps), -1.0, dtype=torch.float32, device="cuda")
st = eng.clips_spanned(a, n, got, span=span)
print("probs differing cols:", (got != p1).any(dim=0).nonzero().flatten().tolist()[:20], "rows:", (got != p1).any(dim=1).nonzero().flatten().tolist()[:20], "max", float((got - p1).abs().max()))
# ---- the recurrent kernel alone: whole vs split at several boundaries, on the SAME gx
eng.encode(a); torch.cuda.synchronize()
ws = eng._workspace(batch, steps)
whole = torch.empty((batch, steps), dtype=torch.float32, device="cuda")
stw = torch.empty((2, batch, 128), dtype=torch.float32, device="cuda")
_lib.check(L.vadx_silero_recur_span(eng.packed.data_ptr(), ws.data_ptr(), steps * tile_bytes, batch, steps, None, whole.data_ptr(), steps, stw.data_ptr(), _lib.stream_ptr(), eng.cfg()))
for cut in (1, 2, 15, 16, 17):
    got = torch.empty((batch, steps), dtype=torch.float32, device="cuda")
    st = torch.empty((2, batch, 128), dtype=torch.float32, device="cuda")
    _lib.check(L.vadx_silero_recur_span(eng.packed.data_ptr(), ws.data_ptr(), cut * tile_bytes, batch, cut, None, got.data_ptr(), steps, st.data_ptr(), _lib.stream_ptr(), eng.cfg()))
    _lib.check(L.vadx_silero_recur_span(eng.packed.data_ptr(), ws.data_ptr() + cut * tile_bytes, (steps - cut) * tile_bytes, batch, steps - cut, st.data_ptr(), got.data_ptr() + 4 * cut, steps, st.data_ptr(), _lib.stream_ptr(), eng.cfg()))
    torch.cuda.synchronize()
    d = (got != whole)
    print("cut", cut, "differing cols", d.any(dim=0).nonzero().flatten().tolist()[:6], "state equal", bool(torch.equal(st, stw)))
