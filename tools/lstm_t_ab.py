"""Two-layer time LSTM (vadx_dfsmn_lstm_t which=0), one launch of N windows (default 960); VADX_LIBRARY selects the build (A/B)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vadx  # noqa: F401
from vadx import dfsmn, weights
net = dfsmn.Iccrn(weights.dfsmn_synthetic(1234))
t = torch
chunks, frames = int(sys.argv[1]) if len(sys.argv) > 1 else 960, 101
tiles = chunks * dfsmn.ft_tiles(frames)
mk = lambda ch, bins: dfsmn.FT(t, net.device, chunks, frames, ch, bins, zero=True)      # noqa: E731
e5, p5 = mk(20, 160), mk(20, 160)
e5.data.normal_()
s5 = net.stats(e5.view(), None, 160, tiles)
run = lambda: net.lstm_t(0, "ch_lstm", e5.view(), net._ln(s5, "ln"), e5.view(), p5.view(), frames, chunks)     # noqa: E731
run(); t.cuda.synchronize()
ev = [(t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)) for _ in range(4)]
for a, b in ev:
    a.record(); run(); b.record()
t.cuda.synchronize()
print("LSTM_T", os.path.basename(os.environ.get("VADX_LIBRARY", "libvadx.so")), chunks, "windows: ms", ["%.2f" % a.elapsed_time(b) for a, b in ev],
      "checksum %.6f" % float(p5.data.double().abs().sum().item()))
