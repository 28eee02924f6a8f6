"""lstm_f / pw_conv / dft_f time against the number of tiles (development aid, GPU box): where the time per tile steps up tells how
many workgroups of the kernel really are resident per CU."""
import sys
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import vadx  # noqa: F401
from vadx import dfsmn, weights, _lib

dev = torch.device("cuda", 0)
net = dfsmn.Iccrn(weights.dfsmn_synthetic(1234), dev)
frames = 16


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return min(a.elapsed_time(b) for a, b in ev)


for tiles in (1024, 6720):
    li = dfsmn.FT(torch, dev, tiles, frames, 40, 81)
    li.data.normal_()
    hf = dfsmn.FT(torch, dev, tiles, frames, 40, 81, zero=False)
    st = net.stats(li.view(), None, 81, tiles)
    name = "cfb_e1"
    t_l = timed(lambda: net.lstm_f(name + ".ceps_unit.ch_lstm_f", li.view(), net._ln(st, name + ".ceps_unit.LN"), hf.view(), 81, tiles))
    print(f"tiles {tiles:5d}  lstm_f<40> {t_l * 1e3:8.1f} us  = {t_l * 1e3 / tiles * 256:7.1f} us x CUs/tile", flush=True)
