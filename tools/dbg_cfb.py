"""Localise a fused-CFB mismatch (development aid): fused vs unfused block on one small input; prints the error of the spectrum li,
of y1 against conv31(ln1_w * gx) rebuilt from the unfused chain's gx, and of the block output, with the worst element's place."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import vadx
from vadx import dfsmn, weights

name, cin, frames, chunks = sys.argv[1] if len(sys.argv) > 1 else "cfb_e1", int(sys.argv[2]) if len(sys.argv) > 2 else 20, 37, 2
w = weights.dfsmn_synthetic(1234)
net = dfsmn.Iccrn(w)
g = torch.Generator().manual_seed(5)
x = torch.randn(chunks, cin, 160, frames, generator=g) * 0.7
xin = dfsmn.to_ft(torch, x, net.device)
a, b = (xin.view(), None) if cin == 20 else (xin.view(0, 20), xin.view(20, 20))
of, ou = dfsmn.FT(torch, net.device, chunks, frames, 20, 160), dfsmn.FT(torch, net.device, chunks, frames, 20, 160)
scf, _ = net.cfb(name, a, b, of.view(), chunks, frames)
scu, _ = net.cfb_unfused(name, a, b, ou.view(), chunks, frames)
torch.cuda.synchronize()
def rep(tag, got, want):
    got, want = got.cpu(), want.cpu()
    d = (got - want).abs()
    k = int(d.argmax())
    idx = np.unravel_index(k, d.shape)
    print(f"{tag:6s} max err {d.max().item():.3e} at {tuple(int(v) for v in idx)} (got {got.flatten()[k].item():.5f} want {want.flatten()[k].item():.5f}); scale {want.abs().max().item():.3f}; "
          f"bad elements {(d > 1e-3 * max(1.0, want.abs().max().item())).sum().item()} of {d.numel()}")
rep("li", dfsmn.from_ft(scf["li"], chunks), dfsmn.from_ft(scu["li"], chunks))
gx = dfsmn.from_ft(scu["gx"], chunks).cpu().double()
w1 = torch.from_numpy(w[f"iccrn.{name}.LN1.w"]).reshape(1, 20, 160, 1).double()
w31 = torch.from_numpy(w[f"iccrn.{name}.conv.weight"]).double()
y1w = torch.nn.functional.conv2d(gx * w1, w31, None, padding=(1, 0)).float()
rep("y1", dfsmn.from_ft(scf["y1"], chunks), y1w)
rep("hf", dfsmn.from_ft(scf["hf"], chunks), dfsmn.from_ft(scu["hf"], chunks))
rep("out", dfsmn.from_ft(of, chunks), dfsmn.from_ft(ou, chunks))
