"""GPU parity: DFSMN near+far building blocks and full path vs the oracle (itself pinned against the
reference's ICCRN / wrapper classes by tests/golden/dfsmn_forward.npz)."""
import numpy as np
import pytest
import torch

import vadx  # noqa: F401
from vadx import dfsmn, weights
from oracle import dfsmn as od

pytestmark = pytest.mark.gpu


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


@pytest.fixture(scope="module")
def wts():
    w = weights.dfsmn_synthetic(1234)
    ow = {k[len("iccrn."):]: T(v) for k, v in w.items() if k.startswith("iccrn.")}
    return w, ow


@pytest.mark.parametrize("name,cin,frames,chunks", [("cfb_e1", 20, 101, 2), ("cfb_d4", 40, 37, 1)])
def test_cfb_block(wts, name, cin, frames, chunks):
    w, ow = wts
    net = dfsmn.Iccrn(w)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(chunks, cin, 160, frames, generator=g) * 0.7
    tb = od.IccrnTables(200).ceps
    want = torch.cat([od.cfb(x[n:n + 1], ow, name, tb) for n in range(chunks)], 0)
    xin = dfsmn.to_ft(torch, x, net.device)
    out = dfsmn.FT(torch, net.device, chunks, frames, 20, 160)
    if cin == 20:
        net.cfb(name, xin.view(), None, out.view(), chunks, frames)
    else:
        net.cfb(name, xin.view(0, 20), xin.view(20, 20), out.view(), chunks, frames)
    got = dfsmn.from_ft(out, chunks).cpu()
    err = (got - want).abs().max().item()
    assert err < 2e-4 * max(1.0, want.abs().max().item()), err
