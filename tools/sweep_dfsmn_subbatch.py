"""DFSMN config-5 pass against the sub-batch size (development aid, GPU box): python tools/sweep_dfsmn_subbatch.py 960 3072 ..."""
import sys
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import vadx  # noqa: F401
import bench_models as bm
from vadx import dfsmn, weights

dev = torch.device("cuda", 0)
for sb in [int(a) for a in sys.argv[1:]] or [3072]:
    try:
        eng = dfsmn.DfsmnEngine(weights.dfsmn_synthetic(1234), device=dev, sub_batch=sb)
        lb, stride = eng.grid()
        W = 15
        n = (W - 1) * stride + eng.L
        near, far = bm.synth_pcm16(torch, dev, 2048, n, seed=1606), bm.synth_pcm16(torch, dev, 2048, n, seed=1607)
        ms = bm.device_ms(torch, lambda: eng.run(near, far, W, stride), 2)
        print("SUB", sb, round(ms, 1), "ms", flush=True)
        del eng, near, far
    except Exception as e:
        print("SUB", sb, "ERR", repr(e)[:300], flush=True)
    torch.cuda.empty_cache()
