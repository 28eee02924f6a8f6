import sys, json, torch
sys.path.insert(0, '/root/repo')
import vadx, bench_models as bm
out = bm.dfsmn_c5(torch, torch.device('cuda', 0), 2, 0, clips=128)
print("RESULT", round(out['ms'], 1), {k: round(v, 1) for k, v in out['kernel_ms'].items() if v > 5})
