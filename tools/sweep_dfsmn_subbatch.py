import sys, json, torch
sys.path.insert(0, "/root/repo")
import vadx, bench_models as bm
dev = torch.device("cuda", 0)
for sb in (3072,):
    try:
        r = bm.dfsmn_c5(torch, dev, 2, 0, sub_batch=sb)
        print("SUB", sb, round(r["ms"], 1), {k: round(v, 1) for k, v in r["kernel_ms"].items() if v > 50}, flush=True)
    except Exception as e:
        print("SUB", sb, "ERR", repr(e)[:200], flush=True)
    torch.cuda.empty_cache()
