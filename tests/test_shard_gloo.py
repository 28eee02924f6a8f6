"""CPU, world_size 2, gloo: the N>1 path (clip partition, barrier + max-over-ranks timing,
ragged result gather).  The per-shard work is host post-processing (no GPU here)."""
import os
import subprocess
import sys
import textwrap

import numpy as np

import vadx  # noqa: F401
from vadx import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 8, 4096, 4099):
        for world in (1, 2, 3, 8):
            spans = [shard.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


WORKER = textwrap.dedent("""
    import os, sys, json, time
    import numpy as np
    sys.path.insert(0, %r)
    import vadx
    from vadx import shard, timestamps
    dist = shard.init("gloo")
    rank, _, world = shard.env_rank()
    rng = np.random.default_rng(0)
    flags = rng.uniform(size=(11, 300)) < 0.4                 # 11 clips: uneven shards
    def detect(part):
        return [timestamps.process_timestamps(timestamps.vad_to_timestamps(f, 0.01), 0.3, 0.2) for f in part]
    shard.fence(dist)
    t0 = time.perf_counter()
    res = shard.run_sharded(flags, detect, dist)
    shard.fence(dist)
    el = shard.max_over_ranks(dist, time.perf_counter() - t0 + rank)      # rank 1 is "slower" by 1 s
    want = detect(flags)
    assert res == want, "gathered results differ from the single-process answer"
    assert el >= 1.0, el
    if rank == 0:
        print(json.dumps({"ok": True, "n": len(res), "world": world}))
    dist.barrier(); dist.destroy_process_group()
""") % ROOT


def test_two_rank_gloo_roundtrip(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", WORLD_SIZE="2")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = [p.communicate(timeout=240) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e.decode()[-2000:]
    assert b'"ok": true' in outs[0][0] and b'"n": 11' in outs[0][0]
