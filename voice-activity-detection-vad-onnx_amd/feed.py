"""Host -> device PCM feed for the int16 engines (on an 8-GPU node the host link, not the kernels, is the limit).

Clips are independent, so a batch held in HOST memory (wav files decoded to int16) is cut into chunks of clips: chunk k + 1 crosses
PCIe on a copy stream into one of two device buffers while the engine's launches for chunk k run on the caller's stream.  The
per-chunk results are what the engine returns for those clips -- bit-identical to running the whole batch resident
(`tests/test_gpu_drivers.py::test_host_feed_is_bitwise_the_resident_path`).  The Silero engine has its own variant
(`vadx.silero.HostFeed`: its chunks fill slices of one workspace); this one serves FSMN / MarbleNet / FireRed / DFSMN.

Pinned host memory (`pin()`) makes the copies asynchronous; pageable memory works but serialises copy and compute."""
from __future__ import annotations

import numpy as np

from . import _lib


def pin(host_pcm):
    """numpy / CPU tensor int16 [B, N] -> pinned CPU tensor (one copy; reuse it across calls)."""
    t = _lib.require_gpu()
    a = host_pcm if t.is_tensor(host_pcm) else t.from_numpy(np.ascontiguousarray(host_pcm, dtype=np.int16))
    if a.dtype != t.int16 or a.dim() != 2 or a.is_cuda:
        raise ValueError("pin expects host int16 PCM [B, N]")
    return a if a.is_pinned() else a.pin_memory()


class HostPcmFeed:
    """Double-buffered upload of `streams` parallel int16 inputs (1; 2 for DFSMN's near / far pair) of [B, n_samples]."""

    def __init__(self, device, n_samples, chunk_clips=256, streams=1):
        self.t = t = _lib.require_gpu()
        self.device = t.device(device)
        self.N, self.chunk, self.streams = int(n_samples), max(1, int(chunk_clips)), int(streams)
        self.buf = [[t.empty((self.chunk, self.N), dtype=t.int16, device=self.device) for _ in range(self.streams)] for _ in range(2)]
        self.copy_stream = t.cuda.Stream(device=self.device)
        # The buffers were just handed out by the caching allocator on the CURRENT stream: a block it recycled may still be read by
        # launches queued there.  The copy stream therefore starts behind everything the allocating stream has queued so far, and the
        # buffers are tied to both streams (record_stream) so that dropping the feed cannot return them while either still uses them.
        with t.cuda.device(self.device):
            alloc = t.cuda.current_stream()
            self.copy_stream.wait_stream(alloc)
            for pair in self.buf:
                for b in pair:
                    b.record_stream(self.copy_stream)
        self.ready = [t.cuda.Event() for _ in range(2)]
        self.free = [t.cuda.Event() for _ in range(2)]
        self.in_use = [False, False]

    def map(self, hosts, fn):
        """hosts: one host int16 [B, N] tensor per stream; fn(*device_chunks [nb, N]) -> whatever the engine returns for those clips.
        Returns the list of per-chunk results in clip order (every launch of fn is on the caller's current stream)."""
        t = self.t
        hosts = [h if t.is_tensor(h) else t.from_numpy(np.ascontiguousarray(h, dtype=np.int16)) for h in hosts]
        if len(hosts) != self.streams:
            raise ValueError(f"HostPcmFeed.map: expected {self.streams} host tensor(s)")
        B = hosts[0].shape[0]
        for h in hosts:
            if h.dtype != t.int16 or h.dim() != 2 or h.is_cuda or tuple(h.shape) != (B, self.N):
                raise ValueError(f"HostPcmFeed.map expects host int16 PCM [{B}, {self.N}]")
        out = []
        with t.cuda.device(self.device):
            comp = t.cuda.current_stream()
            for pair in self.buf:                            # a caller on another stream than the allocating one: tie the buffers to it too
                for b in pair:
                    b.record_stream(comp)
            for k2, b0 in enumerate(range(0, B, self.chunk)):
                nb, k = min(self.chunk, B - b0), k2 & 1
                with t.cuda.stream(self.copy_stream):
                    if self.in_use[k]:                       # the launches that last read this buffer (also across calls)
                        self.copy_stream.wait_event(self.free[k])
                    for s, h in enumerate(hosts):
                        self.buf[k][s][:nb].copy_(h[b0:b0 + nb], non_blocking=True)
                    self.ready[k].record(self.copy_stream)
                comp.wait_event(self.ready[k])
                out.append(fn(*[self.buf[k][s][:nb] for s in range(self.streams)]))
                self.free[k].record(comp)
                self.in_use[k] = True
        return out


def cat_results(chunks):
    """Concatenate per-chunk engine results along dim 0: tensors, or tuples / lists of tensors (non-tensor members, e.g. MarbleNet's
    signal_len, are taken from the first chunk)."""
    t = _lib.require_gpu()
    first = chunks[0]
    if t.is_tensor(first):
        return t.cat(chunks, 0)
    return tuple(t.cat([c[j] for c in chunks], 0) if t.is_tensor(first[j]) else first[j] for j in range(len(first)))
