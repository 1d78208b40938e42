"""Sharding the gated-GCN forward across GPUs: one process per GPU, RCCL over xGMI.

Graphs are independent (the batched adjacency is block-diagonal, ``models/bert_amir5.py:589``), so
the data path needs no exchange: every rank gets a contiguous range of graphs balanced by nnz,
weights are replicated, and the ONLY collective is one all-gather of the per-shard pooled outputs
``out [B_r, H]``.  ``torch.distributed`` backend "nccl" is RCCL on ROCm; the same code
runs on "gloo" with CPU tensors (tests/test_shard_gloo.py).
"""
import ctypes

import numpy as np
import torch
import torch.distributed as dist


def partition_graphs(nnz_per_graph, world):
    """Contiguous graph ranges [(lo, hi)] * world, balanced by nnz (prefix-sum cut points).

    Every rank gets at least one graph when there are at least ``world`` graphs."""
    nnz = np.asarray(nnz_per_graph, dtype=np.int64)
    B = int(nnz.shape[0])
    if world <= 0:
        raise ValueError("world must be positive")
    if B < world:
        raise ValueError("cannot shard %d graphs over %d ranks" % (B, world))
    csum = np.concatenate([[0], np.cumsum(nnz)])
    cuts = [0]
    for r in range(1, world):
        target = csum[-1] * r / world
        c = int(np.searchsorted(csum, target, side="left"))
        c = max(c, cuts[-1] + 1)            # at least one graph per rank
        c = min(c, B - (world - r))         # leave one graph for every later rank
        cuts.append(c)
    cuts.append(B)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def shard_csr_host(rowptr, colidx, T, lo, hi):
    """Host CSR arrays of graphs [lo, hi) re-based to local node ids."""
    r0, r1 = lo * T, hi * T
    e0, e1 = int(rowptr[r0]), int(rowptr[r1])
    return (rowptr[r0:r1 + 1] - e0).astype(np.int32), (colidx[e0:e1] - r0).astype(np.int32)


class PooledGather:
    """All-gather of per-rank pooled outputs [B_r, H] into [sum B_r, H] (uneven B_r allowed).

    ``start`` launches the collective asynchronously and returns a handle; ``finish`` waits for it
    and returns the gathered tensor.  Two slots alternate -- each with its OWN output buffer and its
    own padded send buffer -- so step i's gather can overlap step i+1's kernels (the collective runs
    on RCCL's own stream).  A slot is never rewritten while a collective launched from it may still
    be reading: ``start`` first waits for the slot's previous collective (a no-op when the caller
    already ``finish``-ed it, as the bench loop does)."""

    def __init__(self, counts, width, device, dtype=torch.float32, group=None):
        self.counts = [int(c) for c in counts]
        self.group = group
        self.world = len(self.counts)
        self.uniform = len(set(self.counts)) == 1
        self.max_count = max(self.counts)
        self.bufs = [torch.empty(self.world * self.max_count, width, device=device, dtype=dtype) for _ in range(2)]
        self.pads = [None, None] if self.uniform else \
            [torch.zeros(self.max_count, width, device=device, dtype=dtype) for _ in range(2)]
        self.inflight = [None, None]   # the last collective launched from each slot
        self.turn = 0
        self.side = None               # the `flag` hand-off's stream (start(..., gate=))

    def start(self, pooled, gate=None):
        """Launch the all-gather of ``pooled``.  ``gate=(signal, n)``: the `flag` hand-off -- ``pooled`` is being written by a
        launch that counts itself done in ``signal[1]`` (``heads.dense_head(..., signal=)``: the n-th launch on these words),
        and the collective is issued from a SIDE stream that ``hipStreamWaitValue32`` holds until ``signal[1] >= n``: the
        stream the step runs on records no event and waits for nothing (the event hand-off cost the sharded step ~20 us of
        its ~110).  ``finish`` then waits on the HOST for the side stream's event."""
        slot = self.turn
        self.turn ^= 1
        prev = self.inflight[slot]
        if prev is not None:           # the slot's buffers are still the operands of that collective
            self._wait(prev)
        buf = self.bufs[slot]
        if gate is not None:
            signal, n = gate
            side = self._side_stream(pooled.device)
            rc = _hip().hipStreamWaitValue32(ctypes.c_void_p(side.cuda_stream), ctypes.c_void_p(signal.data_ptr() + 4),
                                            ctypes.c_uint32(n & 0xFFFFFFFF), 0, 0xFFFFFFFF)   # 0 = hipStreamWaitValueGte
            if rc != 0:
                raise RuntimeError("hipStreamWaitValue32 failed (%d)" % rc)
            with torch.cuda.stream(side):
                src = self._source(pooled, slot)
                work = dist.all_gather_into_tensor(buf, src, group=self.group, async_op=True)
                work.wait()            # (the side stream waits for the collective; the host does not)
                done = torch.cuda.Event()
                done.record(side)
            handle = (work, buf, src, done)
        else:
            src = self._source(pooled, slot)
            work = dist.all_gather_into_tensor(buf, src, group=self.group, async_op=True)
            handle = (work, buf, src)  # src is kept alive by the handle until finish()
        self.inflight[slot] = handle
        return handle

    def _source(self, pooled, slot):
        if not self.uniform:  # pad to the largest shard so one fixed-size collective serves every rank
            pad = self.pads[slot]
            pad[:pooled.shape[0]].copy_(pooled)
            return pad
        return pooled if pooled.is_contiguous() else pooled.contiguous()

    def _side_stream(self, device):
        if self.side is None:
            self.side = torch.cuda.Stream(device=device)
        return self.side

    @staticmethod
    def _wait(handle):
        if len(handle) > 3:            # flag hand-off: the side stream's event, waited for on the host -- by polling: a sleeping
            while not handle[3].query():   # hipEventSynchronize comes back tens of microseconds late, and the next replay with it
                pass
        else:
            handle[0].wait()

    def finish(self, handle):
        buf = handle[1]
        self._wait(handle)
        for s in range(2):
            if self.inflight[s] is handle:
                self.inflight[s] = None
        if self.uniform:
            return buf
        parts = [buf[r * self.max_count: r * self.max_count + c] for r, c in enumerate(self.counts)]
        return torch.cat(parts, dim=0)


_HIP = []


def _hip():
    """libamdhip64 (already in the process: torch links it) for hipStreamWaitValue32, which torch does not expose."""
    if not _HIP:
        lib = ctypes.CDLL("libamdhip64.so")
        lib.hipStreamWaitValue32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint, ctypes.c_uint32]
        lib.hipStreamWaitValue32.restype = ctypes.c_int
        _HIP.append(lib)
    return _HIP[0]


def sharded_forward(local_forward, counts, width, device, group=None):
    """Run ``local_forward() -> pooled [B_r, H]`` on this rank and all-gather the result."""
    g = PooledGather(counts, width, device, group=group)
    return g.finish(g.start(local_forward()))
