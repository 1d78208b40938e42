"""Sharding the gated-GCN forward across GPUs: one process per GPU, RCCL over xGMI.

Graphs are independent (the batched adjacency is block-diagonal, ``models/bert_amir5.py:589``), so
the data path needs no exchange: every rank gets a contiguous range of graphs balanced by nnz,
weights are replicated, and the ONLY collective is one all-gather of the per-shard pooled outputs
``out [B_r, H]``.  ``torch.distributed`` backend "nccl" is RCCL on ROCm; the same code
runs on "gloo" with CPU tensors (tests/test_shard_gloo.py).
"""
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

    def start(self, pooled):
        slot = self.turn
        self.turn ^= 1
        prev = self.inflight[slot]
        if prev is not None:           # the slot's buffers are still the operands of that collective
            prev.wait()
        buf = self.bufs[slot]
        src = pooled
        if not self.uniform:  # pad to the largest shard so one fixed-size collective serves every rank
            pad = self.pads[slot]
            pad[:pooled.shape[0]].copy_(pooled)
            src = pad
        elif not src.is_contiguous():
            src = src.contiguous()
        work = dist.all_gather_into_tensor(buf, src, group=self.group, async_op=True)
        self.inflight[slot] = work
        return work, buf, src          # src is kept alive by the handle until finish()

    def finish(self, handle):
        work, buf = handle[0], handle[1]
        work.wait()
        for s in range(2):
            if self.inflight[s] is work:
                self.inflight[s] = None
        if self.uniform:
            return buf
        parts = [buf[r * self.max_count: r * self.max_count + c] for r, c in enumerate(self.counts)]
        return torch.cat(parts, dim=0)


def sharded_forward(local_forward, counts, width, device, group=None):
    """Run ``local_forward() -> pooled [B_r, H]`` on this rank and all-gather the result."""
    g = PooledGather(counts, width, device, group=group)
    return g.finish(g.start(local_forward()))
