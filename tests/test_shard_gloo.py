"""N > 1 path on CPU: world_size-2 gloo run of the sharding + all-gather logic.  The local
forward is the ORACLE here (tests may use it; the product's local forward is the HIP block and
needs a GPU) -- what is under test is partitioning, CSR re-basing and the collective."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ed_gated_gcn_amd import shard, synth
from oracle import ref_dense


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem():
    B, T, H = 11, 9, 16
    rng = np.random.default_rng(4)
    lens = rng.integers(2, T + 1, size=B)
    adj = synth.dependency_batch(B, T, 3.0, seed=9, lengths=lens)
    x = rng.standard_normal((B, T, H)).astype(np.float32)
    g1 = rng.uniform(0.1, 0.9, (B, H)).astype(np.float32)
    g2 = rng.uniform(0.1, 0.9, (B, H)).astype(np.float32)
    w1, b1 = synth.layer_params(H, H, seed=1)
    w2, b2 = synth.layer_params(H, H, seed=2)
    return B, T, H, adj, x, g1, g2, w1, b1, w2, b2


def _worker(rank, world, port, ret):
    """Every rank reports ("ok", payload) or ("error", traceback): a failing worker is surfaced by the
    parent at once instead of leaving it blocked on the queue."""
    import traceback
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        try:
            ret.put((rank, "ok", _worker_body(rank, world)))
        finally:
            dist.destroy_process_group()
    except BaseException:   # noqa: B902 -- report, then exit non-zero
        ret.put((rank, "error", traceback.format_exc()))
        raise


def _worker_body(rank, world):
    B, T, H, adj, x, g1, g2, w1, b1, w2, b2 = _problem()
    parts = shard.partition_graphs(adj.reshape(B, -1).sum(1), world)
    lo, hi = parts[rank]
    rowptr, colidx, _ = synth.csr_from_dense_host(adj)
    lrp, lci = shard.shard_csr_host(rowptr, colidx, T, lo, hi)
    # the re-based shard is exactly the CSR of the shard's own dense adjacency
    erp, eci, _ = synth.csr_from_dense_host(adj[lo:hi])
    assert np.array_equal(lrp, erp) and np.array_equal(lci, eci)
    t = torch.from_numpy

    def local_forward():
        r = ref_dense.gated_block(t(x[lo:hi]), t(adj[lo:hi].astype(np.float32)), t(g1[lo:hi]), t(g2[lo:hi]),
                                  t(w1), t(b1), t(w2), t(b2))
        return r["out"]

    counts = [h - l for l, h in parts]
    assert len(set(counts)) > 1, "the problem is meant to exercise UNEVEN shards (padded send buffers)"
    full = shard.sharded_forward(local_forward, counts, H, "cpu")
    # overlapped form: several gathers in flight; each slot has its own send and receive buffers
    g = shard.PooledGather(counts, H, "cpu")
    base = local_forward()
    h1 = g.start(base)
    h2 = g.start(base * 2)
    a, b = g.finish(h1).clone(), g.finish(h2).clone()
    assert torch.equal(b, a * 2)
    # a third start() reuses slot 0 without the caller having finished it: start() itself must wait
    h3 = g.start(base * 3)
    h4 = g.start(base * 4)
    h5 = g.start(base * 5)          # slot 0 again, h3 still un-finished by the caller
    c5 = g.finish(h5).clone()
    c4 = g.finish(h4).clone()
    assert torch.equal(c4, a * 4) and torch.equal(c5, a * 5)
    return full.numpy()


def test_partition_is_contiguous_balanced_and_complete():
    nnz = np.array([10, 200, 30, 30, 30, 100, 5, 5, 90, 100])
    for world in (1, 2, 3, 4, 8, 10):
        parts = shard.partition_graphs(nnz, world)
        assert parts[0][0] == 0 and parts[-1][1] == len(nnz)
        assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
        assert all(hi > lo for lo, hi in parts)
    two = shard.partition_graphs(nnz, 2)
    loads = [nnz[lo:hi].sum() for lo, hi in two]
    assert abs(loads[0] - loads[1]) <= nnz.max()
    with pytest.raises(ValueError):
        shard.partition_graphs(nnz[:3], 4)
    # uniform graphs (config 2): equal split
    assert shard.partition_graphs(np.full(4096, 128), 8) == [(i * 512, (i + 1) * 512) for i in range(8)]


def test_world2_gloo_shards_and_gathers():
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    results, errors = {}, []
    try:
        for _ in range(world):   # one report per rank, error or not
            rank, status, payload = ret.get(timeout=120)
            if status == "ok":
                results[rank] = payload
            else:
                errors.append("rank %d:\n%s" % (rank, payload))
                break            # the other rank may be stuck in a collective: do not wait for it
    finally:
        for p in procs:
            p.join(timeout=10 if errors else 120)
            if p.is_alive():
                p.kill()         # the exact process this test started
                p.join()
    assert not errors, "\n".join(errors)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    got = results[0]
    B, T, H, adj, x, g1, g2, w1, b1, w2, b2 = _problem()
    t = torch.from_numpy
    ref = ref_dense.gated_block(t(x), t(adj.astype(np.float32)), t(g1), t(g2), t(w1), t(b1), t(w2), t(b2))["out"].numpy()
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-6)   # sharded == unsharded
