"""GPU parity tests (run with ``-m gpu`` on an MI355X).  Everything here goes through the
C ABI of libggcn_hip.so (the Python package only marshals pointers) and is checked against

* the committed golden fixtures the REFERENCE produced (tests/golden, oracle/make_golden.py),
* the CPU oracle on seeded inputs at sizes it finishes in seconds,
* size-independent properties at BASELINE.json's full sizes.

Tolerances (BASELINE.json north_star: 1e-4 fp32): the fp32-MFMA linear is an exact fp32 FMA
chain -> 2e-5 (summation order differs from MKL); the bf16x3 linear -> 1e-4.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import csr_ref, ref_dense  # noqa: E402  (tests may use the oracle)

TOL = {"fp32": 2e-5, "bf16x3": 1e-4, "f16mx8": 1e-4, "f16mx6": 1e-4}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def pkg():
    import ed_gated_gcn_amd as p
    p.load_library()  # fails loudly if the HIP library was not built
    return p


# (precision, fused): the one-launch layer kernel exists for the two split-precision linears
MODES = [("fp32", False), ("bf16x3", False), ("bf16x3", True), ("f16mx8", False), ("f16mx8", True), ("f16mx6", True)]
MODE_IDS = ["fp32", "bf16x3-unfused", "bf16x3-fused", "f16mx8-unfused", "f16mx8-fused", "f16mx6-fused"]
# block-level tests add the whole block as ONE launch (ggcn_block_fused: W1.W2 folded, gcn1 optional);
# "fused" there means one launch per layer
BLOCK_MODES = MODES + [("bf16x3", "block"), ("f16mx8", "block"), ("f16mx6", "block")]
BLOCK_IDS = MODE_IDS + ["bf16x3-block", "f16mx8-block", "f16mx6-block"]


def _block(pkg, x, adj, g1, g2, gc1, gc2, fused, want_gcn1=True):
    """The block through the path named by `fused` (False / True: per-layer launches; "block": one launch)."""
    return pkg.gated_gcn_block(x, adj, g1, g2, gc1, gc2, want_gcn1=want_gcn1, one_launch=(fused == "block"))


def _layer(pkg, dev, w, b, precision, fused=True):
    if precision == "f16mx6" and not pkg._capi.has_f16mx6():
        pytest.skip("f16mx6 is an experiment: libggcn_hip.so is built without it (make -C ed-gated-gcn_amd/csrc F16MX6=1)")
    m = pkg.GraphConvolution(w.shape[0], w.shape[1], opt=None, bias=b is not None).to(dev)
    m.precision = precision
    m.fused = bool(fused)
    m.fused_max_t = 256          # the tests exercise the 256-row graph slot too (the product default is 128)
    with torch.no_grad():
        m.weight.copy_(torch.from_numpy(w))
        if b is not None:
            m.bias.copy_(torch.from_numpy(b))
    return m.eval()


# ---------------------------------------------------------------- CSR builder: bit exact
@pytest.mark.parametrize("dtype", [torch.float32, torch.uint8, torch.bool, torch.int32, torch.int64,
                                   torch.float64, torch.float16])
def test_csr_from_dense_bit_exact(pkg, dev, dtype):
    from ed_gated_gcn_amd import synth
    lens = np.random.default_rng(3).integers(1, 32, size=37)
    adj = synth.dependency_batch(37, 31, 3.0, seed=5, lengths=lens)
    rowptr, colidx, vals = synth.csr_from_dense_host(adj)
    # the reference hands over a non-contiguous slice of [B,ORI_ML,ORI_ML] (bert_amir5.py:589)
    big = torch.zeros(37, 40, 40, dtype=dtype, device=dev)
    big[:, :31, :31] = torch.from_numpy(adj).to(dev).to(dtype)
    view = big[:, :31, :31]
    assert not view.is_contiguous()
    csr = pkg.BatchedCSR.from_dense(view)
    torch.cuda.synchronize()
    got_rowptr = csr.rowptr.cpu().numpy()
    nnz = int(got_rowptr[-1])
    assert np.array_equal(got_rowptr, rowptr)
    assert np.array_equal(csr.colidx[:nnz].cpu().numpy(), colidx)
    assert csr.vals is None                      # 0/1 adjacency detected on the device
    want_mask = (adj.reshape(-1, 31).astype(np.uint32) << np.arange(31, dtype=np.uint32)).sum(axis=1)
    assert np.array_equal(csr.rowmask.cpu().numpy().view(np.uint32), want_mask)
    kept = pkg.BatchedCSR.from_dense(view, binary=False)
    assert np.array_equal(kept.vals[:nnz].cpu().numpy(), vals)
    # row masks rebuilt from the CSR arrays by the library
    from ed_gated_gcn_amd import _capi
    lib = pkg.load_library()
    m2 = torch.zeros(37 * 31, dtype=torch.int32, device=dev)
    _capi.check(lib.ggcn_csr_rowmask(_capi.ptr(csr.rowptr), _capi.ptr(csr.colidx), 37, 31, _capi.ptr(m2),
                                     _capi.stream_of(dev)), "ggcn_csr_rowmask")
    assert torch.equal(m2, csr.rowmask)


def test_csr_weighted_and_large_t(pkg, dev):
    rng = np.random.default_rng(11)
    adj = (rng.random((3, 300, 300)) < 0.03).astype(np.float32) * rng.uniform(0.5, 2, (3, 300, 300)).astype(np.float32)
    adj[:, np.arange(300), np.arange(300)] = 1.0
    rowptr, colidx, vals = csr_ref.csr_from_dense(adj)
    csr = pkg.BatchedCSR.from_dense(torch.from_numpy(adj).to(dev))
    nnz = int(rowptr[-1])
    assert np.array_equal(csr.rowptr.cpu().numpy(), rowptr)
    assert np.array_equal(csr.colidx[:nnz].cpu().numpy(), colidx)
    assert np.array_equal(csr.vals[:nnz].cpu().numpy(), vals)   # weights detected -> values kept
    assert csr.rowmask is None                                   # T > 256: no row masks


def test_csr_scan_across_many_tiles(pkg, dev):
    """N = 5000*7 rows > several 1024-row scan tiles, ragged degrees."""
    rng = np.random.default_rng(2)
    adj = (rng.random((5000, 7, 7)) < 0.4).astype(np.uint8)
    rowptr, colidx, _ = csr_ref.csr_from_dense(adj.astype(np.float32))
    csr = pkg.BatchedCSR.from_dense(torch.from_numpy(adj).to(dev), binary=True)
    assert csr.vals is None
    assert np.array_equal(csr.rowptr.cpu().numpy(), rowptr)
    assert np.array_equal(csr.colidx[:int(rowptr[-1])].cpu().numpy(), colidx)


# ---------------------------------------------------------------- layer vs reference goldens
@pytest.mark.parametrize("precision,fused", MODES, ids=MODE_IDS)
def test_layer_golden_config1(pkg, dev, golden_dir, precision, fused):
    """BASELINE.json configs[0]: one 32-token sentence, hidden=300, reference CPU forward."""
    g = np.load(os.path.join(golden_dir, "gcn_config1.npz"))
    m = _layer(pkg, dev, g["weight"], g["bias"], precision, fused)
    with torch.no_grad():
        out = m(torch.from_numpy(g["text"]).to(dev), torch.from_numpy(g["adj"]).to(dev).float())
    np.testing.assert_allclose(out.cpu().numpy(), g["out"], rtol=0, atol=TOL[precision])


@pytest.mark.parametrize("precision,fused", MODES, ids=MODE_IDS)
def test_layer_golden_sweep(pkg, dev, golden_dir, precision, fused):
    g = np.load(os.path.join(golden_dir, "gcn_sweep.npz"))
    for i, desc in enumerate(g["cases"]):
        k = "c%d_" % i
        bias = g[k + "bias"] if (k + "bias") in g.files else None
        m = _layer(pkg, dev, g[k + "weight"], bias, precision, fused)
        adj = torch.from_numpy(g[k + "adj"]).to(dev)          # bool / int64 / float32 / weighted
        with torch.no_grad():
            out = m(torch.from_numpy(g[k + "text"]).to(dev), adj)
        np.testing.assert_allclose(out.cpu().numpy(), g[k + "out"], rtol=0, atol=TOL[precision],
                                   err_msg="%s (%s)" % (desc, precision))


@pytest.mark.parametrize("precision,fused", BLOCK_MODES, ids=BLOCK_IDS)
def test_gated_block_golden_bertamir55(pkg, dev, golden_dir, precision, fused):
    g = np.load(os.path.join(golden_dir, "amir55_block.npz"))
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    gc1 = _layer(pkg, dev, g["p_gc1.weight"], g["p_gc1.bias"], precision, fused)
    gc2 = _layer(pkg, dev, g["p_gc2.weight"], g["p_gc2.bias"], precision, fused)
    with torch.no_grad():
        r = _block(pkg, t("lstm_out"), t("adj"), t("gate1"), t("gate2"), gc1, gc2, fused)
    tol = TOL[precision]
    np.testing.assert_allclose(r["gcn1"].cpu().numpy(), g["gcn1"], rtol=0, atol=tol)
    np.testing.assert_allclose(r["x"].cpu().numpy(), g["gate2"][:, None, :] * g["gc2_out"], rtol=0, atol=tol)
    np.testing.assert_allclose(r["out"].cpu().numpy(), g["out"], rtol=0, atol=tol)
    assert abs(float(r["xy"]) - float(g["xy"])) <= 1e-4 * max(1.0, abs(float(g["xy"])))


# ---------------------------------------------------------------- vs the oracle on seeded inputs
def _oracle_block(x, adj, g1, g2, w1, b1, w2, b2):
    t = torch.from_numpy
    return ref_dense.gated_block(t(x), t(adj), t(g1), t(g2), t(w1), t(b1), t(w2), t(b2))


@pytest.mark.parametrize("precision,fused", BLOCK_MODES, ids=BLOCK_IDS)
@pytest.mark.parametrize("B,T,H,padded", [(64, 32, 768, False), (16, 31, 256, True), (3, 100, 200, True),
                                          (2, 231, 64, True), (1, 1, 8, False), (7, 5, 300, True),
                                          (9, 32, 100, False), (5, 17, 34, True)])
def test_gated_block_vs_oracle(pkg, dev, precision, fused, B, T, H, padded):
    from ed_gated_gcn_amd import synth
    rng = np.random.default_rng(100 + B + T)
    lens = rng.integers(max(1, T // 3), T + 1, size=B) if padded else None
    adj = synth.dependency_batch(B, T, min(4.0, T), seed=7, lengths=lens)
    x = rng.standard_normal((B, T, H)).astype(np.float32)
    g1 = (1 / (1 + np.exp(-rng.standard_normal((B, H))))).astype(np.float32)
    g2 = (1 / (1 + np.exp(-rng.standard_normal((B, H))))).astype(np.float32)
    w1, b1 = synth.layer_params(H, H, seed=1)
    w2, b2 = synth.layer_params(H, H, seed=2)
    ref = _oracle_block(x, adj.astype(np.float32), g1, g2, w1, b1, w2, b2)
    gc1, gc2 = _layer(pkg, dev, w1, b1, precision, fused), _layer(pkg, dev, w2, b2, precision, fused)
    td = lambda a: torch.from_numpy(a).to(dev)
    with torch.no_grad():
        r = _block(pkg, td(x), td(adj), td(g1), td(g2), gc1, gc2, fused)
    tol = TOL[precision]
    for k in ("gcn1", "x1", "y1", "x", "out"):
        np.testing.assert_allclose(r[k].cpu().numpy(), ref[k].numpy(), rtol=0, atol=tol, err_msg=k)
    assert abs(float(r["xy"]) - float(ref["xy"])) <= 1e-4 * max(1.0, abs(float(ref["xy"])))


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "f16mx8"])
@pytest.mark.parametrize("M,K,F", [(256, 768, 768), (1000, 300, 300), (33, 17, 5), (513, 768, 34), (7, 9216, 256)])
def test_linear_vs_float64(pkg, dev, precision, M, K, F):
    """gcn.py:34 alone, against a float64 product (odd shapes hit every edge/tail path)."""
    rng = np.random.default_rng(M + K + F)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w, _ = __import__("ed_gated_gcn_amd").synth.layer_params(K, F, seed=3)
    m = _layer(pkg, dev, w, None, precision)
    with torch.no_grad():
        y = m.linear(torch.from_numpy(x).to(dev)).cpu().numpy()
    ref = x.astype(np.float64) @ w.astype(np.float64)
    scale = np.sqrt(K) * np.sqrt(np.mean(x.astype(np.float64) ** 2) * np.mean(w.astype(np.float64) ** 2))
    # fp32: a K-long fp32 FMA chain (~sqrt(K)*2^-24 rms, a few sigma at the max); bf16x3: ~2^-16 per product;
    # f16mx8: the fp8 correction leaves ~2^-15 per product
    bound = {"fp32": 1e-5, "bf16x3": 3e-5, "f16mx8": 6e-5}[precision] * max(1.0, scale)
    assert np.max(np.abs(y - ref)) <= bound


@pytest.mark.parametrize("precision", ["bf16x3", "f16mx8"])
@pytest.mark.parametrize("B,T,H", [(130, 32, 768), (33, 31, 256), (6, 7, 96)])
def test_fused_layer_equals_unfused(pkg, dev, B, T, H, precision):
    """Same main loop, same operand planes: the one-launch layer and linear+aggregate differ only by
    the two-plane (2^-17) MFMA aggregation and one reciprocal per node."""
    from ed_gated_gcn_amd import synth
    rng = np.random.default_rng(B)
    adj = synth.dependency_batch(B, T, min(4.0, T), seed=3, lengths=rng.integers(1, T + 1, size=B))
    x = torch.from_numpy(rng.standard_normal((B, T, H)).astype(np.float32)).to(dev)
    g1 = torch.sigmoid(torch.from_numpy(rng.standard_normal((B, H)).astype(np.float32))).to(dev)
    g2 = torch.sigmoid(torch.from_numpy(rng.standard_normal((B, H)).astype(np.float32))).to(dev)
    w, b = synth.layer_params(H, H, seed=8)
    a = torch.from_numpy(adj).to(dev)
    outs = []
    for fused in (True, False):
        m = _layer(pkg, dev, w, b, precision, fused)
        with torch.no_grad():
            outs.append(m.forward_gated(x, a, store_gate=g2, pool_gate_a=g1, pool_gate_b=g2,
                                        want_pool_a=True, want_pool_b=True))
    for u, v in zip(*outs):
        assert torch.max(torch.abs(u - v)).item() <= 4e-5


def test_unaligned_and_strided_inputs(pkg, dev):
    """K, F not multiples of 4 -> scalar paths; text given as a non-contiguous view."""
    from ed_gated_gcn_amd import synth
    rng = np.random.default_rng(9)
    B, T, K, F = 5, 9, 13, 7
    adj = synth.dependency_batch(B, T, 3.0, seed=1)
    x_big = rng.standard_normal((B, T, K + 3)).astype(np.float32)
    w, b = synth.layer_params(K, F, seed=4)
    ref = ref_dense.graph_convolution(torch.from_numpy(x_big[:, :, :K].copy()), torch.from_numpy(adj),
                                      torch.from_numpy(w), torch.from_numpy(b)).numpy()
    for precision in ("fp32", "bf16x3", "f16mx8"):
        m = _layer(pkg, dev, w, b, precision)
        with torch.no_grad():
            out = m(torch.from_numpy(x_big).to(dev)[:, :, :K], torch.from_numpy(adj).to(dev))
        np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=0, atol=TOL[precision])


def test_errors_are_loud(pkg, dev):
    m = pkg.GraphConvolution(8, 8, opt=None).to(dev)
    x = torch.zeros(2, 4, 8, device=dev)
    with pytest.raises(RuntimeError):
        with torch.no_grad():
            m(x.double(), torch.zeros(2, 4, 4, device=dev))    # only float32 / float16 features
    with pytest.raises(RuntimeError):
        with torch.no_grad():
            m(x, torch.zeros(2, 5, 5, device=dev))             # shape mismatch
    with pytest.raises(RuntimeError):
        with torch.no_grad():
            m(x.cpu(), torch.zeros(2, 4, 4))                   # no CPU path


# ---------------------------------------------------------------- full-size properties (config 2)
@pytest.fixture(scope="module")
def config2(pkg, dev):
    from ed_gated_gcn_amd import synth
    B, T, H = 4096, 32, 768
    adj = synth.dependency_batch(B, T, 4.0)
    rowptr, colidx, _ = synth.csr_from_dense_host(adj)
    csr = pkg.BatchedCSR.from_arrays(rowptr, colidx, B, T, dev)
    gen = torch.Generator(device="cpu").manual_seed(synth.SEED)
    x = torch.randn(B, T, H, generator=gen).to(dev)
    g1 = torch.sigmoid(torch.randn(B, H, generator=gen)).to(dev)
    g2 = torch.sigmoid(torch.randn(B, H, generator=gen)).to(dev)
    w1, b1 = synth.layer_params(H, H, seed=1)
    w2, b2 = synth.layer_params(H, H, seed=2)
    return dict(B=B, T=T, H=H, adj=adj, csr=csr, x=x, g1=g1, g2=g2, w1=w1, b1=b1, w2=w2, b2=b2)


@pytest.mark.parametrize("precision,fused", BLOCK_MODES, ids=BLOCK_IDS)
def test_config2_full_size_properties(pkg, dev, config2, precision, fused):
    c = config2
    gc1 = _layer(pkg, dev, c["w1"], c["b1"], precision, fused)
    gc2 = _layer(pkg, dev, c["w2"], c["b2"], precision, fused)
    with torch.no_grad():
        r = _block(pkg, c["x"], c["csr"], c["g1"], c["g2"], gc1, gc2, fused)
        # (1) graphs are independent: a 64-graph slice run alone gives the same numbers
        sl = slice(1000, 1064)
        from ed_gated_gcn_amd import synth
        rp, ci, _ = synth.csr_from_dense_host(c["adj"][sl])
        sub = pkg.BatchedCSR.from_arrays(rp, ci, 64, c["T"], dev)
        rs = _block(pkg, c["x"][sl].contiguous(), sub, c["g1"][sl].contiguous(), c["g2"][sl].contiguous(), gc1, gc2, fused)
    for k in ("gcn1", "x", "out", "x1", "y1"):
        assert torch.equal(r[k][sl], rs[k]), k
    # (2) the slice equals the CPU oracle
    ref = _oracle_block(c["x"][sl].cpu().numpy(), c["adj"][sl].astype(np.float32), c["g1"][sl].cpu().numpy(),
                        c["g2"][sl].cpu().numpy(), c["w1"], c["b1"], c["w2"], c["b2"])
    for k in ("gcn1", "x1", "y1", "x", "out"):
        np.testing.assert_allclose(rs[k].cpu().numpy(), ref[k].numpy(), rtol=0, atol=TOL[precision], err_msg=k)
    # (3) pooled outputs are the max over tokens of what was stored (bert_amir5.py:635-640)
    assert torch.equal(r["out"], r["x"].max(dim=1)[0])
    assert torch.equal(r["x1"], (r["gcn1"] * c["g1"][:, None, :]).max(dim=1)[0])
    assert torch.equal(r["y1"], (r["gcn1"] * c["g2"][:, None, :]).max(dim=1)[0])
    # (4) xy regulariser against a float64 evaluation of the same pooled tensors
    xy64 = (r["x1"].double() * r["y1"].double()).sum(1).mean().item()
    assert abs(float(r["xy"]) - xy64) <= 1e-5 * abs(xy64)
    # (5) dense-adjacency entry point == CSR entry point, bit for bit
    with torch.no_grad():
        d = gc1(c["x"][:256].contiguous(), torch.from_numpy(c["adj"][:256]).to(dev).float())
    assert torch.equal(d, r["gcn1"][:256])


@pytest.mark.parametrize("precision,fused", MODES, ids=MODE_IDS)
def test_config2_linearity_and_identity(pkg, dev, config2, precision, fused):
    """The layer is affine in text: f(a*x + b*y) - bias = a*(f(x)-bias) + b*(f(y)-bias); and with an
    identity adjacency (padding rows) it is text@W / 2 + bias."""
    c = config2
    gc1 = _layer(pkg, dev, c["w1"], c["b1"], precision, fused)
    bias = torch.from_numpy(c["b1"]).to(dev)
    B = 512
    x, y = c["x"][:B].contiguous(), c["x"][B:2 * B].contiguous()
    from ed_gated_gcn_amd import synth
    rp, ci, _ = synth.csr_from_dense_host(c["adj"][:B])
    csr = pkg.BatchedCSR.from_arrays(rp, ci, B, c["T"], dev)
    with torch.no_grad():
        fx, fy = gc1(x, csr) - bias, gc1(y, csr) - bias
        fz = gc1(0.5 * x - 2.0 * y, csr) - bias
        eye = torch.eye(c["T"], device=dev).expand(B, -1, -1)
        fi = gc1(x, eye)
        lin = gc1.linear(x.reshape(B * c["T"], -1)).view(B, c["T"], -1)
    tol = TOL[precision] * 3
    assert torch.max(torch.abs(fz - (0.5 * fx - 2.0 * fy))).item() <= tol
    # fused: hidden enters the aggregation MFMA as two bf16 planes (residual <= 2^-17 |hidden|)
    np.testing.assert_allclose(fi.cpu().numpy(), (lin / 2 + bias).cpu().numpy(), rtol=0,
                               atol=2e-5 if fused else 2e-6)


def test_config4_long_document_sample(pkg, dev):
    """BASELINE.json configs[3] shape (T=512, avg degree 6, hidden=1024) on 4 graphs, fp32."""
    from ed_gated_gcn_amd import synth
    B, T, H = 4, 512, 1024
    adj = synth.dependency_batch(B, T, 6.0, seed=21)
    rng = np.random.default_rng(21)
    x = rng.standard_normal((B, T, H)).astype(np.float32)
    w, b = synth.layer_params(H, H, seed=5)
    ref = ref_dense.graph_convolution(torch.from_numpy(x), torch.from_numpy(adj), torch.from_numpy(w),
                                      torch.from_numpy(b)).numpy()
    for precision in ("fp32", "bf16x3", "f16mx8"):
        m = _layer(pkg, dev, w, b, precision)
        with torch.no_grad():
            out = m(torch.from_numpy(x).to(dev), torch.from_numpy(adj).to(dev))
        np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=0, atol=TOL[precision])


@pytest.mark.parametrize("fused", [False, True], ids=["unfused", "fused"])
def test_f16mx8_is_deterministic_and_degrades_gracefully(pkg, dev, fused):
    """Same inputs -> same bits (no atomics on the value path, no races in the staged pipeline), and
    activations far outside the fp8 window of the correction (|x| > 448 saturates it, |x| < 2^-9 flushes it)
    still give a finite result with the accuracy of the fp16 product (2^-11 relative per factor)."""
    from ed_gated_gcn_amd import synth
    rng = np.random.default_rng(77)
    B, T, H = 64, 32, 256
    adj = torch.from_numpy(synth.dependency_batch(B, T, 4.0, seed=2)).to(dev)
    w, b = synth.layer_params(H, H, seed=6)
    m = _layer(pkg, dev, w, b, "f16mx8", fused)
    x = torch.from_numpy(rng.standard_normal((B, T, H)).astype(np.float32)).to(dev)
    with torch.no_grad():
        first = m(x, adj).clone()
        for _ in range(5):
            assert torch.equal(m(x, adj), first)
    # Inside its window (|x| <= 448) the default arithmetic meets the parity gate at any magnitude (1e-4, relative
    # to values of O(1)); tiny activations lose the correction terms to fp8's underflow but then the absolute error is far
    # below the gate.  OUTSIDE the window the product silently has fp16 accuracy -- so it must not be silent: the sticky
    # flag carries GGCN_RANGE_WINDOW and check_range() raises (bf16x3 takes the same data inside the gate).
    mz = _layer(pkg, dev, w, None, "f16mx8", fused)
    mz.check_range()
    for scale in (400.0 / float(x.abs().max()), 2.0 ** -14):
        xs = x * scale
        ref = ref_dense.graph_convolution(xs.cpu(), adj.cpu(), torch.from_numpy(w), None)
        with torch.no_grad():
            out = mz(xs, adj).cpu()
        assert float((out - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))
        mz.check_range()                                   # in the window: nothing to report
    xs = x * 3000.0
    ref = ref_dense.graph_convolution(xs.cpu(), adj.cpu(), torch.from_numpy(w), None)
    with torch.no_grad():
        out = mz(xs, adj).cpu()
    assert torch.isfinite(out).all()
    assert float((out - ref).abs().max()) <= 2e-3 * float(ref.abs().max())      # fp16-product accuracy: graceful ...
    with pytest.raises(RuntimeError, match="448"):                                # ... and REPORTED
        mz.check_range()
    mb = _layer(pkg, dev, w, None, "bf16x3", fused)
    with torch.no_grad():
        out = mb(xs, adj).cpu()
    assert float((out - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    # hidden values beyond fp16 (in-window activations, large weights): the one-launch layer's fp16 aggregation planes
    # would turn them into NaN -- the bound max|x| * max_f sum_k |w[k,f]| from the weight image's trailer reports it
    if fused:
        wl = (w * 400.0).astype(np.float32)
        ml = _layer(pkg, dev, wl, None, "f16mx8", fused)
        xl = x * (300.0 / float(x.abs().max()))
        with torch.no_grad():
            ml(xl, adj)
        with pytest.raises(RuntimeError, match="hidden"):
            ml.check_range()
        with torch.no_grad():
            ml(x * 1e-3, adj)                                # the same weights with small activations: provably in range
        ml.check_range()


@pytest.mark.parametrize("precision", ["bf16x3", "f16mx8"])
@pytest.mark.parametrize("B,T,H", [(4, 512, 1024), (3, 100, 200), (5, 31, 72), (2, 9, 13)])
def test_fp16_features_config4(pkg, dev, B, T, H, precision):
    """BASELINE.json configs[3]: fp16 features, fp32 accumulation.  The reference cannot run half
    inputs, so the oracle is its fp32 forward on the fp16-ROUNDED inputs; the HIP path
    additionally rounds `hidden` and the output to fp16 -> atol 2e-3."""
    from ed_gated_gcn_amd import synth
    rng = np.random.default_rng(B * T)
    adj = synth.dependency_batch(B, T, min(6.0, T), seed=21)
    x16 = torch.from_numpy(rng.standard_normal((B, T, H)).astype(np.float32)).half()
    g = torch.sigmoid(torch.from_numpy(rng.standard_normal((B, H)).astype(np.float32)))
    w, b = synth.layer_params(H, H, seed=5)
    ref = ref_dense.graph_convolution(x16.float(), torch.from_numpy(adj), torch.from_numpy(w), torch.from_numpy(b))
    ref_gated = (ref * g[:, None, :])
    m = _layer(pkg, dev, w, b, precision)
    with torch.no_grad():
        out, pa, _ = m.forward_gated(x16.to(dev), torch.from_numpy(adj).to(dev), store_gate=g.to(dev),
                                     pool_gate_a=g.to(dev), want_pool_a=True)
    assert out.dtype == torch.float16 and pa.dtype == torch.float32
    np.testing.assert_allclose(out.float().cpu().numpy(), ref_gated.numpy(), rtol=0, atol=2e-3)
    np.testing.assert_allclose(pa.cpu().numpy(), ref_gated.max(dim=1)[0].numpy(), rtol=0, atol=2e-3)
    mf = _layer(pkg, dev, w, b, "fp32")
    with pytest.raises(RuntimeError):
        with torch.no_grad():
            mf(x16.to(dev), torch.from_numpy(adj).to(dev))


# ---------------------------------------------------------------- backward
def _grad_close(got, want, name, rel=2e-4):
    scale = float(want.abs().max()) + 1e-12
    err = float((got.cpu() - want).abs().max())
    assert err <= rel * scale, "%s: max|diff| %.3g vs scale %.3g" % (name, err, scale)


@pytest.mark.parametrize("precision,fused", MODES, ids=MODE_IDS)
@pytest.mark.parametrize("B,T,K,F,weighted", [(8, 32, 256, 256, False), (3, 100, 64, 48, True), (5, 17, 34, 20, False), (6, 20, 64, 96, False)])
def test_layer_backward_vs_oracle_autograd(pkg, dev, precision, fused, B, T, K, F, weighted):
    """train.py:115-121 trains through gc1/gc2: gradients of the HIP layer (transposed-CSR
    aggregation + MFMA dX + library dW) against torch autograd on the oracle's dense forward."""
    from ed_gated_gcn_amd import synth
    rng = np.random.default_rng(B + T + K)
    adj = synth.dependency_batch(B, T, min(4.0, T), seed=5, lengths=rng.integers(2, T + 1, size=B)).astype(np.float32)
    if weighted:
        adj = adj * rng.uniform(0.5, 1.5, adj.shape).astype(np.float32)   # asymmetric weights
    x = torch.from_numpy(rng.standard_normal((B, T, K)).astype(np.float32))
    w, b = synth.layer_params(K, F, seed=6)
    R = torch.from_numpy(rng.standard_normal((B, T, F)).astype(np.float32))
    xr, wr, br = x.clone().requires_grad_(), torch.from_numpy(w).requires_grad_(), torch.from_numpy(b).requires_grad_()
    (ref_dense.graph_convolution(xr, torch.from_numpy(adj), wr, br) * R).sum().backward()
    m = _layer(pkg, dev, w, b, precision, fused).train()
    xg = x.to(dev).requires_grad_()
    out = m(xg, torch.from_numpy(adj).to(dev))
    (out * R.to(dev)).sum().backward()
    _grad_close(xg.grad, xr.grad, "d text")
    _grad_close(m.weight.grad, wr.grad, "d weight")
    _grad_close(m.bias.grad, br.grad, "d bias")


@pytest.mark.parametrize("B,T,F,drop", [(7, 32, 256, 0.0), (5, 17, 20, 0.0), (3, 31, 1028, 0.0), (6, 32, 64, 0.3), (1, 1, 8, 0.0)])
def test_gate_pool_backward_with_the_transposed_aggregation_in_one_launch(pkg, dev, B, T, F, drop):
    """ggcn_gate_pool_backward_agg (graphs of up to 32 nodes, 0/1 adjacency as row masks): dH = A^T.D.dY straight from the
    stored layer output, dY never in memory -- against ggcn_gate_pool_backward[_drop] + ggcn_aggregate_t on the same inputs:
    the gate gradients and the per-graph bias sums bit for bit (the same pass), dH within a few fp32 ulps of its scale (the
    same sums over t ascending; the two-call path multiplies by 1/(deg+1) inside its own loop), ragged lengths, negative
    gates, dead columns behind 1024, a one-node graph; and the shapes it must refuse."""
    from ed_gated_gcn_amd import _capi, synth
    lib = pkg.load_library()
    rng = np.random.default_rng(B * 100 + T + F)
    lens = rng.integers(1, T + 1, size=B)
    adj = synth.dependency_batch(B, T, min(4.0, T), seed=3, lengths=lens).astype(np.float32)
    csr = pkg.BatchedCSR.from_dense(torch.from_numpy(adj).to(dev))
    assert csr.is_binary and csr.rowmask is not None
    out = torch.from_numpy(rng.standard_normal((B * T, F)).astype(np.float32)).to(dev)
    sg = torch.sigmoid(torch.from_numpy(rng.standard_normal((B, F)).astype(np.float32))).to(dev)
    ga = torch.from_numpy(rng.uniform(-1.0, 1.0, (B, F)).astype(np.float32)).to(dev)
    gb = torch.sigmoid(torch.from_numpy(rng.standard_normal((B, F)).astype(np.float32))).to(dev)
    d_out = torch.from_numpy(rng.standard_normal((B * T, F)).astype(np.float32)).to(dev)
    d_pa = torch.from_numpy(rng.standard_normal((B, F)).astype(np.float32)).to(dev)
    d_pb = torch.from_numpy(rng.standard_normal((B, F)).astype(np.float32)).to(dev)
    p, st = _capi.ptr, _capi.stream_of(dev)
    seed = 1234567

    def outs():
        return [torch.full((B, F), float("nan"), device=dev) for _ in range(4)]

    with torch.cuda.device(dev):
        dy = torch.empty(B * T, F, device=dev)
        r_sg, r_ga, r_gb, r_bs = outs()
        if drop == 0.0:
            _capi.check(lib.ggcn_gate_pool_backward(p(out), F, p(sg), p(ga), p(gb), p(d_out), F, p(d_pa), p(d_pb), B, T, F, p(dy), F,
                                                    p(r_sg), p(r_ga), p(r_gb), p(r_bs), st), "ggcn_gate_pool_backward")
        else:
            _capi.check(lib.ggcn_gate_pool_backward_drop(p(out), F, p(sg), p(ga), p(gb), p(d_out), F, p(d_pa), p(d_pb), B, T, F, p(dy), F,
                                                         p(r_sg), p(r_ga), p(r_gb), p(r_bs), drop, seed, 0, 1, 0, st),
                        "ggcn_gate_pool_backward_drop")
        csr_t, inv = csr.transposed(), csr.inv_denominators()
        dh_ref = torch.empty(B * T, F, device=dev)
        _capi.check(lib.ggcn_aggregate_t(p(dy), F, p(csr_t.rowptr), p(csr_t.colidx), p(csr_t.vals), p(inv), B, T, F, p(dh_ref), F, st),
                    "ggcn_aggregate_t")
        dh = torch.full((B * T, F), float("nan"), device=dev)
        n_sg, n_ga, n_gb, n_bs = outs()
        amax = torch.zeros(1, device=dev) if F % 256 == 0 else None      # (the maximum rides along for whole wavefronts of columns)
        rc = lib.ggcn_gate_pool_backward_agg(p(out), F, p(sg), p(ga), p(gb), p(d_out), F, p(d_pa), p(d_pb), p(csr.rowmask), B, T, F,
                                             p(dh), F, p(n_sg), p(n_ga), p(n_gb), p(n_bs), drop, seed, 0, 1, 0, p(amax), st)
        if F % 4 != 0:
            assert rc == _capi.GGCN_EUNSUPPORTED if hasattr(_capi, "GGCN_EUNSUPPORTED") else rc != 0
            return
        _capi.check(rc, "ggcn_gate_pool_backward_agg")
        for name, u, v in (("d_sg", n_sg, r_sg), ("d_ga", n_ga, r_ga), ("d_gb", n_gb, r_gb), ("d_bsum", n_bs, r_bs)):
            assert torch.equal(u, v), name
        scale = float(dh_ref.abs().max()) + 1e-30
        assert float((dh - dh_ref).abs().max()) <= 4e-7 * scale, float((dh - dh_ref).abs().max()) / scale
        assert bool(torch.isfinite(dh).all())
        if amax is not None:
            assert float(amax) == float(dh.abs().max())      # the launch's own max |dH| (what ggcn_linear_scaled scales by)
        else:
            one = torch.zeros(1, device=dev)
            assert lib.ggcn_gate_pool_backward_agg(p(out), F, p(sg), p(ga), p(gb), p(d_out), F, p(d_pa), p(d_pb), p(csr.rowmask), B, T, F,
                                                   p(dh), F, None, None, None, None, 0.0, 0, 0, 0, 0, p(one), st) != 0
        # float64 statement of the same: dH[s] = sum_t A[t,s] dY[t] / (deg_t + 1)
        a64 = torch.from_numpy(adj).double()
        w64 = 1.0 / (a64.sum(2) + 1.0)
        want = torch.einsum("bts,btf->bsf", a64 * w64[:, :, None], dy.view(B, T, F).double().cpu())
        assert float((dh.view(B, T, F).double().cpu() - want).abs().max()) <= 2e-6 * scale
        # refusals: more than 32 nodes, unaligned leading dimension
        assert lib.ggcn_gate_pool_backward_agg(p(out), F, p(sg), p(ga), p(gb), p(d_out), F, p(d_pa), p(d_pb), p(csr.rowmask), 1, 33, F,
                                               p(dh), F, None, None, None, None, 0.0, 0, 0, 0, 0, None, st) != 0
        assert lib.ggcn_gate_pool_backward_agg(p(out), F + 1, p(sg), p(ga), p(gb), None, F, None, None, p(csr.rowmask), 1, 1, F,
                                               p(dh), F, None, None, None, None, 0.0, 0, 0, 0, 0, None, st) != 0


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,F,directed", [(7, 32, 256, False), (5, 17, 20, True), (3, 31, 1028, True), (64, 32, 768, False), (1, 1, 8, False), (9, 32, 64, True)])
def test_gate_pool_backward_on_the_matrix_cores(pkg, dev, B, T, F, directed):
    """ggcn_gate_pool_backward_mma: dH_g = A_g^T . (D.dY_g) as an MFMA chain per (graph, 32 columns) against the scalar one-launch
    form (ggcn_gate_pool_backward_agg) and a float64 statement -- DIRECTED graphs too (A^T is not A: the operand comes from the
    transposed row masks), ragged lengths, negative gates, dead columns past F, a one-node graph, absent optional operands;
    max |dH| rides along; what it must refuse."""
    from ed_gated_gcn_amd import _capi, synth
    lib = pkg.load_library()
    rng = np.random.default_rng(B * 100 + T + F)
    lens = rng.integers(1, T + 1, size=B)
    adj = synth.dependency_batch(B, T, min(4.0, T), seed=3, lengths=lens).astype(np.float32)
    if directed:   # drop a random half of the arcs' reverse directions (self loops stay)
        keep = rng.random(adj.shape) < 0.5
        adj = np.where(np.triu(np.ones((T, T), bool), 1)[None] & keep, 0.0, adj).astype(np.float32)
    csr = pkg.BatchedCSR.from_dense(torch.from_numpy(adj).to(dev))
    assert csr.is_binary and csr.graph_ops is not None and csr.graph_ops_t is not None
    # the transposed masks really are the transpose
    mt = torch.empty_like(csr.rowmask)
    _capi.check(lib.ggcn_rowmask_transpose(_capi.ptr(csr.rowmask), B, T, _capi.ptr(mt), _capi.stream_of(dev)), "transpose")
    bits = ((mt.view(B, T).cpu().numpy().astype(np.int64)[:, :, None] >> np.arange(T)[None, None, :]) & 1).astype(np.float32)
    np.testing.assert_array_equal(bits, np.transpose(adj, (0, 2, 1)))
    out = torch.from_numpy(rng.standard_normal((B * T, F)).astype(np.float32)).to(dev)
    sg = torch.sigmoid(torch.from_numpy(rng.standard_normal((B, F)).astype(np.float32))).to(dev)
    ga = torch.from_numpy(rng.uniform(-1.0, 1.0, (B, F)).astype(np.float32)).to(dev)
    gb = torch.sigmoid(torch.from_numpy(rng.standard_normal((B, F)).astype(np.float32))).to(dev)
    d_out = torch.from_numpy(rng.standard_normal((B * T, F)).astype(np.float32)).to(dev)
    d_pa = torch.from_numpy(rng.standard_normal((B, F)).astype(np.float32)).to(dev)
    d_pb = torch.from_numpy(rng.standard_normal((B, F)).astype(np.float32)).to(dev)
    p, st = _capi.ptr, _capi.stream_of(dev)

    def outs():
        return [torch.full((B, F), float("nan"), device=dev) for _ in range(4)]

    def run_mma(sg_, ga_, gb_, do_, pa_, pb_, with_amax=True):
        dh = torch.full((B * T, F), float("nan"), device=dev)
        o = outs()
        amax = torch.zeros(1, device=dev) if with_amax else None
        _capi.check(lib.ggcn_gate_pool_backward_mma(p(out), F, p(sg_), p(ga_), p(gb_), p(do_), F, p(pa_), p(pb_), p(csr.graph_ops), p(csr.graph_ops_t),
                                                    B, T, F, p(dh), F, p(o[0]) if sg_ is not None else None, p(o[1]) if pa_ is not None else None,
                                                    p(o[2]) if pb_ is not None else None, p(o[3]), p(amax), st), "ggcn_gate_pool_backward_mma")
        return dh, o, amax

    def run_ref(sg_, ga_, gb_, do_, pa_, pb_):
        dh = torch.full((B * T, F), float("nan"), device=dev)
        o = outs()
        _capi.check(lib.ggcn_gate_pool_backward_agg(p(out), F, p(sg_), p(ga_), p(gb_), p(do_), F, p(pa_), p(pb_), p(csr.rowmask), B, T, F, p(dh), F,
                                                    p(o[0]) if sg_ is not None else None, p(o[1]) if pa_ is not None else None,
                                                    p(o[2]) if pb_ is not None else None, p(o[3]), 0.0, 0, 0, 0, 0, None, st), "ggcn_gate_pool_backward_agg")
        return dh, o

    if F % 4 != 0:
        return
    for case in ((sg, ga, gb, d_out, d_pa, d_pb), (None, ga, None, None, d_pa, None), (sg, None, gb, d_out, None, d_pb)):
        dh, o, amax = run_mma(*case)
        dh_ref, o_ref = run_ref(*case)
        scale = float(dh_ref.abs().max()) + 1e-30
        assert bool(torch.isfinite(dh).all())
        assert float((dh - dh_ref).abs().max()) <= 4e-7 * scale, float((dh - dh_ref).abs().max()) / scale
        assert float(amax) == float(dh.abs().max())
        for name, u, v, used in (("d_sg", o[0], o_ref[0], case[0] is not None), ("d_ga", o[1], o_ref[1], case[4] is not None),
                                 ("d_gb", o[2], o_ref[2], case[5] is not None), ("d_bsum", o[3], o_ref[3], True)):
            if used:
                assert float((u - v).abs().max()) <= 2e-6 * (float(v.abs().max()) + 1e-30), name
    # float64 statement for the full case: dH[s] = sum_t A[t,s] dY[t] / (deg_t + 1), dY from the scalar two-call kernel
    dy = torch.empty(B * T, F, device=dev)
    r = outs()
    _capi.check(lib.ggcn_gate_pool_backward(p(out), F, p(sg), p(ga), p(gb), p(d_out), F, p(d_pa), p(d_pb), B, T, F, p(dy), F,
                                            p(r[0]), p(r[1]), p(r[2]), p(r[3]), st), "ggcn_gate_pool_backward")
    a64 = torch.from_numpy(adj).double()
    w64 = 1.0 / (a64.sum(2) + 1.0)
    want = torch.einsum("bts,btf->bsf", a64 * w64[:, :, None], dy.view(B, T, F).double().cpu())
    dh, o, _ = run_mma(sg, ga, gb, d_out, d_pa, d_pb, with_amax=False)
    assert float((dh.view(B, T, F).double().cpu() - want).abs().max()) <= 2e-6 * (float(want.abs().max()) + 1e-30)
    assert torch.equal(o[1], r[1]) and torch.equal(o[2], r[2])      # d_ga, d_gb: the same winners, the same products
    # refusals
    assert lib.ggcn_gate_pool_backward_mma(p(out), F, None, None, None, p(d_out), F, None, None, p(csr.graph_ops), p(csr.graph_ops_t), 1, 33, F,
                                           p(dh), F, None, None, None, None, None, st) != 0
    assert lib.ggcn_gate_pool_backward_mma(p(out), F, None, None, None, p(d_out), F, None, None, p(csr.graph_ops), None, B, T, F,
                                           p(dh), F, None, None, None, None, None, st) != 0


@pytest.mark.gpu
@pytest.mark.parametrize("M,K,F,scale", [(4096, 768, 768, 1.0), (300, 96, 64, 3.0e-9), (1000, 256, 128, 2.0e7), (128, 32, 4, 1.0e-30)])
def test_scaled_f16mx8_linear_for_gradients_of_any_magnitude(pkg, dev, M, K, F, scale):
    """ggcn_linear_scaled (the backward's dX = dH . W^T, train.py:120): the two-unit f16mx8 product on rows far outside fp16's
    range -- a power of two derived on the device from max |x| brings them inside first.  Against float64, relative to the
    result's scale, whatever the magnitude (1e-30 ... 1e7); a column of exact zeros and an outlier row keep their meaning;
    a non-finite maximum leaves the data alone."""
    from ed_gated_gcn_amd import _capi
    lib = pkg.load_library()
    rng = np.random.default_rng(M + K)
    x = torch.from_numpy((rng.standard_normal((M, K)) * scale).astype(np.float32)).to(dev)
    x[:, 3] = 0.0
    x[M // 2] *= 50.0                                            # an outlier row sets the scale; the others keep ~2^-16 of IT
    w = torch.from_numpy(rng.uniform(-0.06, 0.06, (K, F)).astype(np.float32)).to(dev)
    st, P = _capi.stream_of(dev), _capi.ptr
    prec = _capi.PREC["f16mx8"]
    pack = torch.empty(lib.ggcn_weight_pack_bytes(K, F, prec), dtype=torch.uint8, device=dev)
    _capi.check(lib.ggcn_weight_pack(P(w), F, K, F, prec, 0, P(pack), st), "pack")
    amax = x.abs().max().reshape(1).contiguous()
    y = torch.full((M, F), float("nan"), device=dev)
    _capi.check(lib.ggcn_linear_scaled(P(x), K, P(pack), P(y), F, M, K, F, P(amax), st), "ggcn_linear_scaled")
    want = x.double().cpu() @ w.double().cpu()
    err = float((y.double().cpu() - want).abs().max())
    assert err <= 6e-5 * float(want.abs().max()), (err, float(want.abs().max()))
    # rows of ordinary size next to the outlier: still ~1e-4 of THEIR scale while the outlier is 50x larger
    rows = [r for r in range(min(M, 64)) if r != M // 2]
    err_r = float((y[rows].double().cpu() - want[rows]).abs().max())
    assert err_r <= 2e-3 * float(want[rows].abs().max())
    # the plain f16mx8 linear on the same tiny data loses it (that is why the backward used bf16x3): only checked where fp16 flushes
    if scale < 1e-8:
        y0 = torch.empty(M, F, device=dev)
        _capi.check(lib.ggcn_linear(P(x), K, P(w), F, P(pack), P(y0), F, M, K, F, prec, st), "ggcn_linear")
        assert float((y0.double().cpu() - want).abs().max()) > 100 * err
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        lib.ggcn_range_flag(P(flag), 1, st)
    inf = torch.full((1,), float("inf"), device=dev)
    _capi.check(lib.ggcn_linear_scaled(P(x), K, P(pack), P(y), F, M, K, F, P(inf), st), "ggcn_linear_scaled(inf)")
    torch.cuda.synchronize()
    assert lib.ggcn_linear_scaled(P(x), K + 1, P(pack), P(y), F, M, K + 1, F, P(amax), st) != 0     # K % 32 != 0: refused


@pytest.mark.parametrize("precision,fused", MODES, ids=MODE_IDS)
def test_gated_block_backward_vs_oracle_autograd(pkg, dev, precision, fused):
    """The whole block of bert_amir5.py:621-640 under autograd: loss touches out, x and xy
    (train.py:115-117: CE + gate_w*xy + ...)."""
    from ed_gated_gcn_amd import synth
    B, T, H = 12, 31, 128
    rng = np.random.default_rng(77)
    adj = synth.dependency_batch(B, T, 3.5, seed=8, lengths=rng.integers(4, T + 1, size=B))
    t = torch.from_numpy
    x = t(rng.standard_normal((B, T, H)).astype(np.float32))
    g1 = torch.sigmoid(t(rng.standard_normal((B, H)).astype(np.float32)))
    g2 = torch.sigmoid(t(rng.standard_normal((B, H)).astype(np.float32)))
    w1, b1 = synth.layer_params(H, H, seed=1)
    w2, b2 = synth.layer_params(H, H, seed=2)
    R1 = t(rng.standard_normal((B, H)).astype(np.float32))
    R2 = t(rng.standard_normal((B, T, H)).astype(np.float32))

    def loss_of(r, R1, R2):
        return (r["out"] * R1).sum() + 0.1 * (r["x"] * R2).sum() + 0.01 * r["xy"]

    gc1, gc2 = _layer(pkg, dev, w1, b1, precision, fused).train(), _layer(pkg, dev, w2, b2, precision, fused).train()
    xg, g1g, g2g = (v.to(dev).requires_grad_() for v in (x, g1, g2))
    r = pkg.gated_gcn_block(xg, t(adj).to(dev), g1g, g2g, gc1, gc2)
    loss_of(r, R1.to(dev), R2.to(dev)).backward()

    # Reference gradients: torch autograd on the oracle's formula.  A max-pool routes its gradient
    # to the argmax row, and a ~1e-5 forward difference (bf16x3) can flip a near-tie, so the
    # reference takes the rows the GPU forward selected (gather instead of max); that the
    # selections are maxima of the reference values too is asserted separately.
    with torch.no_grad():
        i_x1 = (r["gcn1"] * g1g[:, None, :]).argmax(dim=1).cpu()
        i_y1 = (r["gcn1"] * g2g[:, None, :]).argmax(dim=1).cpu()
        i_out = r["x"].argmax(dim=1).cpu()
    leaves = [v.clone().requires_grad_() for v in (x, g1, g2, t(w1), t(b1), t(w2), t(b2))]
    lx, lg1, lg2, lw1, lb1, lw2, lb2 = leaves
    a32 = t(adj.astype(np.float32))
    gcn1 = ref_dense.graph_convolution(lx, a32, lw1, lb1)
    x2 = lg2[:, None, :] * ref_dense.graph_convolution(gcn1, a32, lw2, lb2)
    pick = lambda v, i: v.gather(1, i[:, None, :]).squeeze(1)
    rr = {"x": x2, "out": pick(x2, i_out),
          "xy": (pick(gcn1 * lg1[:, None, :], i_x1) * pick(gcn1 * lg2[:, None, :], i_y1)).sum(1).mean()}
    assert float((x2.max(dim=1)[0] - rr["out"]).detach().abs().max()) <= 1e-4      # the picks are (near-)maxima
    loss_of(rr, R1, R2).backward()
    got = [xg.grad, g1g.grad, g2g.grad, gc1.weight.grad, gc1.bias.grad, gc2.weight.grad, gc2.bias.grad]
    for name, gv, lv in zip(("x", "gate1", "gate2", "w1", "b1", "w2", "b2"), got, leaves):
        _grad_close(gv, lv.grad, name, rel=5e-4)
    # forward values under autograd equal the inference path
    with torch.no_grad():
        ri = pkg.gated_gcn_block(xg.detach(), t(adj).to(dev), g1g.detach(), g2g.detach(), gc1, gc2, want_gcn1=True,
                                 one_launch=False)
    for k in ("gcn1", "x", "out", "x1", "y1"):
        assert torch.max(torch.abs(ri[k] - r[k].detach())).item() <= 4e-5, k


# ---------------------------------------------------------------- config 5: the classifier end to end
def _ace_batch(rng, B, ORI_ML, BERT_ML, vocab=None):
    from ed_gated_gcn_amd import synth
    sent_len = rng.integers(5, ORI_ML + 1, size=B)
    sent_len[0] = ORI_ML
    bert_len = np.minimum(sent_len + rng.integers(2, 10, size=B), BERT_ML)
    adj = synth.dependency_batch(B, ORI_ML, 3.5, seed=12, lengths=sent_len).astype(np.float32)
    transform = np.zeros((B, ORI_ML, BERT_ML), dtype=np.float32)
    for b in range(B):                                   # word <- word pieces (data_utils.py:749-766 shape)
        for tkn in range(int(sent_len[b])):
            transform[b, tkn, 1 + min(tkn, BERT_ML - 2)] = 1.0
    ids = np.zeros((B, BERT_ML), dtype=np.int64) if vocab is None else rng.integers(0, vocab, size=(B, BERT_ML))
    return {
        "sentence_length": torch.from_numpy(sent_len), "cls_text_sep_length": torch.from_numpy(bert_len),
        "cls_text_sep_indices": torch.from_numpy(ids),
        "cls_text_sep_segments_ids": torch.zeros(B, BERT_ML, dtype=torch.long),
        "transform": torch.from_numpy(transform),
        "anchor_index": torch.from_numpy(np.array([int(rng.integers(0, n)) for n in sent_len])),
        "dist_to_target": torch.from_numpy(rng.integers(0, 6, size=(B, ORI_ML))),
        "dependency_graph": torch.from_numpy(adj),
    }


def test_config5_classifier_golden_reference_logits(pkg, dev, golden_dir):
    """G4: the HIP-backed classifier with the reference's seeded parameters and encoder stand-in
    against the logits the REFERENCE BertAmir55 produced (BASELINE configs[4], 1e-3)."""
    import types
    from oracle.ref_amir55 import BertAmir55Oracle, EncoderStandIn
    g = np.load(os.path.join(golden_dir, "amir55_full.npz"))
    oracle = BertAmir55Oracle(EncoderStandIn(int(g["seed_encoder"])), int(g["n_class"]))
    oracle.seeded_init(torch.Generator().manual_seed(int(g["seed_params"])))
    opt = types.SimpleNamespace(device=dev, dropout=0.25, polarities_dim=int(g["n_class"]))
    model = pkg.GatedGCNEventDetector(EncoderStandIn(int(g["seed_encoder"])), opt)
    model.load_state_dict(oracle.state_dict())           # same keys as the reference's state_dict
    model = model.to(dev).eval()
    inputs = {k[3:]: torch.from_numpy(g[k]).to(dev) for k in g.files if k.startswith("in_")}
    with torch.no_grad():
        logits, xy, kl, scores = model(inputs)
    np.testing.assert_allclose(logits.cpu().numpy(), g["logits"], rtol=0, atol=1e-3)
    np.testing.assert_allclose(scores.cpu().numpy(), g["scores"], rtol=0, atol=1e-2)
    assert abs(float(xy) - float(g["xy"])) <= 1e-3 * abs(float(g["xy"]))
    assert abs(float(kl) - float(g["kl"])) <= 1e-4
    # no precision was asked for: the weights bound every fp16-rounded value of the block far below 65504 (the input is
    # the BiLSTM's, |x| < 1), so inference took the faster kernel -- with a proof, not a promise
    assert model._auto_precision and model.gc1.precision == model.gc2.precision == "f16mx8"
    # weights that no longer prove it: the full-range default, same answer as the oracle restatement with those weights
    with torch.no_grad():
        model.gc1.weight.mul_(3.0e3)
        oracle.gc1.weight.mul_(3.0e3)
        logits_b, _, _, _ = model(inputs)
        ref_b = oracle.eval()({k: v.cpu() for k, v in inputs.items()})[0]
    assert model.gc1.precision == model.gc2.precision == "bf16x3"
    assert float((logits_b.cpu() - ref_b).abs().max()) <= 1e-3 * max(1.0, float(ref_b.abs().max()))
    # an explicit choice is never overridden
    opt2 = types.SimpleNamespace(device=dev, dropout=0.25, polarities_dim=int(g["n_class"]), ggcn_precision="bf16x3")
    m2 = pkg.GatedGCNEventDetector(EncoderStandIn(int(g["seed_encoder"])), opt2).to(dev).eval()
    with torch.no_grad():
        m2(inputs)
    assert not m2._auto_precision and m2.gc1.precision == "bf16x3"
    # ... and neither is a precision assigned to the layers after construction: it ends the automatic choice
    m3 = pkg.GatedGCNEventDetector(EncoderStandIn(int(g["seed_encoder"])), types.SimpleNamespace(
        device=dev, dropout=0.25, polarities_dim=int(g["n_class"]))).to(dev).eval()
    with torch.no_grad():
        m3(inputs)
    assert m3._auto_precision and m3.gc1.precision in ("f16mx8", "bf16x3")
    m3.gc1.precision = m3.gc2.precision = "fp32"
    with torch.no_grad():
        m3(inputs)
        m3.train(); m3(inputs); m3.eval()
    assert not m3._auto_precision and m3.gc1.precision == m3.gc2.precision == "fp32"
    assert isinstance(model.gc2._block_ops, dict) and len(model.gc2._block_ops) == 2   # one folded W12 per precision (f16mx8, then bf16x3), kept across switches


@pytest.mark.parametrize("cls_name,oracle_name,fixture", [("GatedGCNEventDetector54", "BertAmir54Oracle", "amir54_full.npz"),
                                                          ("GCNEventDetectorNoGate", "BertAmir55NoGateOracle", "amir55nogate_full.npz")])
def test_other_live_classifiers_golden_reference_logits(pkg, dev, golden_dir, cls_name, oracle_name, fixture):
    """G5 / G6: the HIP-backed mirrors of BertAmir54 (bert_amir5.py:434) and BertAmir55NoGate (:654) -- every model
    train.py:268-282 can select around this block -- load the reference's state_dict and reproduce the logits the REFERENCE
    classes produced (1e-3), in inference and, for the training branch, against the oracle restatement with dropout off."""
    import types
    import oracle.ref_amir55 as ra
    g = np.load(os.path.join(golden_dir, fixture))
    oracle = getattr(ra, oracle_name)(ra.EncoderStandIn(int(g["seed_encoder"])), int(g["n_class"]))
    oracle.seeded_init(torch.Generator().manual_seed(int(g["seed_params"])))
    opt = types.SimpleNamespace(device=dev, dropout=0.25, polarities_dim=int(g["n_class"]))
    model = getattr(pkg, cls_name)(ra.EncoderStandIn(int(g["seed_encoder"])), opt)
    model.load_state_dict(oracle.state_dict())           # same keys as the reference's state_dict
    model = model.to(dev).eval()
    inputs = {k[3:]: torch.from_numpy(g[k]).to(dev) for k in g.files if k.startswith("in_")}
    with torch.no_grad():
        logits, xy, kl, scores = model(inputs)
    np.testing.assert_allclose(logits.cpu().numpy(), g["logits"], rtol=0, atol=1e-3)
    np.testing.assert_allclose(scores.cpu().numpy(), g["scores"], rtol=0, atol=1e-2)
    assert abs(float(xy) - float(g["xy"])) <= 1e-3 * max(1.0, abs(float(g["xy"])))
    assert abs(float(kl) - float(g["kl"])) <= 1e-4
    # the autograd branch (train mode with p = 0: MIOpen's LSTM backward wants training mode): same numbers, and the parameters
    # receive gradients through the HIP layers
    model.train()
    model.dropout.p = 0.0
    logits2, xy2, kl2, scores2 = model(inputs)
    np.testing.assert_allclose(logits2.detach().cpu().numpy(), g["logits"], rtol=0, atol=1e-3)
    (logits2.sum() + kl2 + (xy2 if torch.is_tensor(xy2) else 0.0)).backward()
    assert model.gc1.weight.grad is not None and torch.isfinite(model.gc1.weight.grad).all()
    assert model.gc2.weight.grad is not None and float(model.gc2.weight.grad.abs().max()) > 0


def test_config5_bert_base_end_to_end(pkg, dev):
    """BASELINE configs[4]: randomly-initialised BERT-base (transformers.BertConfig(), no fetch)
    on PyTorch-ROCm + HIP gated GCN on a synthetic ACE-2005-shaped batch; logits within 1e-3 of
    the CPU oracle classifier run with the same state_dict."""
    import types
    transformers = pytest.importorskip("transformers")
    from oracle.ref_amir55 import BertAmir55Oracle
    torch.manual_seed(5)
    cfg = transformers.BertConfig()                       # bert-base shape: 12 layers, 768 hidden
    hf = transformers.BertModel(cfg).eval()
    B, ORI_ML, BERT_ML, NCLS = 8, 31, 65, 34              # constant.py:230-238 (ACE34)
    rng = np.random.default_rng(3)
    inputs = _ace_batch(rng, B, ORI_ML, BERT_ML, vocab=cfg.vocab_size)
    oracle = BertAmir55Oracle(pkg.LegacyBertAdapter(hf), NCLS)
    oracle.seeded_init(torch.Generator().manual_seed(9))
    oracle.eval()
    with torch.no_grad():
        ref_logits, ref_xy, ref_kl, ref_scores = oracle(inputs)
    opt = types.SimpleNamespace(device=dev, dropout=0.25, polarities_dim=NCLS)
    import copy
    model = pkg.GatedGCNEventDetector(pkg.LegacyBertAdapter(copy.deepcopy(hf)), opt)
    model.load_state_dict(oracle.state_dict())
    model = model.to(dev).eval()
    with torch.no_grad():
        logits, xy, kl, scores = model({k: v.to(dev) for k, v in inputs.items()})
    assert logits.shape == (B, NCLS) and scores.shape == (B, ORI_ML)
    np.testing.assert_allclose(logits.cpu().numpy(), ref_logits.numpy(), rtol=0, atol=1e-3)
    assert abs(float(xy) - float(ref_xy)) <= 1e-3 * abs(float(ref_xy))
    assert abs(float(kl) - float(ref_kl)) <= 1e-4


def test_hipgraph_capture_replays_bit_identically(pkg, dev):
    """include/ggcn.h promises enqueue-only entry points: the whole block captures into a hipGraph."""
    from ed_gated_gcn_amd import synth
    from ed_gated_gcn_amd.graphs import CapturedGatedBlock
    B, T, H = 64, 31, 256
    adj = synth.dependency_batch(B, T, 3.5, seed=4, lengths=np.random.default_rng(1).integers(5, T + 1, size=B))
    rp, ci, _ = synth.csr_from_dense_host(adj)
    csr = pkg.BatchedCSR.from_arrays(rp, ci, B, T, dev)
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(B, T, H, generator=gen).to(dev)
    g1, g2 = torch.rand(B, H, generator=gen).to(dev), torch.rand(B, H, generator=gen).to(dev)
    w1, b1 = synth.layer_params(H, H, seed=1)
    w2, b2 = synth.layer_params(H, H, seed=2)
    gc1, gc2 = _layer(pkg, dev, w1, b1, "bf16x3"), _layer(pkg, dev, w2, b2, "bf16x3")
    cap = CapturedGatedBlock(x, csr, g1, g2, gc1, gc2)
    x2 = torch.randn(B, T, H, generator=gen).to(dev)
    with torch.no_grad():
        ref = pkg.gated_gcn_block(x2, csr, g1, g2, gc1, gc2)
    got = cap(x2, g1, g2)
    torch.cuda.synchronize()
    assert ref["gcn1"] is None and got["gcn1"] is None    # the one-launch block writes gcn1 only on request
    for k in ("x", "out", "x1", "y1", "xy"):
        assert torch.equal(ref[k], got[k]), k


# ---------------------------------------------------------------- sub-word pooling
def _transform_like_reference(B, T, L, rng, ori_ml=None, bert_ml=None):
    """data_utils.py:749-766: word i covers l_i consecutive sub-word positions starting at offset 1
    ([CLS] first), each with weight 1/l_i; padded to [ORI_ML, BERT_ML]."""
    ori_ml, bert_ml = ori_ml or T, bert_ml or L
    tr = np.zeros((B, ori_ml, bert_ml), dtype=np.float32)
    for b in range(B):
        n_words = int(rng.integers(1, T + 1))
        off = 1
        for i in range(n_words):
            l = int(rng.integers(1, 5))
            if off + l >= L:
                break
            tr[b, i, off:off + l] = 1.0 / l
            off += l
    return tr


@pytest.mark.parametrize("B,T,L,D,pad", [(4, 31, 60, 9216, True), (3, 7, 300, 1000, False), (2, 5, 9, 37, True),
                                         (1, 1, 1, 4, False)])
def test_subword_pool_matches_bmm(pkg, dev, B, T, L, D, pad):
    """bert_amir5.py:600 on the non-zeros only == the dense bmm (same products; the order of a row's
    few additions may differ from the BLAS kernel's -> 1e-6 relative)."""
    rng = np.random.default_rng(B * 1000 + L)
    full = _transform_like_reference(B, T, L, rng, T + 3 if pad else None, L + 5 if pad else None)
    x = rng.standard_normal((B, L, D)).astype(np.float32)
    tr_full = torch.from_numpy(full).to(dev)
    tr = tr_full[:, :T, :L]                       # the reference's non-contiguous slice (:585)
    ref = torch.bmm(torch.from_numpy(full[:, :T, :L].copy()), torch.from_numpy(x))
    got = pkg.subword_pool(tr, torch.from_numpy(x).to(dev))
    assert got.shape == (B, T, D)
    np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), rtol=1e-6, atol=1e-6)


def test_subword_pool_dense_rows_and_backward(pkg, dev):
    """A transform with arbitrary (also negative) entries and empty rows; the backward is the transposed
    product and matches autograd through torch.bmm."""
    rng = np.random.default_rng(5)
    B, T, L, D = 3, 6, 20, 520
    a = (rng.standard_normal((B, T, L)) * (rng.random((B, T, L)) < 0.3)).astype(np.float32)
    a[1, 2, :] = 0.0
    x = rng.standard_normal((B, L, D)).astype(np.float32)
    dy = rng.standard_normal((B, T, D)).astype(np.float32)
    xr = torch.from_numpy(x).requires_grad_(True)
    ref = torch.bmm(torch.from_numpy(a), xr)
    ref.backward(torch.from_numpy(dy))
    xg = torch.from_numpy(x).to(dev).requires_grad_(True)
    got = pkg.subword_pool(torch.from_numpy(a).to(dev), xg)
    got.backward(torch.from_numpy(dy).to(dev))
    np.testing.assert_allclose(got.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-5, atol=1e-5)
    assert float(got.detach()[1, 2].abs().max()) == 0.0
    with pytest.raises(RuntimeError):
        pkg.subword_pool(torch.from_numpy(a), torch.from_numpy(x))          # no CPU path
    with pytest.raises(RuntimeError):
        pkg.subword_pool(torch.from_numpy(a).to(dev), torch.from_numpy(x).to(dev)[:, :5])  # shape mismatch


@pytest.mark.parametrize("one_launch", [False, True], ids=["two-launches", "one-launch"])
@pytest.mark.parametrize("precision", ["bf16x3", "f16mx8"])
@pytest.mark.parametrize("B,T,H", [(5, 7, 96), (9, 32, 72), (130, 31, 256), (3, 1, 8)])
def test_regulariser_folded_into_the_layer_launches(pkg, dev, B, T, H, precision, one_launch):
    """bert_amir5.py:638 without launches of its own (partials from layer 1's epilogue, reduced by layer
    2's launch) == ggcn_gate_overlap on the same x1, y1 == the oracle; and it is deterministic."""
    from ed_gated_gcn_amd import synth
    from ed_gated_gcn_amd.gated_block import gate_overlap
    rng = np.random.default_rng(B + H)
    adj = torch.from_numpy(synth.dependency_batch(B, T, min(3.0, T), seed=4, lengths=rng.integers(1, T + 1, size=B)))
    x = torch.from_numpy(rng.standard_normal((B, T, H)).astype(np.float32))
    g1 = torch.sigmoid(torch.from_numpy(rng.standard_normal((B, H)).astype(np.float32)))
    g2 = torch.sigmoid(torch.from_numpy(rng.standard_normal((B, H)).astype(np.float32)))
    (w1, b1), (w2, b2) = synth.layer_params(H, H, seed=1), synth.layer_params(H, H, seed=2)
    l1, l2 = _layer(pkg, dev, w1, b1, precision), _layer(pkg, dev, w2, b2, precision)
    with torch.no_grad():
        r = pkg.gated_gcn_block(x.to(dev), adj.to(dev), g1.to(dev), g2.to(dev), l1, l2, one_launch=one_launch)
        again = pkg.gated_gcn_block(x.to(dev), adj.to(dev), g1.to(dev), g2.to(dev), l1, l2, one_launch=one_launch)
        standalone = gate_overlap(r["x1"], r["y1"])
    ref = ref_dense.gated_block(x, adj.float(), g1, g2, torch.from_numpy(w1), torch.from_numpy(b1),
                                torch.from_numpy(w2), torch.from_numpy(b2))
    assert float(r["xy"]) == float(again["xy"])
    scale = max(1.0, abs(float(ref["xy"])))
    assert abs(float(r["xy"]) - float(standalone)) <= 2e-6 * scale      # same products, another summation tree
    assert abs(float(r["xy"]) - float(ref["xy"])) <= 1e-4 * scale


@pytest.mark.parametrize("precision", ["bf16x3", "fp32"])
@pytest.mark.parametrize("N,K,F", [(20000, 768, 768), (5000, 132, 260), (37, 8, 12), (513, 256, 34), (1, 4, 4),
                                   (301, 7, 9), (64, 33, 2), (70001, 256, 512)])
def test_weight_gradient_vs_float64(pkg, dev, precision, N, K, F):
    """dW = X^T . dH (backward of gcn.py:34) through the C ABI, both forms, against float64; twice for
    bitwise reproducibility (fixed-order slab sums, no atomics).  (70001 x 256 x 512: whole 256-column tiles from 65 536 rows up
    take the 256 x 256 tile of the TN form, one workgroup per CU; a row count that is no multiple of anything.)"""
    from ed_gated_gcn_amd import _capi
    lib = pkg.load_library()
    rng = np.random.default_rng(N + K)
    x = rng.standard_normal((N, K)).astype(np.float32)
    g = (rng.standard_normal((N, F)) * 1e-3).astype(np.float32)      # gradient-sized values
    ref = x.astype(np.float64).T @ g.astype(np.float64)
    xd, gd = torch.from_numpy(x).to(dev), torch.from_numpy(g).to(dev)
    prec = _capi.PREC[precision]
    ws = torch.empty(lib.ggcn_dweight_workspace_bytes(N, K, F, prec), dtype=torch.uint8, device=dev)
    outs = []
    for _ in range(2):
        dw = torch.full((K, F), float("nan"), device=dev)
        _capi.check(lib.ggcn_dweight(_capi.ptr(xd), K, _capi.ptr(gd), F, N, K, F, _capi.ptr(dw), F, prec,
                                     _capi.ptr(ws), _capi.stream_of(dev)), "ggcn_dweight")
        outs.append(dw.cpu().numpy())
    assert np.array_equal(outs[0], outs[1])
    scale = np.sqrt(N) * 1e-3
    bound = {"fp32": 2e-6, "bf16x3": 3e-5}[precision] * max(scale, 1e-3)
    if precision == "fp32" and N > 50000:
        bound *= 2.0     # (fp32 chains three times as long as the shapes the bound was set on)
    assert np.max(np.abs(outs[0] - ref)) <= bound
    rc = lib.ggcn_dweight(_capi.ptr(xd), K, _capi.ptr(gd), F, N, K, F, _capi.ptr(dw), F, _capi.PREC["f16mx8"],
                          _capi.ptr(ws), _capi.stream_of(dev))
    assert rc != 0 and b"range" in lib.ggcn_last_error()


@pytest.mark.parametrize("T,degree,weighted", [(512, 10.0, False), (300, 5.0, True), (64, 3.0, True), (49, 48.0, False)])
def test_fp16_long_graphs_lds_slab_form(pkg, dev, T, degree, weighted):
    """fp16 features, 48 < T <= 768: the graph's 128-byte column slab and its CSR live in LDS.  Covers
    > 4096 edges per graph (indices stay in global memory), edge weights, and a dense graph."""
    from ed_gated_gcn_amd import synth
    rng = np.random.default_rng(T)
    B, H = 3, 136
    adj = synth.dependency_batch(B, T, min(degree, T), seed=T, lengths=np.array([T, T - 7, T // 2]))
    if weighted:
        adj = adj * rng.uniform(0.25, 2.0, size=adj.shape).astype(np.float32)
    x16 = torch.from_numpy(rng.standard_normal((B, T, H)).astype(np.float32)).half()
    g = torch.sigmoid(torch.from_numpy(rng.standard_normal((B, H)).astype(np.float32)))
    w, b = synth.layer_params(H, H, seed=5)
    ref = ref_dense.graph_convolution(x16.float(), torch.from_numpy(adj), torch.from_numpy(w), torch.from_numpy(b))
    m = _layer(pkg, dev, w, b, "bf16x3")
    with torch.no_grad():
        out, pa, pb = m.forward_gated(x16.to(dev), torch.from_numpy(adj).to(dev), store_gate=g.to(dev),
                                      pool_gate_a=g.to(dev), want_pool_a=True, want_pool_b=True)
    np.testing.assert_allclose(out.float().cpu().numpy(), (ref * g[:, None, :]).numpy(), rtol=0, atol=3e-3)
    np.testing.assert_allclose(pa.cpu().numpy(), (ref * g[:, None, :]).max(dim=1)[0].numpy(), rtol=0, atol=3e-3)
    np.testing.assert_allclose(pb.cpu().numpy(), ref.max(dim=1)[0].numpy(), rtol=0, atol=3e-3)


@pytest.mark.parametrize("B,T,K,F,degree,weighted", [(3, 512, 128, 256, 6.0, False), (2, 300, 64, 136, 5.0, True),
                                                       (9, 129, 192, 128, 4.0, False), (2, 512, 64, 72, 20.0, False),
                                                       (4, 400, 1024, 1024, 6.0, False)])
def test_fp16_long_graphs_one_launch(pkg, dev, B, T, K, F, degree, weighted):
    """ggcn_layer_fused_h (half features, precision "f16", 129..512 nodes): linear + neighbour sums in one launch with
    `hidden` kept in LDS -- against the oracle on the fp16-rounded inputs (config 4's gate, SURVEY 8d: 2e-3 + the fp16
    rounding of the output), against the two-launch path of the same precision, with ragged lengths, edge weights,
    > 4096 edges per graph (indices from global memory), pools only, and the shapes it must refuse."""
    from ed_gated_gcn_amd import synth
    rng = np.random.default_rng(T + K)
    lens = np.array([T] + [int(v) for v in rng.integers(T // 3, T + 1, size=B - 1)])
    adj = synth.dependency_batch(B, T, min(degree, T), seed=T, lengths=lens).astype(np.float32)
    if weighted:
        adj = adj * rng.uniform(0.25, 2.0, size=adj.shape).astype(np.float32)
    x16 = torch.from_numpy(rng.standard_normal((B, T, K)).astype(np.float32)).half()
    g1 = torch.from_numpy(rng.uniform(-1.0, 1.0, (B, F)).astype(np.float32))    # negative gates: max picks the other end
    g2 = torch.sigmoid(torch.from_numpy(rng.standard_normal((B, F)).astype(np.float32)))
    w, b = synth.layer_params(K, F, seed=5)
    ref = ref_dense.graph_convolution(x16.float(), torch.from_numpy(adj), torch.from_numpy(w), torch.from_numpy(b))
    m, two = _layer(pkg, dev, w, b, "f16"), _layer(pkg, dev, w, b, "f16", fused=False)
    xd, ad = x16.to(dev), torch.from_numpy(adj).to(dev)
    csr = pkg.BatchedCSR.from_dense(ad)
    assert m.takes_long_path(xd, csr) and not two.takes_long_path(xd, csr)
    kw = dict(store_gate=g2.to(dev), pool_gate_a=g1.to(dev), pool_gate_b=g2.to(dev), want_pool_a=True, want_pool_b=True)
    with torch.no_grad():
        out, pa, pb = m.forward_gated(xd, csr, **kw)
        out2, pa2, pb2 = two.forward_gated(xd, csr, **kw)
        _, pa3, pb3 = m.forward_gated(xd, csr, pool_gate_a=g1.to(dev), pool_gate_b=g2.to(dev), want_out=False,
                                      want_pool_a=True, want_pool_b=True)
    assert out.dtype == torch.float16
    scale = max(1.0, float(ref.abs().max()))
    tol = 2e-3 * scale
    np.testing.assert_allclose(out.float().cpu().numpy(), (ref * g2[:, None, :]).numpy(), rtol=0, atol=tol + scale * 2.0 ** -11)
    np.testing.assert_allclose(pa.cpu().numpy(), (ref * g1[:, None, :]).max(dim=1)[0].numpy(), rtol=0, atol=tol)
    np.testing.assert_allclose(pb.cpu().numpy(), (ref * g2[:, None, :]).max(dim=1)[0].numpy(), rtol=0, atol=tol)
    # same arithmetic as the two launches (fp16 MFMA, hidden rounded to fp16, fp32 sums): only the summation order differs
    assert float((out.float() - out2.float()).abs().max()) <= scale * 2.0 ** -10
    assert float((pa - pa2).abs().max()) <= 1e-4 * scale and float((pb - pb2).abs().max()) <= 1e-4 * scale
    assert torch.equal(pa3, pa) and torch.equal(pb3, pb)


def test_long_graph_mfma_sums_edge_structures(pkg, dev):
    """The MFMA form of the long-graph neighbour sums walks the EDGE SLOTS of a 32-row block 16 at a time, builds the
    selection matrix from row pointers alone and gathers source rows by transposed LDS reads: the structures that stress
    that -- a hub row whose edges span many steps, rows without edges, a graph without any edge, a graph with exactly the
    4096 edges the staging area holds (and one with 4097: lane sums from global ids), a last row block that straddles T, an
    output width whose last tile has dead columns, a gather whose four rows share one bank class -- against the oracle, the
    two launches and the lane form; with and without the row output."""
    from ed_gated_gcn_amd import synth
    rng = np.random.default_rng(77)
    B, T, K, F = 6, 333, 128, 200
    adj = np.zeros((B, T, T), dtype=np.float32)
    adj[0, 5, :] = 1.0                                   # hub: 333 edges in one row (21 steps)
    adj[0, 40:60, 7] = 1.0                               # twenty rows with one and the same source
    adj[0, 200, [4, 8, 12, 16, 20, 24, 28, 32]] = 1.0    # sources of one bank class (row & 3 == 0)
    # graph 1: no edge at all
    adj[2] = (rng.random((T, T)) < 0.02).astype(np.float32)
    adj[2, 100:140, :] = 0.0                             # forty rows without edges (a whole 32-row block among them)
    for g, want in ((3, 4096), (4, 4097)):               # the staging area's capacity, and one more
        a = (rng.random((T, T)) < 0.03).astype(np.float32)
        idx = np.flatnonzero(a.ravel())
        off = np.flatnonzero(a.ravel() == 0)
        if len(idx) > want:
            a.ravel()[rng.choice(idx, len(idx) - want, replace=False)] = 0.0
        else:
            a.ravel()[rng.choice(off, want - len(idx), replace=False)] = 1.0
        assert int(a.sum()) == want
        adj[g] = a
    adj[5] = synth.dependency_batch(1, T, 6.0, seed=9).astype(np.float32)[0]
    x16 = torch.from_numpy(rng.standard_normal((B, T, K)).astype(np.float32)).half()
    g1 = torch.from_numpy(rng.uniform(-1.0, 1.0, (B, F)).astype(np.float32))
    g2 = torch.sigmoid(torch.from_numpy(rng.standard_normal((B, F)).astype(np.float32)))
    w, b = synth.layer_params(K, F, seed=11)
    ref = ref_dense.graph_convolution(x16.float(), torch.from_numpy(adj), torch.from_numpy(w), torch.from_numpy(b))
    one, two = _layer(pkg, dev, w, b, "f16"), _layer(pkg, dev, w, b, "f16", fused=False)
    xd = x16.to(dev)
    csr = pkg.BatchedCSR.from_dense(torch.from_numpy(adj).to(dev))
    assert one.takes_long_path(xd, csr)
    kw = dict(store_gate=g2.to(dev), pool_gate_a=g1.to(dev), pool_gate_b=g2.to(dev), want_pool_a=True, want_pool_b=True)
    with torch.no_grad():
        got = one.forward_gated(xd, csr, **kw)
        ref2 = two.forward_gated(xd, csr, **kw)
        with _lane_sums():
            lane = one.forward_gated(xd, csr, **kw)
        _, pa3, pb3 = one.forward_gated(xd, csr, pool_gate_a=g1.to(dev), pool_gate_b=g2.to(dev), want_out=False,
                                        want_pool_a=True, want_pool_b=True)
    scale = max(1.0, float(ref.abs().max()))
    out, pa, pb = got
    np.testing.assert_allclose(out.float().cpu().numpy(), (ref * g2[:, None, :]).numpy(), rtol=0, atol=2e-3 * scale + scale * 2.0 ** -11)
    np.testing.assert_allclose(pa.cpu().numpy(), (ref * g1[:, None, :]).max(dim=1)[0].numpy(), rtol=0, atol=2e-3 * scale)
    np.testing.assert_allclose(pb.cpu().numpy(), (ref * g2[:, None, :]).max(dim=1)[0].numpy(), rtol=0, atol=2e-3 * scale)
    for name, u, v in zip(("out", "pool_a", "pool_b"), lane, ref2):
        assert torch.equal(u, v), "lane sums vs two launches: %s" % name
    _assert_same_sums(got, ref2, scale, "edge structures")
    assert torch.equal(pa3, pa) and torch.equal(pb3, pb)


def test_config4_full_size_properties(pkg, dev):
    """BASELINE.json configs[3] at full size (256 graphs x 512 tokens, degree 6, hidden 1024, fp16 features) through the
    one-launch layer: (1) the same sums as linear + aggregate (bit-identical with neighbour sums by lanes; with the
    sums on the MFMAs -- the default for unweighted graphs -- the same exact terms added in fp32 in another order); (2) graphs are independent -- an 8-graph slice run
    alone gives the same numbers; (3) that slice equals the oracle on the fp16-rounded inputs; (4) the pools are the
    max over tokens of gate x what was stored (gate >= 0); (5) doubling the store gate doubles the output exactly."""
    from ed_gated_gcn_amd import synth
    B, T, H = 256, 512, 1024
    adj = synth.dependency_batch(B, T, 6.0)
    rp, ci, _ = synth.csr_from_dense_host(adj)
    csr = pkg.BatchedCSR.from_arrays(rp, ci, B, T, dev)
    gen = torch.Generator().manual_seed(synth.SEED)
    x = torch.randn(B, T, H, generator=gen).half().to(dev)
    g1 = torch.sigmoid(torch.randn(B, H, generator=gen)).to(dev)
    g2 = torch.sigmoid(torch.randn(B, H, generator=gen)).to(dev)
    w, b = synth.layer_params(H, H, seed=1)
    one, two = _layer(pkg, dev, w, b, "f16"), _layer(pkg, dev, w, b, "f16", fused=False)
    assert one.takes_long_path(x, csr)
    kw = dict(want_pool_a=True, want_pool_b=True)
    with torch.no_grad():
        out, pa, pb = one.forward_gated(x, csr, store_gate=g2, pool_gate_a=g1, pool_gate_b=g2, **kw)
        out2, pa2, pb2 = two.forward_gated(x, csr, store_gate=g2, pool_gate_a=g1, pool_gate_b=g2, **kw)
        with _lane_sums():
            outl, pal, pbl = one.forward_gated(x, csr, store_gate=g2, pool_gate_a=g1, pool_gate_b=g2, **kw)
        assert torch.equal(outl, out2) and torch.equal(pal, pa2) and torch.equal(pbl, pb2)       # (1) lane sums: bit for bit
        _assert_same_sums((out, pa, pb), (out2, pa2, pb2), max(1.0, float(out2.float().abs().max())), "config 4")   # (1) MFMA sums
        sl = slice(100, 108)
        rps, cis, _ = synth.csr_from_dense_host(adj[sl])
        sub = pkg.BatchedCSR.from_arrays(rps, cis, 8, T, dev)
        outs, pas, pbs = one.forward_gated(x[sl].contiguous(), sub, store_gate=g2[sl].contiguous(),
                                           pool_gate_a=g1[sl].contiguous(), pool_gate_b=g2[sl].contiguous(), **kw)
        assert torch.equal(out[sl], outs) and torch.equal(pa[sl], pas) and torch.equal(pb[sl], pbs)   # (2)
        plain, _, pmax = one.forward_gated(x, csr, pool_gate_b=g2, want_pool_b=True)                  # no store gate
        twice, _, _ = one.forward_gated(x, csr, store_gate=2.0 * torch.ones_like(g2))
    ref = ref_dense.graph_convolution(x[sl].float().cpu(), torch.from_numpy(adj[sl].astype(np.float32)), torch.from_numpy(w),
                                      torch.from_numpy(b))
    scale = max(1.0, float(ref.abs().max()))
    np.testing.assert_allclose(outs.float().cpu().numpy(), (ref * g2[sl].cpu()[:, None, :]).numpy(), rtol=0,
                               atol=2e-3 * scale + scale * 2.0 ** -11)                                 # (3)
    np.testing.assert_allclose(pbs.cpu().numpy(), (ref * g2[sl].cpu()[:, None, :]).max(dim=1)[0].numpy(), rtol=0, atol=2e-3 * scale)
    # (4) fp32 pools vs the fp16-rounded stored rows: equal up to that rounding
    assert float((pmax - (plain.float() * g2[:, None, :]).max(dim=1)[0]).abs().max()) <= scale * 2.0 ** -10
    normal = plain.float().abs() >= 2.0 ** -13      # doubling commutes with the fp16 rounding except among subnormals
    assert torch.equal(twice.float()[normal], 2.0 * plain.float()[normal])                              # (5)
    assert float((twice.float() - 2.0 * plain.float()).abs().max()) <= 2.0 ** -23


def test_fp16_long_graphs_random_shapes_match_two_launches_bitwise(pkg, dev):
    """The one-launch long-graph layer stages both operands by LDS-DMA behind hand-placed waits: a race would show as
    a rare wrong tile.  40 random shapes (T 129..512, K multiple of 64, F multiple of 8, ragged lengths, weighted or
    not, B not a multiple of 8), each compared BIT FOR BIT with linear + aggregate of the same arithmetic (neighbour
    sums by lanes), the same launch repeated three times; then the default form (sums on the MFMAs where it applies)
    against the same reference and against itself."""
    from ed_gated_gcn_amd import synth
    rng = np.random.default_rng(2024)
    for case in range(40):
        B = int(rng.integers(1, 12))
        T = int(rng.integers(129, 513))
        K = 64 * int(rng.integers(1, 9))
        F = 8 * int(rng.integers(1, 41))
        deg = float(rng.uniform(2.0, 14.0))
        weighted = bool(rng.integers(0, 2))
        lens = rng.integers(T // 4, T + 1, size=B)
        adj = synth.dependency_batch(B, T, min(deg, T), seed=case, lengths=lens).astype(np.float32)
        if weighted:
            adj = adj * rng.uniform(0.25, 2.0, size=adj.shape).astype(np.float32)
        x = torch.from_numpy(rng.standard_normal((B, T, K)).astype(np.float32)).half().to(dev)
        g = torch.from_numpy(rng.uniform(-1.0, 1.0, (B, F)).astype(np.float32)).to(dev)
        w, b = synth.layer_params(K, F, seed=case)
        one, two = _layer(pkg, dev, w, b, "f16"), _layer(pkg, dev, w, b, "f16", fused=False)
        csr = pkg.BatchedCSR.from_dense(torch.from_numpy(adj).to(dev))
        assert one.takes_long_path(x, csr), (B, T, K, F)
        kw = dict(store_gate=g, pool_gate_a=g, pool_gate_b=g.abs(), want_pool_a=True, want_pool_b=True)
        what = "case %d (B=%d T=%d K=%d F=%d deg=%.1f weighted=%s)" % (case, B, T, K, F, deg, weighted)
        with torch.no_grad():
            ref = two.forward_gated(x, csr, **kw)
            with _lane_sums():
                for rep in range(3):
                    got = one.forward_gated(x, csr, **kw)
                    for name, u, v in zip(("out", "pool_a", "pool_b"), got, ref):
                        assert torch.equal(u, v), "%s rep %d, lane sums: %s differs, max %g" % (
                            what, rep, name, float((u.float() - v.float()).abs().max()))
            # the default form (MFMA sums for unweighted graphs whose ids fit the staging area): the same terms in another
            # order against the two launches; launch after launch bit for bit
            first = one.forward_gated(x, csr, **kw)
            _assert_same_sums(first, ref, max(1.0, float(ref[0].float().abs().max())), what)
            for rep in range(2):
                got = one.forward_gated(x, csr, **kw)
                for name, u, v in zip(("out", "pool_a", "pool_b"), got, first):
                    assert torch.equal(u, v), "%s rep %d: %s differs between launches" % (what, rep, name)


POISONS = (0x7F800000, 0xFF800000, 0xFFFFFFFF, 0x7C007C00)   # +inf, -inf, NaN / id 65535, fp16 +inf pairs


class _lane_sums:
    """The long-graph layer's neighbour sums by lanes (fp32 additions in CSR order: the arithmetic of ggcn_aggregate_h, bit for
    bit) instead of on the MFMAs (the same fp32 sums of the same exact terms in the matrix pipe's order).  The library reads
    the switch on every call."""
    def __enter__(self):
        os.environ["GGCN_LONG_LANE_SUMS"] = "1"
    def __exit__(self, *exc):
        os.environ.pop("GGCN_LONG_LANE_SUMS", None)


def _assert_same_sums(got, ref, scale, what):
    """One-launch long-graph layer with MFMA neighbour sums against the two launches: identical terms (1.0 x the fp16 hidden
    values), fp32 accumulation in another order -- the stored fp16 rows may differ by one rounding step where the two fp32
    sums straddle a rounding boundary, the fp32 pools by a few fp32 ulps of the sum."""
    out, pa, pb = got
    out2, pa2, pb2 = ref
    if out is not None:
        d = (out.float() - out2.float()).abs()
        assert bool((d <= 2.0 ** -10 * out2.float().abs() + 2.0 ** -24).all()), "%s: out differs by more than one fp16 step (max %g)" % (what, float(d.max()))
        assert float((d > 0).float().mean()) < 2e-3, "%s: %.2g of the stored values differ" % (what, float((d > 0).float().mean()))
    for name, u, v in (("pool_a", pa, pa2), ("pool_b", pb, pb2)):
        if u is not None:
            assert float((u - v).abs().max()) <= 2e-6 * scale, "%s: %s differs by %g" % (what, name, float((u - v).abs().max()))


def _poison_lds(pkg, dev, pattern):
    """Known garbage in every CU's LDS before the next launch (ggcn_debug_poison_lds, include/ggcn.h)."""
    from ed_gated_gcn_amd import _capi
    lib = pkg.load_library()
    with torch.cuda.device(dev):
        _capi.check(lib.ggcn_debug_poison_lds(pattern, _capi.stream_of(dev)), "ggcn_debug_poison_lds")


@pytest.mark.parametrize("B,T,K,F,degree,weighted", [(9, 129, 192, 128, 4.0, False), (3, 512, 128, 256, 6.0, False),
                                                       (2, 300, 64, 136, 5.0, True), (2, 512, 64, 72, 20.0, False),
                                                       (3, 200, 64, 64, 4.0, True)])
def test_long_graph_layer_is_a_function_of_what_it_was_given(pkg, dev, B, T, K, F, degree, weighted):
    """Round 3 saw ONE run of the (9, 129, 192, 128) case return +inf in a pool of layer_fused_long_kernel and could not
    make it happen again.  An exact +inf there is a max/min initialiser (fused_long.hip: vmax = -inf, vmin = +inf) or stale
    memory reaching the reduction, i.e. a read of something this launch did not write: LDS left by EARLIER kernels (the
    hardware never clears it), `colidx` / `vals` entries behind the last edge (torch.empty), the previous contents of the
    output buffers.  This test makes every one of those sources hostile and deterministic -- every CU's LDS filled with
    +inf / -inf / NaN / fp16-inf patterns right before the launch, the CSR tails filled with out-of-range ids and NaN,
    the outputs pre-filled with NaN -- and demands results BIT-IDENTICAL to a launch behind zeroed LDS and clean tails,
    with the first case being the one that failed (odd stage count: the last stage's LDS-DMA repeats land in W buffer 1,
    under the pools' exchange area; third row pass with one live 8-lane group)."""
    from ed_gated_gcn_amd import _capi, synth
    lib = pkg.load_library()
    rng = np.random.default_rng(T + K)
    lens = np.array([T] + [int(v) for v in rng.integers(T // 3, T + 1, size=B - 1)])
    adj = synth.dependency_batch(B, T, min(degree, T), seed=T, lengths=lens).astype(np.float32)
    if weighted:
        adj = adj * rng.uniform(0.25, 2.0, size=adj.shape).astype(np.float32)
    x = torch.from_numpy(rng.standard_normal((B, T, K)).astype(np.float32)).half().to(dev)
    g1 = torch.from_numpy(rng.uniform(-1.0, 1.0, (B, F)).astype(np.float32)).to(dev)
    g2 = torch.sigmoid(torch.from_numpy(rng.standard_normal((B, F)).astype(np.float32))).to(dev)
    w, b = synth.layer_params(K, F, seed=5)
    m = _layer(pkg, dev, w, b, "f16")
    csr = pkg.BatchedCSR.from_dense(torch.from_numpy(adj).to(dev))
    nnz = int(csr.rowptr[-1].item())
    p, st = _capi.ptr, _capi.stream_of(dev)
    pack = m._packed_weight(lib, st)
    bias = m.bias.detach()

    def launch(colidx, vals, want_out=True):
        out = torch.full((B * T, F), float("nan"), dtype=torch.float16, device=dev) if want_out else None
        pa = torch.full((B, F), float("nan"), device=dev)
        pb = torch.full((B, F), float("nan"), device=dev)
        _capi.check(lib.ggcn_layer_fused_h(p(x), K, p(pack), p(csr.rowptr), p(colidx), p(vals), p(bias), B, T, K, F, p(g2),
                                           p(g1), p(g2), p(out), F, p(pa), p(pb), st), "ggcn_layer_fused_h")
        return out, pa, pb

    _poison_lds(pkg, dev, 0)
    ref = launch(csr.colidx, csr.vals)
    assert all(bool(torch.isfinite(t.float()).all()) for t in ref)
    # hostile tails: an id that is no node of any graph, a weight that poisons any sum it enters
    col_bad = csr.colidx.clone()
    col_bad[nnz:] = 0x7FFFFFFF
    val_bad = None
    if csr.vals is not None:
        val_bad = csr.vals.clone()
        val_bad[nnz:] = float("nan")
    for pattern in POISONS:
        for want_out in (True, False):
            _poison_lds(pkg, dev, pattern)
            got = launch(col_bad, val_bad, want_out)
            for name, u, v in zip(("out", "pool_a", "pool_b"), got, ref):
                if u is not None:
                    assert torch.equal(u, v), "LDS pattern %#x, %s: differs from the clean launch in %d places" % (
                        pattern, name, int((u != v).sum()))
    # against the oracle too (config 4's gate on the fp16-rounded inputs)
    oref = ref_dense.graph_convolution(x.float().cpu(), torch.from_numpy(adj), torch.from_numpy(w), torch.from_numpy(b))
    scale = max(1.0, float(oref.abs().max()))
    np.testing.assert_allclose(ref[1].cpu().numpy(), (oref * g1.cpu()[:, None, :]).max(dim=1)[0].numpy(), rtol=0, atol=2e-3 * scale)


def test_lds_resident_paths_do_not_read_stale_lds(pkg, dev):
    """The same demand on every other kernel that keeps state in LDS: the one-launch block and layer (graphs of <= 32
    nodes: operand blocks, gates and store staging behind the stage buffers), the 64-/128-row and the eight-wavefront
    forms (33..256 nodes), the LDS-slab aggregation (T <= 48) and the long-graph aggregation -- each run behind four
    hostile LDS patterns and compared bit for bit with the run behind zeroed LDS."""
    from ed_gated_gcn_amd import synth
    rng = np.random.default_rng(77)

    def case(B, T, H, precision, fused, half=False, block=False):
        lens = rng.integers(max(1, T // 3), T + 1, size=B)
        adj = torch.from_numpy(synth.dependency_batch(B, T, min(4.0, T), seed=T + B, lengths=lens).astype(np.float32)).to(dev)
        x = torch.from_numpy(rng.standard_normal((B, T, H)).astype(np.float32)).to(dev)
        x = x.half() if half else x
        g1 = torch.from_numpy(rng.uniform(-1.0, 1.0, (B, H)).astype(np.float32)).to(dev)
        g2 = torch.sigmoid(torch.from_numpy(rng.standard_normal((B, H)).astype(np.float32))).to(dev)
        w1, b1 = synth.layer_params(H, H, seed=1)
        w2, b2 = synth.layer_params(H, H, seed=2)
        l1, l2 = _layer(pkg, dev, w1, b1, precision, fused=fused), _layer(pkg, dev, w2, b2, precision, fused=fused)
        csr = pkg.BatchedCSR.from_dense(adj)

        def run():
            with torch.no_grad():
                if block:
                    r = pkg.gated_gcn_block(x, csr, g1, g2, l1, l2, want_gcn1=True, one_launch=True)
                    return [r[k] for k in ("gcn1", "x1", "y1", "x", "out")] + [r["xy"].reshape(1)]
                return list(l1.forward_gated(x, csr, store_gate=g2, pool_gate_a=g1, pool_gate_b=g2, want_pool_a=True, want_pool_b=True))
        run()                      # weight images, operand blocks, caches: made outside the compared launches
        _poison_lds(pkg, dev, 0)
        ref = run()
        for pattern in POISONS:
            _poison_lds(pkg, dev, pattern)
            got = run()
            for k, (u, v) in enumerate(zip(got, ref)):
                assert torch.equal(u, v), "B=%d T=%d H=%d %s fused=%s block=%s, LDS pattern %#x: output %d differs" % (
                    B, T, H, precision, fused, block, pattern, k)

    case(37, 32, 256, "f16mx8", True, block=True)      # ggcn_block_fused
    case(37, 19, 256, "bf16x3", True, block=True)
    if pkg._capi.has_f16mx6():
        case(10, 32, 320, "f16mx6", True, block=True)  # layer_fused6_kernel (RAW stages reused for the operands)
    case(21, 31, 200, "f16mx8", True)                  # ggcn_layer_fused, ragged T < 32, F % 32 != 0
    case(9, 100, 256, "f16mx8", True)                  # 128-row slots
    case(9, 60, 256, "bf16x3", True)                   # 64-row slots
    case(5, 231, 512, "f16mx8", True)                  # eight wavefronts per graph (edge lists out of an fp32 tile)
    case(5, 150, 256, "bf16x3", True)
    case(17, 40, 256, "f16mx8", False)                 # linear + LDS-slab aggregation
    case(3, 300, 256, "f16", False, half=True)         # long-graph aggregation (aggregate_narrow)


def test_fp16_long_graph_entry_refuses_what_it_cannot_run(pkg, dev):
    from ed_gated_gcn_amd import _capi, synth
    lib = pkg.load_library()
    B, T = 2, 200
    adj = synth.dependency_batch(B, T, 4.0)
    rp, ci, _ = synth.csr_from_dense_host(adj)
    csr = pkg.BatchedCSR.from_arrays(rp, ci, B, T, dev)
    p, st = _capi.ptr, _capi.stream_of(dev)
    def call(K, F, T_=T, x_off=0):
        x = torch.zeros(B * T * K + 8, dtype=torch.float16, device=dev)[x_off:]
        w = torch.zeros(K, F, device=dev)
        pack = torch.empty(lib.ggcn_weight_pack_bytes(K, F, 3), dtype=torch.uint8, device=dev)
        assert lib.ggcn_weight_pack(p(w), F, K, F, 3, 0, p(pack), st) == 0
        out = torch.empty(B * T, F, dtype=torch.float16, device=dev)
        return lib.ggcn_layer_fused_h(p(x), K, p(pack), p(csr.rowptr), p(csr.colidx), None, None, B, T_, K, F, None, None,
                                      None, p(out), F, None, None, st)
    assert call(64, 64) == 0
    assert call(96, 64) == 3 and b"K % 64" in lib.ggcn_last_error()
    assert call(64, 60) == 3
    assert call(64, 64, x_off=4) == 3          # X not 16-byte aligned
    assert call(64, 64, T_=513) == 3
    torch.cuda.synchronize()


def test_empty_batch_like_the_reference(pkg, dev):
    """B = 0 is a valid input of gcn.py:30-45 (empty output, no kernel launch); the reference block's mean over
    an empty batch is nan."""
    from ed_gated_gcn_amd import synth
    H = 16
    w, b = synth.layer_params(H, H, seed=1)
    l1, l2 = _layer(pkg, dev, w, b, "f16mx8"), _layer(pkg, dev, w, b, "f16mx8")
    x = torch.zeros(0, 5, H, device=dev)
    adj = torch.zeros(0, 5, 5, device=dev)
    ref = ref_dense.graph_convolution(x.cpu(), adj.cpu(), torch.from_numpy(w), torch.from_numpy(b))
    with torch.no_grad():
        out = l1(x, adj)
        r = pkg.gated_gcn_block(x, adj, torch.zeros(0, H, device=dev), torch.zeros(0, H, device=dev), l1, l2)
    assert tuple(out.shape) == tuple(ref.shape) == (0, 5, H)
    assert tuple(r["x"].shape) == (0, 5, H) and tuple(r["out"].shape) == (0, H) and bool(torch.isnan(r["xy"]))


# ---------------------------------------------------------------- gates that dropout zeroed (bert_amir5.py:623-625)
@pytest.mark.parametrize("precision,fused", MODES, ids=MODE_IDS)
@pytest.mark.parametrize("T", [64, 32, 9])
def test_zero_gates_pool_to_signed_zero_not_minus_inf(pkg, dev, precision, fused, T):
    """A gate entry that is exactly 0 (train-mode dropout on the gate) makes y*gate = -0.0 on every row
    whose y is negative; max_t of a column of -0.0 must be 0, never the -inf preset of the chunked
    aggregation's atomic max (T > 48 with fp32 features takes that path)."""
    from ed_gated_gcn_amd import synth
    B, H = 5, 64
    rng = np.random.default_rng(11)
    adj = synth.dependency_batch(B, T, 3.0, seed=2)
    x = rng.standard_normal((B, T, H)).astype(np.float32)
    w, b = synth.layer_params(H, H, seed=3)
    b[:] = -50.0                                   # every y is negative on every row
    g1 = rng.uniform(0.1, 0.9, (B, H)).astype(np.float32)
    g2 = rng.uniform(0.1, 0.9, (B, H)).astype(np.float32)
    g1[:, ::3] = 0.0
    g2[::2, :] = 0.0
    g2[1, 5] = -0.0
    t = torch.from_numpy
    y = ref_dense.graph_convolution(t(x), t(adj.astype(np.float32)), t(w), t(b))      # gcn.py:30-45
    ref = {"pool_a": torch.max(y * t(g1)[:, None, :], 1)[0],                           # bert_amir5.py:627,635
           "pool_b": torch.max(y * t(g2)[:, None, :], 1)[0]}
    assert float(y.max()) < 0
    m = _layer(pkg, dev, w, b, precision, fused)
    with torch.no_grad():
        out, pa, pb = m.forward_gated(t(x).to(dev), t(adj).to(dev), pool_gate_a=t(g1).to(dev), pool_gate_b=t(g2).to(dev),
                                      want_pool_a=True, want_pool_b=True)
    torch.cuda.synchronize()
    for got, want in ((pa, ref["pool_a"]), (pb, ref["pool_b"])):
        got = got.cpu().numpy()
        assert np.isfinite(got).all()
        np.testing.assert_allclose(got, want.numpy(), rtol=0, atol=50 * TOL[precision])   # |y| ~ 50
    assert (pa.cpu().numpy()[:, ::3] == 0).all() and (pb.cpu().numpy()[::2] == 0).all()


# ---------------------------------------------------------------- the block as ONE launch (ggcn_block_fused)
@pytest.mark.parametrize("plane", [0, 1], ids=["bf16-pairs", "fp16-pairs"])
def test_folded_second_layer_operand_matches_numpy(pkg, dev, plane):
    """ggcn_graph_operands2: the block's second layer applies D.A twice (bert_amir5.py:626,639; gcn.py:35,41); the launch
    applies the precomputed (D.A)^2 instead.  The device blocks -- hi + lo fragments of (D.A)^2 * 2^10 in MFMA operand order,
    rowsum(D.A) in accumulator order -- decoded on the host against numpy's float64 (D.A) @ (D.A), ragged graphs included."""
    from ed_gated_gcn_amd import synth
    B, T = 11, 29
    lens = np.random.default_rng(3).integers(1, T + 1, size=B)
    adj = synth.dependency_batch(B, T, 4.0, seed=9, lengths=lens)
    csr = pkg.BatchedCSR.from_dense(torch.from_numpy(adj).to(dev))
    raw = csr.graph_ops2(plane).cpu().numpy().reshape(B, 4224)
    a = adj.astype(np.float64)
    da = a / (a.sum(-1, keepdims=True) + 1.0)
    want = da @ da
    def decode16(u16):
        if plane == 1:
            return u16.view(np.float16).astype(np.float64)
        return (u16.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    got = np.zeros((B, 32, 32))
    for s_ in range(2):
        hi = decode16(raw[:, s_ * 1024:(s_ + 1) * 1024].copy().view(np.uint16).reshape(B, 64, 8))
        lo = decode16(raw[:, 2048 + s_ * 1024:2048 + (s_ + 1) * 1024].copy().view(np.uint16).reshape(B, 64, 8))
        for lane in range(64):
            r, h = lane & 31, lane >> 5
            for e in range(8):
                got[:, r, 16 * s_ + 8 * (e >> 2) + 4 * h + (e & 3)] = (hi[:, lane, e] + lo[:, lane, e]) / 1024.0
    assert np.abs(got[:, :T, :T] - want).max() <= (2.0 ** -16 if plane == 0 else 2.0 ** -20)
    assert np.abs(got[:, T:, :]).max() == 0 and np.abs(got[:, :, T:]).max() == 0          # the 32-row slot beyond T: zeros
    rho = raw[:, 4096:4224].copy().view(np.float32).reshape(B, 2, 16)
    for hh in range(2):
        for rr in range(16):
            row = (rr & 3) + 8 * (rr >> 2) + 4 * hh
            if row < T:
                np.testing.assert_allclose(rho[:, hh, rr], da.sum(-1)[:, row], rtol=1e-6)


@pytest.mark.parametrize("precision", ["bf16x3", "f16mx8"])
@pytest.mark.parametrize("B,T,H,bias", [(64, 32, 768, True), (33, 31, 256, True), (6, 7, 96, False), (130, 32, 300, True)])
def test_one_launch_block_equals_two_launches(pkg, dev, B, T, H, bias, precision):
    """gc2(gc1(x)) through the folded weight W1.W2 (one launch, gcn1 never stored) against the two-launch
    path and the oracle; gcn1 on request is the same tensor layer 1 alone produces; weights that change are
    re-folded; a second call is bit-identical."""
    from ed_gated_gcn_amd import synth
    rng = np.random.default_rng(B * H)
    adj = torch.from_numpy(synth.dependency_batch(B, T, min(4.0, T), seed=5, lengths=rng.integers(2, T + 1, size=B))).to(dev)
    x = torch.from_numpy(rng.standard_normal((B, T, H)).astype(np.float32)).to(dev)
    g1 = torch.sigmoid(torch.from_numpy(rng.standard_normal((B, H)).astype(np.float32))).to(dev)
    g2 = torch.sigmoid(torch.from_numpy(rng.standard_normal((B, H)).astype(np.float32))).to(dev)
    (w1, b1), (w2, b2) = synth.layer_params(H, H, seed=1), synth.layer_params(H, H, seed=2)
    l1, l2 = _layer(pkg, dev, w1, b1 if bias else None, precision), _layer(pkg, dev, w2, b2 if bias else None, precision)
    with torch.no_grad():
        one = pkg.gated_gcn_block(x, adj, g1, g2, l1, l2)
        one_g = pkg.gated_gcn_block(x, adj, g1, g2, l1, l2, want_gcn1=True)
        two = pkg.gated_gcn_block(x, adj, g1, g2, l1, l2, one_launch=False)
    assert one["gcn1"] is None
    tol = TOL[precision]
    for k in ("x1", "y1", "x", "out"):
        assert torch.equal(one[k], one_g[k]), k
        assert float((one[k] - two[k]).abs().max()) <= tol, k
    assert torch.equal(one["x1"], two["x1"]) and torch.equal(one["y1"], two["y1"])   # layer 1 is the same arithmetic
    assert torch.equal(one_g["gcn1"], two["gcn1"])
    assert abs(float(one["xy"]) - float(two["xy"])) <= 2e-6 * max(1.0, abs(float(two["xy"])))
    assert torch.equal(one["out"], one["x"].max(dim=1)[0])
    t = torch.from_numpy
    zeros = np.zeros(H, np.float32)
    ref = ref_dense.gated_block(x.cpu(), adj.cpu().float(), g1.cpu(), g2.cpu(), t(w1), t(b1 if bias else zeros),
                                t(w2), t(b2 if bias else zeros))
    for k in ("x1", "y1", "x", "out"):
        np.testing.assert_allclose(one[k].cpu().numpy(), ref[k].numpy(), rtol=0, atol=tol, err_msg=k)
    # a weight update invalidates the folded operands
    with torch.no_grad():
        l2.weight.mul_(0.5)
        upd = pkg.gated_gcn_block(x, adj, g1, g2, l1, l2)
        upd2 = pkg.gated_gcn_block(x, adj, g1, g2, l1, l2, one_launch=False)
    assert float((upd["x"] - upd2["x"]).abs().max()) <= tol
    assert float((upd["x"] - one["x"]).abs().max()) > 10 * tol


# ---------------------------------------------------------------- the eval form: only what train.py:227 keeps
@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["f16mx8", "bf16x3"])
@pytest.mark.parametrize("B,T,H", [(16, 32, 256), (5, 31, 200), (9, 17, 96), (6, 60, 128), (3, 100, 256), (2, 231, 256), (3, 160, 128), (2, 256, 128), (5, 33, 96)])
def test_eval_form_out_is_the_full_blocks_bit_for_bit(pkg, dev, B, T, H, precision):
    """train.py:227 keeps the logits of an evaluation batch, and those need `out` alone (bert_amir5.py:640,643):
    want=("out",) launches only the W12 column tiles of the one-launch block (graphs of <= 32 nodes; longer graphs: the two
    layers without layer 1's pools, the regulariser and the [B,T,H] store of x) -- `out` is the full block's bit for bit and
    equals the oracle; what was not asked for is not handed out."""
    from ed_gated_gcn_amd import synth
    rng = np.random.default_rng(B * H + T)
    adj = torch.from_numpy(synth.dependency_batch(B, T, min(4.0, T), seed=5, lengths=rng.integers(2, T + 1, size=B))).to(dev)
    x = torch.from_numpy(rng.standard_normal((B, T, H)).astype(np.float32)).to(dev)
    g1 = torch.sigmoid(torch.from_numpy(rng.standard_normal((B, H)).astype(np.float32))).to(dev)
    g2 = torch.sigmoid(torch.from_numpy(rng.standard_normal((B, H)).astype(np.float32))).to(dev)
    (w1, b1), (w2, b2) = synth.layer_params(H, H, seed=1), synth.layer_params(H, H, seed=2)
    l1, l2 = _layer(pkg, dev, w1, b1, precision), _layer(pkg, dev, w2, b2, precision)
    with torch.no_grad():
        full = pkg.gated_gcn_block(x, adj, g1, g2, l1, l2)
        ev = pkg.gated_gcn_block(x, adj, g1, g2, l1, l2, want=("out",))
        ev_nog1 = pkg.gated_gcn_block(x, adj, None, g2, l1, l2, want=("out",)) if T <= 32 else ev   # the eval form never reads gate1
        xo = pkg.gated_gcn_block(x, adj, g1, g2, l1, l2, want=("x", "out"))
        l1only = pkg.gated_gcn_block(x, adj, g1, g2, l1, l2, want=("x1", "y1", "xy"))
    from ed_gated_gcn_amd.gated_block import takes_folded_eval_path
    folded = takes_folded_eval_path(x, l1._as_csr(adj, x), l1, l2)
    assert folded == (T > 32)   # (_layer() sends every graph of <= 256 nodes to the one-launch layer)
    if not folded:   # T <= 32: the same W12 tiles of the same kernel; unfolded longer graphs: the same two layer launches
        assert torch.equal(ev["out"], full["out"]) and torch.equal(ev_nog1["out"], full["out"])
        assert torch.equal(xo["out"], full["out"]) and torch.equal(xo["x"], full["x"])
    else:         # 33..256 nodes: Z = D.A.X, then one layer launch through W12 (another association of the same sums) -- never W1
        assert float((ev["out"] - full["out"]).abs().max()) <= TOL[precision]
        assert float((xo["x"] - full["x"]).abs().max()) <= TOL[precision] and torch.equal(xo["out"], ev["out"])
        with torch.no_grad():
            two = pkg.gated_gcn_block(x, adj, g1, g2, l1, l2, want=("out",), one_launch=False)  # the unfolded eval: gc1 then gc2
        assert torch.equal(two["out"], full["out"])
    assert all(ev[k] is None for k in ("x1", "y1", "xy", "x", "gcn1"))
    assert xo["x1"] is None and xo["xy"] is None
    assert torch.equal(l1only["x1"], full["x1"]) and torch.equal(l1only["y1"], full["y1"]) and l1only["x"] is None and l1only["out"] is None
    assert float(l1only["xy"]) == float(full["xy"])
    t = torch.from_numpy
    ref = ref_dense.gated_block(x.cpu(), adj.cpu().float(), g1.cpu(), g2.cpu(), t(w1), t(b1), t(w2), t(b2))
    np.testing.assert_allclose(ev["out"].cpu().numpy(), ref["out"].numpy(), rtol=0, atol=TOL[precision])
    np.testing.assert_allclose(xo["x"].cpu().numpy(), ref["x"].numpy(), rtol=0, atol=TOL[precision])
    with pytest.raises(ValueError):
        pkg.gated_gcn_block(x, adj, g1, g2, l1, l2, want=("logits",))
    xg = x.clone().requires_grad_(True)
    with pytest.raises(RuntimeError, match="autograd"):
        pkg.gated_gcn_block(xg, adj, g1, g2, l1, l2, want=("out",))


@pytest.mark.gpu
def test_eval_form_through_the_c_abi(pkg, dev):
    """ggcn_block_fused with x1 = y1 = gcn1 = overlap_partial = NULL (include/ggcn.h: the eval form): pool_out alone, or with
    x; half-given layer-1 outputs are refused."""
    from ed_gated_gcn_amd import _capi, synth
    from ed_gated_gcn_amd.gated_block import _block_operands
    lib = pkg.load_library()
    B, T, H = 12, 32, 128
    rng = np.random.default_rng(3)
    adj = synth.dependency_batch(B, T, 4.0, seed=2)
    rp, ci, _ = synth.csr_from_dense_host(adj)
    csr = pkg.BatchedCSR.from_arrays(rp, ci, B, T, dev)
    x = torch.from_numpy(rng.standard_normal((B * T, H)).astype(np.float32)).to(dev)
    g1, g2 = torch.rand(B, H, device=dev), torch.rand(B, H, device=dev)
    (w1, b1), (w2, b2) = synth.layer_params(H, H, seed=1), synth.layer_params(H, H, seed=2)
    l1, l2 = _layer(pkg, dev, w1, b1, "f16mx8"), _layer(pkg, dev, w2, b2, "f16mx8")
    st = _capi.stream_of(dev)
    pack1, pack12, mid = _block_operands(l1, l2, lib, st, precision="f16mx8")
    P = _capi.ptr

    def call(gate1, gcn1, xo, x1, y1, out, part):
        return lib.ggcn_block_fused(P(x), H, P(pack1), P(pack12), P(csr.graph_ops), P(csr.graph_ops2(1)), P(l1.bias.detach()), P(mid),
                                    P(l2.bias.detach()), B, T, H, H, P(gate1), P(g2), P(gcn1), H, P(xo), H, P(x1), P(y1), P(out), P(part),
                                    _capi.PREC["f16mx8"], st)
    e = lambda *s: torch.full(s, float("nan"), device=dev)   # noqa: E731
    full = dict(xo=e(B * T, H), x1=e(B, H), y1=e(B, H), out=e(B, H), part=e(B, 2))
    assert call(g1, None, full["xo"], full["x1"], full["y1"], full["out"], full["part"]) == 0
    out_only, out_x, xo2 = e(B, H), e(B, H), e(B * T, H)
    assert call(None, None, None, None, None, out_only, None) == 0
    assert call(None, None, xo2, None, None, out_x, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(out_only, full["out"]) and torch.equal(out_x, full["out"]) and torch.equal(xo2, full["xo"])
    assert call(g1, None, None, full["x1"], None, out_only, None) != 0        # x1 without y1
    assert b"go together" in lib.ggcn_last_error()
    assert call(None, None, None, None, None, None, None) != 0               # nothing requested


@pytest.mark.gpu
def test_classifier_eval_logits_only_and_the_window_of_the_automatic_precision(pkg, dev, golden_dir):
    """opt.ggcn_eval_logits_only: the inference forward returns the same logits bit for bit (and None for what train.py:227
    drops).  ADVICE r4: the automatic precision must not pick f16mx8 where gc2's main loop splits a gcn1 the weights only
    bound beyond the accuracy window |x| <= 448 -- graphs of more than 32 nodes run two launches; the one-launch block
    never splits gcn1 and keeps f16mx8."""
    import types
    from oracle.ref_amir55 import BertAmir55Oracle, EncoderStandIn
    g = np.load(os.path.join(golden_dir, "amir55_full.npz"))
    oracle = BertAmir55Oracle(EncoderStandIn(int(g["seed_encoder"])), int(g["n_class"]))
    oracle.seeded_init(torch.Generator().manual_seed(int(g["seed_params"])))
    inputs = {k[3:]: torch.from_numpy(g[k]).to(dev) for k in g.files if k.startswith("in_")}
    models = []
    for flag in (False, True):
        opt = types.SimpleNamespace(device=dev, dropout=0.25, polarities_dim=int(g["n_class"]), ggcn_eval_logits_only=flag)
        m = pkg.GatedGCNEventDetector(EncoderStandIn(int(g["seed_encoder"])), opt)
        m.load_state_dict(oracle.state_dict())
        models.append(m.to(dev).eval())
    with torch.no_grad():
        logits, xy, kl, scores = models[0](inputs)
        logits_e, xy_e, kl_e, scores_e = models[1](inputs)
    assert torch.equal(logits, logits_e) and xy_e is None and kl_e is None and scores_e is None
    np.testing.assert_allclose(logits_e.cpu().numpy(), g["logits"], rtol=0, atol=1e-3)
    # training is unaffected by the flag
    models[1].train(); models[1].dropout.p = 0.0
    lt, xyt, klt, sct = models[1](inputs)
    assert sct is not None and torch.is_tensor(xyt)
    models[1].eval()
    # ---- the window: scale gc1 so that 448 < m1 = max(colsum|W1| + |b1|) < 32752
    m = models[0]
    with torch.no_grad():
        c1 = float((m.gc1.weight.abs().sum(0) + m.gc1.bias.abs()).max())
        s = 2000.0 / c1
        m.gc1.weight.mul_(s); m.gc1.bias.mul_(s)
        m.gc2.weight.mul_(1.0 / s)       # keeps |gcn1.W2| and |x.W12| where they were
        oracle.gc1.weight.mul_(s); oracle.gc1.bias.mul_(s); oracle.gc2.weight.mul_(1.0 / s)
        logits_s, _, _, _ = m(inputs)                    # 31-node graphs: the one-launch block never splits gcn1
        assert m._auto_precision and m.gc1.precision == "f16mx8"
        ref_s = oracle.eval()({k: v.cpu() for k, v in inputs.items()})[0]
        assert float((logits_s.cpu() - ref_s).abs().max()) <= 1e-3 * max(1.0, float(ref_s.abs().max()))
        m.gc1.check_range()                              # nothing tripped
        # the same weights on 40-node graphs (two launches: gc2 splits gcn1): bf16x3, and no range report later
        x40 = torch.rand(4, 40, 2 * m.hidden_dim, device=dev) * 2 - 1
        from ed_gated_gcn_amd import synth
        adj40 = torch.from_numpy(synth.dependency_batch(4, 40, 4.0, seed=1)).to(dev)
        csr40 = m.gc1._as_csr(adj40, x40)
        assert m._proved_precision(csr40, x40) == "bf16x3"
        csr31 = m.gc1._as_csr(inputs["dependency_graph"][:, :31, :31].contiguous(), torch.zeros(inputs["dependency_graph"].shape[0], 31, 2 * m.hidden_dim, device=dev))
        assert m._proved_precision(csr31, torch.zeros(inputs["dependency_graph"].shape[0], 31, 2 * m.hidden_dim, device=dev)) == "f16mx8"
        assert m._proved_precision(csr31) == "bf16x3"    # without the input nothing proves the one-launch path


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,H,C", [(512, 32, 768, 34), (7, 20, 96, 5), (3, 100, 128, 64)])
def test_dense_head_rides_with_the_regularisers_final_sum(pkg, dev, B, T, H, C):
    """ggcn_dense_head: logits = out @ Wt (+ bias) -- the share of bert_amir5.py:643's dense that reads the block's output -- in
    the launch that finishes xy (bert_amir5.py:638): against float64 / torch, row by row independent of the batch (what the
    sharded run all-gathers), xy equal to ggcn_overlap_reduce's within the order of additions."""
    from ed_gated_gcn_amd import synth
    rng = np.random.default_rng(B + C)
    adj = torch.from_numpy(synth.dependency_batch(B, T, min(4.0, T), seed=5)).to(dev)
    x = torch.from_numpy(rng.standard_normal((B, T, H)).astype(np.float32)).to(dev)
    g1, g2 = torch.rand(B, H, device=dev), torch.rand(B, H, device=dev)
    (w1, b1), (w2, b2) = synth.layer_params(H, H, seed=1), synth.layer_params(H, H, seed=2)
    l1, l2 = _layer(pkg, dev, w1, b1, "f16mx8"), _layer(pkg, dev, w2, b2, "f16mx8")
    wt = (torch.randn(H, C, generator=torch.Generator().manual_seed(7)) / H ** 0.5).to(dev)
    bias = torch.randn(C, generator=torch.Generator().manual_seed(8)).to(dev)
    with torch.no_grad():
        plain = pkg.gated_gcn_block(x, adj, g1, g2, l1, l2)
        r = pkg.gated_gcn_block(x, adj, g1, g2, l1, l2, dense_head=(wt, bias))
        r_nb = pkg.gated_gcn_block(x, adj, g1, g2, l1, l2, dense_head=(wt, None), want=("out",))
    for k in ("x1", "y1", "x", "out"):
        assert torch.equal(r[k], plain[k]), k
    assert abs(float(r["xy"]) - float(plain["xy"])) <= 2e-6 * max(1.0, abs(float(plain["xy"])))
    want = plain["out"].double().cpu() @ wt.double().cpu() + bias.double().cpu()
    assert float((r["logits"].double().cpu() - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))
    assert float((r_nb["logits"].double().cpu() - (want - bias.double().cpu())).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))
    assert r_nb["xy"] is None
    # a row's logits are the same bits in any sub-batch (the sharded run compares gathered logits bitwise with the unsharded ones)
    lo, hi = B // 3, max(B // 3 + 1, 2 * B // 3)
    sub = pkg.dense_head(plain["out"][lo:hi], wt, bias)
    assert torch.equal(sub, r["logits"][lo:hi])
    # ggcn_dense_head_signal: the same launch counting itself done in memory (the sharded step's `flag` hand-off): same bits,
    # signal[1] = number of finished launches, signal[0] (its arrival counter) back at zero
    signal = torch.zeros(2, dtype=torch.int32, device=dev)
    with torch.no_grad():
        for n in (1, 2, 3):
            rs = pkg.gated_gcn_block(x, adj, g1, g2, l1, l2, dense_head=(wt, bias, signal))
            torch.cuda.synchronize()
            assert signal.tolist() == [0, n]
    assert torch.equal(rs["logits"], r["logits"]) and float(rs["xy"]) == float(r["xy"])
    lib = pkg.load_library()
    from ed_gated_gcn_amd import _capi
    P = _capi.ptr
    assert lib.ggcn_dense_head(P(plain["out"]), H, P(wt), C, None, B, H, C, P(r["logits"]), C, P(plain["out"]), H, None, _capi.stream_of(dev)) != 0   # partials without xy
    assert lib.ggcn_dense_head_signal(P(plain["out"]), H, P(wt), C, None, B, H, C, P(r["logits"]), C, None, 0, None, None, _capi.stream_of(dev)) != 0   # no signal words
    big = torch.zeros(H, 65, device=dev)
    with pytest.raises(RuntimeError, match="at most 64"):
        pkg.dense_head(plain["out"], big)
    xg = x.clone().requires_grad_(True)
    with pytest.raises(RuntimeError, match="inference"):
        pkg.gated_gcn_block(xg, adj, g1, g2, l1, l2, dense_head=(wt, bias))


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,F", [(5, 231, 256), (3, 129, 128), (2, 256, 64), (4, 200, 256)])
def test_precomputed_edge_lists_of_the_eight_wavefront_layer(pkg, dev, B, T, F):
    """ggcn_graph_edge_lists: the per-row edge lists of graphs of 129..256 nodes made once per adjacency tensor (the kernel's LDS
    image, brought in by LDS-DMA) instead of by every workgroup -- the same launch with and without them is bit-identical,
    dense rows (more than 16 neighbours: they walk their mask words either way) and a full 256-node slot included; the blocks
    hold what a host loop over the dense adjacency says."""
    from ed_gated_gcn_amd import _capi, synth
    lib = pkg.load_library()
    rng = np.random.default_rng(T + F)
    lens = rng.integers(T // 2, T + 1, size=B); lens[0] = T
    adj = synth.dependency_batch(B, T, 4.0, seed=4, lengths=lens)
    adj[0, 3, :40] = 1; adj[0, :40, 3] = 1            # a hub: 40 neighbours, beyond the 16 a list holds
    csr = pkg.BatchedCSR.from_dense(torch.from_numpy(adj.astype(np.float32)).to(dev))
    lists = csr.edge_lists
    assert lists is not None and lists.numel() == B * 11264
    blk = lists.view(B, 11264).cpu().numpy()
    deg = blk[:, 8192:9216].copy().view(np.int32); inv = blk[:, 9216:10240].copy().view(np.float32)
    want_deg = np.zeros((B, 256), np.int32); want_deg[:, :T] = adj.sum(2).astype(np.int32)
    np.testing.assert_array_equal(deg, want_deg)
    np.testing.assert_array_equal(inv, (1.0 / (want_deg + 1)).astype(np.float32))
    assert not blk[:, 10240:10368].any()
    ids = blk[:, :8192].copy().view(np.uint16).reshape(B, 256, 16)
    first = [int(np.flatnonzero(adj[1, 5])[e]) for e in range(min(16, int(adj[1, 5].sum())))]
    assert [int(v) >> 7 for v in ids[1, 5, :len(first)]] == first      # (offset = row * 128 + a chunk flip below 128)
    x = torch.from_numpy(rng.standard_normal((B * T, 64)).astype(np.float32)).to(dev)
    w, b = synth.layer_params(64, F, seed=1)
    m = _layer(pkg, dev, w, b, "f16mx8")
    st, P = _capi.stream_of(dev), _capi.ptr
    pack = m._packed_weight(lib, st)
    g = torch.rand(B, F, device=dev)
    res = []
    for ops in (None, lists):
        out = torch.full((B * T, F), float("nan"), device=dev); pa = torch.empty(B, F, device=dev); pb = torch.empty(B, F, device=dev)
        _capi.check(lib.ggcn_layer_fused(P(x), 64, P(pack), P(csr.rowmask), P(ops), P(m.bias.detach()), B, T, 64, F, P(g), P(g), None,
                                         P(out), F, P(pa), P(pb), None, None, None, _capi.PREC["f16mx8"], st), "ggcn_layer_fused")
        res.append((out, pa))
    torch.cuda.synchronize()
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and bool(torch.isfinite(res[1][0]).all())
    assert lib.ggcn_graph_edge_lists(P(csr.rowmask), B, 100, P(lists), st) != 0      # 33..128 nodes: another kernel, no lists


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,H", [(16, 32, 256), (37, 23, 256), (5, 32, 768), (3, 7, 64)])
def test_eight_wavefront_shared_x_experiment_is_bit_identical_to_the_block(pkg, dev, B, T, H):
    """ggcn_lab_block_fused8 (one workgroup of eight wavefronts shares a row block's X planes between its W1
    and W12 column tiles; since late round 5 also what ggcn_block_fused runs for large batches): the same tiles, arithmetic and order as the four-wavefront kernel,
    so x, x1, y1, out and the regulariser's partials are the same bits -- whole and ragged batches, both XCD mappings, with and
    without the [N,F] output."""
    from ed_gated_gcn_amd import _capi, synth
    from ed_gated_gcn_amd.gated_block import _block_operands
    lib = pkg.load_library()
    rng = np.random.default_rng(B + T + H)
    adj = synth.dependency_batch(B, T, min(4.0, T), seed=6, lengths=rng.integers(1, T + 1, size=B))
    rp, ci, _ = synth.csr_from_dense_host(adj)
    csr = pkg.BatchedCSR.from_arrays(rp, ci, B, T, dev)
    x = torch.from_numpy(rng.standard_normal((B * T, H)).astype(np.float32)).to(dev)
    g1, g2 = torch.rand(B, H, device=dev), torch.rand(B, H, device=dev)
    (w1, b1), (w2, b2) = synth.layer_params(H, H, seed=1), synth.layer_params(H, H, seed=2)
    l1, l2 = _layer(pkg, dev, w1, b1, "f16mx8"), _layer(pkg, dev, w2, b2, "f16mx8")
    st, P = _capi.stream_of(dev), _capi.ptr
    pack1, pack12, mid = _block_operands(l1, l2, lib, st, precision="f16mx8")
    bb1, bb2 = l1.bias.detach(), l2.bias.detach()
    npart = (H + 63) // 64

    def bufs():
        e = lambda *s: torch.full(s, float("nan"), device=dev)   # noqa: E731
        return dict(xo=e(B * T, H), x1=e(B, H), y1=e(B, H), out=e(B, H), part=e(B, npart))
    ref = bufs()
    _capi.check(lib.ggcn_block_fused(P(x), H, P(pack1), P(pack12), P(csr.graph_ops), P(csr.graph_ops2(1)), P(bb1), P(mid), P(bb2), B, T, H, H,
                                     P(g1), P(g2), None, H, P(ref["xo"]), H, P(ref["x1"]), P(ref["y1"]), P(ref["out"]), P(ref["part"]),
                                     _capi.PREC["f16mx8"], st), "ggcn_block_fused")
    for rowmajor in (False, True):
        if rowmajor:
            os.environ["GGCN_LAB_BLOCK8_ROWMAJOR"] = "1"
        try:
            for with_x in (True, False):
                r = bufs()
                _capi.check(lib.ggcn_lab_block_fused8(P(x), H, P(pack1), P(pack12), P(csr.graph_ops), P(csr.graph_ops2(1)), P(bb1), P(mid), P(bb2),
                                                      B, T, H, H, P(g1), P(g2), P(r["xo"]) if with_x else None, H, P(r["x1"]), P(r["y1"]),
                                                      P(r["out"]), P(r["part"]), None, st), "ggcn_lab_block_fused8")
                torch.cuda.synchronize()
                for k in ("x1", "y1", "out", "part") + (("xo",) if with_x else ()):
                    a, b = ref[k], r[k]
                    assert bool(((a == b) | (torch.isnan(a) & torch.isnan(b))).all()), (k, rowmajor, with_x)
        finally:
            os.environ.pop("GGCN_LAB_BLOCK8_ROWMAJOR", None)


@pytest.mark.parametrize("B,T,H", [(2048, 32, 768), (2100, 29, 768), (1024, 32, 768)])
def test_large_batches_take_the_eight_wavefront_block_and_nothing_changes(pkg, dev, B, T, H):
    """ggcn_block_fused hands batches that make >= 6 rounds of one workgroup per CU (or >= 3 whole rounds: 1024 x 768) -- all
    outputs, f16mx8, whole 256-column groups -- to the eight-wavefront kernel of fused_block8.hip (2-4 % less time in steady state
    at the power cap); GGCN_BLOCK_FORM=4 keeps the four-wavefront kernel (ggcn_block_fused_form says which).  Same tiles, same arithmetic, same order: every output the same bits, whole and ragged batches; against the oracle on
    a slice; smaller batches, the eval form and a request for gcn1 stay where they were."""
    from ed_gated_gcn_amd import synth
    rng = np.random.default_rng(B + T)
    adj_np = synth.dependency_batch(B, T, min(4.0, T), seed=3, lengths=rng.integers(1, T + 1, size=B))
    adj = torch.from_numpy(adj_np).to(dev)
    x = torch.from_numpy(rng.standard_normal((B, T, H)).astype(np.float32)).to(dev)
    g1, g2 = torch.rand(B, H, device=dev), torch.rand(B, H, device=dev)
    (w1, b1), (w2, b2) = synth.layer_params(H, H, seed=1), synth.layer_params(H, H, seed=2)
    l1, l2 = _layer(pkg, dev, w1, b1, "f16mx8"), _layer(pkg, dev, w2, b2, "f16mx8")
    csr = pkg.BatchedCSR.from_dense(adj)
    lib = pkg.load_library()
    assert lib.ggcn_block_fused_form(B, T, H, H) == 8 and lib.ggcn_block_fused_form(512, T, H, H) == 4
    assert lib.ggcn_block_fused_form(1536, 32, 768, 768) == 4 and lib.ggcn_block_fused_form(B, T, H, 200) == 4     # 4.5 rounds; no whole slices
    with torch.no_grad():
        r8 = pkg.gated_gcn_block(x, csr, g1, g2, l1, l2)
        os.environ["GGCN_BLOCK_FORM"] = "4"
        try:
            assert lib.ggcn_block_fused_form(B, T, H, H) == 4
            r4 = pkg.gated_gcn_block(x, csr, g1, g2, l1, l2)
        finally:
            os.environ.pop("GGCN_BLOCK_FORM", None)
        r8b = pkg.gated_gcn_block(x, csr, g1, g2, l1, l2)
        with_gcn1 = pkg.gated_gcn_block(x, csr, g1, g2, l1, l2, want_gcn1=True)       # (the four-wavefront kernel: gcn1 is its output)
    for k in ("x1", "y1", "x", "out"):
        assert torch.equal(r8[k], r4[k]) and torch.equal(r8[k], r8b[k]) and torch.equal(r8[k], with_gcn1[k]), k
    assert float(r8["xy"]) == float(r4["xy"]) == float(r8b["xy"])
    sl = slice(0, 40)
    ref = _oracle_block(x[sl].cpu().numpy(), adj_np[sl], g1[sl].cpu().numpy(), g2[sl].cpu().numpy(), w1, b1, w2, b2)
    for k in ("x1", "y1", "x", "out"):
        np.testing.assert_allclose(r8[k][sl].cpu().numpy(), ref[k].numpy(), rtol=0, atol=TOL["f16mx8"] * max(1.0, float(ref[k].abs().max())))


# ---------------------------------------------------------------- N > 1 product path on one device
def _shard_worker(rank, world, port, ret):
    import traceback
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    try:
        import torch.distributed as dist
        import ed_gated_gcn_amd as pkg
        from ed_gated_gcn_amd import shard, synth
        dist.init_process_group("gloo", rank=rank, world_size=world)
        try:
            dev = torch.device("cuda:0")
            B, T, H = 37, 32, 256
            rng = np.random.default_rng(5)
            adj = synth.dependency_batch(B, T, 4.0, seed=3, lengths=rng.integers(3, T + 1, size=B))
            x = torch.from_numpy(rng.standard_normal((B, T, H)).astype(np.float32))
            g1 = torch.sigmoid(torch.from_numpy(rng.standard_normal((B, H)).astype(np.float32)))
            g2 = torch.sigmoid(torch.from_numpy(rng.standard_normal((B, H)).astype(np.float32)))
            parts = shard.partition_graphs(adj.reshape(B, -1).sum(1), world)
            lo, hi = parts[rank]
            counts = [h - l for l, h in parts]
            ls = []
            for s in (1, 2):
                w, b = synth.layer_params(H, H, seed=s)
                m = pkg.GraphConvolution(H, H, None).to(dev)
                with torch.no_grad():
                    m.weight.copy_(torch.from_numpy(w)); m.bias.copy_(torch.from_numpy(b))
                ls.append(m.eval())
            rp, ci, _ = synth.csr_from_dense_host(adj)
            lrp, lci = shard.shard_csr_host(rp, ci, T, lo, hi)
            csr = pkg.BatchedCSR.from_arrays(lrp, lci, hi - lo, T, dev)
            gather = shard.PooledGather(counts, H, dev)
            pending, got = [], []
            with torch.no_grad():
                for k in range(3):      # bench.py's loop: at most two gathers in flight over the two slots
                    r = pkg.gated_gcn_block((x[lo:hi] * (k + 1)).to(dev), csr, g1[lo:hi].to(dev), g2[lo:hi].to(dev), *ls)
                    while len(pending) > 1:
                        got.append(gather.finish(pending.pop(0)).cpu().clone())
                    pending.append(gather.start(r["out"]))
                while pending:
                    got.append(gather.finish(pending.pop(0)).cpu().clone())
                if rank == 0:           # the unsharded product path on the same device
                    full = pkg.BatchedCSR.from_arrays(rp, ci, B, T, dev)
                    want = [pkg.gated_gcn_block((x * (k + 1)).to(dev), full, g1.to(dev), g2.to(dev), *ls)["out"].cpu()
                            for k in range(3)]
                    for k in range(3):
                        assert torch.equal(got[k], want[k]), "step %d: sharded != unsharded" % k
            ret.put((rank, "ok", None))
        finally:
            dist.destroy_process_group()
    except BaseException:   # noqa: B902
        ret.put((rank, "error", traceback.format_exc()))
        raise


def test_two_rank_sharded_block_same_device(pkg, dev):
    """BASELINE configs[2] in small: two processes (gloo rendezvous, both on cuda:0) run the HIP block on their
    nnz-balanced shards and all-gather the pooled outputs; graphs are independent, so the result equals the
    unsharded block bit for bit.  Uneven shards exercise the padded send buffers of PooledGather."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    errors = []
    try:
        for _ in range(2):
            rank, status, payload = ret.get(timeout=240)
            if status != "ok":
                errors.append("rank %d:\n%s" % (rank, payload))
                break
    finally:
        for p in procs:
            p.join(timeout=10 if errors else 120)
            if p.is_alive():
                p.kill()
                p.join()
    assert not errors, "\n".join(errors)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]


def test_bench_self_launches_its_ranks_and_the_gathered_logits_match_the_unsharded_batch(pkg, dev):
    """`python bench.py --gpus 2` typed as is (what a driver without torch.distributed.run would run): the parent
    touches no GPU, starts two rank processes and relays rank 0's JSON line.  The ranks run bench.py's OWN N > 1 step
    loop -- hipGraph replay of the sharded block, the logits head, one asynchronous all-gather per step -- here over
    gloo with both ranks on cuda:0 (one-GPU box; on the 8-GPU node the backend is RCCL), and `--check-gather` compares
    the last gathered logits with the unsharded 96-graph batch: graphs are independent, so bit for bit."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--same-device",
           "--graphs", "96", "--steps", "6", "--warmup", "2", "--precondition", "4", "--no-alt", "--no-cpu-baseline",
           "--check-gather"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["scaling"] == "strong" and r["steps"] == 6
    assert r["rccl"]["world_size_seen"] == 2 and r["rccl"]["backend"] == "gloo"
    assert r["rccl"]["gather_bytes_per_rank"] == max(r["config"]["graphs_per_gpu"]) * 34 * 4
    assert sum(r["config"]["graphs_per_gpu"]) == 96
    assert r["config"]["hipgraph_replay"] is True, r["config"]["capture_note"]
    assert r["gather_check"]["bitwise_equal"] is True and r["gather_check"]["rows"] == 96, r["gather_check"]
    assert r["value"] > 0 and "roofline" in r


@pytest.mark.parametrize("gather_mode", ["async", "flag", "graph"])
def test_bench_distributed_loop_over_rccl_with_a_process_group_of_one(pkg, dev, gather_mode):
    """The only collective of the path (BASELINE configs[2]: all-gather of per-shard logits over xGMI) runs on RCCL, and the
    build box has one GPU: `bench.py --gpus 1 --force-dist --backend nccl` runs bench.py's N > 1 step loop -- RCCL process
    group bound to the device (`init_process_group("nccl", device_id=...)`, HSA_ENABLE_IPC_MODE_LEGACY=0), hipGraph replay
    of the block, the logits head, PooledGather's asynchronous `all_gather_into_tensor` on RCCL's stream beside the
    replay, all-reduce of the timing, barrier, destroy -- with a world of one, so that the driver's 8-GPU run is not the
    first time RCCL sees this code.  The gathered logits equal the unsharded batch bit for bit."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist", "--backend", "nccl",
           "--graphs", "512", "--steps", "12", "--warmup", "3", "--precondition", "8", "--no-alt", "--no-cpu-baseline",
           "--no-config4", "--check-gather", "--gather-mode", gather_mode]   # graph: the collective captured inside the step's hipGraph (opt-in)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 1 and r["steps"] == 12
    assert r["rccl"]["backend"] == "nccl" and r["rccl"]["world_size_seen"] == 1 and r["rccl"]["gather_mode"] == gather_mode
    assert r["rccl"]["gather_bytes_per_rank"] == 512 * 34 * 4
    assert r["config"]["hipgraph_replay"] is True, r["config"]["capture_note"]
    assert r["gather_check"]["bitwise_equal"] is True and r["gather_check"]["rows"] == 512, r["gather_check"]
    assert r["value"] > 0 and "roofline" in r


def test_range_report_reaches_a_process_that_never_asks(pkg, dev):
    """the lazy report needs a LATER forward to surface; a process with a single forward (or none after the
    violation) used to exit silently.  Now the snapshot is taken behind the first guarded forward and whatever is still
    unreported at interpreter exit is printed to stderr (tools/exit_probe.py: one forward with |x| ~ 1000, no check)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "exit_probe.py")], capture_output=True, text=True, timeout=300, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "forward done" in out.stdout
    assert "448" in out.stderr and "bf16x3" in out.stderr, out.stderr[-2000:]


# ---------------------------------------------------------------- the drop-in forward(text, adj): syncs, cache, defaults
def test_dropin_forward_is_sync_free_and_shares_the_conversion(pkg, dev):
    """models/gcn.py:30-45 has no device synchronisation; neither has forward(text, dense adj) here when the
    0/1 adjacency is promised (opt.ggcn_binary_adj) -- and without the promise the ONE read-back of the
    weighted flag is paid once per adjacency tensor: gc1 and gc2 of a forward share the conversion."""
    import types
    from ed_gated_gcn_amd import csr as csr_mod, synth
    B, T, H = 16, 31, 64
    big = torch.zeros(B, 40, 40, device=dev)
    big[:, :T, :T] = torch.from_numpy(synth.dependency_batch(B, T, 3.0, seed=2)).to(dev).float()
    adj = big[:, :T, :T]                                  # the reference's non-contiguous slice (bert_amir5.py:589)
    x = torch.randn(B, T, H, device=dev)
    w, b = synth.layer_params(H, H, seed=1)
    assert pkg.GraphConvolution(H, H, None).precision == "f16mx8"        # the drop-in default = the benched arithmetic (range-flagged)
    promised = types.SimpleNamespace(ggcn_binary_adj=True)
    gc1, gc2 = pkg.GraphConvolution(H, H, promised).to(dev), pkg.GraphConvolution(H, H, promised).to(dev)
    plain1, plain2 = pkg.GraphConvolution(H, H, None).to(dev), pkg.GraphConvolution(H, H, None).to(dev)
    with torch.no_grad():
        for m in (gc1, gc2, plain1, plain2):
            m.weight.copy_(torch.from_numpy(w)); m.bias.copy_(torch.from_numpy(b))
        ref = gc2(gc1(x, adj), adj)                       # warm-up: packs the weights
        want = plain2(plain1(x, adj), adj)                # un-promised: converts (and reads the flag back) once
        torch.cuda.synchronize()
        n_cached = len(csr_mod._RECENT)
        torch.cuda.set_sync_debug_mode("error")
        try:
            got = gc2(gc1(x, adj), adj)                   # promised 0/1 adjacency: zero host syncs
            again = plain2(plain1(x, adj), adj)           # same adjacency tensor again: cache hit, zero host syncs
        finally:
            torch.cuda.set_sync_debug_mode("default")
    assert len(csr_mod._RECENT) == n_cached
    assert torch.equal(got, ref) and torch.equal(again, want) and torch.equal(got, want)
    # an in-place edit of the adjacency is a new adjacency: the cache must not serve the old conversion
    with torch.no_grad():
        big[:, 0, 1] = 1.0
        big[:, 1, 0] = 1.0
        edited = plain1(x, adj)
        fresh = plain1(x, adj.clone())
    assert torch.equal(edited, fresh)



def test_default_precision_reports_values_beyond_the_fp16_range(pkg, dev):
    """The default arithmetic (f16mx8) needs |v| < 65504 where the reference's fp32 matmul (gcn.py:34) has no limit: the
    kernels set a sticky per-device flag (one v_max3 per two values in the split, ggcn_range_flag), `check_range()` reads
    it now, and without being asked a later forward raises once a polled snapshot has reached the host -- no device
    synchronisation on the forward path.  In-range data never trips it; bf16x3 takes the same data without complaint."""
    from ed_gated_gcn_amd import range_guard, synth
    B, T, H = 8, 20, 64
    adj = torch.from_numpy(synth.dependency_batch(B, T, 3.0, seed=4)).to(dev).float()
    w, b = synth.layer_params(H, H, seed=2)
    m = pkg.GraphConvolution(H, H, None).to(dev)
    with torch.no_grad():
        m.weight.copy_(torch.from_numpy(w)); m.bias.copy_(torch.from_numpy(b))
    m.check_range()                                       # clears whatever earlier tests left
    x = torch.randn(B, T, H, device=dev)
    with torch.no_grad():
        m(x, adj)
    m.check_range()                                       # ordinary data: nothing to report
    big = x.clone()
    big[3, 5, 7] = 7.0e4
    with torch.no_grad():
        m(big, adj)
    with pytest.raises(RuntimeError, match="f16mx8"):
        m.check_range()
    m.check_range()                                       # reported once, cleared
    # the lazy form: a violation surfaces on a later forward by itself
    torch.cuda.set_sync_debug_mode("error")
    try:
        with torch.no_grad():
            with pytest.raises(RuntimeError, match="bf16x3"):
                for it in range(4 * range_guard.POLL_EVERY + 200):
                    m(big if it == 0 else x, adj)
                    if it % 8 == 7:
                        range_guard._STATE[dev.index]["event"].synchronize() if range_guard._STATE[dev.index]["pending"] else None
    finally:
        torch.cuda.set_sync_debug_mode("default")
    m.check_range()
    # an infinity counts; the one-launch block reports too
    inf = x.clone()
    inf[0, 0, 0] = float("inf")
    g = torch.rand(B, H, device=dev)
    m2 = pkg.GraphConvolution(H, H, None).to(dev)
    with torch.no_grad():
        m2.weight.copy_(torch.from_numpy(w)); m2.bias.copy_(torch.from_numpy(b))
        pkg.gated_gcn_block(inf, adj, g, g, m, m2)
    with pytest.raises(RuntimeError):
        m.check_range()
    # f16mx6 (same fp16 main product; an experiment, built on request) reports too
    if pkg._capi.has_f16mx6():
        m.precision = m2.precision = "f16mx6"
        xb = torch.randn(8, 32, H, device=dev)
        xb[2, 3, 4] = -9.0e4
        adj32 = torch.from_numpy(synth.dependency_batch(8, 32, 3.0, seed=5)).to(dev).float()
        with torch.no_grad():
            assert m.kernel_precision(xb.reshape(-1, H), pkg.BatchedCSR.from_dense(adj32)) == "f16mx6"
            m(xb, adj32)
        with pytest.raises(RuntimeError):
            m.check_range()
    else:
        m.precision = "f16mx6"
        with pytest.raises(RuntimeError, match="F16MX6=1"):
            m(x, adj)
    # bf16x3: the fp32 range, no flag
    m.precision = m2.precision = "bf16x3"
    with torch.no_grad():
        out = m(big, adj)
    m.check_range()
    ref = ref_dense.graph_convolution(big.cpu(), adj.cpu(), torch.from_numpy(w), torch.from_numpy(b))
    assert float((out.cpu() - ref).abs().max()) <= 1e-4 * float(ref.abs().max())

def test_forward_under_inference_mode(pkg, dev):
    """torch.inference_mode() tensors track no version counter: the adjacency cache and the weight-image keys must not
    read `_version` from them (a batch moved to the device inside the context is such a tensor)."""
    from ed_gated_gcn_amd import synth
    B, T, H = 8, 31, 64
    w, b = synth.layer_params(H, H, seed=1)
    m = _layer(pkg, dev, w, b, "bf16x3")
    adj_np = synth.dependency_batch(B, T, 3.0, seed=2).astype(np.float32)
    x_cpu = torch.randn(B, T, H)
    with torch.no_grad():
        want = m(x_cpu.to(dev), torch.from_numpy(adj_np).to(dev))
    with torch.inference_mode():
        x, adj = x_cpu.to(dev), torch.from_numpy(adj_np).to(dev)     # inference tensors
        assert adj.is_inference()
        got = m(x, adj)
        got2 = m(got, adj)                                           # gc2 of a forward: the same adjacency again
        g = torch.sigmoid(torch.randn(B, H, device=dev))
        r = pkg.gated_gcn_block(x, adj, g, g, m, m)
    assert torch.equal(got, want) and got2.shape == got.shape and torch.isfinite(r["out"]).all()


def test_f16mx8_range_validation_is_loud(pkg, dev):
    """f16mx8 saturates beyond the fp16 range; validate_range() is the explicit up-front check of a batch and of the weights (the
    kernels' own sticky flag: test_default_precision_reports_values_beyond_the_fp16_range)."""
    import types
    H = 64
    m = pkg.GraphConvolution(H, H, types.SimpleNamespace(ggcn_precision="f16mx8")).to(dev)
    assert m.precision == "f16mx8"
    with torch.no_grad():
        m.weight.normal_(0, 0.05); m.bias.zero_()
    x = torch.randn(3, 7, H, device=dev)
    rep = m.validate_range(x)
    assert abs(rep["text_absmax"] - float(x.abs().max())) < 1e-6 and rep["weight_absmax"] < 1.0
    x[1, 2, 3] = 7.0e4
    with pytest.raises(RuntimeError, match="outside the range"):
        m.validate_range(x)
    x[1, 2, 3] = float("nan")
    with pytest.raises(RuntimeError, match="non-finite"):
        m.validate_range(x)
    assert m.validate_range(x.half().nan_to_num(0.0))["text_absmax"] > 0      # fp16 features too


# ---------------------------------------------------------------- backward helpers that used to leave the library
@pytest.mark.parametrize("B,T,deg,weighted", [(37, 31, 3.0, False), (5, 100, 4.0, True), (3, 513, 6.0, False), (4, 1, 1.0, False)])
def test_csr_transpose_on_device_is_exact(pkg, dev, B, T, deg, weighted):
    """ggcn_csr_transpose (the backward applies A^T to a CSR that was collated on the host) against numpy."""
    from ed_gated_gcn_amd import synth
    rng = np.random.default_rng(B + T)
    adj = synth.dependency_batch(B, T, min(deg, T), seed=8, lengths=rng.integers(1, T + 1, size=B)).astype(np.float32)
    adj[:, 0, T - 1] = 1.0                                  # make it asymmetric
    if weighted:
        adj *= rng.uniform(0.5, 2.0, size=adj.shape).astype(np.float32)
    rp, ci, va = synth.csr_from_dense_host(adj)
    csr = pkg.BatchedCSR.from_arrays(rp, ci, B, T, dev, vals=va if weighted else None)
    t = csr.transposed()
    torch.cuda.synchronize()
    erp, eci, eva = synth.csr_from_dense_host(np.ascontiguousarray(adj.transpose(0, 2, 1)))
    assert np.array_equal(t.rowptr.cpu().numpy(), erp)
    assert np.array_equal(t.colidx.cpu().numpy()[:len(eci)], eci)
    if weighted:
        assert np.array_equal(t.vals.cpu().numpy()[:len(eva)], eva)
    else:
        assert t.vals is None
    assert t.transposed() is csr


@pytest.mark.parametrize("M,F", [(4096, 768), (37, 5), (1, 300), (100000, 64)])
def test_colsum_is_exact_enough_and_deterministic(pkg, dev, M, F):
    from ed_gated_gcn_amd import _capi
    lib = pkg.load_library()
    x = torch.randn(M, F, device=dev)
    ws = torch.empty(lib.ggcn_colsum_workspace_bytes(F), dtype=torch.uint8, device=dev)
    outs = []
    for _ in range(2):
        o = torch.full((F,), float("nan"), device=dev)
        _capi.check(lib.ggcn_colsum(_capi.ptr(x), F, M, F, _capi.ptr(o), _capi.ptr(ws), _capi.stream_of(dev)), "ggcn_colsum")
        outs.append(o.cpu())
    assert torch.equal(outs[0], outs[1])
    ref = x.double().sum(0).cpu()
    assert float((outs[0].double() - ref).abs().max()) <= 1e-6 * np.sqrt(M) * 4


def test_backward_through_a_collated_batch_uses_no_host_transpose(pkg, dev):
    """Training through a CSR that GraphBatcher collated (no dense tensor behind it): the transposed CSR, dW of an
    unaligned width and db all come from the library; gradients match the oracle's autograd."""
    from ed_gated_gcn_amd import synth
    from ed_gated_gcn_amd.batcher import GraphBatcher
    B, T, K, F = 6, 17, 30, 22
    rng = np.random.default_rng(3)
    adj = synth.dependency_batch(B, T, 3.0, seed=6, lengths=rng.integers(3, T + 1, size=B))
    gb = GraphBatcher()
    for i in range(B):
        gb.add(i, adj[i])
    csr = gb.collate(list(range(B)), T, dev)
    w, b = synth.layer_params(K, F, seed=4)
    x = rng.standard_normal((B, T, K)).astype(np.float32)
    for precision in ("fp32", "bf16x3"):
        m = _layer(pkg, dev, w, b, precision).train()
        xg = torch.from_numpy(x).to(dev).requires_grad_()
        out = m(xg, csr)
        (out * out).sum().backward()
        xr = torch.from_numpy(x).requires_grad_()
        wr, br = torch.from_numpy(w).requires_grad_(), torch.from_numpy(b).requires_grad_()
        yr = ref_dense.graph_convolution(xr, torch.from_numpy(adj.astype(np.float32)), wr, br)
        (yr * yr).sum().backward()
        _grad_close(xg.grad, xr.grad, "x")
        _grad_close(m.weight.grad, wr.grad, "w")
        _grad_close(m.bias.grad, br.grad, "b")


# ---------------------------------------------------------------- gate MLPs, scores/kl head, collated batches
def _golden_gate_seq(g, name, dev):
    seq = torch.nn.Sequential(torch.nn.Sigmoid(), torch.nn.Linear(256, 256), torch.nn.Sigmoid(), torch.nn.Linear(256, 256),
                              torch.nn.Sigmoid()).to(dev)
    with torch.no_grad():
        for i in (1, 3):
            seq[i].weight.copy_(torch.from_numpy(g["p_%s.%d.weight" % (name, i)]))
            seq[i].bias.copy_(torch.from_numpy(g["p_%s.%d.bias" % (name, i)]))
    return seq


def test_gate_mlps_one_launch_golden_and_oracle(pkg, dev, golden_dir):
    """bert_amir5.py:562-571,621-622 as one launch: the gates the REFERENCE computed (fixture G3), and seeded
    shapes against the same nn.Sequential on the CPU (ragged B, H that is not a multiple of 256)."""
    g = np.load(os.path.join(golden_dir, "amir55_block.npz"))
    s1, s2 = _golden_gate_seq(g, "gate1", dev), _golden_gate_seq(g, "gate2", dev)
    with torch.no_grad():
        g1, g2 = pkg.gate_mlps(torch.from_numpy(g["aspect"]).to(dev), s1, s2)
    np.testing.assert_allclose(g1.cpu().numpy(), g["gate1"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(g2.cpu().numpy(), g["gate2"], rtol=0, atol=2e-6)
    for B, H in ((1, 8), (19, 300), (256, 256), (5, 1024)):
        gen = torch.Generator().manual_seed(B + H)
        mk = lambda: torch.nn.Sequential(torch.nn.Sigmoid(), torch.nn.Linear(H, H), torch.nn.Sigmoid(), torch.nn.Linear(H, H),
                                         torch.nn.Sigmoid())
        c1, c2 = mk(), mk()
        a = torch.randn(B, H, generator=gen) * 2
        with torch.no_grad():
            want1, want2 = c1(a), c2(a)
            import copy
            got1, got2 = pkg.gate_mlps(a.to(dev), copy.deepcopy(c1).to(dev), copy.deepcopy(c2).to(dev))
        assert float((got1.cpu() - want1).abs().max()) <= 3e-6 and float((got2.cpu() - want2).abs().max()) <= 3e-6


def test_scores_and_kl_head_golden_and_oracle(pkg, dev, golden_dir):
    """bert_amir5.py:645-648 as one launch (+ the 1-workgroup mean): the REFERENCE's scores and kl (fixture G3)
    from its x, aspect and logits; and seeded shapes against the literal torch ops."""
    g = np.load(os.path.join(golden_dir, "amir55_block.npz"))
    fc = torch.nn.Linear(512, 34).to(dev)
    with torch.no_grad():
        fc.weight.copy_(torch.from_numpy(g["p_fc.0.weight"])); fc.bias.copy_(torch.from_numpy(g["p_fc.0.bias"]))
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    xg = t("gate2")[:, None, :] * t("gc2_out")
    for B, T, H, C in ((3, 7, 64, 5), (9, 31, 256, 34), (2, 100, 96, 12), (1, 1, 8, 2)):
        gen = torch.Generator().manual_seed(B * T)
        x = torch.randn(B, T, H, generator=gen)
        a = torch.randn(B, H, generator=gen)
        lg = torch.randn(B, C, generator=gen)
        lin = torch.nn.Linear(2 * H, C)
        dist = torch.randint(0, 12, (B, T), generator=gen)
        with torch.no_grad():
            ow = lin(torch.cat([x, a.repeat(1, T).view(B, T, H)], dim=2))
            want_s = (lg.repeat(1, T).view(B, T, -1) * ow).sum(2)
            want_kl = (torch.softmax(want_s, 1) * torch.softmax(dist.float(), 1)).sum(1).mean()
            import copy
            got_s, got_kl = pkg.scores_and_kl(x.to(dev), a.to(dev), lg.to(dev), copy.deepcopy(lin).to(dev), dist.to(dev))
        assert float((got_s.cpu() - want_s).abs().max()) <= 2e-5 * max(1.0, float(want_s.abs().max()))
        assert abs(float(got_kl) - float(want_kl)) <= 1e-6
    if "dist" in g.files:
        with torch.no_grad():
            s, kl = pkg.scores_and_kl(xg, t("aspect"), t("logits"), fc, t("dist"))
        np.testing.assert_allclose(s.cpu().numpy(), g["scores"], rtol=0, atol=2e-5)
        assert abs(float(kl) - float(g["kl"])) <= 1e-6
    else:   # the fixture holds no dist: scores only
        with torch.no_grad():
            s, _ = pkg.scores_and_kl(xg, t("aspect"), t("logits"), fc, torch.zeros(4, 31, device=dev))
        np.testing.assert_allclose(s.cpu().numpy(), g["scores"], rtol=0, atol=2e-5)


@pytest.mark.parametrize("precision,fused", [("bf16x3", "block"), ("f16mx8", "block"), ("bf16x3", True), ("fp32", False)],
                         ids=["bf16x3-block", "f16mx8-block", "bf16x3-fused", "fp32"])
def test_collated_batch_through_the_block_golden(pkg, dev, golden_dir, precision, fused):
    """The wire format end to end on the GPU: per-sample graphs cached once by GraphBatcher,
    collated for the batch (device = GPU), through the HIP block, against the REFERENCE's tensors (fixture G3)."""
    from ed_gated_gcn_amd.batcher import GraphBatcher
    g = np.load(os.path.join(golden_dir, "amir55_block.npz"))
    B, T = g["adj"].shape[0], g["adj"].shape[1]
    gb = GraphBatcher()
    ORI_ML = 40
    for i in range(B):                                     # what the dataset holds: [ORI_ML, ORI_ML], identity-padded
        dense = np.eye(ORI_ML, dtype=np.int64)
        dense[:T, :T] = g["adj"][i]
        gb.add("s%d" % i, dense)
    csr = gb.collate(["s%d" % i for i in range(B)], T, dev)
    assert csr.rowmask is not None and csr.rowmask.is_cuda and csr.is_binary
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    gc1 = _layer(pkg, dev, g["p_gc1.weight"], g["p_gc1.bias"], precision, fused)
    gc2 = _layer(pkg, dev, g["p_gc2.weight"], g["p_gc2.bias"], precision, fused)
    with torch.no_grad():
        r = _block(pkg, t("lstm_out"), csr, t("gate1"), t("gate2"), gc1, gc2, fused)
        dense_r = _block(pkg, t("lstm_out"), t("adj"), t("gate1"), t("gate2"), gc1, gc2, fused)
    tol = TOL[precision]
    np.testing.assert_allclose(r["gcn1"].cpu().numpy(), g["gcn1"], rtol=0, atol=tol)
    np.testing.assert_allclose(r["x"].cpu().numpy(), g["gate2"][:, None, :] * g["gc2_out"], rtol=0, atol=tol)
    np.testing.assert_allclose(r["out"].cpu().numpy(), g["out"], rtol=0, atol=tol)
    for k in ("gcn1", "x", "out", "x1", "y1"):
        assert torch.equal(r[k], dense_r[k]), k            # collated CSR == device-side conversion of the dense slice


def test_training_dropout_masks_the_gates_per_token(pkg, dev):
    """bert_amir5.py:621-625 drops entries of the REPEATED [B,T,H] gates: with p = 0.5 a pooled feature is zero only
    when all T tokens lose it (2^-T), not with probability p as one shared [B,H] mask would give."""
    import types
    from ed_gated_gcn_amd import synth

    class _Bert(torch.nn.Module):
        def forward(self, ids, seg, output_all_encoded_layers=True):
            gen = torch.Generator(device=ids.device).manual_seed(1)
            return ([torch.randn(ids.shape[0], ids.shape[1], 768, device=ids.device, generator=gen) for _ in range(12)],
                    torch.zeros(ids.shape[0], 768, device=ids.device))
    opt = types.SimpleNamespace(dropout=0.5, polarities_dim=34, device=dev)
    m = pkg.GatedGCNEventDetector(_Bert(), opt)
    gen = torch.Generator().manual_seed(0)
    for p in m.parameters():
        if p.dim() > 1:
            torch.nn.init.xavier_uniform_(p, generator=gen)
        else:
            torch.nn.init.uniform_(p, -0.05, 0.05, generator=gen)
    m = m.to(dev)
    with torch.no_grad():
        m.gc1.bias.fill_(5.0)                              # gcn1 > 0 everywhere: x1 = 0 iff every token's gate entry was dropped
    rng = np.random.default_rng(0)
    inputs = _ace_batch(rng, 32, 31, 60)
    inputs = {k: v.to(dev) for k, v in inputs.items()}
    seen = {}
    orig = m.gc1.forward_gated

    def spy(*a, **k):                                      # x1 = the first pool of layer 1 (bert_amir5.py:627-635)
        r = orig(*a, **k)
        assert k.get("dropout") is not None and k["dropout"][2] == (0, 1, 2), "the layer launch draws the gates' keep factors"
        seen["x1"] = r[1].detach()
        return r
    m.train()
    m.gc1.forward_gated = spy
    try:
        logits, xy, kl, scores = m(inputs)
    finally:
        m.gc1.forward_gated = orig
    (logits.sum() + xy + kl).backward()                    # the HIP layers train under this path
    assert m.gc1.weight.grad is not None and torch.isfinite(m.gc1.weight.grad).all()
    zero_frac = float((seen["x1"] == 0).float().mean())
    assert zero_frac < 0.02, "pooled features vanish with probability %.2f: the dropout mask is shared by the tokens" % zero_frac


# ---------------------------------------------------------------- training-mode dropout of the gates inside the layer launches
def _drop_mask(pkg, dev, rows, F, p, seed, stream):
    from ed_gated_gcn_amd import _capi
    lib = pkg.load_library()
    m = torch.empty(rows, F, dtype=torch.float32, device=dev)
    _capi.check(lib.ggcn_dropout_mask(rows, F, float(p), int(seed), stream, _capi.ptr(m), _capi.stream_of(dev)), "ggcn_dropout_mask")
    return m


def test_gate_dropout_masks_are_bernoulli_independent_and_reproducible(pkg, dev):
    """ggcn_dropout_mask = the keep factors the layer epilogue and the backward pass draw (csrc/dropout_hash.h): values in
    {0, 1/(1-p)}, dropped fraction p, two independent streams per seed, the same numbers for the same (seed, element)."""
    rows, F = 4096, 768
    for p in (0.1, 0.5):
        m1, m2 = _drop_mask(pkg, dev, rows, F, p, 1234, 1), _drop_mask(pkg, dev, rows, F, p, 1234, 2)
        for m in (m1, m2):
            vals = torch.unique(m).cpu().tolist()
            assert len(vals) == 2 and vals[0] == 0.0 and abs(vals[1] - 1.0 / (1.0 - p)) < 1e-6, vals
            assert abs(float((m == 0).float().mean()) - p) < 2e-3
            assert abs(float(m.mean()) - 1.0) < 5e-3                      # an unbiased gate, like F.dropout
        both = float(((m1 == 0) & (m2 == 0)).float().mean())
        assert abs(both - p * p) < 2e-3                                   # the two gates' draws are independent
        assert float((m1[1:] == 0).float().mul((m1[:-1] == 0).float()).mean()) - p * p < 2e-3   # and so are tokens
        assert torch.equal(m1, _drop_mask(pkg, dev, rows, F, p, 1234, 1))
        assert not torch.equal(m1, _drop_mask(pkg, dev, rows, F, p, 1235, 1))
    assert torch.equal(_drop_mask(pkg, dev, 64, 32, 0.5, 7, 0), torch.ones(64, 32, device=dev))


@pytest.mark.parametrize("T", [31, 60, 100, 200], ids=["T31", "T60-64row", "T100-128row", "T200-eight-wavefronts"])
@pytest.mark.parametrize("precision", ["bf16x3", "f16mx8", "f16mx6"])
def test_block_layers_with_gate_dropout_vs_oracle_on_the_exported_masks(pkg, dev, precision, T):
    """bert_amir5.py:621-640 in training mode: gates repeated to [B,T,H], dropped per token (:621-625), then :626-640.
    The two layer launches draw the keep factors themselves (stream 1 = gate1, stream 2 = gate2 in both layers); the oracle
    gets the same factors from ggcn_dropout_mask and evaluates the reference's formulas -- forward and, through torch
    autograd, backward."""
    from ed_gated_gcn_amd import synth
    if precision == "f16mx6" and T > 32:
        pytest.skip("the fp6 experiment takes graphs of <= 32 nodes")
    B, H, p, seed = 12, 128, 0.5, 2 ** 40 + 99
    rng = np.random.default_rng(5)
    adj = synth.dependency_batch(B, T, 3.5, seed=8, lengths=rng.integers(4, T + 1, size=B))
    t = torch.from_numpy
    x = t(rng.standard_normal((B, T, H)).astype(np.float32))
    g1 = torch.sigmoid(t(rng.standard_normal((B, H)).astype(np.float32)))
    g2 = torch.sigmoid(t(rng.standard_normal((B, H)).astype(np.float32)))
    w1, b1 = synth.layer_params(H, H, seed=1)
    w2, b2 = synth.layer_params(H, H, seed=2)
    R1 = t(rng.standard_normal((B, H)).astype(np.float32))
    R2 = t(rng.standard_normal((B, T, H)).astype(np.float32))
    gc1, gc2 = _layer(pkg, dev, w1, b1, precision).train(), _layer(pkg, dev, w2, b2, precision).train()
    xg, g1g, g2g = (v.to(dev).requires_grad_() for v in (x, g1, g2))
    adj_d = t(adj).to(dev)
    gcn1, x1, y1 = gc1.forward_gated(xg, adj_d, pool_gate_a=g1g, pool_gate_b=g2g, want_pool_a=True, want_pool_b=True,
                                     dropout=(p, seed, (0, 1, 2)))
    xo, out, _ = gc2.forward_gated(gcn1, adj_d, store_gate=g2g, pool_gate_a=g2g, want_pool_a=True, dropout=(p, seed, (2, 2, 0)))
    xy = (x1 * y1).sum(1).mean()
    ((out * R1.to(dev)).sum() + 0.1 * (xo * R2.to(dev)).sum() + 0.01 * xy).backward()

    k1 = _drop_mask(pkg, dev, B * T, H, p, seed, 1).view(B, T, H).cpu()
    k2 = _drop_mask(pkg, dev, B * T, H, p, seed, 2).view(B, T, H).cpu()
    xr, g1r, g2r = x.clone().requires_grad_(), g1.clone().requires_grad_(), g2.clone().requires_grad_()
    w1r, b1r, w2r, b2r = (t(v).clone().requires_grad_() for v in (w1, b1, w2, b2))
    a = t(adj).float()
    gate1 = g1r[:, None, :] * k1                                           # :621-624: repeat, then dropout
    gate2 = g2r[:, None, :] * k2
    gcn1_r = ref_dense.graph_convolution(xr, a, w1r, b1r)                  # :626
    # a max-pool routes its gradient to the argmax row and a ~1e-5 forward difference can flip a near-tie: the reference
    # takes the rows the GPU forward selected (that they are maxima of the reference values too is checked below)
    with torch.no_grad():
        i1 = (gcn1.detach().cpu() * gate1).argmax(1)
        i2 = (gcn1.detach().cpu() * gate2).argmax(1)
        io = xo.detach().cpu().argmax(1)
    x1_r = (gcn1_r * gate1).gather(1, i1[:, None, :])[:, 0]                # :627-635
    y1_r = (gcn1_r * gate2).gather(1, i2[:, None, :])[:, 0]                # :631-636
    xy_r = (x1_r * y1_r).sum(1).mean()                                     # :638
    xo_r = gate2 * ref_dense.graph_convolution(gcn1_r, a, w2r, b2r)        # :639
    out_r = xo_r.gather(1, io[:, None, :])[:, 0]                           # :640
    ((out_r * R1).sum() + 0.1 * (xo_r * R2).sum() + 0.01 * xy_r).backward()
    tol = TOL[precision]
    for name, got, want in (("gcn1", gcn1, gcn1_r), ("x1", x1, x1_r), ("y1", y1, y1_r), ("x", xo, xo_r), ("out", out, out_r)):
        np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), rtol=0, atol=2 * tol, err_msg=name)
    with torch.no_grad():   # the gathered rows are maxima of the reference's own values (up to the tolerance)
        assert float(((gcn1_r * gate1).max(1)[0] - x1_r).abs().max()) <= 2 * tol
        assert float((xo_r.max(1)[0] - out_r).abs().max()) <= 2 * tol
    assert abs(float(xy.detach()) - float(xy_r.detach())) <= 1e-3 * max(1.0, abs(float(xy_r.detach())))
    for name, got, want in (("d x", xg.grad, xr.grad), ("d gate1", g1g.grad, g1r.grad), ("d gate2", g2g.grad, g2r.grad),
                            ("d W1", gc1.weight.grad, w1r.grad), ("d b1", gc1.bias.grad, b1r.grad),
                            ("d W2", gc2.weight.grad, w2r.grad), ("d b2", gc2.bias.grad, b2r.grad)):
        _grad_close(got, want, name, rel=5e-4)
    # half of the gate entries really were dropped, per token: a pooled feature of an all-positive gcn1 would vanish only
    # when all T tokens lose it
    assert 0.45 < float((k1 == 0).float().mean()) < 0.55


# ---------------------------------------------------------------- one launch per layer for 32 < T <= 256 (LitBank: ORI_ML = 100, ACE cased: 231)
@pytest.mark.parametrize("dtype", [torch.float32, torch.uint8, torch.int64])
@pytest.mark.parametrize("T", [33, 64, 65, 100, 128, 129, 231, 256])
def test_wide_row_masks_bit_exact(pkg, dev, T, dtype):
    """ceil(T/32) mask words per node from the dense slice (one kernel), from a CSR (ggcn_csr_rowmask) and from
    the host collation agree with numpy bit for bit."""
    from ed_gated_gcn_amd import _capi, synth
    B = 7
    lens = np.random.default_rng(T).integers(1, T + 1, size=B)
    adj = synth.dependency_batch(B, T, 4.0, seed=T, lengths=lens)
    adj[:, 0, T - 1] = 1                                    # asymmetric, touches the last word
    W = (T + 31) // 32
    want = np.zeros((B * T, W), dtype=np.uint32)
    rr, cc = np.nonzero(adj.reshape(B * T, T))
    np.bitwise_or.at(want, (rr, cc // 32), (np.uint32(1) << (cc % 32).astype(np.uint32)))
    big = torch.zeros(B, T + 9, T + 9, dtype=dtype, device=dev)
    big[:, :T, :T] = torch.from_numpy(adj).to(dev).to(dtype)
    csr = pkg.BatchedCSR.from_dense(big[:, :T, :T])
    torch.cuda.synchronize()
    assert csr.is_binary and np.array_equal(csr.rowmask.cpu().numpy().view(np.uint32).reshape(B * T, W), want)
    rp, ci, _ = synth.csr_from_dense_host(adj)
    host = pkg.BatchedCSR.from_arrays(rp, ci, B, T, dev)
    assert torch.equal(host.rowmask, csr.rowmask)
    lib = pkg.load_library()
    m2 = torch.zeros(B * T * W, dtype=torch.int32, device=dev)
    _capi.check(lib.ggcn_csr_rowmask(_capi.ptr(host.rowptr), _capi.ptr(host.colidx), B, T, _capi.ptr(m2),
                                     _capi.stream_of(dev)), "ggcn_csr_rowmask")
    assert torch.equal(m2, csr.rowmask)
    assert np.array_equal(csr.rowptr.cpu().numpy(), rp)     # the CSR arrays behind the masks are the same graph


@pytest.mark.parametrize("precision", ["bf16x3", "f16mx8"])
@pytest.mark.parametrize("B,T,K,F,padded", [(5, 100, 256, 256, True), (9, 64, 768, 768, False), (3, 33, 72, 40, True),
                                            (4, 128, 300, 300, False), (6, 96, 64, 512, True), (2, 65, 9, 13, True),
                                            (17, 48, 128, 128, True), (5, 231, 256, 256, True), (3, 256, 768, 768, False),
                                            (4, 129, 72, 40, True), (2, 200, 9, 13, True), (7, 160, 300, 520, True)])
def test_wide_graph_layer_one_launch_vs_oracle(pkg, dev, precision, B, T, K, F, padded):
    """ggcn_layer_fused on graphs of 33..256 nodes (64- / 128- / 256-row slots, SB x SB adjacency blocks) against the
    oracle (gcn.py:30-45 + both gates and pools), against the unfused path, and through a host-collated CSR."""
    from ed_gated_gcn_amd import synth
    rng = np.random.default_rng(B * T + F)
    lens = rng.integers(max(1, T // 4), T + 1, size=B) if padded else None
    adj = synth.dependency_batch(B, T, 4.0, seed=B + T, lengths=lens)
    x = rng.standard_normal((B, T, K)).astype(np.float32)
    w, b = synth.layer_params(K, F, seed=4)
    g1 = rng.uniform(-1.0, 1.0, (B, F)).astype(np.float32)            # negative gates: the min side of the pools
    g2 = rng.uniform(0.1, 0.9, (B, F)).astype(np.float32)
    t = torch.from_numpy
    y = ref_dense.graph_convolution(t(x), t(adj.astype(np.float32)), t(w), t(b))
    want = {"out": y * t(g2)[:, None, :], "pa": torch.max(y * t(g1)[:, None, :], 1)[0], "pb": torch.max(y * t(g2)[:, None, :], 1)[0]}
    fused, unfused = _layer(pkg, dev, w, b, precision, True), _layer(pkg, dev, w, b, precision, False)
    xd, ad, g1d, g2d = t(x).to(dev), t(adj).to(dev), t(g1).to(dev), t(g2).to(dev)
    csr = pkg.BatchedCSR.from_dense(ad)
    assert fused.takes_fused_path(xd, csr)
    kw = dict(store_gate=g2d, pool_gate_a=g1d, pool_gate_b=g2d, want_pool_a=True, want_pool_b=True)
    with torch.no_grad():
        got = fused.forward_gated(xd, ad, **kw)
        ref2 = unfused.forward_gated(xd, ad, **kw)
        rp, ci, _ = synth.csr_from_dense_host(adj)
        via_host = fused.forward_gated(xd, pkg.BatchedCSR.from_arrays(rp, ci, B, T, dev), **kw)
        pools_only = fused.forward_gated(xd, ad, pool_gate_a=g1d, pool_gate_b=g2d, want_out=False, want_pool_a=True, want_pool_b=True)
    tol = TOL[precision]
    for name, gv, uv, hv in zip(("out", "pa", "pb"), got, ref2, via_host):
        np.testing.assert_allclose(gv.cpu().numpy(), want[name].numpy(), rtol=0, atol=tol, err_msg=name)
        assert float((gv - uv).abs().max()) <= tol, name
        assert torch.equal(gv, hv), name
    assert pools_only[0] is None and torch.equal(pools_only[1], got[1]) and torch.equal(pools_only[2], got[2])


def test_ace_length_graphs_take_the_eight_wavefront_layer(pkg, dev):
    """Graphs of 193..256 nodes (ACE cased, ORI_ML = 231: constant.py:267) go through the one-launch layer on their own --
    no fused_max_t set -- and so do graphs of 129..192 nodes (f16mx8; bf16x3: 161..192) in batches that fill whole rounds of
    workgroups; small batches of such graphs keep linear + aggregate.
    Both paths give the oracle's numbers; a row with more neighbours than an edge list holds (16) walks its mask words."""
    from ed_gated_gcn_amd import synth
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    H = 256
    w, b = synth.layer_params(H, H, seed=3)
    m = pkg.GraphConvolution(H, H, None).to(dev)
    m.precision = "f16mx8"
    assert m.fused_max_t == 128
    with torch.no_grad():
        m.weight.copy_(torch.from_numpy(w)); m.bias.copy_(torch.from_numpy(b))
    def csr_of(B, T, degree=4.0):
        adj = synth.dependency_batch(B, T, degree, seed=T)
        rp, ci, _ = synth.csr_from_dense_host(adj)
        return adj, pkg.BatchedCSR.from_arrays(rp, ci, B, T, dev)
    took = {}
    for B, T in ((2 * cus, 231), (8, 231), (8, 193), (8, 192), (2 * cus, 176), (2 * cus - cus // 2, 176), (2 * cus, 160), (8, 129), (8, 100)):
        adj, csr = csr_of(B, T)
        took[(B, T)] = m.takes_fused_path(torch.empty(B, T, H, device=dev), csr)
    assert took[(2 * cus, 231)] and took[(8, 231)] and took[(8, 193)] and took[(8, 100)]
    assert took[(2 * cus, 176)] and took[(2 * cus, 160)]     # 129..192 nodes: whole rounds of workgroups only
    assert not took[(8, 192)] and not took[(2 * cus - cus // 2, 176)] and not took[(8, 129)]
    m.precision = "bf16x3"                               # (its main loop has no compiled-in block count: 161 and up)
    adj, csr = csr_of(2 * cus, 160)
    assert not m.takes_fused_path(torch.empty(2 * cus, 160, H, device=dev), csr)
    m.precision = "f16mx8"
    # 129-, 144-, 160-node graphs in a batch of two rounds: the second row group's main loop is compiled for 1 block
    for T in (129, 144, 160, 200):
        B = 2 * cus
        adj, csr = csr_of(B, T)
        rng = np.random.default_rng(T)
        x = torch.from_numpy(rng.standard_normal((B, T, H)).astype(np.float32))
        g = torch.from_numpy(rng.uniform(-1.0, 1.0, (B, H)).astype(np.float32))
        assert m.takes_fused_path(x.to(dev), csr)
        with torch.no_grad():
            out, pa, _ = m.forward_gated(x.to(dev), csr, store_gate=g.to(dev), pool_gate_a=g.to(dev), want_pool_a=True)
        sl = slice(B - 3, B)
        ref = ref_dense.graph_convolution(x[sl], torch.from_numpy(adj[sl].astype(np.float32)), torch.from_numpy(w), torch.from_numpy(b))
        scale = max(1.0, float(ref.abs().max()))
        np.testing.assert_allclose(out[sl].cpu().numpy(), (ref * g[sl, None, :]).numpy(), rtol=0, atol=TOL["f16mx8"] * scale)
        np.testing.assert_allclose(pa[sl].cpu().numpy(), (ref * g[sl, None, :]).max(dim=1)[0].numpy(), rtol=0, atol=TOL["f16mx8"] * scale)
    # dense rows (> 16 neighbours: the mask-word walk) in a small batch
    adj, csr = csr_of(6, 231, degree=24.0)
    rng = np.random.default_rng(7)
    x = torch.from_numpy(rng.standard_normal((6, 231, H)).astype(np.float32))
    with torch.no_grad():
        out, pa, _ = m.forward_gated(x.to(dev), csr, want_pool_a=True)
    ref = ref_dense.graph_convolution(x, torch.from_numpy(adj.astype(np.float32)), torch.from_numpy(w), torch.from_numpy(b))
    scale = max(1.0, float(ref.abs().max()))
    assert int(adj.sum(-1).max()) > 16
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=0, atol=TOL["f16mx8"] * scale)
    np.testing.assert_allclose(pa.cpu().numpy(), ref.max(dim=1)[0].numpy(), rtol=0, atol=TOL["f16mx8"] * scale)
    # numbers: the auto-selected launch against the oracle on a slice of the big batch
    B, T = 2 * cus, 231
    adj, csr = csr_of(B, T)
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.standard_normal((B, T, H)).astype(np.float32))
    g = torch.sigmoid(torch.from_numpy(rng.standard_normal((B, H)).astype(np.float32)))
    with torch.no_grad():
        out, pa, _ = m.forward_gated(x.to(dev), csr, store_gate=g.to(dev), pool_gate_a=g.to(dev), want_pool_a=True)
    ref = ref_dense.graph_convolution(x[:4], torch.from_numpy(adj[:4].astype(np.float32)), torch.from_numpy(w), torch.from_numpy(b))
    scale = max(1.0, float(ref.abs().max()))
    np.testing.assert_allclose(out[:4].cpu().numpy(), (ref * g[:4, None, :]).numpy(), rtol=0, atol=TOL["f16mx8"] * scale)
    np.testing.assert_allclose(pa[:4].cpu().numpy(), (ref * g[:4, None, :]).max(dim=1)[0].numpy(), rtol=0, atol=TOL["f16mx8"] * scale)


@pytest.mark.parametrize("T", [100, 231], ids=["litbank", "ace-cased"])
@pytest.mark.parametrize("precision", ["bf16x3", "f16mx8"])
def test_wide_graph_block_litbank_shape(pkg, dev, precision, T):
    """The gated block at LitBank's shape (constant.py:227: ORI_ML = 100; hidden 256) and at ACE cased's
    (constant.py:267: ORI_ML = 231): two one-launch layers, xy folded into them, against the oracle block;
    training through the same path gives the oracle's gradients."""
    from ed_gated_gcn_amd import synth
    B, H = 12, 256
    rng = np.random.default_rng(7)
    adj = synth.dependency_batch(B, T, 3.5, seed=3, lengths=rng.integers(10, T + 1, size=B))
    x = rng.standard_normal((B, T, H)).astype(np.float32)
    g1 = (1 / (1 + np.exp(-rng.standard_normal((B, H))))).astype(np.float32)
    g2 = (1 / (1 + np.exp(-rng.standard_normal((B, H))))).astype(np.float32)
    (w1, b1), (w2, b2) = synth.layer_params(H, H, seed=1), synth.layer_params(H, H, seed=2)
    ref = _oracle_block(x, adj.astype(np.float32), g1, g2, w1, b1, w2, b2)
    gc1, gc2 = _layer(pkg, dev, w1, b1, precision), _layer(pkg, dev, w2, b2, precision)
    td = lambda a: torch.from_numpy(a).to(dev)
    assert gc1.takes_fused_path(td(x), pkg.BatchedCSR.from_dense(td(adj)))
    with torch.no_grad():
        r = pkg.gated_gcn_block(td(x), td(adj), td(g1), td(g2), gc1, gc2, want_gcn1=True)
    for k in ("gcn1", "x1", "y1", "x", "out"):
        np.testing.assert_allclose(r[k].cpu().numpy(), ref[k].numpy(), rtol=0, atol=TOL[precision], err_msg=k)
    assert abs(float(r["xy"]) - float(ref["xy"])) <= 1e-4 * max(1.0, abs(float(ref["xy"])))
    xg = td(x).requires_grad_()
    gc1.train(); gc2.train()
    rt = pkg.gated_gcn_block(xg, td(adj), td(g1), td(g2), gc1, gc2)
    (rt["x"] * rt["x"]).sum().backward()                  # no arg-max in the loss: near-ties cannot re-route the gradient
    xr = torch.from_numpy(x).requires_grad_()
    w1r, b1r = torch.from_numpy(w1).requires_grad_(), torch.from_numpy(b1).requires_grad_()
    rr = ref_dense.gated_block(xr, torch.from_numpy(adj.astype(np.float32)), torch.from_numpy(g1), torch.from_numpy(g2),
                               w1r, b1r, torch.from_numpy(w2), torch.from_numpy(b2))
    (rr["x"] * rr["x"]).sum().backward()
    _grad_close(xg.grad, xr.grad, "x", rel=5e-4)
    _grad_close(gc1.weight.grad, w1r.grad, "w1", rel=5e-4)
    _grad_close(gc1.bias.grad, b1r.grad, "b1", rel=5e-4)


# ---------------------------------------------------------------- fp16 features: the plain fp16 MFMA linear (config 4)
@pytest.mark.parametrize("M,K,F", [(512, 1024, 1024), (300, 96, 200), (33, 72, 13), (1000, 64, 512), (7, 9, 5)])
def test_f16_linear_for_half_features(pkg, dev, M, K, F):
    """precision='f16' (half features only): X exact, W rounded to fp16, fp32 accumulation, fp16 output -- against
    float64 on the fp16-rounded inputs with the fp16-rounded weights (tight) and with the fp32 weights (config 4's
    gate, SURVEY 8d: atol 2e-3); float32 features are refused."""
    from ed_gated_gcn_amd import synth
    rng = np.random.default_rng(M + K)
    x = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).half()
    w, b = synth.layer_params(K, F, seed=2)
    m = _layer(pkg, dev, w, b, "f16")
    with torch.no_grad():
        y = m.linear(x.to(dev))
    assert y.dtype == torch.float16
    w16 = torch.from_numpy(w).half().double()
    ref_tight = x.double() @ w16
    ref_gate = x.double() @ torch.from_numpy(w).double()
    scale = max(1.0, float(ref_gate.abs().max()))
    assert float((y.cpu().double() - ref_tight).abs().max()) <= 6e-4 * scale        # fp16 output rounding (2^-11) + fp32 sums
    assert float((y.cpu().double() - ref_gate).abs().max()) <= 2e-3 * scale
    with pytest.raises(RuntimeError, match="float16 features only"):
        m(torch.randn(2, 3, K, device=dev), torch.eye(3, device=dev).expand(2, -1, -1))


def test_f16_precision_layer_config4_sample(pkg, dev):
    """BASELINE configs[3] in small with precision='f16': 4 graphs x 512 tokens, hidden 1024, fp16 features; the gated
    layer against the fp32 reference on the fp16-rounded inputs (atol 2e-3) and against precision='f16mx8'."""
    from ed_gated_gcn_amd import synth
    B, T, H = 4, 512, 1024
    rng = np.random.default_rng(9)
    adj = synth.dependency_batch(B, T, 6.0, seed=1)
    x = torch.from_numpy(rng.standard_normal((B, T, H)).astype(np.float32)).half()
    w, b = synth.layer_params(H, H, seed=1)
    g1 = torch.sigmoid(torch.from_numpy(rng.standard_normal((B, H)).astype(np.float32)))
    t = torch.from_numpy
    y = ref_dense.graph_convolution(x.float(), t(adj.astype(np.float32)), t(w), t(b))
    want_pa = torch.max(y * g1[:, None, :], 1)[0]
    rp, ci, _ = synth.csr_from_dense_host(adj)
    csr = pkg.BatchedCSR.from_arrays(rp, ci, B, T, dev)
    outs = {}
    for prec in ("f16", "f16mx8"):
        m = _layer(pkg, dev, w, b, prec)
        with torch.no_grad():
            out, pa, _ = m.forward_gated(x.to(dev), csr, pool_gate_a=g1.to(dev), want_pool_a=True)
        assert out.dtype == torch.float16
        np.testing.assert_allclose(out.float().cpu().numpy(), y.numpy(), rtol=0, atol=2e-3)
        np.testing.assert_allclose(pa.cpu().numpy(), want_pa.numpy(), rtol=0, atol=2e-3)
        outs[prec] = out
    assert float((outs["f16"].float() - outs["f16mx8"].float()).abs().max()) <= 2e-3


# ---------------------------------------------------------------- real-valued adjacency in one launch (<= 32 nodes)
@pytest.mark.parametrize("precision", ["bf16x3", "f16mx8"])
@pytest.mark.parametrize("B,T,K,F,signed", [(8, 32, 256, 256, False), (37, 31, 768, 768, False), (5, 17, 34, 20, False),
                                            (3, 1, 64, 64, False), (64, 32, 128, 384, True), (9, 20, 96, 100, True)])
def test_weighted_adjacency_layer_one_launch_vs_oracle(pkg, dev, precision, B, T, K, F, signed):
    """gcn.py:33-41 take ANY real `adj` (denom = rowsum + 1; adj.hidden / denom): graphs of <= 32 nodes with real weights run
    as ONE launch (ggcn_layer_fused_weighted on ggcn_graph_operands_weighted blocks) and agree with the oracle -- and, far
    inside the gate, with linear + aggregate (fp32 sums of the same terms).  Ragged lengths; gates and both pools."""
    from ed_gated_gcn_amd import synth
    rng = np.random.default_rng(B * 1000 + T)
    lens = rng.integers(1, T + 1, size=B)
    adj = synth.dependency_batch(B, T, min(3.0, T), seed=B + T, lengths=lens).astype(np.float32)
    wts = rng.uniform(0.05, 2.0, size=adj.shape).astype(np.float32)
    if signed:
        wts *= np.where(rng.random(adj.shape) < 0.2, -0.25, 1.0).astype(np.float32)   # some negative weights, row sums stay > 0
    adj = adj * wts
    x = rng.standard_normal((B, T, K)).astype(np.float32)
    w, b = synth.layer_params(K, F, seed=3)
    gs, ga, gb = (torch.sigmoid(torch.from_numpy(rng.standard_normal((B, F)).astype(np.float32))).to(dev) for _ in range(3))
    m = _layer(pkg, dev, w, b, precision)
    xd, ad = torch.from_numpy(x).to(dev), torch.from_numpy(adj).to(dev)
    csr = pkg.BatchedCSR.from_dense(ad)
    assert not csr.is_binary and m.takes_weighted_path(xd, csr) and not m.takes_fused_path(xd, csr)
    with torch.no_grad():
        out, pa, pb = m.forward_gated(xd, csr, store_gate=gs, pool_gate_a=ga, pool_gate_b=gb, want_pool_a=True, want_pool_b=True)
        plain = m(xd, ad)
        m.fused = False
        out2, pa2, pb2 = m.forward_gated(xd, csr, store_gate=gs, pool_gate_a=ga, pool_gate_b=gb, want_pool_a=True, want_pool_b=True)
    y = ref_dense.graph_convolution(torch.from_numpy(x), torch.from_numpy(adj), torch.from_numpy(w), torch.from_numpy(b))
    scale = max(1.0, float(y.abs().max()))
    tol = TOL[precision] * scale
    np.testing.assert_allclose(plain.cpu().numpy(), y.numpy(), rtol=0, atol=tol)
    np.testing.assert_allclose(out.cpu().numpy(), (y * gs.cpu()[:, None, :]).numpy(), rtol=0, atol=tol)
    np.testing.assert_allclose(pa.cpu().numpy(), (y * ga.cpu()[:, None, :]).max(1).values.numpy(), rtol=0, atol=tol)
    np.testing.assert_allclose(pb.cpu().numpy(), (y * gb.cpu()[:, None, :]).max(1).values.numpy(), rtol=0, atol=tol)
    # against linear + aggregate: the same products, the aggregation through hi / lo parts of `hidden` and of D.A_w instead
    # of fp32 FMAs (fp16 pairs carry 22 bits, bf16 pairs 16 -- the precision of the block's W12 tiles in that mode)
    close = (4e-6 if precision == "f16mx8" else 6e-5) * scale
    np.testing.assert_allclose(out.cpu().numpy(), out2.cpu().numpy(), rtol=0, atol=close)
    np.testing.assert_allclose(pa.cpu().numpy(), pa2.cpu().numpy(), rtol=0, atol=close)


def test_weighted_adjacency_outside_the_plane_type_keeps_two_launches(pkg, dev):
    """Mixed-sign weights can make rowsum + 1 tiny: an entry of D.A_w beyond the fp16 planes' range sets the builder's flag and
    the layer stays with linear + aggregate (same results as ever); bf16 planes take it.  The gated block with a real-valued
    adjacency runs through the weighted layers and agrees with the oracle."""
    from ed_gated_gcn_amd import synth
    B, T, H = 6, 32, 128
    rng = np.random.default_rng(5)
    adj = synth.dependency_batch(B, T, 3.0, seed=9).astype(np.float32)
    adj *= rng.uniform(0.1, 1.0, size=adj.shape).astype(np.float32)
    bad = adj.copy()
    bad[2, 5, :] = 0.0
    bad[2, 5, 5], bad[2, 5, 6] = 100.0, -100.99   # rowsum + 1 = 0.01: entries of +-10^4
    x = rng.standard_normal((B, T, H)).astype(np.float32)
    w, b = synth.layer_params(H, H, seed=1)
    xd = torch.from_numpy(x).to(dev)
    m16, mb = _layer(pkg, dev, w, b, "f16mx8"), _layer(pkg, dev, w, b, "bf16x3")
    csr_bad = pkg.BatchedCSR.from_dense(torch.from_numpy(bad).to(dev))
    assert not m16.takes_weighted_path(xd, csr_bad) and mb.takes_weighted_path(xd, csr_bad)
    y = ref_dense.graph_convolution(torch.from_numpy(x), torch.from_numpy(bad), torch.from_numpy(w), torch.from_numpy(b))
    with torch.no_grad():
        for m, prec in ((m16, "f16mx8"), (mb, "bf16x3")):
            got = m(xd, csr_bad)
            np.testing.assert_allclose(got.cpu().numpy(), y.numpy(), rtol=0, atol=TOL[prec] * max(1.0, float(y.abs().max())))
    # the block (bert_amir5.py:626-640) on a real-valued adjacency
    g1 = torch.sigmoid(torch.from_numpy(rng.standard_normal((B, H)).astype(np.float32)))
    g2 = torch.sigmoid(torch.from_numpy(rng.standard_normal((B, H)).astype(np.float32)))
    w2, b2 = synth.layer_params(H, H, seed=2)
    l1, l2 = _layer(pkg, dev, w, b, "f16mx8"), _layer(pkg, dev, w2, b2, "f16mx8")
    ad = torch.from_numpy(adj).to(dev)
    assert l1.takes_weighted_path(xd, pkg.BatchedCSR.from_dense(ad))
    with torch.no_grad():
        r = pkg.gated_gcn_block(xd, ad, g1.to(dev), g2.to(dev), l1, l2)
    ref = _oracle_block(x, adj, g1.numpy(), g2.numpy(), w, b, w2, b2)
    for k in ("x1", "y1", "x", "out"):
        np.testing.assert_allclose(r[k].cpu().numpy(), ref[k].numpy(), rtol=0, atol=TOL["f16mx8"] * max(1.0, float(ref[k].abs().max())))
    assert abs(float(r["xy"]) - float(ref["xy"])) <= 1e-4 * max(1.0, abs(float(ref["xy"])))


@pytest.mark.parametrize("plane", [0, 1], ids=["bf16-pairs", "fp16-pairs"])
def test_weighted_adjacency_operand_matches_numpy(pkg, dev, plane):
    """ggcn_graph_operands_weighted through the C ABI: the device blocks -- hi + lo fragments of D.A_w * 2^10 in MFMA operand order
    -- decoded on the host against numpy's float64 w_ij / (rowsum_i + 1) (gcn.py:35,41), ragged graphs, negative weights; a NULL
    weight array means ones; the flag reports entries outside the plane type and nothing else."""
    from ed_gated_gcn_amd import synth, _capi
    lib = pkg.load_library()
    B, T = 11, 29
    rng = np.random.default_rng(4)
    lens = rng.integers(1, T + 1, size=B)
    adj = synth.dependency_batch(B, T, 4.0, seed=9, lengths=lens).astype(np.float32)
    wadj = adj * rng.uniform(0.01, 3.0, size=adj.shape).astype(np.float32) * np.where(rng.random(adj.shape) < 0.15, -0.3, 1.0).astype(np.float32)
    rp, ci, va = synth.csr_from_dense_host(wadj)
    csr = pkg.BatchedCSR.from_arrays(rp, ci, B, T, dev, vals=va)
    p = _capi.ptr

    def build(vals):
        ops = torch.zeros(lib.ggcn_graph_operands2_bytes(B), dtype=torch.uint8, device=dev)
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        assert lib.ggcn_graph_operands_weighted(p(csr.rowptr), p(csr.colidx), p(vals), B, T, plane, p(ops), p(flag), None) == 0
        torch.cuda.synchronize()
        return ops.cpu().numpy().reshape(B, 4224), int(flag.item())

    def decode(raw):
        def d16(u16):
            if plane == 1:
                return u16.view(np.float16).astype(np.float64)
            return (u16.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
        got = np.zeros((B, 32, 32))
        for s_ in range(2):
            hi = d16(raw[:, s_ * 1024:(s_ + 1) * 1024].copy().view(np.uint16).reshape(B, 64, 8))
            lo = d16(raw[:, 2048 + s_ * 1024:2048 + (s_ + 1) * 1024].copy().view(np.uint16).reshape(B, 64, 8))
            for lane in range(64):
                r, h = lane & 31, lane >> 5
                for e in range(8):
                    got[:, r, 16 * s_ + 8 * (e >> 2) + 4 * h + (e & 3)] = (hi[:, lane, e] + lo[:, lane, e]) / 1024.0
        return got

    raw, flag = build(csr.vals)
    a = wadj.astype(np.float64)
    want = a / (a.sum(-1, keepdims=True) + 1.0)
    got = decode(raw)
    assert flag == 0
    assert np.abs(got[:, :T, :T] - want).max() <= (2.0 ** -15 if plane == 0 else 2.0 ** -19) * max(1.0, np.abs(want).max())
    assert np.abs(got[:, T:, :]).max() == 0 and np.abs(got[:, :, T:]).max() == 0
    assert np.all(raw[:, 4096:4224].copy().view(np.float32) == 0)          # no `mid` bias rides along
    raw1, flag1 = build(None)                                               # NULL weights: ones
    a1 = (wadj != 0).astype(np.float64)
    assert flag1 == 0 and np.abs(decode(raw1)[:, :T, :T] - a1 / (a1.sum(-1, keepdims=True) + 1.0)).max() <= 2.0 ** -15
    big = csr.vals.clone()
    e0 = int(csr.rowptr[3 * T + 2].item())
    row_sum = float(csr.vals[e0:int(csr.rowptr[3 * T + 3].item())].sum().item())
    big[e0] += -row_sum - 1.0 + 1e-3                                        # that row's rowsum + 1 becomes 1e-3
    _, flag2 = build(big)
    assert flag2 == (1 if plane == 1 else 0)                                # entries of ~10^3 * 2^10: beyond fp16, inside bf16
    with pytest.raises(Exception):
        _capi.check(lib.ggcn_graph_operands_weighted(p(csr.rowptr), p(csr.colidx), p(csr.vals), B, 33, plane, p(torch.zeros(64, dtype=torch.uint8, device=dev)), None, None), "T > 32")
