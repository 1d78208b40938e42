"""The oracle against the fixtures the REFERENCE produced (oracle/make_golden.py).

This is what pins parity: tests/golden/*.npz hold outputs of the reference's own
GraphConvolution / BertAmir55 classes; every oracle function must reproduce them.
"""
import os

import numpy as np
import pytest
import torch

from oracle import csr_ref, ref_dense


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_dense_restatement_is_bit_equal_on_config1(golden_dir):
    g = _load(golden_dir, "gcn_config1.npz")
    torch.set_num_threads(1)  # fixtures were generated single-threaded
    out = ref_dense.graph_convolution(torch.from_numpy(g["text"]), torch.from_numpy(g["adj"]),
                                      torch.from_numpy(g["weight"]), torch.from_numpy(g["bias"]))
    assert out.shape == (1, 32, 300)
    # same ATen ops in the same order -> identical bits on the same build;
    # 1e-6 leaves room for a BLAS that blocks differently on another host.
    np.testing.assert_allclose(out.numpy(), g["out"], rtol=0, atol=1e-6)


def _sweep_cases(golden_dir):
    g = _load(golden_dir, "gcn_sweep.npz")
    for i, desc in enumerate(g["cases"]):
        k = "c%d_" % i
        bias = g[k + "bias"] if (k + "bias") in g.files else None
        yield str(desc), g[k + "text"], g[k + "adj"], g[k + "weight"], bias, g[k + "out"]


def test_dense_restatement_sweep(golden_dir):
    n = 0
    for desc, text, adj, w, b, out in _sweep_cases(golden_dir):
        got = ref_dense.graph_convolution(torch.from_numpy(text), torch.from_numpy(adj),
                                          torch.from_numpy(w),
                                          None if b is None else torch.from_numpy(b))
        np.testing.assert_allclose(got.numpy(), out, rtol=0, atol=1e-6, err_msg=desc)
        n += 1
    assert n == 8


def test_c_csr_restatement_sweep(golden_dir):
    """The CSR form (global node ids, one nnz per padding row) equals the dense reference."""
    for desc, text, adj, w, b, out in _sweep_cases(golden_dir):
        B, T, K = text.shape
        rowptr, colidx, vals = csr_ref.csr_from_dense(adj.astype(np.float32))
        binary = bool(np.all(vals == 1.0))
        got = csr_ref.gcn_layer_csr(text.reshape(B * T, K), w, b, rowptr, colidx,
                                    None if binary else vals)
        np.testing.assert_allclose(got.reshape(out.shape), out, rtol=1e-5, atol=2e-6, err_msg=desc)
        # F9: every row keeps at least its self loop, padding rows exactly one entry
        deg = np.diff(rowptr)
        assert deg.min() >= 1
        assert np.all(colidx // T == np.repeat(np.arange(B * T) // T, deg))


def test_gated_block_against_bertamir55(golden_dir):
    g = _load(golden_dir, "amir55_block.npz")
    t = lambda k: torch.from_numpy(g[k])
    r = ref_dense.gated_block(t("lstm_out"), t("adj"), t("gate1"), t("gate2"),
                              t("p_gc1.weight"), t("p_gc1.bias"), t("p_gc2.weight"), t("p_gc2.bias"))
    np.testing.assert_allclose(r["gcn1"].numpy(), g["gcn1"], rtol=0, atol=1e-6)
    gate2 = g["gate2"][:, None, :]
    np.testing.assert_allclose(r["x"].numpy(), gate2 * g["gc2_out"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(r["out"].numpy(), g["out"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(float(r["xy"]), float(g["xy"]), rtol=1e-6)


def test_c_gate_pool_and_overlap(golden_dir):
    g = _load(golden_dir, "amir55_block.npz")
    B, T, H = g["gcn1"].shape
    _, x1 = csr_ref.gate_pool(g["gcn1"].reshape(B * T, H), g["gate1"], B, T)
    _, y1 = csr_ref.gate_pool(g["gcn1"].reshape(B * T, H), g["gate2"], B, T)
    assert abs(csr_ref.gate_overlap(x1, y1) - float(g["xy"])) < 1e-5 * abs(float(g["xy"]))
    gated, out = csr_ref.gate_pool(g["gc2_out"].reshape(B * T, H), g["gate2"], B, T)
    np.testing.assert_allclose(out, g["out"], rtol=0, atol=1e-6)


def test_gate_mlp_matches_fixture(golden_dir):
    """bert_amir5.py:562-571: Sigmoid-Linear-Sigmoid-Linear-Sigmoid on the anchor state."""
    g = _load(golden_dir, "amir55_block.npz")
    a = torch.from_numpy(g["aspect"])
    for name in ("gate1", "gate2"):
        h = torch.sigmoid(a)
        h = torch.sigmoid(h @ torch.from_numpy(g["p_%s.1.weight" % name]).T + torch.from_numpy(g["p_%s.1.bias" % name]))
        h = torch.sigmoid(h @ torch.from_numpy(g["p_%s.3.weight" % name]).T + torch.from_numpy(g["p_%s.3.bias" % name]))
        np.testing.assert_allclose(h.numpy(), g[name], rtol=0, atol=1e-6)


def test_full_classifier_restatement_reproduces_bertamir55(golden_dir):
    """G4: BertAmir55Oracle with the seeded train.py:75-84 parameters and the seeded encoder
    stand-in must give the logits / xy / kl / scores the REFERENCE class produced."""
    from oracle.ref_amir55 import BertAmir55Oracle, EncoderStandIn
    g = _load(golden_dir, "amir55_full.npz")
    torch.set_num_threads(1)
    model = BertAmir55Oracle(EncoderStandIn(int(g["seed_encoder"])), int(g["n_class"]))
    model.seeded_init(torch.Generator().manual_seed(int(g["seed_params"])))
    model.eval()
    inputs = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("in_")}
    with torch.no_grad():
        logits, xy, kl, scores = model(inputs)
    np.testing.assert_allclose(logits.numpy(), g["logits"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(scores.numpy(), g["scores"], rtol=0, atol=2e-5)
    assert abs(float(xy) - float(g["xy"])) <= 1e-6 * abs(float(g["xy"]))
    assert abs(float(kl) - float(g["kl"])) <= 1e-6


@pytest.mark.parametrize("cls_name,fixture", [("BertAmir54Oracle", "amir54_full.npz"), ("BertAmir55NoGateOracle", "amir55nogate_full.npz")])
def test_other_live_classifiers_restatements_reproduce_the_reference(golden_dir, cls_name, fixture):
    """G5 / G6: the restatements of BertAmir54 (bert_amir5.py:434) and BertAmir55NoGate (:654) -- the other models
    train.py:268-282 can select -- against the outputs the REFERENCE classes produced for the same seeds and inputs."""
    import oracle.ref_amir55 as ra
    g = _load(golden_dir, fixture)
    torch.set_num_threads(1)
    model = getattr(ra, cls_name)(ra.EncoderStandIn(int(g["seed_encoder"])), int(g["n_class"]))
    model.seeded_init(torch.Generator().manual_seed(int(g["seed_params"])))
    model.eval()
    inputs = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("in_")}
    with torch.no_grad():
        logits, xy, kl, scores = model(inputs)
    np.testing.assert_allclose(logits.numpy(), g["logits"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(scores.numpy(), g["scores"], rtol=0, atol=2e-5)
    assert abs(float(xy) - float(g["xy"])) <= 1e-6 * max(1.0, abs(float(g["xy"])))
    assert abs(float(kl) - float(g["kl"])) <= 1e-6
