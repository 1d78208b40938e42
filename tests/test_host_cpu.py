"""CPU-side checks: the C-ABI library loads and exports every symbol of include/ggcn.h,
the host logic (synthetic batcher, CSR validation, module contract) behaves, and the
product path refuses to run without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import ed_gated_gcn_amd as pkg
from ed_gated_gcn_amd import _capi, synth
from ed_gated_gcn_amd.csr import BatchedCSR
from ed_gated_gcn_amd.gcn import GraphConvolution

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "ggcn.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ggcn_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_expected_entry_points():
    names = _declared_functions()
    for must in ("ggcn_csr_from_dense", "ggcn_linear", "ggcn_aggregate", "ggcn_gate_overlap",
                 "ggcn_weight_pack", "ggcn_abi_version", "ggcn_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(pkg.lib_path())
    for name in _declared_functions():
        assert hasattr(lib, name), "libggcn_hip.so does not export %s" % name


def test_binding_covers_every_declared_symbol_and_abi_matches():
    assert sorted(_capi.PROTOTYPES) == _declared_functions()
    lib = pkg.load_library()
    assert lib.ggcn_abi_version() == _capi.ABI_VERSION
    # pure host helpers (no GPU needed)
    assert lib.ggcn_weight_pack_bytes(768, 768, 0) == 768 * 768 * 2 * 2
    assert lib.ggcn_weight_pack_bytes(300, 300, 0) == 320 * 320 * 2 * 2
    # f16mx8: per (32 columns x 32 k) 2 KiB fp16 + 1 KiB fp8 (residual) + 256 B block scales
    # (+ a 16-byte trailer: max_f sum_k |w[k,f]|, the factor of the range flag's hidden-value bound)
    assert lib.ggcn_weight_pack_bytes(768, 768, 2) == 24 * 24 * 3328 + 16
    assert lib.ggcn_has_f16mx6() in (0, 1)
    assert lib.ggcn_csr_workspace_bytes(131072) == 128 * 4
    assert lib.ggcn_overlap_workspace_bytes(4096) == 4096 * 4


def test_bad_arguments_return_codes_not_crashes():
    lib = pkg.load_library()
    rc = lib.ggcn_linear(None, 8, None, 8, None, None, 8, 4, 8, 8, 0, None)
    assert rc == 1 and b"null" in lib.ggcn_last_error()
    rc = lib.ggcn_aggregate(None, 8, None, None, None, None, 1, 4, 8, None, None, None, None, 8, None, None, None)
    assert rc == 1
    rc = lib.ggcn_csr_from_dense(None, 0, 1, 4, 16, 4, 1, None, None, None, 16, None, None, None, None)
    assert rc == 1
    rc = lib.ggcn_layer_fused(None, 8, None, None, None, None, 1, 4, 8, 8, None, None, None, None, 8, None, None, None, None,
                              None, 0, None)
    assert rc == 1
    rc = lib.ggcn_graph_operands(None, 1, 4, None, None)
    assert rc == 1 and lib.ggcn_graph_operands_bytes(3) == 3 * 2176
    rc = lib.ggcn_csr_rowmask(None, None, 1, 4, None, None)
    assert rc == 1


def test_module_contract_matches_reference_layer():
    m = GraphConvolution(6, 10, opt=None)                 # gcn.py:14
    assert list(m.state_dict().keys()) == ["weight", "bias"]
    assert tuple(m.weight.shape) == (6, 10) and tuple(m.bias.shape) == (10,)   # gcn.py:18,21
    assert m.in_features == 6 and m.out_features == 10
    m2 = GraphConvolution(6, 10, opt=None, bias=False)     # gcn.py:23
    assert m2.bias is None and list(m2.state_dict().keys()) == ["weight"]
    # train.py:75-84 walks parameters of children and re-initialises them in place
    for p in m.parameters():
        assert p.requires_grad
    m.load_state_dict({"weight": torch.zeros(6, 10), "bias": torch.ones(10)})


def test_no_cpu_fallback():
    m = GraphConvolution(8, 8, opt=None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 4, 8), torch.zeros(1, 4, 4))
    with pytest.raises(RuntimeError, match="GPU"):
        BatchedCSR.from_dense(torch.zeros(1, 4, 4))


def test_synthetic_batch_matches_the_contract():
    adj = synth.dependency_batch(64, 32, 4.0)
    assert adj.shape == (64, 32, 32) and adj.dtype == np.uint8
    assert np.all(adj.sum(axis=(1, 2)) == 128)                     # nnz = deg*T exactly
    assert np.all(adj == adj.transpose(0, 2, 1))                    # symmetric (graph.py:73-74)
    assert np.all(adj[:, np.arange(32), np.arange(32)] == 1)        # self loops (graph.py:66)
    rowptr, colidx, vals = synth.csr_from_dense_host(adj)
    assert rowptr[-1] == 64 * 128 and np.all(vals == 1)
    # block diagonal with global ids
    rows = np.repeat(np.arange(64 * 32), np.diff(rowptr))
    assert np.all(rows // 32 == colidx // 32)
    # padded variant: padding rows keep exactly their self loop
    lens = np.array([8, 32, 17, 9])
    p = synth.dependency_batch(4, 32, 4.0, lengths=lens)
    for b, n in enumerate(lens):
        assert np.all(p[b, n:, :].sum(axis=1) == 1) and np.all(p[b, :, n:].sum(axis=0) == 1)
        # nnz parity: an odd remainder cannot be filled by symmetric edge pairs
        assert round(4.0 * n) + (32 - n) - p[b].sum() in (0, 1)
    # deterministic
    assert np.array_equal(adj, synth.dependency_batch(64, 32, 4.0))


def test_algorithmic_bytes_match_survey():
    # SURVEY 8d / BASELINE.md: config 2, one gate per layer
    assert synth.algorithmic_bytes_per_layer(4096, 32, 768, 524288) == 822873092


def test_from_arrays_validates():
    adj = synth.dependency_batch(2, 4, 2.0)
    rowptr, colidx, _ = synth.csr_from_dense_host(adj)
    with pytest.raises(RuntimeError, match="rowptr"):
        BatchedCSR.from_arrays(rowptr[:-1], colidx, 2, 4, "cpu")
    bad = colidx.copy()
    bad[0] = 5  # row 0 (graph 0) pointing into graph 1
    with pytest.raises(RuntimeError, match="block-diagonal"):
        BatchedCSR.from_arrays(rowptr, bad, 2, 4, "cpu")
    ok = BatchedCSR.from_arrays(rowptr, colidx, 2, 4, "cpu")
    assert ok.nnz == len(colidx) and ok.n_nodes == 8
    # row masks: bit j of word b*T+i <=> adj[b,i,j] != 0
    want = (adj.reshape(8, 4).astype(np.uint32) << np.arange(4, dtype=np.uint32)).sum(axis=1)
    assert np.array_equal(ok.rowmask.numpy().view(np.uint32), want)
    assert ok.vals is None
    big = synth.dependency_batch(1, 40, 3.0)
    rp, ci, _ = synth.csr_from_dense_host(big)
    wide = BatchedCSR.from_arrays(rp, ci, 1, 40, "cpu").rowmask.numpy().view(np.uint32).reshape(40, 2)   # 2 words per node
    bits = ((wide[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(40, 64)[:, :40]
    assert np.array_equal(bits, (big[0] != 0).astype(np.uint32))
    huge = synth.dependency_batch(1, 260, 3.0)
    rp, ci, _ = synth.csr_from_dense_host(huge)
    assert BatchedCSR.from_arrays(rp, ci, 1, 260, "cpu").rowmask is None   # T > 256: no masks


def test_graph_batcher_matches_dense_slice():
    """per-sample cache + collate == CSR of the reference's adj[:, :T, :T]."""
    from ed_gated_gcn_amd.batcher import GraphBatcher
    ORI_ML = 31                                           # constant.py:237
    lens = np.array([9, 31, 17, 24, 5])
    dense = synth.dependency_batch(5, ORI_ML, 3.0, seed=2, lengths=lens)   # identity on padding (graph.py:66)
    bt = GraphBatcher()
    for i in range(5):
        bt.add("s%d" % i, dense[i])
    ids = ["s3", "s0", "s4"]
    T = int(lens[[3, 0, 4]].max())                        # batch max length (bert_amir5.py:581)
    got = bt.collate(ids, T, "cpu")
    sl = dense[[3, 0, 4]][:, :T, :T]                      # bert_amir5.py:589
    rp, ci, _ = synth.csr_from_dense_host(sl)
    assert np.array_equal(got.rowptr.numpy(), rp) and np.array_equal(got.colidx.numpy(), ci)
    assert got.vals is None and got.B == 3 and got.T == T
    want_mask = (sl.reshape(-1, T).astype(np.uint32) << np.arange(T, dtype=np.uint32)).sum(axis=1)
    assert np.array_equal(got.rowmask.numpy().view(np.uint32), want_mask)
    # weighted sample keeps its values, the others read as ones
    w = dense[1].astype(np.float32) * 0.5
    bt.add("w", w)
    mix = bt.collate(["w", "s1"], ORI_ML, "cpu")
    rp2, ci2, v2 = synth.csr_from_dense_host(np.stack([w, dense[1].astype(np.float32)]))
    assert np.array_equal(mix.colidx.numpy(), ci2) and np.allclose(mix.vals.numpy(), v2)
    with pytest.raises(ValueError):
        bt.collate(["s0"], 40, "cpu")


def test_product_package_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under ed-gated-gcn_amd/ (Python or HIP/C++) may
    import, include, link or load it, and the C ABI takes no torch types."""
    pkg_dir = os.path.join(ROOT, "ed-gated-gcn_amd")
    bad = []
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                if re.search(r"\boracle\b", text) or "ggcn_oracle" in text:
                    bad.append(os.path.join(dirpath, f))
    assert not bad, "oracle referenced from the product tree: %s" % bad
    header = open(os.path.join(ROOT, "include", "ggcn.h")).read()
    code = re.sub(r"/\*.*?\*/", "", header, flags=re.S)          # prototypes without the comments
    assert "torch" not in code.lower() and "tensor" not in code.lower()
    assert 'extern "C"' in code


def test_default_precision_and_range_check_need_a_gpu():
    """The module's default arithmetic is the benched one (f16mx8, range-flagged by the kernels); asking for the flag of a
    module that is not on a GPU fails loudly instead of answering from the host."""
    import pytest
    import ed_gated_gcn_amd as pkg
    m = pkg.GraphConvolution(8, 8, None)
    assert m.precision == "f16mx8"
    with pytest.raises(RuntimeError, match="GPU"):
        m.check_range()

