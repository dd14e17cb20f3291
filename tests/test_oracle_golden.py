"""Pins the oracle (oracle/) to golden vectors produced by the reference itself.

Both checkers are held to the fixtures of tests/golden/make_golden.py:
  * CanonicalBackend (C, canonical arithmetic) — the bit-exact target of the HIP kernels;
  * TorchRefBackend  (reference op sequence in torch) — the timed CPU baseline.
kNN contract (SURVEY.md §7 hard part 1): identical neighbour *sets* wherever the k-th / (k+1)-th
distance gap exceeds 1e-5, and on >= 99.9 % of all rows; torch.topk's order among exact ties is not
a contract.
"""
import numpy as np
import pytest
import torch

from conftest import knn_rows_equal_as_sets, load_golden

KNN_FIXTURES = ["g1_self_a", "g1_self_a_rp", "g1_self_dil", "g1_self_dil_rp", "g1_window", "g2_xy",
                "g2_xy_norp", "g3_chunked"]


def _knn_inputs(g):
    x = torch.from_numpy(g["x"]).squeeze(-1).contiguous()
    y = torch.from_numpy(g["y"]).squeeze(-1).contiguous() if g["y"].size else None
    rp = torch.from_numpy(g["relpos"]).squeeze(0).contiguous() if g["relpos"].size else None
    return x, y, rp, int(g["k"]), int(g["dilation"])


@pytest.mark.parametrize("name", KNN_FIXTURES)
@pytest.mark.parametrize("backend", ["canonical", "torch_ref"])
def test_knn_matches_reference(oracle_lib, name, backend):
    from oracle.ref_ops import TorchRefBackend
    be = oracle_lib.CanonicalBackend if backend == "canonical" else TorchRefBackend
    g = load_golden(name)
    x, y, rp, k, d = _knn_inputs(g)
    got = be.knn_graph(x, y, rp, k * d, normalize=True).numpy()
    assert got.shape == g["nn_full"].shape and got.dtype == np.int32
    same = knn_rows_equal_as_sets(got, g["nn_full"])
    safe = g["kth_gap"] > 1e-5
    assert same[safe].all(), "%d rows differ although their k-th gap is > 1e-5" % (~same[safe]).sum()
    assert same.mean() >= 0.999
    # dilated view = every d-th of the full list (torch_edge.py:126-136); compare as sets too
    dil = got[:, :, ::d]
    same_d = knn_rows_equal_as_sets(dil, g["edge_index"][0])
    assert same_d[safe].all()
    # ordered equality wherever all k gaps are comfortable (spot-check of the ascending order)
    ordered = (got == g["nn_full"]).all(-1)
    assert ordered.mean() >= 0.99


def test_canonical_order_is_distance_then_index(oracle_lib):
    """Exact ties: duplicated candidates must come out in ascending index order."""
    x = torch.randn(1, 8, 10, generator=torch.Generator().manual_seed(0))
    y = x[:, :, [0, 0, 1, 1, 2, 2, 3, 3]].contiguous()     # every candidate twice
    idx = oracle_lib.CanonicalBackend.knn_graph(x.contiguous(), y, None, 4).numpy()
    for n in range(4):                                        # query n's two copies are its nearest
        assert list(idx[0, n, :2]) == [2 * n, 2 * n + 1]


def test_pairwise_distance_helpers(oracle_lib):
    from oracle.ref_ops import TorchRefBackend
    g = load_golden("g_distance")
    x = torch.from_numpy(g["x"]).transpose(2, 1).contiguous()   # (B,C,N)
    y = torch.from_numpy(g["y"]).transpose(2, 1).contiguous()
    for be in (oracle_lib.CanonicalBackend, TorchRefBackend):
        np.testing.assert_allclose(be.pairwise_distance(x, None, 0, 40).numpy(), g["pairwise"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(be.pairwise_distance(x, None, 7, 19).numpy(), g["part"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(be.pairwise_distance(x, y, 0, 40).numpy(), g["xy"], rtol=1e-5, atol=1e-5)
        self_knn = be.knn_graph(x, None, None, 5, normalize=False).numpy()
        assert knn_rows_equal_as_sets(self_knn, g["knn_unnormalised"][0]).all()
        xy_knn = be.knn_graph(x, y, None, 5, normalize=False).numpy()
        assert knn_rows_equal_as_sets(xy_knn, g["xy_knn_unnormalised"][0]).all()


@pytest.mark.parametrize("backend", ["canonical", "torch_ref"])
def test_mr_aggregate_and_gather(oracle_lib, backend):
    from oracle.ref_ops import TorchRefBackend
    be = oracle_lib.CanonicalBackend if backend == "canonical" else TorchRefBackend
    g = load_golden("g4_mrconv")
    for tag in ("self", "xy"):
        x = torch.from_numpy(g[tag + "_x"]).squeeze(-1).contiguous()
        y = torch.from_numpy(g[tag + "_y"]).squeeze(-1).contiguous() if tag == "xy" else None
        idx = torch.from_numpy(g[tag + "_idx"]).contiguous()
        k = idx.shape[2]
        gathered = be.gather_fwd(x if y is None else y, idx)
        np.testing.assert_array_equal(gathered.numpy(), g[tag + "_gather"])
        pre, arg = be.mr_fwd(x, y, idx, None, k, 1, want_arg=True)
        np.testing.assert_array_equal(pre.numpy(), g[tag + "_pre"].squeeze(-1))
        gout = torch.from_numpy(g[tag + "_gout"]).squeeze(-1).contiguous()
        if arg is not None:   # scatter formulation from the recorded arg-max ids
            dx2, dy2 = be.mr_bwd_arg(gout, arg, (x if y is None else y).shape[2], y is not None)
            np.testing.assert_allclose(dx2.numpy(), g[tag + "_dx"].squeeze(-1), rtol=1e-5, atol=1e-6)
            if y is not None:
                np.testing.assert_allclose(dy2.numpy(), g[tag + "_dy"].squeeze(-1), rtol=1e-5, atol=1e-6)
        dx, dy = be.mr_bwd(gout, x, y, idx, None, k, 1)
        np.testing.assert_allclose(dx.numpy(), g[tag + "_dx"].squeeze(-1), rtol=1e-5, atol=1e-6)
        if y is not None:
            np.testing.assert_allclose(dy.numpy(), g[tag + "_dy"].squeeze(-1), rtol=1e-5, atol=1e-6)
        # gather backward == scatter-add of ones
        ones = torch.ones_like(gathered)
        cnt = be.gather_bwd(ones, idx, (y if y is not None else x).shape[2])
        ref = np.zeros(cnt.shape[2])
        np.add.at(ref, g[tag + "_idx"][0].reshape(-1), 1)
        np.testing.assert_array_equal(cnt[0, 0].numpy(), ref)


BTI_CASES = [("synapse26", 3, 26), ("synapse6", 3, 6), ("ica26", 3, 26), ("ravir8", 2, 8), ("ravir4", 2, 4),
             ("incl26", 3, 26)]


def bti_luts(name):
    """(lut_a, lut_c) int32 tensors for a fixture, built by the product's own LUT builder."""
    from nextou_amd.loss.bti_loss import BTI_Loss
    from nextou_amd.nnUNetTrainer.nnUNetTrainer_NexToU_BTI_ICA_NoMirroring import nnUNetTrainer_NexToU_BTI_ICA_NoMirroring as ICA
    from nextou_amd.nnUNetTrainer.nnUNetTrainer_NexToU_BTI_Synapse import nnUNetTrainer_NexToU_BTI_Synapse as SYN
    if name.startswith("synapse"):
        inc, exc = [], SYN.exclusion_list
    elif name.startswith("ica"):
        inc, exc = [], ICA.exclusion_list
    elif name.startswith("ravir"):
        inc, exc = [], [[1, 2]]
    else:
        inc, exc = [[1, 2], [[3], [4]]], [[1, 3]]
    return inc, exc


@pytest.mark.parametrize("name,dim,conn", BTI_CASES)
def test_bti_critical_map_bit_exact(oracle_lib, name, dim, conn):
    from nextou_amd.loss.bti_loss import BTI_Loss
    g = load_golden("g7_bti")
    inc, exc = bti_luts(name)
    loss = BTI_Loss(dim=dim, connectivity=conn, inclusion=inc, exclusion=exc, min_thick=1)
    (lut_a, lut_c), = loss._luts_on(torch.device("cpu"))
    logits = torch.from_numpy(g[name + "_logits"])
    labels = oracle_lib.CanonicalBackend.argmax_labels(logits)
    np.testing.assert_array_equal(labels.numpy(), g[name + "_labels"])
    crit = oracle_lib.CanonicalBackend.bti_critical(labels, lut_a, lut_c, conn, 1)
    np.testing.assert_array_equal(crit.numpy(), g[name + "_critical"])
    assert 0 < crit.float().mean() < 1   # the fixture exercises both outcomes


def near_tie_expectations(labels, want, logits, gap):
    """What the canonical ``argmax(softmax)`` restatement owes the reference on tests/golden/g7d_near_ties.npz (reference
    bti_loss.py:131-133 on formula.near_tie_logits):

    * gaps <= 2^-25 (exact ties included): every float32 ``exp`` returns exactly 1 for both logits, so their softmax values are the
      same float on any device and torch.argmax returns the first index — the restatement must agree on EVERY such voxel;
    * gaps in (2^-25, 2^-21]: whether the two softmax values coincide depends on the last bit of ATen's vectorised exp and of its sum;
      the canonical arithmetic (correctly rounded exp, class-order float32 sum, IEEE division) reproduces ATen-CPU's outcome on all but
      ~0.2 % of these voxels (measured 78 of 43 715, against 1 071 on which the reference departs from the plain arg-max) — the declared deviation, gated at 0.5 %;
    * nowhere may a label be anything but the first index of the near-tie pair or the arg-max of the logits."""
    plain = logits.argmax(1).to(torch.uint8)[0]
    labels, want = labels.reshape(-1), torch.from_numpy(want).reshape(-1)
    sure = gap <= 2.0 ** -25
    assert torch.equal(labels[sure], want[sure])
    assert int(sure.sum()) > 10000 and int((want[sure] != plain[sure]).sum()) > 2000      # the fixture exercises the rule
    band = ~sure
    miss = int((labels[band] != want[band]).sum())
    assert miss <= 0.005 * int(band.sum()), (miss, int(band.sum()))
    assert int((want[band] != plain[band]).sum()) > 500 and int((labels[band] != plain[band]).sum()) > 500
    first = (logits[0] >= logits[0].max(0, keepdim=True).values - 2.0 ** -21).float().argmax(0).to(torch.uint8)
    assert bool(((labels == plain) | (labels == first)).all())
    return miss, int(band.sum())


def test_argmax_of_softmax_near_ties(oracle_lib):
    """VERDICT r3 missing #7: ``argmax(softmax(x))`` is not ``argmax(x)`` in float32.  The oracle restates the reference's rule
    (first index among EQUAL float32 softmax values) and is held to the reference's own labels on planted near ties."""
    import formula
    g = load_golden("g7d_near_ties")
    logits, gap = formula.near_tie_logits("g7d.near_ties")
    labels = oracle_lib.CanonicalBackend.argmax_labels(logits)
    near_tie_expectations(labels, g["labels"], logits, gap)
    # the op-sequence port IS the reference's two ops on the CPU: identical everywhere
    from oracle.ref_ops import TorchRefBackend as ReferenceOps
    np.testing.assert_array_equal(ReferenceOps.argmax_labels(logits).numpy(), g["labels"])
    # the case of the verdict: reference 0, plain arg-max 1
    x = torch.tensor([1e-3, float(np.nextafter(np.float32(1e-3), np.float32(1))), -1.0]).reshape(1, 3, 1)
    assert int(oracle_lib.CanonicalBackend.argmax_labels(x)) == 0 == int(torch.argmax(torch.softmax(x, 1), 1))


def test_bti_conv_formulation_equals_bit_logic(oracle_lib):
    """The torch restatement of the reference's float64-conv loop agrees with the bit-logic oracle."""
    from nextou_amd.loss.bti_loss import BTI_Loss, _label_set
    from oracle.ref_ops import bti_critical_ref
    g = load_golden("g7_bti")
    for name, dim, conn in BTI_CASES:
        inc, exc = bti_luts(name)
        inter = [(True, torch.tensor(_label_set(a)), torch.tensor(_label_set(c))) for a, c in inc] + \
                [(False, torch.tensor(_label_set(a)), torch.tensor(_label_set(c))) for a, c in exc]
        P = torch.from_numpy(g[name + "_labels"]).unsqueeze(1).double()
        crit = bti_critical_ref(P, inter, dim, conn, 1)
        np.testing.assert_array_equal(crit.squeeze(1).numpy().astype(np.uint8), g[name + "_critical"])


def test_oracle_c_entry_points_normalise_strided_inputs(oracle_lib):
    """liboracle.so reads raw pointers as dense row-major tensors.  Channels-last logits (the full-resolution stage
    runs NDHWC on the GPU) or any other strided view must give the answer of their row-major copy — the binding
    normalises at the boundary; before that a channels-last tensor was silently read in the wrong order."""
    ora = oracle_lib.CanonicalBackend
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(2, 5, 3, 6, 4, generator=g)
    want = ora.argmax_labels(logits)
    assert torch.equal(want, logits.argmax(1).to(torch.uint8))
    cl = logits.contiguous(memory_format=torch.channels_last_3d)
    assert not cl.is_contiguous()
    assert torch.equal(ora.argmax_labels(cl), want)
    assert torch.equal(ora.argmax_labels(logits.double()), want)                  # dtype normalised as well
    x = torch.randn(2, 6, 40, generator=g)
    xt = x.transpose(1, 2).contiguous().transpose(1, 2)                              # same values, (B, N, C) storage
    assert not xt.is_contiguous()
    assert torch.equal(ora.knn_graph(xt, None, None, 5), ora.knn_graph(x, None, None, 5))
    idx = ora.knn_graph(x, None, None, 5)
    a, _ = ora.mr_fwd(x, None, idx, None, 5, 1)
    b, _ = ora.mr_fwd(xt, None, idx.to(torch.int64), None, 5, 1)
    assert torch.equal(a, b)
    labels = torch.randint(0, 4, (2, 3, 5, 4), generator=g, dtype=torch.uint8)
    lut_a = torch.tensor([0, 1, 0, 0], dtype=torch.int32)
    lut_c = torch.tensor([0, 0, 1, 0], dtype=torch.int32)
    lt = labels.permute(0, 3, 1, 2).contiguous().permute(0, 2, 3, 1)
    assert not lt.is_contiguous()
    assert torch.equal(ora.bti_critical(lt, lut_a, lut_c, 26, 1), ora.bti_critical(labels, lut_a, lut_c, 26, 1))
