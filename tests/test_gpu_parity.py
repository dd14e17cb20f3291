"""Parity tests proper: the HIP path (through the C-ABI) against the CPU oracle on the same inputs.

Bars (BASELINE.json north_star): kNN neighbour indices bit-exact vs the canonical oracle; integer /
bit work (labels, critical map, gather, MR-aggregate forward, which is exact max/sub arithmetic)
bit-exact; float accumulations (MR-aggregate backward, whose scatter order is not fixed) within a
stated fp32 tolerance; block / model outputs vs reference goldens within the tolerances of
test_modules_golden.py.  Full cfg-2 sizes are covered through size-independent properties plus
oracle checks on row subsets.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import formula
import model_cases as mc
from conftest import knn_rows_equal_as_sets, load_golden

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from nextou_amd import _lib, graph_ops
    _lib.lib()  # fails loudly if the .so is missing
    maps = open("/proc/self/maps").read()
    assert "libnextou_hip.so" in maps, "HIP extension not loaded into this process"
    return graph_ops


@pytest.fixture(scope="module")
def ora():
    import oracle
    oracle.lib()
    return oracle.CanonicalBackend


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def _knn_case(ops, ora, B, C, N, M, k, relpos, seed, algos=("fused", "naive"), normalize=True):
    x = _rand((B, C, N), seed)
    y = None if M is None else _rand((B, C, M), seed + 1)
    rp = _rand((N, M or N), seed + 2, 0.05) if relpos else None
    want = ora.knn_graph(x, y, rp, k, normalize=normalize).numpy()
    for algo in algos:
        if algo == "fused" and k > 32:
            continue
        got = ops.knn_graph(x.to(DEV), None if y is None else y.to(DEV), None if rp is None else rp.to(DEV), k,
                            algo=algo, normalize=normalize).cpu().numpy()
        bad = (got != want).any(-1)
        assert not bad.any(), "%s kNN differs from the oracle in %d of %d rows (first: row %s got %s want %s)" % (
            algo, bad.sum(), bad.size, np.argwhere(bad)[0], got[bad][0], want[bad][0])


# ---------------------------------------------------------------------------------------------
# K1
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["g1_self_a", "g1_self_a_rp", "g1_self_dil", "g1_self_dil_rp", "g1_window",
                                  "g2_xy", "g2_xy_norp", "g3_chunked"])
def test_knn_golden_fixtures(ops, ora, name):
    """HIP == oracle bit for bit, and == the reference's sets wherever the k-th gap is > 1e-5."""
    g = load_golden(name)
    x = torch.from_numpy(g["x"]).squeeze(-1).contiguous()
    y = torch.from_numpy(g["y"]).squeeze(-1).contiguous() if g["y"].size else None
    rp = torch.from_numpy(g["relpos"]).squeeze(0).contiguous() if g["relpos"].size else None
    kt = int(g["k"]) * int(g["dilation"])
    want = ora.knn_graph(x, y, rp, kt).numpy()
    for algo in ("fused", "naive"):
        got = ops.knn_graph(x.to(DEV), None if y is None else y.to(DEV), None if rp is None else rp.to(DEV), kt,
                            algo=algo).cpu().numpy()
        np.testing.assert_array_equal(got, want)
        same = knn_rows_equal_as_sets(got, g["nn_full"])
        assert same[g["kth_gap"] > 1e-5].all() and same.mean() >= 0.999


@pytest.mark.parametrize("B,C,N,M,k,relpos", [
    (2, 12, 64, None, 9, False),      # one partial tile
    (3, 7, 50, None, 1, True),        # odd channel count, k = 1, ragged N
    (2, 6, 33, 17, 5, True),          # tiny ragged xy
    (1, 24, 300, 100, 16, True),      # several query tiles, ragged candidates
    (2, 36, 200, 200, 32, False),     # k = 32 (largest fused list)
    (2, 36, 257, 129, 28, True),      # k = 28 bucketed into 32
    (4, 132, 168, None, 7, True),     # cfg-2 Swin window shape
    (2, 264, 512, 1344, 28, True),    # cfg-2 Pool s3 shape, fewer queries
    (1, 324, 1344, None, 32, True),   # cfg-2 Pool s4 self graph
    (2, 12, 40, None, 40, False),     # k == M (everything selected), naive only above 32
    (1, 8, 70, 70, 33, True),         # k > 32 -> naive algorithm
    (1, 4, 9, None, 3, False),        # smaller than one wave
])
def test_knn_shape_sweep_bit_exact(ops, ora, B, C, N, M, k, relpos):
    _knn_case(ops, ora, B, C, N, M, k, relpos, seed=1000 + N + (M or 0) + k)


@pytest.mark.parametrize("B,C,N,k,relpos", [
    (600, 12, 168, 7, True),     # whole window per workgroup (B * ceil(N/32) >= 512), list bucket 7
    (100, 33, 191, 14, True),    # whole window, ragged N and C, bucket 14
    (90, 8, 192, 28, False),     # whole window at the 192-point limit, bucket 28
    (3, 5, 64, 32, True),        # N <= 64: one 64-wide tile pair, K = 32
    (2, 3, 1, 1, False),         # a single point
    (1, 4, 33, 9, True),         # N <= 64 ragged, K bucketed into 14
    (16, 324, 168, 14, True),    # cfg-2 stage-4 Swin windows: candidate ranges split over workgroups + merge
    (12, 40, 130, 32, True),     # split, ragged last range, K = 32
    (11, 17, 97, 20, False),     # split, odd everything, K bucketed into 28
    (2, 324, 168, 28, True),     # cfg-2 stage 5 (too few windows: stays on prep + fused + merge)
])
def test_knn_single_launch_window_kernel_bit_exact(ops, ora, B, C, N, k, relpos):
    """Self graphs of <= 192 points take knn_window_kernel (normalisation, squared norms, MFMA distances and selection in one launch;
    knn_graph.hip plan_window); the same ids as the oracle and as the three-launch path it replaces."""
    _knn_case(ops, ora, B, C, N, None, k, relpos, seed=4000 + N + k, algos=("fused",))
    x = _rand((B, C, N), 4000 + N + k).to(DEV)
    rp = _rand((N, N), 4002 + N + k, 0.05).to(DEV) if relpos else None
    one = ops.knn_graph(x, None, rp, k, algo="fused")
    # (the switch is read once per process by the library: the three-launch path is reached through an explicit y = x graph)
    three = ops.knn_graph(x, x.clone(), rp, k, algo="fused")
    assert torch.equal(one, three)


@pytest.mark.parametrize("B,C,N,k", [(600, 12, 168, 7), (90, 8, 192, 28), (100, 33, 191, 14)])
def test_knn_window_kernel_one_group_variant_bit_exact(ops, ora, monkeypatch, B, C, N, k):
    """Whole windows run as two wave groups of three candidate tiles by default; NEXTOU_KNN_WIN_G=1 (read per call) keeps the one-group
    kernel with six tiles per wave for A/B — same ids."""
    monkeypatch.setenv("NEXTOU_KNN_WIN_G", "1")
    _knn_case(ops, ora, B, C, N, None, k, True, seed=5000 + N + k, algos=("fused",))
    x = _rand((B, C, N), 5000 + N + k).to(DEV)
    one = ops.knn_graph(x, None, None, k, algo="fused")
    monkeypatch.delenv("NEXTOU_KNN_WIN_G")
    assert torch.equal(one, ops.knn_graph(x, None, None, k, algo="fused"))


def test_knn_auto_selects_and_rejects(ops):
    x = _rand((1, 8, 40), 3).to(DEV)
    assert ops.knn_graph(x, k=33).shape == (1, 40, 33)           # auto -> naive
    with pytest.raises(RuntimeError, match="K <= 32"):
        ops.knn_graph(x, k=33, algo="fused")
    with pytest.raises(RuntimeError, match="out of range"):
        ops.knn_graph(x, k=41)


def test_knn_exact_ties_break_by_index(ops, ora):
    """Duplicated candidates (exact distance ties) and all-equal rows."""
    x = _rand((2, 16, 96), 5)
    y = torch.cat([x[:, :, :48], x[:, :, :48]], 2).contiguous()      # every candidate twice
    for k in (4, 16):
        want = ora.knn_graph(x, y, None, k).numpy()
        for algo in ("fused", "naive"):
            got = ops.knn_graph(x.to(DEV), y.to(DEV), None, k, algo=algo).cpu().numpy()
            np.testing.assert_array_equal(got, want)
    const = torch.ones(1, 6, 40)
    got = ops.knn_graph(const.to(DEV), None, None, 8).cpu().numpy()
    np.testing.assert_array_equal(got, np.broadcast_to(np.arange(8), (1, 40, 8)))


def test_knn_unnormalised_and_pairwise(ops, ora):
    _knn_case(ops, ora, 2, 10, 70, 30, 6, False, seed=77, normalize=False)
    _knn_case(ops, ora, 2, 10, 70, None, 6, True, seed=78, normalize=False)
    x, y = _rand((2, 10, 70), 9), _rand((2, 10, 30), 10)
    for yy, (r0, r1) in ((None, (0, 70)), (None, (13, 41)), (y, (0, 70))):
        want = ora.pairwise_distance(x, yy, r0, r1)
        got = ops.pairwise_sq_distance(x.to(DEV), None if yy is None else yy.to(DEV), r0, r1).cpu()
        assert torch.equal(got, want)
    g = load_golden("g_distance")
    from nextou_amd.network_architecture import torch_edge
    xr, yr = torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["y"]).to(DEV)
    np.testing.assert_allclose(torch_edge.pairwise_distance(xr).cpu().numpy(), g["pairwise"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(torch_edge.part_pairwise_distance(xr, 7, 19).cpu().numpy(), g["part"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(torch_edge.xy_pairwise_distance(xr, yr).cpu().numpy(), g["xy"], rtol=1e-5, atol=1e-5)


def test_knn_full_size_cfg2(ops, ora):
    """cfg-2 sizes: fused == naive on the GPU for every row; oracle on row subsets; properties."""
    # Swin s2: 1024 windows x 168 points, C = 132, k = 7
    x = _rand((1024, 132, 168), 21)
    rp = _rand((168, 168), 22, 0.05)
    fused = ops.knn_graph(x.to(DEV), None, rp.to(DEV), 7, algo="fused")
    naive = ops.knn_graph(x.to(DEV), None, rp.to(DEV), 7, algo="naive")
    assert torch.equal(fused, naive)
    sub = [0, 1, 511, 1023]
    want = ora.knn_graph(x[sub].contiguous(), None, rp, 7)
    assert torch.equal(fused[sub].cpu(), want)
    f = fused.cpu().numpy()
    assert f.min() >= 0 and f.max() < 168
    assert (np.sort(f, -1)[..., 1:] != np.sort(f, -1)[..., :-1]).all()      # distinct per row
    # Pool s3: N = 10752 queries, M = 1344 pooled candidates, C = 264, k = 28
    xq, yc, rp = _rand((2, 264, 10752), 23), _rand((2, 264, 1344), 24), _rand((10752, 1344), 25, 0.05)
    fused = ops.knn_graph(xq.to(DEV), yc.to(DEV), rp.to(DEV), 28, algo="fused")
    naive = ops.knn_graph(xq.to(DEV), yc.to(DEV), rp.to(DEV), 28, algo="naive")
    assert torch.equal(fused, naive)
    rows = torch.arange(0, 10752, 41)
    want = ora.knn_graph(xq[:, :, rows].contiguous(), yc, rp[rows].contiguous(), 28)
    assert torch.equal(fused[:, rows].cpu(), want)
    # ascending distance along k (recomputed in fp64 on the GPU)
    xn, yn = F.normalize(xq.to(DEV).double(), dim=1), F.normalize(yc.to(DEV).double(), dim=1)
    d = (xn * xn).sum(1).unsqueeze(2) - 2 * torch.einsum("bcn,bcm->bnm", xn, yn) + (yn * yn).sum(1).unsqueeze(1)
    d = d + rp.to(DEV).double()
    picked = d.gather(2, fused.long())
    assert (picked[:, :, 1:] - picked[:, :, :-1]).min() > -1e-5
    kth = torch.kthvalue(d, 28, dim=2).values
    assert (picked[:, :, -1] - kth).abs().max() < 1e-5                     # really the 28 smallest


# ---------------------------------------------------------------------------------------------
# K2 + gather
# ---------------------------------------------------------------------------------------------
def _idx(B, N, M, K, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.stack([torch.stack([torch.randperm(M, generator=g)[:K] for _ in range(N)]) for _ in range(B)]).to(torch.int32)


@pytest.mark.parametrize("B,C,N,M,K,stride_extra,step", [
    (2, 12, 48, None, 5, 0, 1),
    (3, 6, 200, 64, 8, 0, 1),
    (2, 33, 168, None, 7, 0, 1),         # window shape, K < bucket
    (2, 24, 3000, 300, 14, 0, 1),        # N tiled over several workgroups, atomics flush
    (1, 10, 1500, None, 28, 0, 1),       # self graph, K bucket 32
    (2, 8, 100, 40, 4, 4, 2),            # dilated view: every 2nd of 8 stored neighbours
    (1, 4, 20000, None, 6, 0, 1),        # self rows too long for LDS -> global-atomic fallback
    (1, 3, 64, 20000, 9, 0, 1),          # source rows too long for LDS (forward fallback)
    (2, 5, 70, 70, 33, 0, 1),            # K > 32 -> generic kernels
    (1, 6, 8000, None, 9, 0, 1),         # 4096 < M <= 9728 (a 160^3 patch's 20^3 self graph): arg tape + one-row 8-byte accumulators
    (1, 5, 1200, 9000, 6, 0, 1),         # the same for a pooled graph's source side
])
def test_mr_aggregate_vs_oracle(ops, ora, B, C, N, M, K, stride_extra, step):
    x = _rand((B, C, N), 31)
    y = None if M is None else _rand((B, C, M), 32)
    idx = _idx(B, N, M or N, K * step + stride_extra, 33)
    want, want_arg = ora.mr_fwd(x, y, idx, None, K, step, want_arg=True)
    xd = x.to(DEV).requires_grad_(True)
    yd = None if y is None else y.to(DEV).requires_grad_(True)
    out = ops.mr_aggregate(xd, idx.to(DEV), yd, k=K, idx_step=step)
    assert torch.equal(out.detach().cpu(), want), "forward must be bit-exact (pure max / sub arithmetic)"
    gout = _rand(out.shape, 34)
    grads = torch.autograd.grad(out, [xd] if yd is None else [xd, yd], gout.to(DEV))
    dx, dy = ora.mr_bwd(gout, x, y, idx, None, K, step)
    # both backward formulations of the library: recorded arg-max scatter and recompute
    if ops._HIP.mr_has_arg(B, C, N, M or N, K):
        _, arg = ops._HIP.mr_fwd(x.to(DEV), None if y is None else y.to(DEV), idx.to(DEV), None, K, step, want_arg=True)
        assert torch.equal(arg.cpu(), want_arg), "recorded arg-max ids must equal the oracle's (first max wins)"
    rdx, rdy = ops._HIP.mr_bwd(gout.to(DEV), x.to(DEV), None if y is None else y.to(DEV), idx.to(DEV), None, K, step)
    assert float((rdx.cpu() - dx).abs().max()) <= 1e-5 * max(1.0, float(dx.abs().max()))
    if rdy is not None:
        assert float((rdy.cpu() - dy).abs().max()) <= 1e-5 * max(1.0, float(dy.abs().max()))
    # scatter-add order differs (LDS / L2 atomics): fp32 tolerance relative to the accumulated magnitude
    tol = 1e-5 * max(1.0, float(dx.abs().max()))
    assert float((grads[0].cpu() - dx).abs().max()) <= tol
    if yd is not None:
        tol = 1e-5 * max(1.0, float(dy.abs().max()))
        assert float((grads[1].cpu() - dy).abs().max()) <= tol


@pytest.mark.parametrize("force,quads,threads", [(None, None, None), (None, "1", "256"), (None, "2", "512"), (None, "4", "128"),
                                                  ("v", None, None)])
@pytest.mark.parametrize("B,C,N,M,K,stride_extra,step", [
    (2, 33, 168, None, 7, 0, 1),          # window, exact 7 x 1 instance, C not a multiple of 4
    (2, 12, 300, 96, 8, 0, 1),
    (2, 24, 3000, 300, 14, 0, 1),
    (1, 20, 700, 128, 16, 0, 1),
    (2, 30, 2500, 1344, 28, 0, 1),        # the pooled stage-3 list length
    (1, 10, 1500, None, 32, 0, 1),
    (2, 9, 500, 200, 19, 0, 1),           # no exact instance: bound-checked 8 x 4
    (2, 8, 100, 40, 4, 4, 2),             # dilated view (idx_step 2): scalar id loads
    (1, 6, 257, 64, 28, 3, 1),            # id rows not 16-byte aligned
])
def test_mr_forward_kernel_variants(ops, ora, monkeypatch, force, quads, threads, B, C, N, M, K, stride_extra, step):
    """Both LDS forward kernels of K2 (the channel-quad kernel in its default plan and with 1 / 2 / 4 quads per workgroup; the
    dword kernel, NEXTOU_MR_FWD=v) against the oracle: values and recorded arg-max ids bit for bit, with duplicated feature
    values so that exact ties of the rounded differences occur and the first-maximum rule is exercised."""
    if force is not None:
        monkeypatch.setenv("NEXTOU_MR_FWD", force)
    if quads is not None:
        monkeypatch.setenv("NEXTOU_QB_QUADS", quads)
        monkeypatch.setenv("NEXTOU_QB_THREADS", threads)
        monkeypatch.setenv("NEXTOU_QB_WGS", "64")
    x = _rand((B, C, N), 41)
    y = None if M is None else _rand((B, C, M), 42)
    src = x if y is None else y
    src[:, :, 1::3] = src[:, :, 0:-1:3][:, :, : src[:, :, 1::3].shape[2]]          # equal neighbours -> ties
    src.mul_(4).round_().div_(4)                                                      # coarse values: more ties
    idx = _idx(B, N, M or N, K * step + stride_extra, 43)
    want, want_arg = ora.mr_fwd(x, y, idx, None, K, step, want_arg=True)
    xd, yd, idd = x.to(DEV), None if y is None else y.to(DEV), idx.to(DEV)
    assert ops._HIP.mr_has_arg(B, C, N, M or N, K)
    got, arg = ops._HIP.mr_fwd(xd, yd, idd, None, K, step, want_arg=True)
    assert torch.equal(got.cpu(), want)
    assert torch.equal(arg.cpu(), want_arg)
    plain, none = ops._HIP.mr_fwd(xd, yd, idd, None, K, step, want_arg=False)
    assert none is None and torch.equal(plain.cpu(), want)


def test_mr_aggregate_center_index_and_golden(ops, ora):
    x = _rand((2, 7, 60), 41)
    idx, ctr = _idx(2, 60, 60, 5, 42), _idx(2, 60, 60, 5, 43)
    want, _ = ora.mr_fwd(x, None, idx, ctr, 5, 1)
    xd = x.to(DEV).requires_grad_(True)
    out = ops.mr_aggregate(xd, idx.to(DEV), None, center_idx=ctr.to(DEV))
    assert torch.equal(out.detach().cpu(), want)
    gout = _rand(out.shape, 44)
    (dx,) = torch.autograd.grad(out, xd, gout.to(DEV))
    want_dx, _ = ora.mr_bwd(gout, x, None, idx, ctr, 5, 1)
    assert float((dx.cpu() - want_dx).abs().max()) <= 1e-5 * max(1.0, float(want_dx.abs().max()))
    g = load_golden("g4_mrconv")   # the reference's own numbers
    for tag in ("self", "xy"):
        xg = torch.from_numpy(g[tag + "_x"]).squeeze(-1).to(DEV).requires_grad_(True)
        yg = torch.from_numpy(g[tag + "_y"]).squeeze(-1).to(DEV).requires_grad_(True) if tag == "xy" else None
        ig = torch.from_numpy(g[tag + "_idx"]).to(DEV)
        pre = ops.mr_aggregate(xg, ig, yg)
        np.testing.assert_array_equal(pre.detach().cpu().numpy(), g[tag + "_pre"].squeeze(-1))
        gg = torch.from_numpy(g[tag + "_gout"]).squeeze(-1).to(DEV)
        grads = torch.autograd.grad(pre, [xg] if yg is None else [xg, yg], gg)
        np.testing.assert_allclose(grads[0].cpu().numpy(), g[tag + "_dx"].squeeze(-1), rtol=1e-5, atol=1e-6)
        if yg is not None:
            np.testing.assert_allclose(grads[1].cpu().numpy(), g[tag + "_dy"].squeeze(-1), rtol=1e-5, atol=1e-6)
        gathered = ops.gather_neighbors(yg if yg is not None else xg, ig)
        np.testing.assert_array_equal(gathered.detach().cpu().numpy(), g[tag + "_gather"])


def test_gather_backward(ops, ora):
    src = _rand((2, 9, 50), 51)
    idx = _idx(2, 80, 50, 6, 52)
    sd = src.to(DEV).requires_grad_(True)
    out = ops.gather_neighbors(sd, idx.to(DEV))
    assert torch.equal(out.detach().cpu(), ora.gather_fwd(src, idx))
    gout = _rand(out.shape, 53)
    (ds,) = torch.autograd.grad(out, sd, gout.to(DEV))
    want = ora.gather_bwd(gout, idx, 50)
    assert float((ds.cpu() - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("B,C,N,M,K", [(64, 132, 168, None, 7), (2, 48, 10752, 1344, 28), (3, 10, 200, 64, 8), (2, 7, 50, None, 5),
                                       (2, 6, 8000, None, 9), (2, 5, 3000, 9700, 6)])       # M > 4096: > 32 KB of accumulators per row
def test_mr_backward_is_bit_reproducible_and_tighter_than_fp32(ops, B, C, N, M, K):
    """The default backward (mr_bwd_fix_kernel: 64-bit fixed-point LDS accumulators) gives the SAME bits run after run — the float-atomic
    scatter it replaces did not (VERDICT r3 weak #1-iii) — and is closer to the float64 scatter than fp32 summation order allows it to be
    in general: <= 2 ulp of its two terms + N * 2^-47 of the tile's largest gradient (half a quantum per addend).  Gradients spanning 12 orders of magnitude across
    channels, clustered winners (many queries share one source), an inf poisons its tile with NaN."""
    g = torch.Generator().manual_seed(B * 7 + N)
    m = M or N
    gout = torch.randn(B, 2 * C, N, generator=g) * torch.logspace(-6, 6, 2 * C).view(1, -1, 1)
    arg = (torch.randint(0, m, (B, C, N), generator=g) // 3 * 3 % m).to(torch.int16)           # runs of equal winners
    go, ar = gout.to(DEV), arg.to(DEV)
    dx, dy = ops._HIP.mr_bwd_arg(go, ar, m, M is not None)
    for _ in range(3):
        dx2, dy2 = ops._HIP.mr_bwd_arg(go, ar, m, M is not None)
        assert torch.equal(dx, dx2) and (dy is None or torch.equal(dy, dy2))
    g64, idx = gout.double(), arg.long() & 0xffff
    scat = torch.zeros(B, C, m, dtype=torch.float64).scatter_add_(2, idx, g64[:, 1::2])
    ident = (gout[:, 0::2] - gout[:, 1::2]).double()              # the identity term is fp32 arithmetic in kernel and reference alike
    want_dx, want_dy = (ident + scat, None) if M is None else (ident, scat)
    mag_dx, mag_dy = (ident.abs() + scat.abs(), None) if M is None else (ident.abs(), scat.abs())
    tile_max = g64[:, 1::2].abs().amax(dim=(1, 2), keepdim=True)  # >= the largest |g| of any workgroup tile of that sample
    for got, want, mag in ((dx, want_dx, mag_dx), (dy, want_dy, mag_dy)):
        if got is None:
            continue
        err = (got.cpu().double() - want).abs()
        bound = 2.0 ** -22 * mag + N * 2.0 ** -47 * tile_max
        assert bool((err <= bound).all()), float((err / bound).max())
    go[0, 3, 5] = float("inf")
    dx3, dy3 = ops._HIP.mr_bwd_arg(go, ar, m, M is not None)
    scattered = dx3 if M is None else dy3
    assert bool(torch.isnan(scattered[0, 1]).any()) and bool(torch.isfinite(scattered[B - 1]).all())


def test_mr_aggregate_full_size_cfg2(ops):
    """cfg-2 Swin s2 and Pool s3 sizes against the materialising torch formulation on the GPU."""
    for (B, C, N, M, K) in ((1024, 132, 168, None, 7), (2, 264, 10752, 1344, 28)):
        x = _rand((B, C, N), 61).to(DEV).requires_grad_(True)
        y = None if M is None else _rand((B, C, M), 62).to(DEV).requires_grad_(True)
        g = torch.Generator().manual_seed(63)
        idx = torch.randint(0, M or N, (B, N, K), generator=g, dtype=torch.int32).to(DEV)
        out = ops.mr_aggregate(x, idx, y)
        from oracle.ref_ops import mr_aggregate_ref    # the reference's materialising op sequence
        ref = mr_aggregate_ref(x, idx, y)
        assert torch.equal(out, ref)
        assert torch.equal(out[:, 0::2], x)                        # interleave: even channels are x
        gout = _rand((B, 2 * C, N), 64).to(DEV)
        got = torch.autograd.grad(out, [x] if y is None else [x, y], gout)
        want = torch.autograd.grad(ref, [x] if y is None else [x, y], gout)
        for a, b in zip(got, want):
            assert float((a - b).abs().max()) <= 1e-4 * max(1.0, float(b.abs().max()))
        # conservation: sum of dx (+ dy) equals the sum of the identity-branch gradient
        total = sum(float(t.double().sum()) for t in got)
        assert abs(total - float(gout[:, 0::2].double().sum())) <= 1e-2 * (1 + abs(total))


# ---------------------------------------------------------------------------------------------
# K5
# ---------------------------------------------------------------------------------------------
def test_bti_kernels_vs_oracle_and_reference(ops, ora):
    from nextou_amd.loss.bti_loss import BTI_Loss
    from test_oracle_golden import BTI_CASES, bti_luts
    g = load_golden("g7_bti")
    for name, dim, conn in BTI_CASES:
        inc, exc = bti_luts(name)
        loss = BTI_Loss(dim=dim, connectivity=conn, inclusion=inc, exclusion=exc, min_thick=1)
        logits = torch.from_numpy(g[name + "_logits"]).to(DEV).requires_grad_(True)
        target = torch.from_numpy(g[name + "_target"]).float().to(DEV)
        labels = ops.argmax_labels(logits)
        np.testing.assert_array_equal(labels.cpu().numpy(), g[name + "_labels"])
        crit = loss.critical_voxels_from_labels(labels)
        np.testing.assert_array_equal(crit.cpu().numpy(), g[name + "_critical"])
        value = loss(logits, target)
        np.testing.assert_allclose(value.item(), float(g[name + "_loss"]), rtol=1e-10)
        (grad,) = torch.autograd.grad(value, logits)
        np.testing.assert_allclose(grad.cpu().numpy(), g[name + "_grad"], rtol=1e-5, atol=1e-7)
        # the same from channels-last logits — what the network's heads hand the loss: read where they lie (ABI v9: no (B, L, V) copy);
        # labels and the critical map identical, the float64 loss to the last bits, the gradient comes back channels-last
        if logits.shape[1] % 2 == 0:
            mf = torch.channels_last_3d if logits.dim() == 5 else torch.channels_last
            cl = logits.detach().contiguous(memory_format=mf).requires_grad_(True)
            assert torch.equal(ops.argmax_labels(cl), labels)
            v2 = loss(cl, target)
            np.testing.assert_allclose(v2.item(), value.item(), rtol=1e-13)
            (g2,) = torch.autograd.grad(v2, cl)
            assert g2.is_contiguous(memory_format=mf) and torch.equal(g2.contiguous(), grad)


@pytest.mark.parametrize("shape,conn,thick", [((2, 7, 9, 11), 26, 1), ((1, 5, 6, 7), 26, 2), ((2, 9, 10, 13), 6, 1),
                                              ((3, 17, 19), 8, 1), ((3, 17, 19), 4, 1), ((1, 33, 21), 8, 2),
                                              ((1, 9, 12, 10), 26, 3),      # the widest box the tiled kernel takes
                                              ((2, 10, 11, 13), 26, 4), ((2, 40, 37), 8, 5)])      # min_thick > 3: bti_critical_naive_kernel
def test_bti_ragged_shapes(ops, ora, shape, conn, thick):
    g = torch.Generator().manual_seed(sum(shape) + conn)
    labels = torch.randint(0, 6, shape, generator=g, dtype=torch.uint8)
    lut_a = torch.tensor([0, 1, 2, 4, 1, 0] + [0] * 250, dtype=torch.int32)
    lut_c = torch.tensor([0, 2, 1, 0, 4, 7] + [0] * 250, dtype=torch.int32)
    want = ora.bti_critical(labels, lut_a, lut_c, conn, thick)
    got = ops.bti_critical_map(labels.to(DEV), lut_a.to(DEV), lut_c.to(DEV), conn, thick)
    assert torch.equal(got.cpu(), want)
    logits = torch.randn((2, 5) + shape[1:], generator=g)
    logits[0, 3] = logits[0, 1]                                  # exact ties -> first index wins
    assert torch.equal(ops.argmax_labels(logits.to(DEV)).cpu(), ora.argmax_labels(logits))


def test_argmax_of_softmax_near_ties_on_gpu(ops, ora):
    """``argmax(softmax(x))`` (reference bti_loss.py:131-133) on logits whose top two entries float32 softmax cannot tell apart: the
    kernel equals the oracle bit for bit (same fixed fma sequence for the band's exp), and is held to the reference's own labels
    exactly as the oracle is (tests/test_oracle_golden.py::near_tie_expectations; golden g7d_near_ties from make_golden.py)."""
    import formula
    from test_oracle_golden import near_tie_expectations
    g = load_golden("g7d_near_ties")
    logits, gap = formula.near_tie_logits("g7d.near_ties")
    want = ora.argmax_labels(logits)
    got = ops.argmax_labels(logits.to(DEV)).cpu()
    assert torch.equal(got, want)
    near_tie_expectations(got, g["labels"], logits, gap)
    odd = logits[:, :, :59999].contiguous()                     # V % 4 != 0: the scalar kernel
    assert torch.equal(ops.argmax_labels(odd.to(DEV)).cpu(), ora.argmax_labels(odd))
    rows = logits.reshape(1, 14, 240, 250).to(DEV).contiguous(memory_format=torch.channels_last)      # channels-last rows kernel
    assert torch.equal(ops.argmax_labels(rows).cpu().reshape(1, -1), want)
    x = torch.tensor([1e-3, float(np.nextafter(np.float32(1e-3), np.float32(1))), -1.0]).reshape(1, 3, 1)
    assert int(ops.argmax_labels(x.to(DEV))) == 0               # the reference says 0, the plain arg-max 1


def test_bti_full_size_cfg4(ops, ora):
    """14 classes at 64x224x192, B = 2: label map and critical map bit-exact vs the oracle."""
    from nextou_amd.harness import synthetic_batch, config_3d_fullres_nextou
    from nextou_amd.loss.bti_loss import BTI_Loss
    from nextou_amd.nnUNetTrainer.nnUNetTrainer_NexToU_BTI_Synapse import nnUNetTrainer_NexToU_BTI_Synapse as SYN
    cfg = config_3d_fullres_nextou()
    _, target = synthetic_batch(cfg, 1, 14, 2, DEV, blob_labels=True)
    labels = target[:, 0].to(torch.uint8).contiguous()
    loss = BTI_Loss(3, 26, [], SYN.exclusion_list, 1)
    crit = loss.critical_voxels_from_labels(labels)
    (lut_a, lut_c), = loss._luts_on(torch.device("cpu"))
    want = ora.bti_critical(labels.cpu(), lut_a, lut_c, 26, 1)
    assert torch.equal(crit.cpu(), want)
    frac = float(crit.float().mean())
    assert 0.0 < frac < 0.5       # blob labels: only interfaces between excluded organs are critical
    # idempotence-style property: a volume with one label has no critical voxel
    assert int(loss.critical_voxels_from_labels(torch.full_like(labels, 5)).sum()) == 0


# ---------------------------------------------------------------------------------------------
# modules / models through the HIP path vs the reference's goldens
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", list(mc.BLOCKS))
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_blocks_on_gpu(ops, name, mode):
    out, dx, g_out, g_dx, tape, entries = mc.run_block(name, mode, DEV, teacher_forced=True)
    assert tape.cursor == len(entries)
    assert float((out - g_out).abs().max()) <= 2e-5 * max(1.0, float(g_out.abs().max()))
    assert float((dx - g_dx).abs().max()) <= 5e-5 * max(1.0, float(g_dx.abs().max()))
    out2, dx2, _, _, tape2, _ = mc.run_block(name, mode, DEV, teacher_forced=False)
    for mine, ref in zip(tape2.entries, entries):
        if mine.dtype == torch.int32:
            assert knn_rows_equal_as_sets(mine.numpy(), ref.numpy()).mean() >= 0.999
        else:
            assert (mine == ref).float().mean() >= 0.999


@pytest.mark.parametrize("name,cfg,batch", [("g8_tiny2d", mc.TINY_2D, 2), ("g8_tiny3d", mc.TINY_3D, 1)])
def test_tiny_models_on_gpu_teacher_forced(ops, name, cfg, batch):
    """Protocol P-B on the MI355X (SURVEY.md §7 hard part 0): kNN ids and pool arg-max injected from
    the reference run, train-mode BN.  Gate: max |logit - logit_ref| <= max(1e-3, 2 x the reference's
    own self-noise floor) — the floor is the reference vs ITSELF under 1e-7 relative input noise
    (below one fp32 ulp), stored in the fixture by make_golden.py; with max |logit| ~ 25-35 it is
    ~1.2e-3 here, i.e. 1e-3 absolute is inside the noise of ANY fp32 re-implementation of the dense
    conv stages (MIOpen vs MKLDNN summation order).  The CPU test of the same fixture
    (test_modules_golden.py) holds the product logic to 1e-3 with identical conv arithmetic."""
    torch.backends.cudnn.benchmark = False
    outs, g, tape, entries, _ = mc.run_model(name, cfg, batch, DEV, teacher_forced=True)
    assert tape.cursor == len(entries)
    worst = 0.0
    for i, o in enumerate(outs):
        if "logits%d" % i in g.files:
            worst = max(worst, float((o - torch.from_numpy(g["logits%d" % i])).abs().max()))
        else:
            worst = max(worst, float((o.reshape(-1)[::97] - torch.from_numpy(g["logits%d_sample" % i])).abs().max()))
    floor, absmax = float(g["self_noise_floor"]), float(g["logit_absmax"])
    print("\n%s teacher-forced max |dlogit| = %.3e (reference self-noise floor %.3e, max |logit| %.1f -> %.1e relative)"
          % (name, worst, floor, absmax, worst / absmax))
    assert worst <= max(1e-3, 2 * floor)


def test_train_step_runs_and_is_repeatable(ops):
    """Harness train step on the GPU: forward is bit-identical across runs (no atomics in it)."""
    from nextou_amd.harness import config_3d_fullres_nextou, downsample_targets, synthetic_batch
    from nextou_amd.nnUNetTrainer.nnUNetTrainer_NexToU_BTI_Synapse import nnUNetTrainer_NexToU_BTI_Synapse
    cfg = config_3d_fullres_nextou(patch_size=(32, 128, 128), base=6, max_features=48, batch_size=2)
    torch.manual_seed(0)
    tr = nnUNetTrainer_NexToU_BTI_Synapse(cfg, 14, device=DEV, log=None).initialize()
    data, target = synthetic_batch(cfg, 1, 14, 2, DEV, blob_labels=True)
    with torch.no_grad():
        a = tr.network(data)
        b = tr.network(data)
    assert all(torch.equal(u, v) for u, v in zip(a, b))
    tgt = downsample_targets(target, a)
    l0 = float(tr.train_step(data, tgt))
    l1 = float(tr.train_step(data, tgt))
    assert np.isfinite(l0) and np.isfinite(l1)


# ---------------------------------------------------------------------------------------------
# BASELINE.json configurations at full size: MI355X forward vs the oracle-backed CPU forward
# ---------------------------------------------------------------------------------------------
def _full_size_forward_parity(cfg, classes, batch, ora):
    """Protocol P-B at full size.  The GPU run (HIP kernels, MIOpen convs) records its kNN ids and pool
    arg-max locations; the same network on the CPU (oracle kernels, MKLDNN convs) replays them.  The gate
    is max(1e-3, 4 x self-noise floor), the floor being the CPU network vs itself under 1e-7 relative
    input noise with the same decisions (He-initialised random weights put max |logit| at ~30)."""
    import copy
    from nextou_amd import graph_ops
    from nextou_amd.nnUNetTrainer.nnUNetTrainer_NexToU import nnUNetTrainer_NexToU
    torch.manual_seed(0)
    tr = nnUNetTrainer_NexToU(cfg, classes, log=None).initialize()
    cpu_net = tr.network.train()
    gpu_net = copy.deepcopy(cpu_net).to(DEV).train()
    x = _rand([batch, 1] + list(cfg.patch_size), 99)
    tape = graph_ops.IndexTape()
    with torch.no_grad(), graph_ops.index_tape(tape):
        gpu = [o.cpu() for o in gpu_net(x.to(DEV))]
    graph_ops.install_cpu_checker(ora)
    try:
        def replay(inp):
            with torch.no_grad(), graph_ops.index_tape(graph_ops.IndexTape(tape.entries)):
                return cpu_net(inp)
        cpu = replay(x)
        noisy = replay(x * (1 + 1e-7 * _rand(x.shape, 100)))
    finally:
        graph_ops.install_cpu_checker(None)
    worst = max(float((a - b).abs().max()) for a, b in zip(gpu, cpu))
    floor = max(float((a - b).abs().max()) for a, b in zip(cpu, noisy))
    absmax = max(float(o.abs().max()) for o in cpu)
    print("\nfull-size forward: max |dlogit| GPU vs CPU = %.3e, self-noise floor %.3e, max |logit| %.1f (%d graph decisions)"
          % (worst, floor, absmax, len(tape.entries)))
    # 4 x: MIOpen/CK implicit-GEMM vs oneDNN direct convolutions differ by a few ulp at EVERY layer, not
    # only at the input where the floor's 1e-7 perturbation enters (observed 1.8-2.0 x the floor)
    assert worst <= max(1e-3, 4 * floor)
    return worst, floor


@pytest.mark.timeout(1200)
def test_cfg1_2d_forward_parity(ops, ora):
    """BASELINE.json configs[0]: 2-D NexToU, 1x512x512, 7 stages, 3 classes."""
    from nextou_amd.harness import config_2d_nextou
    _full_size_forward_parity(config_2d_nextou(), 3, 1, ora)


@pytest.mark.timeout(2400)
def test_cfg2_3d_forward_parity(ops, ora):
    """BASELINE.json configs[1]: 3-D NexToU 64x224x192, base 33 / max 324, 14 classes (batch 1 forward)."""
    from nextou_amd.harness import config_3d_fullres_nextou
    _full_size_forward_parity(config_3d_fullres_nextou(), 14, 1, ora)


def test_bf16_autocast_keeps_graph_ops_in_fp32(ops):
    """cfg 5 regime: conv stages under bf16 autocast, kNN / MR aggregate always in fp32."""
    from nextou_amd.harness import config_3d_fullres_nextou, downsample_targets, synthetic_batch
    from nextou_amd.nnUNetTrainer.nnUNetTrainer_NexToU import nnUNetTrainer_NexToU
    cfg = config_3d_fullres_nextou(patch_size=(32, 128, 128), base=6, max_features=48, batch_size=2)
    torch.manual_seed(0)
    tr = nnUNetTrainer_NexToU(cfg, 14, device=DEV, log=None).initialize()
    data, target = synthetic_batch(cfg, 1, 14, 2, DEV)
    seen = []
    real = ops._HIP.knn_graph
    ops._HIP.knn_graph = staticmethod(lambda x, *a, **k: (seen.append(x.dtype), real(x, *a, **k))[1])
    try:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            outs = tr.network(data)
            loss = tr.loss([o.float() for o in outs], downsample_targets(target, outs))
        loss.backward()
    finally:
        del ops._HIP.knn_graph          # (an instance attribute would shadow the class's for every later test that patches the class)
    assert len(seen) == 14 and all(d == torch.float32 for d in seen)      # every graph build ran in fp32
    assert outs[0].dtype == torch.bfloat16                                 # ... while the conv stages ran in bf16
    assert torch.isfinite(loss) and all(torch.isfinite(o.float()).all() for o in outs)
    assert all(torch.isfinite(p.grad).all() for p in tr.network.parameters() if p.grad is not None)


def test_critical_cross_entropy_kernel(ops, ora):
    """Fused float64 CE x critical map: forward sums and fp32 logit gradients vs the reference op sequence."""
    for (B, L, sp) in ((2, 14, (12, 20, 18)), (3, 5, (33, 21)), (1, 3, (7,))):
        g = torch.Generator().manual_seed(sum(sp) + L)
        logits = torch.randn((B, L) + sp, generator=g) * 3
        target = torch.randint(0, L, (B,) + sp, generator=g, dtype=torch.uint8)
        critical = (torch.rand((B,) + sp, generator=g) < 0.3).to(torch.uint8)
        want = ora.bti_ce_fwd(logits, target, critical)
        xd = logits.to(DEV).requires_grad_(True)
        got = ops.critical_cross_entropy(xd, target.to(DEV), critical.to(DEV))
        assert got.dtype == torch.float64 and got.shape == (B,)
        np.testing.assert_allclose(got.detach().cpu().numpy(), want.numpy(), rtol=1e-12)
        w = torch.rand(B, generator=g, dtype=torch.float64) + 0.5
        (grad,) = torch.autograd.grad((got * w.to(DEV)).sum(), xd)
        ref = ora.bti_ce_bwd(logits, target, critical, w)
        np.testing.assert_allclose(grad.cpu().numpy(), ref.numpy(), rtol=1e-6, atol=1e-9)
        assert float(grad.cpu()[critical.unsqueeze(1).expand_as(logits) == 0].abs().max()) == 0.0
    # repeatable bit for bit (no atomics in the reduction)
    a = ops.critical_cross_entropy(xd, target.to(DEV), critical.to(DEV))
    b = ops.critical_cross_entropy(xd, target.to(DEV), critical.to(DEV))
    assert torch.equal(a, b)


# ---------------------------------------------------------------------------------------------
# K6  norm + LeakyReLU
def _norm_reference(x, w, b, rm, rv, training, momentum, eps, slope, instance, gy):
    """float64 torch reference of batch_norm | instance_norm -> leaky_relu and its gradients (CPU)."""
    xd = x.double().requires_grad_(True)
    wd = None if w is None else w.double().requires_grad_(True)
    bd = None if b is None else b.double().requires_grad_(True)
    rmd, rvd = (None, None) if rm is None else (rm.double().clone(), rv.double().clone())
    if instance:
        z = F.instance_norm(xd, None, None, wd, bd, True, momentum, eps)
    else:
        z = F.batch_norm(xd, rmd, rvd, wd, bd, training, momentum, eps)
    y = F.leaky_relu(z, slope) if slope != 1.0 else z
    grads = torch.autograd.grad(y, [t for t in (xd, wd, bd) if t is not None], gy.double())
    return y.detach(), grads, rmd, rvd


@pytest.mark.parametrize("shape,instance,training,slope,affine", [
    ((2, 33, 8, 28, 24), False, True, 0.01, True),     # 16-byte path, column tiles
    ((64, 12, 4, 7, 6), False, True, 0.01, True),      # window batch: short rows, row tiles
    ((3, 5, 3, 5), False, True, 0.01, True),           # S = 15: scalar path
    ((2, 7, 1031), False, True, 1.0, True),            # prime S, bare norm
    ((2, 24, 6, 10, 12), False, False, 0.01, True),    # inference: running statistics
    ((2, 24, 6, 10, 12), False, True, 0.2, False),     # no affine parameters
    ((2, 18, 344), True, True, 0.01, True),            # instance norm (Pool-GNN BasicConv)
    ((1, 6, 5, 9), True, True, 1.0, False),
])
def test_norm_act_vs_float64_reference(ops, shape, instance, training, slope, affine):
    """K6 against the float64 op sequence.  Bars: outputs and input gradients within 4 fp32 ulp of the output /
    gradient scale (the stock fp32 kernels are held to the same bar beside it), parameter gradients rtol 2e-5,
    running statistics rtol 1e-6; bit-identical when repeated."""
    g = torch.Generator().manual_seed(len(shape) * 1000 + shape[1])
    C = shape[1]
    x = torch.randn(shape, generator=g) * 1.7 + 0.4
    w = (torch.rand(C, generator=g) + 0.5) if affine else None
    b = (torch.randn(C, generator=g) * 0.3) if affine else None
    rm, rv = (None, None) if instance else (torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5)
    gy = torch.randn(shape, generator=g)
    want_y, want_g, want_rm, want_rv = _norm_reference(x, w, b, rm, rv, training, 0.1, 1e-5, slope, instance, gy)

    def run():
        xd = x.to(DEV).requires_grad_(True)
        wd = None if w is None else w.to(DEV).requires_grad_(True)
        bd = None if b is None else b.to(DEV).requires_grad_(True)
        rmd, rvd = (None, None) if rm is None else (rm.to(DEV), rv.to(DEV))
        y = ops.norm_act(xd, wd, bd, rmd, rvd, training, 0.1, 1e-5, slope, instance=instance)
        grads = torch.autograd.grad(y, [t for t in (xd, wd, bd) if t is not None], gy.to(DEV))
        return y.detach(), grads, rmd, rvd

    y, grads, rmd, rvd = run()
    ulp = 2.0 ** -23
    assert float((y.cpu().double() - want_y).abs().max()) <= 4 * ulp * float(want_y.abs().max())
    assert float((grads[0].cpu().double() - want_g[0]).abs().max()) <= 8 * ulp * max(float(want_g[0].abs().max()), 1.0)
    for got, want in zip(grads[1:], want_g[1:]):
        np.testing.assert_allclose(got.cpu().double().numpy(), want.numpy(), rtol=2e-5, atol=2e-5 * float(want.abs().max()))
    if rm is not None:
        np.testing.assert_allclose(rmd.cpu().double().numpy(), want_rm.numpy(), rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(rvd.cpu().double().numpy(), want_rv.numpy(), rtol=1e-6, atol=1e-7)
    y2, grads2, _, _ = run()
    assert torch.equal(y, y2) and all(torch.equal(a, b_) for a, b_ in zip(grads, grads2))


def test_norm_act_bf16_and_modules(ops):
    """bf16 tensors (cfg 5's autocast conv stages) and the fused module classes against the stock modules on the GPU."""
    from nextou_amd.network_architecture.norm_act import fuse_norm_act
    g = torch.Generator().manual_seed(5)
    x = (torch.randn((2, 16, 4, 12, 16), generator=g) * 2).to(DEV)
    gy = torch.randn((2, 16, 4, 12, 16), generator=g).to(DEV)
    w, b = (torch.rand(16, generator=g) + 0.5).to(DEV), torch.randn(16, generator=g).to(DEV)
    xb = x.bfloat16().requires_grad_(True)
    yb = ops.norm_act(xb, w, b, None, None, True, 0.1, 1e-5, 0.01)
    (gb,) = torch.autograd.grad(yb, xb, gy.bfloat16())
    assert yb.dtype == torch.bfloat16 and gb.dtype == torch.bfloat16
    xf = xb.detach().float().requires_grad_(True)      # same (bf16-rounded) inputs in fp32
    yf = F.leaky_relu(F.batch_norm(xf, None, None, w, b, True, 0.1, 1e-5), 0.01)
    (gf,) = torch.autograd.grad(yf, xf, gy.bfloat16().float())
    assert float((yb.float() - yf).abs().max().detach()) <= 2.0 ** -8 * float(yf.abs().max().detach())  # one bf16 rounding
    assert float((gb.float() - gf).abs().max()) <= 2.0 ** -7 * float(gf.abs().max())

    for make in (lambda: torch.nn.Sequential(torch.nn.Conv3d(4, 12, 3, padding=1), torch.nn.BatchNorm3d(12),
                                             torch.nn.LeakyReLU(0.01, inplace=True)),
                 lambda: torch.nn.Sequential(torch.nn.Conv2d(4, 12, 1), torch.nn.InstanceNorm2d(12, affine=True),
                                             torch.nn.LeakyReLU(0.01)),
                 lambda: torch.nn.Sequential(torch.nn.Conv3d(4, 12, 1), torch.nn.BatchNorm3d(12))):
        torch.manual_seed(3)
        stock = make().to(DEV)
        torch.manual_seed(3)
        fused = make().to(DEV)
        assert fuse_norm_act(fused) == 1
        assert list(fused.state_dict()) == list(stock.state_dict())
        dims = 3 if isinstance(stock[0], torch.nn.Conv3d) else 2
        inp = torch.randn((3, 4) + (6, 10, 8)[:dims], device=DEV)
        for mode in ("train", "eval"):
            getattr(stock, mode)(), getattr(fused, mode)()
            a, b_ = stock(inp), fused(inp)
            assert float((a - b_).abs().max()) <= 1e-5 * float(a.abs().max())
            ga = torch.autograd.grad(a.square().sum(), list(stock.parameters()))
            gb_ = torch.autograd.grad(b_.square().sum(), list(fused.parameters()))
            scale = max(float(u.abs().max()) for u in ga)   # the conv bias gradient ahead of a norm is pure round-off
            for u, v in zip(ga, gb_):
                assert float((u - v).abs().max()) <= 2e-5 * scale
        for (k1, v1), (k2, v2) in zip(stock.state_dict().items(), fused.state_dict().items()):
            assert k1 == k2 and float((v1.double() - v2.double()).abs().max()) <= 1e-5, k1


def test_norm_act_full_size_cfg2(ops):
    """The largest norm call of cfg 2 (2 x 33 x 64 x 224 x 192): size-independent properties — per-channel mean 0 /
    variance 1 of the normalised tensor, exact LeakyReLU sign structure, gradient orthogonality
    (sum gx = 0 and sum gx * x_hat = 0 per channel, the two projections batch norm's backward removes)."""
    g = torch.Generator(device=DEV).manual_seed(11)
    x = torch.randn((2, 33, 64, 224, 192), generator=g, device=DEV) * 3 + 1
    x.requires_grad_(True)
    y = ops.norm_act(x, None, None, None, None, True, 0.1, 1e-5, 1.0)
    m = y.double().mean(dim=(0, 2, 3, 4))
    v = y.double().var(dim=(0, 2, 3, 4), unbiased=False)
    assert float(m.abs().max()) < 1e-6 and float((v - 1).abs().max()) < 1e-5
    ya = ops.norm_act(x, None, None, None, None, True, 0.1, 1e-5, 0.01)
    assert torch.equal(ya, torch.where(y > 0, y, y * 0.01))
    gy = torch.randn(x.shape, generator=g, device=DEV)
    (gx,) = torch.autograd.grad(ya, x, gy)
    s1 = gx.double().sum(dim=(0, 2, 3, 4))
    s2 = (gx.double() * y.double()).sum(dim=(0, 2, 3, 4))
    n = x[:, 0].numel()
    assert float(s1.abs().max()) / n < 1e-7 and float(s2.abs().max()) / n < 1e-7


# ---------------------------------------------------------------------------------------------
# inference
def test_sliding_window_inference_on_gpu(ops):
    """Batched tiles + mirror copies give the same logits as one forward per copy; eval mode, deep supervision off."""
    from nextou_amd.inference import predict_sliding_window
    torch.manual_seed(1)
    net = mc.build_model(mc.TINY_3D).to(DEV)
    for m in net.modules():                      # non-trivial running statistics
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    image = torch.randn(1, 40, 160, 128, device=DEV)
    one = predict_sliding_window(net, image, mc.TINY_3D["patch"], 0.5, True, (0, 1, 2), batch_size=1)
    many = predict_sliding_window(net, image, mc.TINY_3D["patch"], 0.5, True, (0, 1, 2), batch_size=8)
    assert one.shape == (mc.TINY_3D["classes"], 40, 160, 128) and bool(torch.isfinite(one).all())
    # MIOpen picks its algorithms per batch size, and the network is discontinuous in its kNN decisions (SURVEY §7
    # hard part 0), so batched == sequential only up to round-off amplified by a few flipped neighbours:
    scale = float(one.abs().max())
    assert float((one - many).abs().mean()) <= 1e-4 * scale, float((one - many).abs().mean()) / scale
    assert float((one.argmax(0) == many.argmax(0)).float().mean()) >= 0.995
    assert net.training and net.decoder.deep_supervision is True


@pytest.mark.parametrize("training", [True, False])
def test_norm_act_folds_the_conv_bias(ops, training):
    """norm_act(x, pre_bias=b) == norm(x + b): outputs, running statistics, and every gradient incl. d/db
    (identically 0 under batch statistics, w * invstd * sum(dz) with running statistics) vs float64."""
    g = torch.Generator().manual_seed(21)
    shape, C = (3, 10, 6, 8, 12), 10
    x = torch.randn(shape, generator=g)
    pb = torch.randn(C, generator=g) * 2
    w, b = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    rm, rv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    gy = torch.randn(shape, generator=g)
    xd = x.double().requires_grad_(True)
    pbd, wd, bd = (t.double().requires_grad_(True) for t in (pb, w, b))
    rmd, rvd = rm.double().clone(), rv.double().clone()
    z = F.batch_norm(xd + pbd.view(1, -1, 1, 1, 1), rmd, rvd, wd, bd, training, 0.1, 1e-5)
    want = F.leaky_relu(z, 0.01)
    wg = torch.autograd.grad(want, (xd, wd, bd, pbd), gy.double())

    xg = x.to(DEV).requires_grad_(True)
    pbg, wgp, bgp = (t.to(DEV).requires_grad_(True) for t in (pb, w, b))
    rmg, rvg = rm.to(DEV), rv.to(DEV)
    y = ops.norm_act(xg, wgp, bgp, rmg, rvg, training, 0.1, 1e-5, 0.01, pre_bias=pbg)
    got = torch.autograd.grad(y, (xg, wgp, bgp, pbg), gy.to(DEV))
    ulp = 2.0 ** -23
    assert float((y.detach().cpu().double() - want.detach()).abs().max()) <= 8 * ulp * float(want.abs().max())
    for a, r in zip(got, wg):
        assert float((a.cpu().double() - r).abs().max()) <= 3e-5 * max(float(r.abs().max()), 1.0)
    if training:
        assert float(got[3].abs().max()) == 0.0
    np.testing.assert_allclose(rmg.cpu().double().numpy(), rmd.numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(rvg.cpu().double().numpy(), rvd.numpy(), rtol=1e-6, atol=1e-7)


# ---------------------------------------------------------------------------------------------
# channels-last stages: K6 NDHWC kernels, per-channel sums, own-bias-gradient convolutions, layout policy
def _cl(t):
    return t.contiguous(memory_format=torch.channels_last_3d if t.dim() == 5 else torch.channels_last)


@pytest.mark.parametrize("shape,training,slope,dtype", [
    ((2, 33, 8, 28, 24), True, 0.01, torch.float32),    # 7 rows x 33 channels per workgroup pass, 16-byte accesses
    ((2, 66, 4, 6, 10), True, 0.01, torch.float32),
    ((3, 14, 5, 7, 9), True, 1.0, torch.float32),       # 13230 elements: not a multiple of 4 -> scalar path
    ((2, 200, 3, 4, 4), True, 0.2, torch.float32),      # one row per pass, 56 idle lanes
    ((2, 12, 16, 20), True, 0.01, torch.float32),       # 2-D channels_last
    ((2, 33, 8, 28, 24), False, 0.01, torch.float32),   # inference
    ((2, 24, 4, 12, 16), True, 0.01, torch.bfloat16),
])
def test_norm_act_channels_last(ops, shape, training, slope, dtype):
    """The NDHWC kernels give what the NCDHW kernels give on the same logical tensor (both are held to the float64
    reference elsewhere): outputs / gradients within a few ulp, statistics within 1e-6, layout preserved."""
    g = torch.Generator().manual_seed(shape[1])
    C = shape[1]
    x = (torch.randn(shape, generator=g) * 1.5 + 0.3).to(dtype)
    w, b = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    pb = torch.randn(C, generator=g)
    rm, rv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    gy = torch.randn(shape, generator=g).to(dtype)

    def run(channels_last):
        xd = x.to(DEV)
        xd = (_cl(xd) if channels_last else xd).requires_grad_(True)
        wd, bd, pbd = (t.to(DEV).requires_grad_(True) for t in (w, b, pb))
        rmd, rvd = rm.to(DEV), rv.to(DEV)
        y = ops.norm_act(xd, wd, bd, rmd, rvd, training, 0.1, 1e-5, slope, pre_bias=pbd)
        gyd = gy.to(DEV)
        grads = torch.autograd.grad(y, (xd, wd, bd, pbd), _cl(gyd) if channels_last else gyd)
        return y.detach(), grads, rmd, rvd

    y0, g0, rm0, rv0 = run(False)
    y1, g1, rm1, rv1 = run(True)
    assert ops._dense_channels_last(y1) is not None and ops._dense_channels_last(g1[0]) is not None
    tol = 2.0 ** -7 if dtype == torch.bfloat16 else 4 * 2.0 ** -23
    assert float((y0.float() - y1.float()).abs().max()) <= tol * float(y0.float().abs().max())
    assert float((g0[0].float() - g1[0].float()).abs().max()) <= 2 * tol * max(float(g0[0].float().abs().max()), 1.0)
    for a, c in zip(g0[1:], g1[1:]):
        np.testing.assert_allclose(c.cpu().numpy(), a.cpu().numpy(), rtol=1e-5, atol=1e-5 * max(float(a.abs().max()), 1e-3))
    np.testing.assert_allclose(rm1.cpu().numpy(), rm0.cpu().numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(rv1.cpu().numpy(), rv0.cpu().numpy(), rtol=1e-6, atol=1e-7)
    y2, g2, _, _ = run(True)
    assert torch.equal(y1, y2) and torch.equal(g1[0], g2[0])


def test_channel_sum_and_own_bias_convolutions(ops):
    from nextou_amd.network_architecture.layout import to_channels_last
    g = torch.Generator().manual_seed(9)
    for shape in ((2, 33, 6, 10, 12), (3, 14, 5, 7, 9), (2, 300, 2, 3, 4), (4, 6, 11)):
        x = torch.randn(shape, generator=g)
        want = x.double().sum(dim=[0] + list(range(2, x.dim())))
        got = ops._HIP.channel_sum(x.to(DEV))
        np.testing.assert_allclose(got.cpu().double().numpy(), want.numpy(), rtol=1e-6, atol=1e-5)
        if x.dim() >= 4 and shape[1] <= 256:
            got = ops._HIP.channel_sum(_cl(x.to(DEV)), channels_last=True)
            np.testing.assert_allclose(got.cpu().double().numpy(), want.numpy(), rtol=1e-6, atol=1e-5)
    # a single-channel image re-strided as channels-last makes the first convolution produce NDHWC
    img = torch.randn(2, 1, 6, 20, 16, generator=g).to(DEV)
    conv = torch.nn.Conv3d(1, 8, (1, 3, 3), padding=(0, 1, 1)).to(DEV)
    out = conv(to_channels_last(img))
    assert ops._dense_channels_last(out) is torch.channels_last_3d
    assert float((out - conv(img)).abs().max()) <= 1e-5 * float(out.abs().max())
    # convolutions with their own bias gradient == stock autograd, both layouts, plain and transposed
    for transposed in (False, True):
        for cl in (False, True):
            torch.manual_seed(4)
            m = (torch.nn.ConvTranspose3d(12, 6, (1, 2, 2), (1, 2, 2)) if transposed else torch.nn.Conv3d(12, 5, 1)).to(DEV)
            x = torch.randn(2, 12, 4, 6, 8, generator=g).to(DEV)
            x = (_cl(x) if cl else x).requires_grad_(True)
            y_ref = m(x)
            gy = torch.randn(y_ref.shape, generator=g).to(DEV)
            ref = torch.autograd.grad(y_ref, (x, m.weight, m.bias), gy)
            out_pad = (0, 0, 0)
            y = ops.conv_own_bias_grad(x, m.weight, m.bias, m.stride, m.padding, m.dilation, transposed, out_pad, 1)
            got = torch.autograd.grad(y, (x, m.weight, m.bias), _cl(gy) if cl else gy)
            assert float((y - y_ref).abs().max()) <= 1e-5 * float(y_ref.abs().max())
            for a, r in zip(got, ref):
                assert float((a - r).abs().max()) <= 2e-5 * max(float(r.abs().max()), 1.0)


def test_channels_last_policy_changes_layout_not_results(ops, monkeypatch):
    """Stage policy 'auto' (every stage NDHWC, graph stages through the fused window / pool kernels; plain stages with
    internally padded channels) vs 'none' (NCDHW, no padding) on the tiny 3-D model: same weights, same recorded kNN /
    arg-max decisions -> same logits and gradients up to conv round-off; the skips really are NDHWC / padded."""
    from nextou_amd.graph_ops import IndexTape, index_tape
    x = torch.randn(1, 1, 32, 128, 128, generator=torch.Generator().manual_seed(3)).to(DEV)
    results = {}
    tape = IndexTape()
    for policy in ("auto", "none"):
        monkeypatch.setenv("NEXTOU_CHANNELS_LAST_STAGES", policy)
        torch.manual_seed(0)
        net = mc.build_model(mc.TINY_3D).to(DEV)
        assert net.encoder.channels_last_stages == (frozenset(range(6)) if policy == "auto" else frozenset())
        from nextou_amd.network_architecture.channel_pad import force_padding
        with force_padding(None if policy == "auto" else False):
            skips = net.encoder(x)
        assert (ops._dense_channels_last(skips[0]) is not None) == (policy == "auto")
        assert (ops._dense_channels_last(skips[2]) is not None) == (policy == "auto")   # graph stages follow the policy
        assert skips[0].shape[1] == (8 if policy == "auto" else 6) and skips[2].shape[1] == 24   # 6 -> 8 inside stage 0 only
        if policy == "auto":
            assert float(skips[0][:, 6:].abs().max()) == 0.0 and float(skips[1][:, 12:].abs().max()) == 0.0
        if policy == "none":
            tape = IndexTape(tape.entries)                          # replay the decisions of the first run
        with index_tape(tape), force_padding(None if policy == "auto" else False):
            outs = net(x)
            loss = sum(o.square().mean() for o in outs)
            grads = torch.autograd.grad(loss, [p for p in net.parameters() if p.requires_grad], allow_unused=True)
        results[policy] = ([o.detach() for o in outs], grads)
    (oa, ga), (on, gn) = results["auto"], results["none"]
    for a, b in zip(oa, on):
        assert float((a - b).abs().max()) <= 2e-4 * float(b.abs().max())
    scale = max(float(t.abs().max()) for t in gn if t is not None)
    for a, b in zip(ga, gn):
        assert (a is None) == (b is None)
        if a is not None:
            assert float((a - b).abs().max()) <= 5e-3 * scale


def test_layout_policy_under_reduced_precision(ops, monkeypatch):
    """Round 1 fenced the channels-last policy off under bf16 autocast (NDHWC at 33 / 66 channels: 309 vs 185 ms on cfg 2).
    With the internal channel padding the same path is the fast one (189 -> 111 ms, profiles/r02_bf16_ndhwc_trace.md), so
    the policy now applies under reduced precision whenever the padding does; NEXTOU_REDUCED_PRECISION_LAYOUT=ncdhw and
    NEXTOU_PAD_CHANNELS=0 both restore NCDHW.  Encoder and decoder take the decision from the same predicate, so a
    mixed-precision forward stays consistent."""
    from nextou_amd.network_architecture.layout import layout_policy_applies, runs_in_fp32
    torch.manual_seed(0)
    net = mc.build_model(mc.TINY_3D).to(DEV)
    assert net.encoder.channels_last_stages == frozenset(range(6))
    x = torch.randn(1, 1, 32, 128, 128, device=DEV)
    assert runs_in_fp32(x) and not runs_in_fp32(x.bfloat16()) and not runs_in_fp32(x.half())
    assert ops._dense_channels_last(net.encoder(x)[0]) is torch.channels_last_3d
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert not runs_in_fp32(x) and layout_policy_applies(x)
        skips = net.encoder(x)
        assert all(ops._dense_channels_last(s_) is torch.channels_last_3d for s_ in skips[:5])
        assert skips[0].dtype == torch.bfloat16 and skips[0].shape[1] == 8 and skips[1].shape[1] == 16     # padded
        assert float(skips[0][:, 6:].float().abs().max()) == 0.0
        outs = net(x)
    assert all(bool(torch.isfinite(o.float()).all()) for o in outs)
    loss = sum(o.float().square().mean() for o in outs)
    grads = torch.autograd.grad(loss, [p for p in net.parameters() if p.requires_grad], allow_unused=True)
    assert all(bool(torch.isfinite(g_).all()) for g_ in grads if g_ is not None)
    for env, val in (("NEXTOU_REDUCED_PRECISION_LAYOUT", "ncdhw"), ("NEXTOU_PAD_CHANNELS", "0")):
        monkeypatch.setenv(env, val)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            assert not layout_policy_applies(x)
            skips = net.encoder(x)
            assert ops._dense_channels_last(skips[0]) is None and skips[0].shape[1] == 6
            outs2 = net(x)
        assert all(bool(torch.isfinite(o.float()).all()) for o in outs2)
        monkeypatch.delenv(env)
    assert ops._dense_channels_last(net.encoder(x)[0]) is torch.channels_last_3d        # fp32: NDHWC as before
