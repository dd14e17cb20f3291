"""No kernel of libnextou_hip.so touches memory outside its operands.

Every input, every output and every workspace of a launch lives in a guard-page buffer (tools/guard_alloc.py: HIP virtual-memory
API, the pages before and after the mapping are unmapped).  First pass: every buffer ENDS on the guard (a read or write past the last
element faults); second pass: every buffer STARTS on the guard (a negative offset faults).  A fault is `Memory access fault by GPU`
and kills the process — the test then fails as a crashed run, exactly like the library convolution of
tools/conv_bwd_fault_repro.py did under the N > 1 step (profiles/r05_n_gt_1.md).  The results of the guarded launches are
compared with the same launch on caching-allocator tensors: bit-identical for the kernels with a fixed summation order, to
round-off for those that accumulate with float atomics.

The wrappers under test are the ctypes bindings `graph_ops._HIP.*` the autograd functions call — they allocate outputs and
workspaces with torch.empty / torch.empty_like, which GuardScope.patched_outputs() redirects into guard-page buffers.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.fixture(scope="module")
def hip():
    from nextou_amd import _lib, graph_ops
    _lib.lib()
    assert "libnextou_hip.so" in open("/proc/self/maps").read(), "HIP extension not loaded into this process"
    return graph_ops._HIP


def _cl(t):
    return t.contiguous(memory_format={4: torch.channels_last, 5: torch.channels_last_3d}[t.dim()])


def _flat(res):
    if res is None:
        return []
    if isinstance(res, torch.Tensor):
        return [res]
    out = []
    for r in res:
        out += _flat(r)
    return out


def _guarded(launch, inputs, exact=True, rtol=2e-5, inplace=()):
    """launch(*inputs) on plain tensors, then with every tensor in guard-page buffers (end-flush, start-flush); compares.
    ``inplace``: indices of inputs the launch updates in place (fresh copies per pass, compared too)."""
    from tools.guard_alloc import GuardScope

    def fresh():
        return [None if t is None else t.clone(memory_format=torch.preserve_format) if isinstance(t, torch.Tensor) else t for t in inputs]
    ins = fresh()
    want = _flat(launch(*ins)) + [ins[i] for i in inplace]
    torch.cuda.synchronize()
    for flush in ("end", "start"):
        scope = GuardScope(flush=flush, align=16)
        try:
            gin = [scope.like(t) if isinstance(t, torch.Tensor) else t for t in fresh()]
            with scope.patched_outputs():
                res = launch(*gin)
            torch.cuda.synchronize()
            got = _flat(res) + [gin[i] for i in inplace]
            assert len(got) == len(want)
            for a, e in zip(got, want):
                assert a.shape == e.shape and a.dtype == e.dtype and a.stride() == e.stride()
                if exact or not a.dtype.is_floating_point:
                    assert torch.equal(a, e), "guarded launch (%s-flush) differs from the plain one" % flush
                else:
                    assert float((a.double() - e.double()).abs().max()) <= rtol * (float(e.double().abs().max()) + 1e-30)
        finally:
            torch.cuda.synchronize()
            scope.close()


def _gen(seed):
    return torch.Generator().manual_seed(seed)


# ---------------------------------------------------------------- K1 kNN graph
@pytest.mark.parametrize("B,C,N,M,K,relpos", [
    (2, 24, 64, None, 9, False),          # self graph, one tile
    (1, 33, 200, None, 18, True),         # ragged N, odd channel count, relative positions
    (2, 48, 343, 49, 16, True),           # pooled candidates (reduce_ratio graph)
    (1, 16, 1100, None, 32, False),       # several candidate slabs + merge
    (2, 324, 168, None, 28, True),        # cfg 2 stage-5 Swin window count
    (1, 8, 4104, None, 9, False),         # M > 4096: the two-level path
])
def test_knn_graph(hip, B, C, N, M, K, relpos):
    g = _gen(N + K)
    x = torch.randn((B, C, N), generator=g).to(DEV)
    y = None if M is None else torch.randn((B, C, M), generator=g).to(DEV)
    rp = (torch.randn((N, N if M is None else M), generator=g) * 0.1).to(DEV) if relpos else None
    from nextou_amd import _lib
    _guarded(lambda x, y, rp: hip.knn_graph(x, y, rp, K, _lib.KNN_AUTO, True), [x, y, rp])
    _guarded(lambda x, y: hip.pairwise_distance(x, y, 0, min(N, 40)), [x, y])
    idx = hip.knn_graph(x, y, rp, K, _lib.KNN_AUTO, True)
    _guarded(lambda idx: hip.edge_index(idx, 2), [idx])


# ---------------------------------------------------------------- K2 aggregation, gather
@pytest.mark.parametrize("B,C,N,M,K,step", [
    (2, 24, 64, None, 9, 1), (1, 33, 200, None, 9, 2), (2, 48, 343, 49, 16, 1), (1, 132, 1000, 125, 8, 2), (2, 66, 512, None, 9, 1),
])
def test_mr_aggregate_and_gather(hip, B, C, N, M, K, step):
    g = _gen(N + C)
    x = torch.randn((B, C, N), generator=g).to(DEV)
    y = None if M is None else torch.randn((B, C, M), generator=g).to(DEV)
    m = N if M is None else M
    idx = torch.randint(0, m, (B, N, K * step), generator=g, dtype=torch.int32).to(DEV)
    _guarded(lambda x, y, idx: hip.mr_fwd(x, y, idx, None, K, step, want_arg=True), [x, y, idx])
    gout = torch.randn((B, 2 * C, N), generator=g).to(DEV)
    _guarded(lambda go, x, y, idx: hip.mr_bwd(go, x, y, idx, None, K, step), [gout, x, y, idx], exact=False)
    _, arg = hip.mr_fwd(x, y, idx, None, K, step, want_arg=True)
    if arg is not None:
        _guarded(lambda go, arg: hip.mr_bwd_arg(go, arg, m, M is not None), [gout, arg], exact=False)
        if M is None and hip.mr_bwd_wants_idx(B, C, N, K):
            _guarded(lambda go, arg, idx: hip.mr_bwd_arg_idx(go, arg, idx, K, step), [gout, arg, idx])
    center = torch.arange(N, dtype=torch.int32).repeat(B, 1).unsqueeze(-1).expand(B, N, K * step).contiguous().to(DEV)
    _guarded(lambda x, y, idx, c: hip.mr_fwd(x, y, idx, c, K, step), [x, None, idx % N, center])
    src = x if y is None else y
    i64 = idx[..., ::step].contiguous()           # (B, N, K) int32
    _guarded(lambda s, i: hip.gather_fwd(s, i), [src, i64])
    go4 = torch.randn((B, C, N, K), generator=g).to(DEV)
    _guarded(lambda go, i: hip.gather_bwd(go, i, m), [go4, i64], exact=False)


# ---------------------------------------------------------------- K3 / K4 windows, pooling, cells, depth taps
@pytest.mark.parametrize("B,C,sp,win,shift,pool", [
    (2, 24, (4, 8, 8), (2, 4, 4), (1, 2, 2), (2, 2, 2)),
    (1, 33, (3, 14, 12), (3, 7, 6), (0, 0, 0), (1, 2, 2)),
    (2, 16, (16, 24), (8, 8), (4, 4), (2, 4)),          # (a pool has at most 8 cells)
    (1, 132, (4, 14, 12), (4, 7, 6), (2, 3, 3), (2, 2, 2)),
])
def test_windows_pools_cells(hip, B, C, sp, win, shift, pool):
    g = _gen(C + sp[-1])
    x = _cl(torch.randn((B, C) + sp, generator=g).to(DEV))
    _guarded(lambda x: hip.window_gather(x, win, shift), [x])
    rows = hip.window_gather(x, win, shift)
    _guarded(lambda r, res: hip.window_scatter(r, res, sp, win, shift), [rows, x])
    _guarded(lambda r: hip.window_scatter(r, None, sp, win, shift), [rows])
    _guarded(lambda x: hip.pool_rows(x, pool), [x])
    vals, cell = hip.pool_rows(x, pool)
    x2 = _cl(torch.randn((B, 2 * C) + sp, generator=g).to(DEV))
    _guarded(lambda x2, cell: hip.cell_gather(x2, cell, pool), [x2, cell])
    src = torch.randn((B, 2 * C, vals.shape[2]), generator=g).to(DEV)
    _guarded(lambda s, cell: hip.cell_scatter(s, cell, sp, pool), [src, cell])
    if len(sp) == 3 and C % 4 == 0:
        _guarded(lambda x: hip.depth_unroll(x), [x])


# ---------------------------------------------------------------- K5 loss kernels
@pytest.mark.parametrize("B,L,sp,layout", [
    (2, 14, (4, 16, 16), "cl"), (2, 14, (4, 16, 16), "nchw"), (1, 3, (5, 9, 7), "nchw"), (2, 4, (33, 31), "cl"), (1, 17, (3, 5, 7), "cl"),
])
def test_loss_kernels(hip, B, L, sp, layout):
    g = _gen(L + sp[-1])
    from nextou_amd.graph_ops import _logits_in_place
    logits = torch.randn((B, L) + sp, generator=g).to(DEV)
    logits = _logits_in_place(_cl(logits) if layout == "cl" else logits)
    t64 = torch.randint(0, L, (B,) + sp, generator=g).to(DEV)
    t8 = t64.to(torch.uint8)
    _guarded(lambda lg: hip.argmax_labels(lg), [logits])
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    for t in (t64, t8, t64.float()):
        _guarded(lambda t, f: hip.labels_u8(t, L, f), [t, flag], inplace=(1,))
    _guarded(lambda lg, t: hip.ce_mean_fwd(lg, t, -100), [logits, t64])
    scale = torch.full((1,), 0.37, device=DEV)
    _guarded(lambda lg, t, s: hip.ce_mean_bwd(lg, t, s, -100), [logits, t64, scale])
    mask = (torch.rand((B,) + sp, generator=g) > 0.3).to(torch.uint8).to(DEV)
    for m in (None, mask):
        _guarded(lambda lg, t, m: hip.dice_stats_fwd(lg, t, m), [logits, t8, m])
        gi = torch.randn((B, L), generator=g, dtype=torch.float64).to(DEV)
        gp = torch.randn((B, L), generator=g, dtype=torch.float64).to(DEV)
        _guarded(lambda lg, t, m, gi, gp: hip.dice_stats_bwd(lg, t, m, gi, gp), [logits, t8, m, gi, gp])
    crit = (torch.rand((B,) + sp, generator=g) > 0.8).to(torch.uint8).to(DEV)
    _guarded(lambda lg, t, c: hip.bti_ce_fwd(lg, t, c), [logits, t8, crit])
    sc = torch.full((B,), 0.5, dtype=torch.float64, device=DEV)       # upstream gradient per sample
    _guarded(lambda lg, t, c, s: hip.bti_ce_bwd(lg, t, c, s), [logits, t8, crit, sc])


@pytest.mark.parametrize("B,sp,conn,thick", [(2, (6, 20, 24), 26, 3), (1, (1, 33, 31), 8, 3), (2, (17, 9, 40), 6, 1), (1, (32, 56, 48), 26, 5)])
def test_bti_critical_map(hip, B, sp, conn, thick):
    from nextou_amd.loss.bti_loss import BTI_Loss
    g = _gen(sp[-1])
    lab = torch.randint(0, 3, (B,) + sp, generator=g).to(torch.uint8)
    lab = (lab if sp[0] > 1 else lab[:, 0]).contiguous().to(DEV)
    dim = 2 if sp[0] == 1 else 3
    a, c = BTI_Loss(dim=dim, connectivity=conn, inclusion=[[1, 2]], exclusion=[[1, 0]], min_thick=thick)._luts_on(DEV)[0]
    _guarded(lambda lab, a, c: hip.bti_critical(lab, a, c, conn, thick), [lab, a, c])


# ---------------------------------------------------------------- K6 normalisation + activation
@pytest.mark.parametrize("B,C,sp,layout,period", [
    (2, 24, (4, 8, 8), "cl", 0), (2, 24, (4, 8, 8), "nchw", 0), (2, 33, (3, 5, 7), "cl", 0), (2, 8, (16, 16), "nchw", 8),
    (1, 132, (2, 7, 6), "cl", 0), (3, 16, (5, 9), "nchw", 16),
])
def test_norm_act(hip, B, C, sp, layout, period):
    g = _gen(C + B)
    x = torch.randn((B, C) + sp, generator=g).to(DEV)
    cl = layout == "cl"
    x = _cl(x) if cl else x
    rows = B * C if period else C              # instance norm: one normalised row per (sample, channel); parameters repeat with `period`
    w = (torch.rand(C, generator=g) + 0.5).to(DEV)
    b = torch.randn(C, generator=g).to(DEV)
    xv = x if cl else x.reshape(1, B * C, -1) if period else x.reshape(B, C, -1)
    rm = None if period else torch.zeros(C, device=DEV)
    rv = None if period else torch.ones(C, device=DEV)
    assert not (cl and period)        # (instance statistics on channels-last volumes go through the rows kernels, tested below)
    _guarded(lambda x, w, b, rm, rv: hip.norm_act_fwd(x, w, b, rm, rv, True, 0.1, 1e-5, 0.01, period, None, cl), [xv, w, b, rm, rv],
             inplace=() if period else (3, 4))
    y, m, i = hip.norm_act_fwd(xv, w, b, rm, rv, True, 0.1, 1e-5, 0.01, period, None, cl)
    assert m.numel() == rows
    gy = torch.randn(xv.shape, generator=g).to(DEV)
    gy = _cl(gy) if cl else gy
    _guarded(lambda x, gy, w, b, m, i: hip.norm_act_bwd(x, gy, w, b, m, i, True, 0.01, period, 0.0, cl), [xv, gy, w, b, m, i])
    _guarded(lambda x: hip.channel_sum(x, cl), [xv])
    if not period:
        _guarded(lambda x, w, b, rm, rv: hip.norm_act_fwd(x, w, b, rm, rv, False, 0.1, 1e-5, 0.01, 0, None, cl), [xv, w, b, rm, rv])
        for dt in (torch.float16,):        # (bf16 has no numpy view for the guard buffers; the 16-bit path is one template)
            xh = xv.to(dt)
            xh = _cl(xh) if cl else xh
            _guarded(lambda x, w, b, rm, rv: hip.norm_act_fwd(x, w, b, rm, rv, True, 0.1, 1e-5, 0.01, 0, None, cl), [xh, w, b, rm, rv],
                     inplace=(3, 4))


# ---------------------------------------------------------------- K7 point-wise convolutions on rows (+ fused pieces)
@pytest.mark.parametrize("B,cin,cout,groups,sp", [
    (2, 24, 96, 1, (4, 8, 8)), (1, 36, 68, 1, (3, 5, 7)), (2, 48, 48, 4, (2, 9, 8)), (2, 132, 528, 1, (4, 7, 6)),
    (2, 528, 132, 1, (4, 7, 6)), (1, 528, 528, 6, (8, 14, 12)), (3, 16, 16, 1, (17, 19)),
])
def test_pointwise_rows(hip, B, cin, cout, groups, sp):
    g = _gen(cin + cout)
    x = _cl(torch.randn((B, cin) + sp, generator=g).to(DEV))
    w2 = (torch.randn((cout, cin // groups), generator=g) * 0.1).to(DEV)
    bias = torch.randn(cout, generator=g).to(DEV)
    _guarded(lambda x, w, b: hip.pw_rows(x, w, b, groups), [x, w2, bias])
    _guarded(lambda x, w: hip.pw_rows(x, w, None, groups), [x, w2])
    gy = _cl(torch.randn((B, cout) + sp, generator=g).to(DEV))
    _guarded(lambda gy, x: hip.pw_wgrad(gy, x, groups), [gy, x])
    # statistics epilogue, then the normalise-on-load prologue of the second GEMM, finalize / apply, backward pieces
    _guarded(lambda x, w: hip.pw_rows_fused(x, w, groups, want_stats=True), [x, w2])
    h, part = hip.pw_rows_fused(x, w2, groups, want_stats=True)
    P = x.numel() // cin
    gam = (torch.rand(cout, generator=g) + 0.5).to(DEV)
    bet = torch.randn(cout, generator=g).to(DEV)
    rm, rv = torch.zeros(cout, device=DEV), torch.ones(cout, device=DEV)
    _guarded(lambda part, gam, bet, rm, rv: hip.norm_finalize(part, P, cout, DEV, gam, bet, None, rm, rv, True, 0.1, 1e-5),
             [part, gam, bet, rm, rv], inplace=(3, 4))
    m, i, sc, sh = hip.norm_finalize(part, P, cout, DEV, gam, bet, None, rm, rv, True, 0.1, 1e-5)
    _guarded(lambda h, res, gam, bet, m, i: hip.norm_apply_rows(h, res, gam, bet, m, i, 0.01), [h, None, gam, bet, m, i])
    if groups == 1:
        w3 = (torch.randn((cin, cout), generator=g) * 0.1).to(DEV)
        _guarded(lambda h, w, sc, sh: hip.pw_rows_fused(h, w, 1, pro=(sc, sh, 0.01), want_stats=True), [h, w3, sc, sh])
        y2, _ = hip.pw_rows_fused(h, w3, 1, pro=(sc, sh, 0.01), want_stats=True)
        m2 = torch.randn(cin, generator=g).to(DEV)
        i2 = (torch.rand(cin, generator=g) + 0.5).to(DEV)
        _guarded(lambda h, res, gam, bet, m, i: hip.norm_apply_rows(h, res, gam, bet, m, i, 1.0), [y2, x, None, None, m2, i2])
        dy = _cl(torch.randn((B, cin) + sp, generator=g).to(DEV))
        _guarded(lambda dy, wt, h, gam, bet, m, i: hip.pw_rows_fused(dy, wt, 1, bwd=(h, gam, bet, m, i, 0.01)),
                 [dy, w3.t().contiguous(), h, gam, bet, m, i])
        da, bpart = hip.pw_rows_fused(dy, w3.t().contiguous(), 1, bwd=(h, gam, bet, m, i, 0.01))
        _guarded(lambda bp: hip.norm_bwd_finalize(bp, P, cout, DEV, True), [bpart])
        coeff, _, _ = hip.norm_bwd_finalize(bpart, P, cout, DEV, True)
        _guarded(lambda h, da, co, gam, bet, m, i: hip.norm_bwd_apply_rows(h, da, co, gam, bet, m, i, 0.01), [h, da, coeff, gam, bet, m, i])
        _guarded(lambda dy, h, sc, sh: hip.pw_wgrad_fused(dy, h, 1, (sc, sh, 0.01)), [dy, h, sc, sh])


# ---------------------------------------------------------------- K2 + K7 fused (windows / pooled)
@pytest.mark.parametrize("B,C,sp,win,K,groups", [(2, 24, (4, 8, 8), (2, 4, 4), 9, 4), (1, 132, (4, 14, 12), (4, 7, 6), 9, 6), (2, 16, (16, 16), (8, 8), 9, 4), (1, 264, (4, 7, 6), (4, 7, 6), 7, 6)])
def test_grouped_window_rows(hip, B, C, sp, win, K, groups):
    g = _gen(C + K)
    x = _cl(torch.randn((B, C) + sp, generator=g).to(DEV))
    shift = tuple(w // 2 for w in win)
    rows = hip.window_gather(x, win, shift)
    nw, _, Nw = rows.shape
    assert hip.mr_grouped_rows_supported(nw, C, groups, Nw, K)
    idx = torch.randint(0, Nw, (nw, Nw, K), generator=g, dtype=torch.int32).to(DEV)
    w2 = (torch.randn((2 * C, 2 * C // groups), generator=g) * 0.1).to(DEV)
    _guarded(lambda r, idx, w: hip.mr_grouped_rows(r, idx, K, 1, w, groups, B, sp, win, shift, True, True, True), [rows, idx, w2])
    _, arg, _, _ = hip.mr_grouped_rows(rows, idx, K, 1, w2, groups, B, sp, win, shift, False, True, False)
    dh = _cl(torch.randn((B, 2 * C) + sp, generator=g).to(DEV))
    _guarded(lambda dh, w, arg: hip.mr_grouped_rows_bwd(dh, w, arg, groups, sp, win, shift), [dh, w2, arg])


@pytest.mark.parametrize("B,C,N,M,K,groups", [(2, 132, 1344, 168, 16, 4), (1, 66, 300, 300, 9, 6), (2, 264, 200, 25, 8, 6)])
def test_grouped_channel_major(hip, B, C, N, M, K, groups, monkeypatch):
    monkeypatch.setenv("NEXTOU_MR_GROUPED_CM_MIN_WORKGROUPS", "0")
    monkeypatch.setenv("NEXTOU_MR_GROUPED_CM_CHUNKS", "1")
    Ng = 2 * C // groups
    assert hip.mr_grouped_cm_tiles(B, C, groups, Ng, N, M, K) > 0
    g = _gen(N + M)
    x = torch.randn((B, C, N), generator=g).to(DEV)
    y = None if M == N else torch.randn((B, C, M), generator=g).to(DEV)
    idx = torch.randint(0, M, (B, N, K), generator=g, dtype=torch.int32).to(DEV)
    w2 = (torch.randn((2 * C, 2 * C // groups), generator=g) * 0.1).to(DEV)
    _guarded(lambda x, y, idx, w: hip.mr_grouped_cm(x, y, idx, K, 1, w, groups, True, True, True), [x, y, idx, w2])
    _, _, h, part = hip.mr_grouped_cm(x, y, idx, K, 1, w2, groups, False, False, True)
    gam = (torch.rand(2 * C, generator=g) + 0.5).to(DEV)
    bet = torch.randn(2 * C, generator=g).to(DEV)
    _guarded(lambda h, gam, bet, part: hip.norm_act_fwd_partials(h, gam, bet, part, 2 * C, 1e-5, 0.01), [h.view(1, B * 2 * C, N), gam, bet, part])


# ---------------------------------------------------------------- decoder concatenation
def test_cat_bias(hip):
    from nextou_amd import graph_ops
    g = _gen(3)
    y = _cl(torch.randn((2, 40, 3, 8, 8), generator=g).to(DEV))
    skip = _cl(torch.randn((2, 36, 3, 8, 8), generator=g).to(DEV))
    bias = torch.randn(40, generator=g).to(DEV)
    _guarded(lambda y, b, s: graph_ops._CatBias.apply(y, b, s), [y, bias, skip])
    _guarded(lambda y, s: graph_ops._CatBias.apply(y, None, s), [y, skip])


# ---------------------------------------------------------------- filter of the data gradient's forward convolution
@pytest.mark.parametrize("co,ci,k,layout", [(66, 72, (3, 3, 3), "cl"), (33, 40, (1, 3, 3), "nchw"), (40, 40, (3, 3), "cl"),
                                             (264, 132, (1, 1, 1), "nchw"), (5, 3, (2, 2, 2), "cl")])
def test_filter_flip_t(hip, co, ci, k, layout):
    g = _gen(co + ci)
    w = torch.randn((co, ci) + k, generator=g).to(DEV)
    w = _cl(w) if layout == "cl" else w
    want = _cl(w.transpose(0, 1).flip(*range(2, 2 + len(k))))
    got = hip.filter_flip_t(w)
    assert got.shape == want.shape and got.stride() == want.stride() and torch.equal(got, want)
    _guarded(lambda w: hip.filter_flip_t(w), [w])


# ---------------------------------------------------------------- step glue (ABI v13)
@pytest.mark.parametrize("C,ld,c_off", [(40, 80, 0), (36, 76, 40), (128, 132, 0)])
def test_narrow_copy_sum(hip, C, ld, c_off):
    g = _gen(C + ld)
    x = _cl(torch.randn((2, ld, 3, 7, 9), generator=g).to(DEV))
    _guarded(lambda x: hip.narrow_copy_sum(x, c_off, C), [x])


@pytest.mark.parametrize("clip", [True, False])
def test_clip_sgd(hip, clip):
    """Parameters, gradients and momentum buffers of every size class (one element, odd counts, chunk boundaries, a channels-last
    filter) on guard pages; the device table, the chunk list and the norm workspace are torch.empty allocations and land there too."""
    from nextou_amd.optim import ClipSGD
    g = _gen(17)
    shapes = [(1,), (37,), (16384,), (16385,), (66, 8, 3, 3, 3), (40001,)]
    ps = [torch.randn(s, generator=g).to(DEV) for s in shapes]
    ps[4] = _cl(ps[4])
    gs = [torch.empty_like(p).copy_(torch.randn(p.shape, generator=g).to(DEV) * 4) for p in ps]
    ms = [torch.empty_like(p).copy_(torch.randn(p.shape, generator=g).to(DEV)) for p in ps]
    n = len(ps)

    def launch(*tensors):
        params = [torch.nn.Parameter(t) for t in tensors[:n]]
        opt = ClipSGD(params, 0.01, momentum=0.99, weight_decay=3e-5, nesterov=True)
        for p, gr, m in zip(params, tensors[n:2 * n], tensors[2 * n:]):
            p.grad = gr
            opt.state[p]["momentum_buffer"] = m
        out = opt.clip_and_step(12.0) if clip else opt.step()
        assert opt.last_path == "own"
        return out.reshape(1).clone() if clip else None
    _guarded(launch, ps + gs + ms, exact=True, inplace=tuple(range(3 * n)))


@pytest.mark.parametrize("cin,cout,c2,sp,stride", [(72, 40, 40, (3, 6, 5), (1, 2, 2)), (20, 12, 8, (2, 3, 3), (2, 2, 2)), (24, 12, 8, (5, 7), (2, 2))])
def test_upconv_cat_rows(hip, cin, cout, c2, sp, stride):
    g = _gen(cin + cout)
    T = 1
    for s_ in stride:
        T *= s_
    sp_out = tuple(d * s_ for d, s_ in zip(sp, stride))
    y2 = _cl(torch.randn((2, T * cout) + sp, generator=g).to(DEV))
    skip = _cl(torch.randn((2, c2) + sp_out, generator=g).to(DEV))
    bias = torch.randn(cout, generator=g).to(DEV)
    go = _cl(torch.randn((2, cout + c2) + sp_out, generator=g).to(DEV))
    _guarded(lambda y2, b, s: hip.upconv_cat_rows(y2, b, s, stride), [y2, bias, skip])
    _guarded(lambda go: hip.upconv_cat_rows_bwd(go, cout, sp, stride), [go])


@pytest.mark.parametrize("cin,cout,c2,sp,stride", [(72, 40, 40, (3, 6, 5), (1, 2, 2)), (20, 12, 8, (2, 3, 3), (2, 2, 2)), (24, 12, 8, (5, 7), (2, 4))])
def test_upconv_cat_direct(hip, cin, cout, c2, sp, stride):
    """Round 6: the up-convolution's GEMM storing into the concatenation buffer (nextou_pw_rows_up) + the skip-half pass."""
    g = _gen(cin + cout + 1)
    T = 1
    for s_ in stride:
        T *= s_
    sp_out = tuple(d * s_ for d, s_ in zip(sp, stride))
    x = _cl(torch.randn((2, cin) + sp, generator=g).to(DEV))
    w2 = torch.randn((T * cout, cin), generator=g).to(DEV)
    skip = _cl(torch.randn((2, c2) + sp_out, generator=g).to(DEV))
    bias = torch.randn(cout, generator=g).to(DEV)
    _guarded(lambda x, w2, b, s: hip.upconv_cat_direct(x, w2, b, s, stride, cout), [x, w2, bias, skip])
