"""Randomised shape sweeps (hypothesis, derandomised so that every run checks the same cases):
HIP kernels vs the oracle on ragged, odd and degenerate sizes."""
import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _ops_ora():
    import oracle
    from nextou_amd import graph_ops
    return graph_ops, oracle.CanonicalBackend


def _rand(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


@settings(max_examples=30, deadline=None, derandomize=True)
@given(B=st.integers(1, 5), C=st.integers(1, 70), N=st.integers(1, 400), M=st.one_of(st.none(), st.integers(1, 500)),
       k=st.integers(1, 40), relpos=st.booleans(), normalize=st.booleans(), seed=st.integers(0, 10 ** 6))
def test_knn_any_shape_bit_exact(B, C, N, M, k, relpos, normalize, seed):
    ops, ora = _ops_ora()
    k = min(k, M if M is not None else N)
    x = _rand((B, C, N), seed)
    y = None if M is None else _rand((B, C, M), seed + 1)
    rp = _rand((N, M or N), seed + 2, 0.05) if relpos else None
    want = ora.knn_graph(x, y, rp, k, normalize=normalize)
    got = ops.knn_graph(x.to(DEV), None if y is None else y.to(DEV), None if rp is None else rp.to(DEV), k,
                        normalize=normalize)
    assert torch.equal(got.cpu(), want)


@settings(max_examples=25, deadline=None, derandomize=True)
@given(B=st.integers(1, 4), C=st.integers(1, 40), N=st.integers(1, 700), M=st.one_of(st.none(), st.integers(1, 300)),
       K=st.integers(1, 34), step=st.integers(1, 3), seed=st.integers(0, 10 ** 6))
def test_mr_aggregate_any_shape(B, C, N, M, K, step, seed):
    ops, ora = _ops_ora()
    x = _rand((B, C, N), seed)
    y = None if M is None else _rand((B, C, M), seed + 1)
    g = torch.Generator().manual_seed(seed + 2)
    idx = torch.randint(0, M or N, (B, N, K * step), generator=g, dtype=torch.int32)
    want, _ = ora.mr_fwd(x, y, idx, None, K, step)
    xd = x.to(DEV).requires_grad_(True)
    yd = None if y is None else y.to(DEV).requires_grad_(True)
    out = ops.mr_aggregate(xd, idx.to(DEV), yd, k=K, idx_step=step)
    assert torch.equal(out.detach().cpu(), want)
    gout = _rand(out.shape, seed + 3)
    grads = torch.autograd.grad(out, [xd] if yd is None else [xd, yd], gout.to(DEV))
    dx, dy = ora.mr_bwd(gout, x, y, idx, None, K, step)
    assert float((grads[0].cpu() - dx).abs().max()) <= 1e-5 * max(1.0, float(dx.abs().max()))
    if dy is not None:
        assert float((grads[1].cpu() - dy).abs().max()) <= 1e-5 * max(1.0, float(dy.abs().max()))


@settings(max_examples=20, deadline=None, derandomize=True)
@given(B=st.integers(1, 3), D=st.integers(1, 9), H=st.integers(1, 17), W=st.integers(1, 33), L=st.integers(2, 20),
       conn3d=st.booleans(), full=st.booleans(), thick=st.integers(1, 2), seed=st.integers(0, 10 ** 6))
def test_bti_any_shape(B, D, H, W, L, conn3d, full, thick, seed):
    ops, ora = _ops_ora()
    g = torch.Generator().manual_seed(seed)
    shape = (B, D, H, W) if conn3d else (B, H, W)
    conn = (26 if full else 6) if conn3d else (8 if full else 4)
    labels = torch.randint(0, L, shape, generator=g, dtype=torch.uint8)
    lut_a = torch.randint(0, 2 ** 31 - 1, (256,), generator=g, dtype=torch.int32)
    lut_c = torch.randint(0, 2 ** 31 - 1, (256,), generator=g, dtype=torch.int32) & ~lut_a
    want = ora.bti_critical(labels, lut_a, lut_c, conn, thick)
    got = ops.bti_critical_map(labels.to(DEV), lut_a.to(DEV), lut_c.to(DEV), conn, thick)
    assert torch.equal(got.cpu(), want)
    logits = torch.randn((B, L) + shape[1:], generator=g)
    want_lab = ora.argmax_labels(logits)
    assert torch.equal(ops.argmax_labels(logits.to(DEV)).cpu(), want_lab)
    # the same logits stored channels-last (the layout the network's heads produce): the rows kernels read them in place
    mf = torch.channels_last_3d if conn3d else torch.channels_last
    xcl = logits.to(DEV).contiguous(memory_format=mf)
    assert torch.equal(ops.argmax_labels(xcl).cpu(), want_lab)
    target = torch.randint(0, L, shape, generator=g, dtype=torch.uint8)
    crit = (torch.rand(shape, generator=g) < 0.3).to(torch.uint8)
    want_ce = ora.bti_ce_fwd(logits, target, crit)
    for xin in (logits.to(DEV), xcl):
        xin = xin.clone().requires_grad_(True)
        got_ce = ops.critical_cross_entropy(xin, target.to(DEV), crit.to(DEV))
        np.testing.assert_allclose(got_ce.detach().cpu().numpy(), want_ce.numpy(), rtol=1e-12, atol=1e-300)
        wgt = torch.rand(B, generator=torch.Generator().manual_seed(seed + 5), dtype=torch.float64) + 0.5
        (grad,) = torch.autograd.grad((got_ce * wgt.to(DEV)).sum(), xin)
        ref = ora.bti_ce_bwd(logits, target, crit, wgt)
        assert float((grad.cpu() - ref).abs().max()) <= 1e-6 * max(float(ref.abs().max()), 1e-30)


@settings(max_examples=12, deadline=None, derandomize=True)
@given(B=st.integers(86, 160), C=st.integers(1, 40), N=st.integers(65, 192), k=st.integers(1, 32), relpos=st.booleans(),
       seed=st.integers(0, 10 ** 6))
def test_knn_whole_window_kernel_any_shape_bit_exact(B, C, N, k, relpos, seed):
    """Self graphs of 65 ... 192 points in batches large enough for the whole-window plan of knn_window_kernel (two wave groups, lists
    merged through LDS): ragged last tiles in both groups, every list bucket."""
    ops, ora = _ops_ora()
    x = _rand((B, C, N), seed)
    rp = _rand((N, N), seed + 2, 0.05) if relpos else None
    want = ora.knn_graph(x, None, rp, k)
    got = ops.knn_graph(x.to(DEV), None, None if rp is None else rp.to(DEV), k)
    assert torch.equal(got.cpu(), want)


@settings(max_examples=20, deadline=None, derandomize=True)
@given(B=st.integers(1, 2), cg2=st.one_of(st.integers(1, 16), st.sampled_from([22, 27])), groups=st.integers(1, 6),
       win=st.tuples(st.integers(1, 4), st.integers(1, 5), st.integers(1, 6)),
       cnt=st.tuples(st.integers(1, 2), st.integers(1, 3), st.integers(1, 3)), k=st.integers(1, 32), shifted=st.booleans(),
       seed=st.integers(0, 10 ** 6))
def test_mr_grouped_rows_any_shape(B, cg2, groups, win, cnt, k, shifted, seed):
    """K2 + K7 forward / backward in one launch each against the launches they replace, on random window / group / list shapes."""
    ops, _ = _ops_ora()
    be = ops._HIP
    C = 2 * cg2 * groups
    Nw = win[0] * win[1] * win[2]
    k = min(k, Nw)
    spatial = tuple(w * c for w, c in zip(win, cnt))
    shift = tuple((s // 2) if shifted else 0 for s in win)
    n_windows = B * cnt[0] * cnt[1] * cnt[2]
    if not be.mr_grouped_rows_supported(n_windows, C, groups, Nw, k):
        return
    g = torch.Generator().manual_seed(seed)
    vol = torch.randn((B, C) + spatial, generator=g).to(DEV).contiguous(memory_format=torch.channels_last_3d)
    windows = ops.window_gather(vol, win, shift)
    idx = torch.randint(0, Nw, (n_windows, Nw, k), generator=g, dtype=torch.int32).to(DEV)
    w = (torch.randn((2 * C, 2 * C // groups), generator=g) * 0.3).to(DEV)
    a, arg, h, part = be.mr_grouped_rows(windows, idx, k, 1, w, groups, B, spatial, win, shift, True, True, True)
    agg, arg0 = be.mr_fwd(windows, None, idx, None, k, 1, want_arg=True)
    a0 = be.window_scatter(agg, None, spatial, win, shift)
    assert torch.equal(a, a0) and torch.equal(arg, arg0)
    want = torch.nn.functional.conv3d(a0.double(), w.double().reshape(2 * C, -1, 1, 1, 1), groups=groups)
    scale = max(float(want.abs().max()), 1e-6)
    assert float((h.double() - want).abs().max()) <= 3e-6 * scale
    hs = h.permute(0, 2, 3, 4, 1).reshape(-1, 2 * C).double()
    got = part.sum(1)
    assert torch.allclose(got[:, 0], hs.sum(0), rtol=0, atol=1e-5 * scale * max(hs.shape[0], 1) ** 0.5 + 1e-9)
    assert torch.allclose(got[:, 1], (hs * hs).sum(0), rtol=1e-5, atol=1e-6 * scale * scale)
    dh = torch.randn(h.shape, generator=g).to(DEV).contiguous(memory_format=torch.channels_last_3d)
    dx = be.mr_grouped_rows_bwd(dh, w, arg, groups, spatial, win, shift)
    n_, k_ = w.shape[0] // groups, w.shape[1]
    wt = w.reshape(groups, n_, k_).transpose(1, 2).reshape(groups * k_, n_).contiguous()
    dx0, _ = be.mr_bwd_arg(be.window_gather(be.pw_rows(dh, wt, None, groups), win, shift), arg, Nw, False)
    assert float((dx - dx0).abs().max()) <= 3e-6 * max(float(dx0.abs().max()), 1e-6)


def _logits(shape, seed, channels_last):
    x = (_rand(shape, seed) * 3).to(DEV)
    mf = {4: torch.channels_last, 5: torch.channels_last_3d}.get(len(shape))
    if channels_last and mf is not None:
        x = x.contiguous(memory_format=mf)
    return x


@settings(max_examples=25, deadline=None, derandomize=True)
@given(B=st.integers(1, 3), L=st.integers(2, 24), sp=st.lists(st.integers(1, 19), min_size=1, max_size=3), channels_last=st.booleans(),
       masked=st.booleans(), seed=st.integers(0, 10 ** 6))
def test_dice_statistics_any_shape(B, L, sp, channels_last, masked, seed):
    """K5d: the three soft-Dice sums per (sample, class) and the logit gradient through them against their float64 definition — odd class
    counts, one-voxel volumes, channels-last and NCDHW logits, optional loss mask."""
    ops, _ = _ops_ora()
    shape = (B, L) + tuple(sp)
    x = _logits(shape, seed, channels_last).requires_grad_(True)
    g = torch.Generator().manual_seed(seed + 1)
    y = torch.randint(0, L, (B, 1) + tuple(sp), generator=g).float().to(DEV)
    mask = (torch.rand((B, 1) + tuple(sp), generator=g) > 0.3).to(DEV) if masked else None
    if not ops.dice_stats_eligible(x, y):
        return
    inter, pred, gt = ops.dice_stats(x, y, mask)
    x64 = x.detach().double().contiguous().requires_grad_(True)
    p = torch.softmax(x64, 1)
    oh = torch.zeros_like(p).scatter_(1, y.long(), 1.0)
    w = mask.double() if masked else torch.ones_like(y, dtype=torch.float64)
    axes = tuple(range(2, x.dim()))
    wi, wp, wg = (p * oh * w).sum(axes), (p * w).sum(axes), (oh * w).sum(axes)
    for got, want in ((inter, wi), (pred, wp), (gt, wg)):
        assert float((got.detach().double() - want.detach()).abs().max()) <= 2e-6 * max(1.0, float(want.detach().abs().max()))
    ci = torch.rand(inter.shape, generator=g, dtype=torch.float64).to(DEV)
    cp = torch.rand(inter.shape, generator=g, dtype=torch.float64).to(DEV)
    gx, = torch.autograd.grad((inter.double() * ci).sum() + (pred.double() * cp).sum(), x)
    gw, = torch.autograd.grad((wi * ci).sum() + (wp * cp).sum(), x64)
    # (float32 products of O(1) coefficients: a few 1e-8 of absolute round-off whatever the gradient's own size)
    assert float((gx.double() - gw).abs().max()) <= 5e-6 * float(gw.abs().max()) + 2e-7


@settings(max_examples=25, deadline=None, derandomize=True)
@given(B=st.integers(1, 3), L=st.integers(2, 24), sp=st.lists(st.integers(1, 23), min_size=1, max_size=3), channels_last=st.booleans(),
       ignore_every=st.integers(0, 5), seed=st.integers(0, 10 ** 6))
def test_mean_cross_entropy_any_shape(B, L, sp, channels_last, ignore_every, seed):
    """K5c: mean cross-entropy and its logit gradient against torch's float64 cross_entropy on random class counts, volumes and layouts,
    with ignored voxels (down to every voxel but one)."""
    import torch.nn.functional as F
    ops, _ = _ops_ora()
    shape = (B, L) + tuple(sp)
    x = _logits(shape, seed, channels_last).requires_grad_(True)
    t = torch.randint(0, L, (B,) + tuple(sp), generator=torch.Generator().manual_seed(seed + 1)).to(DEV)
    if ignore_every:
        flat = t.view(-1)
        flat[::ignore_every + 1] = -100
        flat[-1] = 0
    if not ops.cross_entropy_mean_eligible(x, t):
        return
    loss = ops.cross_entropy_mean(x, t)
    gx, = torch.autograd.grad(loss * 1.7, x)
    x64 = x.detach().double().requires_grad_(True)
    want = F.cross_entropy(x64, t)
    gw, = torch.autograd.grad(want * 1.7, x64)
    assert abs(float(loss) - float(want)) <= 2e-6 * max(abs(float(want)), 1e-6)
    assert float((gx.double() - gw).abs().max()) <= 2e-6 * float(gw.abs().max()) + 1e-7


@settings(max_examples=20, deadline=None, derandomize=True)
@given(n=st.integers(1, 5000), L=st.integers(1, 255), dtype=st.sampled_from(["f32", "i64", "u8"]), bad=st.sampled_from([None, "low", "high"]),
       seed=st.integers(0, 10 ** 6))
def test_label_map_conversion_and_range_flag(n, L, dtype, bad, seed):
    """nextou_labels_u8: uint8 copy of a float32 / int64 / uint8 label map with the out-of-range flag OR-ed on the device."""
    ops, _ = _ops_ora()
    t = torch.randint(0, L, (n,), generator=torch.Generator().manual_seed(seed))
    pos = seed % n
    if bad == "high":
        t[pos] = L if L < 255 or dtype != "u8" else 255
    elif bad == "low" and dtype != "u8":
        t[pos] = -1
    expect_bad = (bad == "high" and (L <= 255 and int(t[pos]) >= L)) or (bad == "low" and dtype != "u8")
    src = {"f32": t.float(), "i64": t.long(), "u8": t.clamp(0, 255).to(torch.uint8)}[dtype].to(DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    out = ops.checked_label_map(src, L, flag)
    assert out.dtype == torch.uint8 and out.shape == src.shape
    assert bool(flag.item()) == bool(expect_bad)
    ok = (t >= 0) & (t < L)
    assert torch.equal(out.cpu()[ok], t[ok].to(torch.uint8))


@settings(max_examples=16, deadline=None, derandomize=True)
@given(B=st.integers(1, 2), sp=st.tuples(st.integers(1, 5), st.integers(1, 9), st.integers(1, 11)), g=st.sampled_from([1, 1, 2, 4, 6]),
       a=st.integers(1, 12), b=st.integers(1, 16), c=st.integers(1, 12), two=st.booleans(), with_res=st.booleans(), training=st.booleans(),
       mode=st.sampled_from(["1", "fwd"]))
def test_pointwise_chain_any_shape(B, sp, g, a, b, c, two, with_res, training, mode):
    """The fused point-wise chain (K7 GEMMs with K6 in their prologue / epilogue, forward and backward) on random channel counts, group
    counts and ragged point counts against the float64 ATen restatement of the reference's op sequence."""
    import os
    import test_gpu_fused as tf
    from hypothesis import assume
    ops, _ = _ops_ora()
    ci, cm = 4 * a * g, 4 * b * g
    co = 4 * c if two else cm
    # (batch statistics over a handful of points are ill-conditioned — and torch.batch_norm refuses a single one; the product only takes
    # the chain from NEXTOU_PW_FUSE_MIN_POINTS = 65 536 points on)
    assume(B * sp[0] * sp[1] * sp[2] >= 16)
    old = os.environ.get("NEXTOU_PW_FUSE")
    os.environ["NEXTOU_PW_FUSE"] = mode
    try:
        tf.check_pointwise_chain(ops, training, B, tuple(sp), ci, cm, co, g, two, with_res, mode)
    finally:
        if old is None:
            os.environ.pop("NEXTOU_PW_FUSE", None)
        else:
            os.environ["NEXTOU_PW_FUSE"] = old


@settings(max_examples=24, deadline=None, derandomize=True)
@given(B=st.integers(1, 24), C=st.integers(1, 140), n4=st.integers(1, 48), k=st.integers(1, 192), relpos=st.booleans(),
       levels=st.sampled_from([0, 0, 2, 5]), seed=st.integers(0, 10 ** 6))
def test_knn_small_kernel_any_shape_bit_exact(B, C, n4, k, relpos, levels, seed):
    """knn_small_kernel (round 6: <= 192-point self graphs, N a multiple of 4, a handful of windows) on ragged tiles, one ... three channel slabs
    and — `levels` > 0: features quantised to a few values, so that whole groups of candidates are at EXACTLY the same distance — the tie path of
    the counting selection (reference torch_edge.py:58-90: topk's order on equal distances is the lower index, as restated by the oracle)."""
    ops, ora = _ops_ora()
    N = 4 * n4
    if B * ((N + 15) // 16) > 384:
        B = max(1, 384 // ((N + 15) // 16))
    k = min(k, N)
    x = _rand((B, C, N), seed)
    if levels:
        x = torch.round(x * levels) / levels
        x[:, 0] += 0.5                                    # no all-zero point: 0 / eps stays what it is, but keep the norms apart from eps
    rp = _rand((N, N), seed + 2, 0.05) if relpos else None
    want = ora.knn_graph(x, None, rp, k)
    got = ops.knn_graph(x.to(DEV), None, None if rp is None else rp.to(DEV), k)
    assert torch.equal(got.cpu(), want)


@settings(max_examples=16, deadline=None, derandomize=True)
@given(B=st.integers(1, 3), D=st.integers(1, 5), H=st.integers(1, 21), W=st.integers(1, 70), C=st.integers(1, 48), two_d=st.booleans(),
       training=st.booleans(), offset=st.sampled_from([0.0, 3.0]), seed=st.integers(0, 10 ** 6))
def test_stem_block_any_shape(B, D, H, W, C, two_d, training, offset, seed):
    """K9 (round 6; reference NexToU_Encoder_Decoder.py:125-136: Conv(1 -> C, [1,]3,3) -> BatchNorm -> LeakyReLU on the image) against float64 autograd:
    ragged rows and row batches, every channel count up to the 48 the kernels take, 2-D and 3-D, batch and running statistics."""
    import torch.nn.functional as F
    from nextou_amd import graph_ops
    g = torch.Generator().manual_seed(seed)
    shape = (B, 1, H, W) if two_d else (B, 1, D, H, W)
    x = (torch.randn(shape, generator=g) + offset).to(DEV)
    st_ = list(x.stride()); st_[1] = 1
    x = x.as_strided(x.shape, st_)
    c_pad = (C + 3) // 4 * 4
    w = (torch.randn((C, 1, 3, 3) if two_d else (C, 1, 1, 3, 3), generator=g) * 0.4).to(DEV).requires_grad_(True)
    cb = torch.randn(C, generator=g).to(DEV).requires_grad_(True)
    gamma = (1 + 0.3 * torch.randn(C, generator=g)).to(DEV).requires_grad_(True)
    beta = (0.2 * torch.randn(C, generator=g)).to(DEV).requires_grad_(True)
    rm, rv = torch.randn(C, generator=g).to(DEV), (0.5 + torch.rand(C, generator=g)).to(DEV)
    cl = torch.channels_last if two_d else torch.channels_last_3d
    gy = torch.randn((B, c_pad) + tuple(shape[2:]), generator=g).to(DEV).contiguous(memory_format=cl)
    rmd, rvd = rm.double().clone(), rv.double().clone()
    if training:
        y = graph_ops._StemBlock.apply(x, w, cb, gamma, beta, rm, rv, True, 0.1, 1e-5, 0.01, c_pad)
        y.backward(gy)
    else:                                                 # running statistics: forward only (stem_block_eligible declines eval-mode training)
        with torch.no_grad():
            y = graph_ops._StemBlock.apply(x, w, cb, gamma, beta, rm, rv, False, 0.1, 1e-5, 0.01, c_pad)
    wd, cbd, gd, bd = (t.detach().double().clone().requires_grad_(True) for t in (w, cb, gamma, beta))
    z = (F.conv2d if two_d else F.conv3d)(x.double().contiguous(), wd, cbd, 1, (1, 1) if two_d else (0, 1, 1))
    count = z.numel() // C
    if training and count == 1:
        return                                            # torch refuses batch statistics over one value per channel
    ry = F.leaky_relu(F.batch_norm(z, rmd, rvd, gd, bd, training, 0.1, 1e-5), 0.01)
    scale = float(ry.abs().max()) + 1e-30
    assert float((y[:, :C].double() - ry).abs().max()) <= 5e-5 * scale
    if c_pad > C:
        assert float(y[:, C:].abs().max()) == 0.0
    assert torch.allclose(rm.double(), rmd, rtol=1e-5, atol=1e-6) and torch.allclose(rv.double(), rvd, rtol=1e-4, atol=1e-6)
    if not training or count < 16:                        # (a handful of values per channel: the gradient through the batch statistics is all cancellation)
        return
    ry.backward(gy[:, :C].double())
    tol = 2e-4
    for name, a, e in (("weight", w.grad, wd.grad), ("gamma", gamma.grad, gd.grad), ("beta", beta.grad, bd.grad)):
        assert float((a.double() - e).abs().max()) <= tol * (float(e.abs().max()) + 1e-30) + 1e-6, name
