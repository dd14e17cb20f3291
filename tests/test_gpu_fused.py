"""Round-3 parity tests on the MI355X for the fused point-wise pipeline: K7 GEMMs with K6's statistics in
their epilogues and K6's normalisation + activation in their operand loads (csrc/pw_gemm.hip, include/nextou_hip.h "K7 + K6
fused"), against (i) the plain K7 / K6 kernels they replace — bit for bit where the arithmetic is the same —, (ii) float64
restatements of the reference's op sequence conv -> batch_norm -> leaky_relu -> conv -> batch_norm -> add (reference
NexToU_Encoder_Decoder.py:368-390, :710-720, :833-842; torch_nn.py:66-92), (iii) the op-by-op module path."""
import copy

import pytest
import torch
import torch.nn.functional as F
from torch import nn

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
CL = torch.channels_last_3d


@pytest.fixture(scope="module")
def ops():
    from nextou_amd import _lib, graph_ops
    _lib.lib()
    assert "libnextou_hip.so" in open("/proc/self/maps").read(), "HIP extension not loaded into this process"
    return graph_ops


def _cl(t):
    return t.to(DEV).contiguous(memory_format=CL)


def _rows(t):                       # (B, C, D, H, W) channels-last -> (P, C) float64
    return t.permute(0, 2, 3, 4, 1).reshape(-1, t.shape[1]).double()


_SHAPES = [
    # (B, spatial, Cin, Cmid, groups of the first conv): cfg-2 graph-stage channel counts, ragged point counts (P % 128 != 0) included
    (2, (4, 7, 6), 132, 528, 1), (1, (3, 5, 7), 264, 264, 6), (2, (8, 14, 12), 264, 1056, 1), (1, (2, 7, 6), 324, 1296, 1),
    (3, (5, 9, 11), 44, 36, 1), (1, (1, 1, 3), 16, 8, 2), (1, (4, 7, 6), 648, 648, 6),
]


@pytest.mark.parametrize("B,sp,ci,cm,g", _SHAPES)
def test_fused_gemm_epilogues_and_prologues(ops, B, sp, ci, cm, g):
    """(a) statistics epilogue: same y as the plain GEMM bit for bit, partial sums == float64 sums of y and y^2;
    (b) finalize: mean / invstd / affine == float64 statistics, running statistics updated as F.batch_norm does;
    (c) operand prologue: GEMM(norm + act on load) == GEMM(K6 apply output) bit for bit;
    (d) gradient-statistics epilogue == the float64 sums K6's backward reduce defines; backward finalize + apply == K6's backward;
    (e) weight gradient with the operand prologue == the plain weight gradient of the materialised activation, bit for bit."""
    hip = ops._HIP
    gen = torch.Generator().manual_seed(ci * 3 + cm)
    x = _cl(torch.randn((B, ci) + sp, generator=gen) * 1.5 + 0.3)
    w1 = (torch.randn((cm, ci // g), generator=gen) * 0.1).to(DEV)
    P = x.numel() // ci
    # (a)
    h_plain = hip.pw_rows(x, w1, None, g)
    h, part = hip.pw_rows_fused(x, w1, g, want_stats=True)
    assert torch.equal(h, h_plain)
    h64 = _rows(h)
    sums = part.sum(1)                                              # (C, 2) float64
    assert torch.allclose(sums[:, 0], h64.sum(0), rtol=1e-6, atol=1e-6 * float(h64.abs().sum(0).max()))
    assert torch.allclose(sums[:, 1], h64.square().sum(0), rtol=1e-6)
    assert torch.equal(part, hip.pw_rows_fused(x, w1, g, want_stats=True)[1])          # fixed order: bit-reproducible
    # (b)
    gamma = (torch.rand((cm,), generator=gen) + 0.5).to(DEV)
    beta = (torch.randn((cm,), generator=gen) * 0.2).to(DEV)
    cbias = (torch.randn((cm,), generator=gen) * 0.1).to(DEV)
    rm, rv = torch.zeros(cm, device=DEV), torch.ones(cm, device=DEV)
    mean, invstd, scale, shift = hip.norm_finalize(part, P, cm, DEV, gamma, beta, cbias, rm, rv, True, 0.1, 1e-5)
    m64, v64 = h64.mean(0), h64.var(0, unbiased=False)
    assert torch.allclose(mean.double(), m64, rtol=1e-5, atol=1e-6)
    assert torch.allclose(invstd.double(), (v64 + 1e-5).rsqrt(), rtol=2e-6)
    assert torch.equal(scale, gamma * invstd)
    assert torch.allclose(shift, beta - mean * scale, rtol=1e-6, atol=1e-6)
    assert torch.allclose(rm.double(), 0.1 * (m64 + cbias.double()), rtol=1e-5, atol=1e-6)     # the folded conv bias enters here only
    assert torch.allclose(rv.double(), 0.9 + 0.1 * h64.var(0, unbiased=P > 1), rtol=1e-5)
    # (c)
    slope = 0.01
    a = hip.norm_apply_rows(h, None, gamma, beta, mean, invstd, slope)
    a_ref, mean_ref, invstd_ref = hip.norm_act_fwd(h, gamma, beta, None, None, True, 0.0, 1e-5, slope, 0, None, channels_last=True)
    # K6's own statistics pass sums every value in float64, the GEMM epilogue 32 points at a time in fp32: same statistics to a few
    # ulp, and given the SAME statistics K6 in pieces is K6 bit for bit
    assert torch.allclose(mean, mean_ref, rtol=1e-5, atol=1e-6) and torch.allclose(invstd, invstd_ref, rtol=2e-6)
    assert torch.equal(hip.norm_apply_rows(h, None, gamma, beta, mean_ref, invstd_ref, slope), a_ref)
    assert float((a - a_ref).abs().max()) <= 1e-5 * float(a_ref.abs().max())
    co = ci
    w2 = (torch.randn((co, cm), generator=gen) * 0.1).to(DEV)
    y_ref = hip.pw_rows(a, w2, None, 1)
    y, part2 = hip.pw_rows_fused(h, w2, 1, pro=(scale, shift, slope), want_stats=True)
    assert torch.equal(y, y_ref)
    y_nostats, none = hip.pw_rows_fused(h, w2, 1, pro=(scale, shift, slope))
    assert none is None and torch.equal(y_nostats, y_ref)
    res = _cl(torch.randn((B, co) + sp, generator=gen))
    mean2, invstd2, _, _ = hip.norm_finalize(part2, P, co, DEV, None, None, None, None, None, True, 0.0, 1e-5)
    out = hip.norm_apply_rows(y, res, None, None, mean2, invstd2, 1.0)
    y64 = _rows(y)
    ref = (y64 - y64.mean(0)) / (y64.var(0, unbiased=False) + 1e-5).sqrt() + _rows(res)
    assert float((_rows(out) - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    # (d)
    dy = _cl(torch.randn((B, co) + sp, generator=gen))
    w2t = w2.t().contiguous()
    da_plain = hip.pw_rows(dy, w2t, None, 1)
    da, partb = hip.pw_rows_fused(dy, w2t, 1, bwd=(h, gamma, beta, mean, invstd, slope))
    assert torch.equal(da, da_plain)
    z64 = h64 * scale.double() + shift.double()
    dz64 = _rows(da) * torch.where(z64 > 0, 1.0, slope)
    xh64 = (h64 - mean.double()) * invstd.double()
    sb = partb.sum(1)
    mag = dz64.abs().sum(0)
    assert bool(((sb[:, 0] - dz64.sum(0)).abs() <= 1e-6 * mag + 1e-9).all())
    assert bool(((sb[:, 1] - (dz64 * xh64).sum(0)).abs() <= 1e-6 * (dz64 * xh64).abs().sum(0) + 1e-9).all())
    coeff, gw, gb = hip.norm_bwd_finalize(partb, P, cm, DEV, True)
    dh = hip.norm_bwd_apply_rows(h, da, coeff, gamma, beta, mean, invstd, slope)
    dh_ref, gw_ref, gb_ref = hip.norm_act_bwd(h, da, gamma, beta, mean, invstd, True, slope, 0, channels_last=True)
    tol = 2e-5 * float(dh_ref.abs().max())
    assert float((dh - dh_ref).abs().max()) <= tol
    assert torch.allclose(gw, gw_ref, rtol=1e-5, atol=1e-5 * float(gw_ref.abs().max()))
    assert torch.allclose(gb, gb_ref, rtol=1e-5, atol=1e-5 * float(gb_ref.abs().max()))
    # (e)
    assert torch.equal(hip.pw_wgrad_fused(dy, h, 1, (scale, shift, slope)), hip.pw_wgrad(dy, a, 1))


def _reference_chain(x, res, w1, g, bn1, slope1, w2, bn2, slope2, training):
    """conv -> batch_norm -> leaky_relu [-> conv -> batch_norm -> leaky_relu] [+ res]: ATen ops in float64 (the reference's op sequence)."""
    dd = torch.float64

    def bn(t, p):
        gamma, beta, rm, rv, cb = p
        if training:
            return F.batch_norm(t, None, None, gamma.to(dd), beta.to(dd), True, 0.0, 1e-5)
        return F.batch_norm(t + cb.to(dd).view(1, -1, 1, 1, 1), rm.to(dd), rv.to(dd), gamma.to(dd), beta.to(dd), False, 0.0, 1e-5)
    t = F.conv3d(x.to(dd), w1.to(dd)[:, :, None, None, None], None, groups=g)
    t = F.leaky_relu(bn(t, bn1), slope1)
    if w2 is not None:
        t = F.conv3d(t, w2.to(dd)[:, :, None, None, None], None)
        t = F.leaky_relu(bn(t, bn2), slope2)
    return t if res is None else t + res.to(dd)


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("B,sp,ci,cm,g,two,with_res", [
    (2, (4, 7, 6), 132, 528, 1, True, True),          # FFN
    (2, (4, 7, 6), 264, 264, 6, True, True),          # SwinGrapher tail: grouped BasicConv -> fc2 -> + shortcut (out 132)
    (1, (3, 5, 7), 132, 132, 1, False, False),        # fc1
    (1, (3, 5, 7), 264, 132, 1, False, True),         # PoolGrapher fc2 + shortcut
    (1, (2, 7, 6), 324, 1296, 1, True, True),
])
@pytest.mark.parametrize("mode", ["1", "fwd"])
def test_pointwise_chain_autograd_vs_float64_reference(ops, monkeypatch, training, B, sp, ci, cm, g, two, with_res, mode):
    """graph_ops._PointwiseChain (forward values, input / weight / norm-parameter / residual gradients) against the float64 ATen
    restatement of the reference's op sequence, in training (batch statistics) and inference (running statistics + the folded conv
    bias), with the gradient statistics fused into the data-gradient GEMM ("1") and with K6's own backward reduce ("fwd")."""
    monkeypatch.setenv("NEXTOU_PW_FUSE", mode)
    co = (132 if (two and g == 6) else ci) if two else cm
    check_pointwise_chain(ops, training, B, sp, ci, cm, co, g, two, with_res, mode)


def check_pointwise_chain(ops, training, B, sp, ci, cm, co, g, two, with_res, mode):
    """(shared with the randomised sweep in tests/test_gpu_property.py; NEXTOU_PW_FUSE must already be ``mode``)"""
    gen = torch.Generator().manual_seed(ci + cm + g + int(two))
    x = _cl(torch.randn((B, ci) + sp, generator=gen)).requires_grad_(True)
    res = _cl(torch.randn((B, co) + sp, generator=gen)).requires_grad_(True) if with_res else None

    def params(c):
        return [(torch.rand((c,), generator=gen) + 0.5).to(DEV).requires_grad_(True), (torch.randn((c,), generator=gen) * 0.2).to(DEV).requires_grad_(True),
                (torch.randn((c,), generator=gen) * 0.1).to(DEV), (torch.rand((c,), generator=gen) + 0.5).to(DEV),
                (torch.randn((c,), generator=gen) * 0.1).to(DEV).requires_grad_(True)]
    w1 = (torch.randn((cm, ci // g), generator=gen) * 0.1).to(DEV).requires_grad_(True)
    bn1 = params(cm)
    w2 = (torch.randn((co, cm), generator=gen) * 0.1).to(DEV).requires_grad_(True) if two else None
    bn2 = params(co) if two else None
    slope1, slope2 = 0.01, 1.0

    def state(p, slope):
        return ops._NormState(training, 0.1, 1e-5, slope, p[2].clone(), p[3].clone())
    args = (x, res, w1.view(cm, ci // g, 1, 1, 1), bn1[0], bn1[1], bn1[4], None if w2 is None else w2.view(co, cm, 1, 1, 1),
            None if bn2 is None else bn2[0], None if bn2 is None else bn2[1], None if bn2 is None else bn2[4], g, state(bn1, slope1),
            None if bn2 is None else state(bn2, slope2), mode != "fwd")
    out = ops._PointwiseChain.apply(*args)
    ref = _reference_chain(x, res, w1, g, bn1, slope1, w2, bn2, slope2, training)
    scale = float(ref.abs().max())
    assert float((out.double() - ref).abs().max()) <= 3e-5 * scale
    gout = _cl(torch.randn(out.shape, generator=gen))
    leaves = [t for t in [x, res, w1, bn1[0], bn1[1], w2] + ([bn2[0], bn2[1]] if two else []) if t is not None]
    got = torch.autograd.grad(out, leaves, gout, allow_unused=True)
    want = torch.autograd.grad(ref, leaves, gout.double(), allow_unused=True)
    for a, b, t in zip(got, want, leaves):
        assert a is not None and b is not None
        assert float((a.double() - b.double()).abs().max()) <= 2e-4 * max(float(b.abs().max()), 1e-6), (tuple(t.shape),)
    # the folded conv biases: exactly zero gradient under batch statistics, the analytic one with running statistics
    gcb = torch.autograd.grad(ops._PointwiseChain.apply(*args), [bn1[4]], gout, allow_unused=True)[0]
    if training:
        assert gcb is not None and float(gcb.abs().max()) == 0.0
    else:
        want_cb = torch.autograd.grad(_reference_chain(x, res, w1, g, bn1, slope1, w2, bn2, slope2, False), [bn1[4]], gout.double())[0]
        assert float((gcb.double() - want_cb).abs().max()) <= 2e-4 * max(float(want_cb.abs().max()), 1e-6)


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_graph_blocks_fused_vs_op_by_op(ops, monkeypatch, mode):
    """The FFN / SwinGrapher / PoolGrapher modules on a channels-last volume: the fused pipeline (default) against the same
    modules run op by op (NEXTOU_PW_FUSE=0: MIOpen convolutions + K6 passes) — outputs, input gradient, parameter gradients and
    running statistics; the launch labels prove which path ran."""
    import ctypes
    import json
    from nextou_amd import _lib, graph_ops
    from nextou_amd.network_architecture import NexToU_Encoder_Decoder as encdec
    from nextou_amd.network_architecture.norm_act import fuse_norm_act
    monkeypatch.setenv("NEXTOU_PW_FUSE_MIN_POINTS", "0")        # the test volumes are small: take the fused path whatever the size
    kw = dict(conv_op=nn.Conv3d, norm_op=nn.BatchNorm3d, norm_op_kwargs={'eps': 1e-5, 'affine': True})
    torch.manual_seed(4)
    blocks = {
        "ffn": (encdec.FFN(132, 528, act="leakyrelu", drop_path=0.0, **kw), (2, 132, 4, 7, 6)),
        "swin": (encdec.SwinGrapher(12, (4, 8, 8), 4, 1, 'mr', 'leakyrelu', 'instance', True, True, 0.2, 1, n=32, relative_pos=True,
                                    window_size=(2, 4, 4), shift_size=[1, 2, 2], dropout_op=None, **kw), (2, 12, 4, 8, 8)),
        "pool": (encdec.PoolGrapher(12, (8, 16, 32), 4, 1, 'mr', 'leakyrelu', 'instance', True, True, 0.2, 2, n=4096, relative_pos=True,
                                    img_min_shape=(2, 4, 4), dropout_op=None, **kw), (2, 12, 8, 16, 32)),
    }
    L_ = _lib.lib()
    for name, (blk, shape) in blocks.items():
        fuse_norm_act(blk)
        blk = blk.to(DEV).train(mode == "train")
        x0 = torch.randn(shape, device=DEV).contiguous(memory_format=CL)
        gy = torch.randn(shape, device=DEV).contiguous(memory_format=CL)
        results = {}
        tape = graph_ops.IndexTape()
        for setting in ("1", "0"):
            monkeypatch.setenv("NEXTOU_PW_FUSE", setting)
            m = copy.deepcopy(blk)
            x = x0.clone().requires_grad_(True)
            L_.nextou_profile_enable(512)
            # the kNN / arg-max decisions of the first run are replayed in the second (the two paths differ by round-off)
            with graph_ops.index_tape(tape if setting == "1" else graph_ops.IndexTape(tape.entries)):
                y = m(x)
            grads = torch.autograd.grad(y, [x] + [p for p in m.parameters() if p.requires_grad], gy, allow_unused=True)
            torch.cuda.synchronize()
            buf = ctypes.create_string_buffer(1 << 16)
            n = L_.nextou_profile_report(buf, len(buf))
            L_.nextou_profile_enable(0)
            names = [r["kernel"] for r in json.loads(buf.value[:n].decode())]
            results[setting] = (y.detach(), grads, {k: v.clone() for k, v in m.state_dict().items() if "running" in k}, names)
        fused_names, plain_names = results["1"][3], results["0"][3]
        assert any(k.startswith("pw_rows") and "stats" in k for k in fused_names) == (mode == "train"), fused_names
        assert any(k.startswith("pw_rows_kernel") for k in fused_names) and not any(k.startswith("pw_rows_kernel") for k in plain_names)
        (y1, g1, r1, _), (y0, g0, r0, _) = results["1"], results["0"]
        assert float((y1 - y0).abs().max()) <= 5e-5 * float(y0.abs().max()), name
        gscale = max(float(b.abs().max()) for b in g0 if b is not None)
        for a, b in zip(g1, g0):
            assert (a is None) == (b is None)
            if a is not None:   # (gradients that are analytically zero — a norm bias in front of the max-relative aggregate — are noise on both sides)
                assert float((a - b).abs().max()) <= 1e-3 * max(float(b.abs().max()), 1e-3 * gscale), name
        for k in r0:
            assert torch.allclose(r1[k], r0[k], rtol=1e-5, atol=1e-6), (name, k)


@pytest.mark.parametrize("B,C,spatial,window,shift,groups,k", [
    (2, 132, (4, 12, 28), (2, 6, 14), (1, 3, 7), 6, 7),      # cfg-2 stage-2 Swin windows (168 points, 6 groups of 44): the 3 x 11 instance
    (1, 132, (2, 6, 14), (2, 6, 14), (0, 0, 0), 6, 14),      # one window, K = 14
    (2, 12, (4, 8, 8), (2, 4, 4), (1, 2, 2), 6, 4),          # groups of 4 channels (generic instance), 32-point windows
    (3, 64, (1, 12, 10), (1, 4, 5), (0, 2, 0), 4, 9),        # 2-D (D = 1), groups of 32, 20-point windows (less than one wave)
    (1, 48, (6, 6, 6), (3, 6, 6), (1, 0, 3), 2, 32),         # 108-point windows, groups of 48, the longest list
    (1, 264, (4, 6, 14), (2, 6, 14), (1, 3, 7), 6, 14),      # cfg-2 stage 3: groups of 88 (the 6 x 22 instance)
    (1, 324, (2, 6, 14), (2, 6, 14), (0, 0, 0), 6, 28),      # cfg-2 stages 4 / 5: groups of 108 (the 7 x 27 instance, 256 threads)
])
def test_mr_aggregate_fused_with_grouped_conv(ops, B, C, spatial, window, shift, groups, k):
    """K2 + K7 in one launch (nextou_mr_grouped_rows): aggregate rows and arg tape bit-identical to
    window_scatter(mr_aggregate(windows)), the product equal to the grouped 1x1 convolution of those rows (bit-identical to
    pw_rows_grp_kernel where that kernel takes the shape), statistics partials = float64 sums of the product."""
    be = ops._HIP
    g = torch.Generator().manual_seed(C + k)
    vol = _cl(torch.randn((B, C) + spatial, generator=g).to(DEV))
    windows = ops.window_gather(vol, window, shift)
    n_windows, _, Nw = windows.shape
    assert be.mr_grouped_rows_supported(n_windows, C, groups, Nw, k)
    nn_idx = ops.knn_graph(windows, None, None, k)
    w = (torch.randn((2 * C, 2 * C // groups), generator=g) * 0.2).to(DEV)
    a, arg, h, part = be.mr_grouped_rows(windows, nn_idx, k, 1, w, groups, B, spatial, window, shift, True, True, True)
    agg, arg0 = be.mr_fwd(windows, None, nn_idx, None, k, 1, want_arg=True)
    a0 = be.window_scatter(agg, None, spatial, window, shift)
    assert torch.equal(a, a0) and torch.equal(arg, arg0)
    h0, _ = be.pw_rows_fused(a0, w, groups, want_stats=True)
    want = F.conv3d(a0.double(), w.double().reshape(2 * C, -1, 1, 1, 1), groups=groups)
    scale = float(want.abs().max())
    assert float((h.double() - want).abs().max()) <= 2e-6 * scale
    if a0.numel() // (2 * C) >= 64 * 2 * 256 and 2 * C // groups <= 64:
        assert torch.equal(h, h0)
    hs = _rows(h)
    got = part.sum(1)
    assert torch.allclose(got[:, 0], hs.sum(0), rtol=0, atol=1e-5 * scale * hs.shape[0] ** 0.5)
    assert torch.allclose(got[:, 1], (hs * hs).sum(0), rtol=1e-5, atol=1e-6 * scale * scale)
    # eval: no aggregate, no tape, no statistics — the same product
    _, _, h_eval, _ = be.mr_grouped_rows(windows, nn_idx, k, 1, w, groups, B, spatial, window, shift, False, False, False)
    assert torch.equal(h_eval, h)
    # backward for the window tensor in one launch: grouped data-gradient GEMM + window gather + arg-tape scatter
    dh = _cl(torch.randn(h.shape, generator=g).to(DEV))
    dx = be.mr_grouped_rows_bwd(dh, w, arg, groups, spatial, window, shift)
    n_, k_ = w.shape[0] // groups, w.shape[1]
    wt = w.reshape(groups, n_, k_).transpose(1, 2).reshape(groups * k_, n_).contiguous()
    ga = be.pw_rows(dh, wt, None, groups)
    dx0, _ = be.mr_bwd_arg(be.window_gather(ga, window, shift), arg, Nw, False)
    ga64 = F.conv_transpose3d(dh.double(), w.double().reshape(2 * C, -1, 1, 1, 1), groups=groups)
    gw64 = ops.window_gather(ga64.float(), window, shift).double()          # (layout only; the float32 round trip is checked below)
    tol = 2e-6 * float(dx0.abs().max())
    assert float((dx - dx0).abs().max()) <= tol
    assert torch.equal(dx, be.mr_grouped_rows_bwd(dh, w, arg, groups, spatial, window, shift))      # fixed-point sums: bit-reproducible
    want_dx = torch.zeros((n_windows, C, Nw), dtype=torch.float64, device=DEV)
    gx64, gm64 = gw64[:, 0::2], gw64[:, 1::2]
    want_dx.scatter_add_(2, arg.long() & 0xffff, gm64)
    want_dx += gx64 - gm64
    assert float((dx.double() - want_dx).abs().max()) <= 1e-5 * float(want_dx.abs().max())


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_swin_block_with_fused_aggregate_vs_three_launches(ops, monkeypatch, mode):
    """SwinGrapher at the cfg-2 stage-2 width: the K2 + K7 kernel in front of the fused chain against mr_aggregate -> window_scatter ->
    chain (NEXTOU_MR_GROUPED=0) — same output, gradients and running statistics (the two differ only in the statistics' summation
    order); the launch labels prove which path ran."""
    import ctypes
    import json
    from nextou_amd import _lib, graph_ops
    from nextou_amd.network_architecture import NexToU_Encoder_Decoder as encdec
    from nextou_amd.network_architecture.norm_act import fuse_norm_act
    monkeypatch.setenv("NEXTOU_PW_FUSE_MIN_POINTS", "0")
    kw = dict(conv_op=nn.Conv3d, norm_op=nn.BatchNorm3d, norm_op_kwargs={'eps': 1e-5, 'affine': True})
    torch.manual_seed(11)
    blk = encdec.SwinGrapher(132, (4, 12, 28), 7, 1, 'mr', 'leakyrelu', 'instance', True, True, 0.2, 1, n=168, relative_pos=True,
                             window_size=(2, 6, 14), shift_size=[1, 3, 7], dropout_op=None, **kw)
    fuse_norm_act(blk)
    blk = blk.to(DEV).train(mode == "train")
    shape = (2, 132, 4, 12, 28)
    x0 = torch.randn(shape, device=DEV).contiguous(memory_format=CL)
    gy = torch.randn(shape, device=DEV).contiguous(memory_format=CL)
    L_ = _lib.lib()
    results = {}
    for setting in ("1", "0"):
        monkeypatch.setenv("NEXTOU_MR_GROUPED", setting)
        m = copy.deepcopy(blk)
        x = x0.clone().requires_grad_(True)
        L_.nextou_profile_enable(512)
        y = m(x)
        grads = torch.autograd.grad(y, [x] + [p for p in m.parameters() if p.requires_grad], gy, allow_unused=True)
        with torch.no_grad():
            y_ng = copy.deepcopy(blk)(x0)
        torch.cuda.synchronize()
        buf = ctypes.create_string_buffer(1 << 16)
        n = L_.nextou_profile_report(buf, len(buf))
        L_.nextou_profile_enable(0)
        names = [r["kernel"] for r in json.loads(buf.value[:n].decode())]
        results[setting] = (y.detach(), grads, {k: v.clone() for k, v in m.state_dict().items() if "running" in k}, names, y_ng)
    assert sum(k.startswith("mr_grp_rows_bwd_kernel") for k in results["1"][3]) == 1
    assert sum(k.startswith("mr_grp_rows_kernel<train>") for k in results["1"][3]) == 1
    assert sum(k.startswith("mr_grp_rows_kernel<eval>") for k in results["1"][3]) == 1
    assert not any(k.startswith(("mr_fwd", "window_scatter")) for k in results["1"][3][:results["1"][3].index(
        next(k for k in results["1"][3] if k.startswith("mr_grp_rows_kernel")))]), results["1"][3]
    assert not any(k.startswith("mr_grp_rows_kernel") for k in results["0"][3])
    (y1, g1, r1, _, n1), (y0, g0, r0, _, n0) = results["1"], results["0"]
    assert float((y1 - y0).abs().max()) <= 1e-6 * float(y0.abs().max())
    assert float((n1 - n0).abs().max()) <= 1e-6 * float(n0.abs().max())
    assert float((n1 - y1).abs().max()) <= 1e-6 * float(y1.abs().max())       # (no_grad takes the eval kernel variant)
    gscale = max(float(b.abs().max()) for b in g0 if b is not None)
    for a, b in zip(g1, g0):
        assert (a is None) == (b is None)
        if a is not None:   # (analytically zero gradients — a bias in front of a batch-statistics norm — are round-off of the largest ones)
            assert float((a - b).abs().max()) <= max(1e-5 * float(b.abs().max()), 5e-7 * gscale)
    for k in r0:
        assert torch.allclose(r1[k], r0[k], rtol=1e-6, atol=1e-7), k


@pytest.mark.parametrize("ci,co", [(132, 528), (132, 264), (132, 132), (264, 132), (528, 132)])
def test_stationary_weights_rows_kernel(ops, monkeypatch, ci, co):
    """pw_rows_sw_kernel (weights in registers, x streamed once; the stage-2 shapes, >= 65 536 points) against pw_rows_kernel and the
    float64 product: bit-identical for K = 132 (same MFMA, same k order), fp32 round-off for K = 264 / 528; plain and with every fused
    epilogue / prologue, on a ragged point count; the statistics partials of the two kernels describe the same sums."""
    hip = ops._HIP
    gen = torch.Generator().manual_seed(ci + co)
    sp = (33, 45, 45)                                              # 66 825 points per sample: not a multiple of 64
    x = _cl(torch.randn((1, ci) + sp, generator=gen))
    w = (torch.randn((co, ci), generator=gen) * 0.1).to(DEV)
    gamma = (torch.rand((ci,), generator=gen) + 0.5).to(DEV)
    beta = (torch.randn((ci,), generator=gen) * 0.2).to(DEV)
    mean = (torch.randn((ci,), generator=gen) * 0.1).to(DEV)
    invstd = (torch.rand((ci,), generator=gen) + 0.5).to(DEV)
    scale = gamma * invstd
    shift = beta - mean * scale
    h = _cl(torch.randn((1, co) + sp, generator=gen))
    g2, b2 = (torch.rand((co,), generator=gen) + 0.5).to(DEV), (torch.randn((co,), generator=gen) * 0.2).to(DEV)
    m2, i2 = (torch.randn((co,), generator=gen) * 0.1).to(DEV), (torch.rand((co,), generator=gen) + 0.5).to(DEV)
    import json
    import ctypes
    from nextou_amd import _lib
    L_ = _lib.lib()

    def run(mode):
        monkeypatch.setenv("NEXTOU_PW_SW", mode)
        monkeypatch.setenv("NEXTOU_PW_KS", "0")         # (round 5's K-split kernel has its own test below)
        L_.nextou_profile_enable(64)
        out = {"plain": hip.pw_rows(x, w, None, 1), "stats": hip.pw_rows_fused(x, w, 1, want_stats=True),
               "pro": hip.pw_rows_fused(x, w, 1, pro=(scale, shift, 0.01)),
               "pro_stats": hip.pw_rows_fused(x, w, 1, pro=(scale, shift, 0.01), want_stats=True),
               "bwd": hip.pw_rows_fused(x, w, 1, bwd=(h, g2, b2, m2, i2, 0.01))}
        torch.cuda.synchronize()
        buf = ctypes.create_string_buffer(1 << 16)
        n = L_.nextou_profile_report(buf, len(buf))
        L_.nextou_profile_enable(0)
        return out, [r["kernel"] for r in json.loads(buf.value[:n].decode())]

    new, names_new = run("2")
    old, names_old = run("0")
    assert all(k.startswith("pw_rows_sw_kernel") for k in names_new), names_new
    assert all(k.startswith("pw_rows_kernel") for k in names_old), names_old
    # one 132-channel slab: the same MFMA chain as pw_rows_kernel, bit for bit; several slabs (K = 264, 528) cut the k range at
    # multiples of 132 instead of 16 — another order of the same fp32 chain, equal to round-off
    x64, w64 = x.permute(0, 2, 3, 4, 1).reshape(-1, ci)[::97].double(), w.double()
    mag = x64.abs() @ w64.abs().t()

    def same(a, b, key):
        if ci == 132:
            assert torch.equal(a, b), key
        else:
            assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()), key
    same(new["plain"], old["plain"], "plain")
    got = new["plain"].permute(0, 2, 3, 4, 1).reshape(-1, co)[::97].double()
    assert bool(((got - x64 @ w64.t()).abs() <= 6e-7 * mag + 1e-30).all())
    for key in ("stats", "pro", "pro_stats", "bwd"):
        same(new[key][0], old[key][0], key)
    assert torch.equal(new["pro"][0], new["pro_stats"][0])
    for key in ("stats", "pro_stats", "bwd"):
        a, b = new[key][1].sum(1), old[key][1].sum(1)              # (C, 2) float64 each; different tilings of the same sums
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-6 * float(b.abs().max())), key
    y64 = new["plain"].permute(0, 2, 3, 4, 1).reshape(-1, co).double()
    s = new["stats"][1].sum(1)
    assert torch.allclose(s[:, 0], y64.sum(0), rtol=1e-6, atol=1e-6 * float(y64.abs().sum(0).max()))
    assert torch.allclose(s[:, 1], y64.square().sum(0), rtol=1e-6)


@pytest.mark.parametrize("co,sp", [(132, (33, 45, 45)), (132, (16, 64, 66)), (96, (33, 45, 45)), (144, (5, 120, 121))])
def test_k_split_stationary_weights_rows_kernel(ops, monkeypatch, co, sp):
    """pw_rows_ks_kernel (K = 528 split over the four waves of a workgroup, weights in registers, x streamed once, partial products summed
    through LDS; >= 65 536 points) against pw_rows_kernel and the float64 product: fp32 round-off (the k range is cut at multiples of 132);
    plain, with the statistics epilogue, with the normalise + activate prologue, with both — the four variants agree bit for bit where
    their arithmetic is the same, the statistics partials describe the float64 sums of the product; ragged and exact point counts,
    fewer than nine channel tiles; repeated launches bit-identical."""
    import ctypes
    import json
    from nextou_amd import _lib
    hip = ops._HIP
    ci = 528
    gen = torch.Generator().manual_seed(co + sp[0])
    x = _cl(torch.randn((1, ci) + sp, generator=gen))
    w = (torch.randn((co, ci), generator=gen) * 0.1).to(DEV)
    scale = (torch.rand((ci,), generator=gen) + 0.5).to(DEV)
    shift = (torch.randn((ci,), generator=gen) * 0.2).to(DEV)
    L_ = _lib.lib()

    def run(ks):
        monkeypatch.setenv("NEXTOU_PW_KS", ks)
        monkeypatch.setenv("NEXTOU_PW_SW", "0")
        L_.nextou_profile_enable(64)
        out = {"plain": hip.pw_rows(x, w, None, 1), "stats": hip.pw_rows_fused(x, w, 1, want_stats=True),
               "pro": hip.pw_rows_fused(x, w, 1, pro=(scale, shift, 0.01)),
               "pro_stats": hip.pw_rows_fused(x, w, 1, pro=(scale, shift, 0.01), want_stats=True)}
        torch.cuda.synchronize()
        buf = ctypes.create_string_buffer(1 << 16)
        n = L_.nextou_profile_report(buf, len(buf))
        L_.nextou_profile_enable(0)
        return out, [r["kernel"] for r in json.loads(buf.value[:n].decode())]

    new, names_new = run("2")           # 2: every instantiated variant (the default, 1, takes the kernel without a prologue only)
    again, _ = run("2")
    old, names_old = run("0")
    assert all(k.startswith("pw_rows_ks_kernel") for k in names_new), names_new
    assert all(k.startswith("pw_rows_kernel") for k in names_old), names_old
    assert torch.equal(new["plain"], new["stats"][0]) and torch.equal(new["pro"][0], new["pro_stats"][0])
    for key in ("plain",):
        assert torch.equal(new[key], again[key])
    for key in ("stats", "pro", "pro_stats"):
        assert torch.equal(new[key][0], again[key][0])
        if new[key][1] is not None:
            assert torch.equal(new[key][1], again[key][1])
    rows = x.permute(0, 2, 3, 4, 1).reshape(-1, ci)
    x64, w64 = rows[::97].double(), w.double()
    mag = x64.abs() @ w64.abs().t()
    got = new["plain"].permute(0, 2, 3, 4, 1).reshape(-1, co)[::97].double()
    assert bool(((got - x64 @ w64.t()).abs() <= 6e-7 * mag + 1e-30).all())
    assert float((new["plain"] - old["plain"]).abs().max()) <= 1e-5 * float(old["plain"].abs().max())
    a64 = torch.nn.functional.leaky_relu(torch.addcmul(shift.double(), rows[::97].double(), scale.double()), 0.01)
    gotp = new["pro"][0].permute(0, 2, 3, 4, 1).reshape(-1, co)[::97].double()
    assert bool(((gotp - a64 @ w64.t()).abs() <= 2e-6 * (a64.abs() @ w64.abs().t()) + 1e-30).all())
    assert float((new["pro"][0] - old["pro"][0]).abs().max()) <= 1e-5 * float(old["pro"][0].abs().max())
    for key in ("stats", "pro_stats"):
        y64 = new[key][0].permute(0, 2, 3, 4, 1).reshape(-1, co).double()
        sums = new[key][1].sum(1)
        assert torch.allclose(sums[:, 0], y64.sum(0), rtol=1e-6, atol=1e-6 * float(y64.abs().sum(0).max()))
        assert torch.allclose(sums[:, 1], y64.square().sum(0), rtol=1e-6)
        b = old[key][1].sum(1)
        assert torch.allclose(sums, b, rtol=1e-5, atol=1e-5 * float(b.abs().max()))


@pytest.mark.parametrize("n,k", [(528, 132), (132, 528), (132, 132), (132, 264)])
def test_stationary_output_wgrad_kernel(ops, monkeypatch, n, k):
    """pw_wgrad_so_kernel (the whole N x K product in one workgroup's accumulators, gy / x streamed once) against the float64 product
    and against pw_wgrad_kernel — two summation orders of the same products: agreement to fp32 round-off of the 66 825-term sums —,
    with and without the normalise + activate operand prologue, on a ragged point count; bit-reproducible."""
    hip = ops._HIP
    gen = torch.Generator().manual_seed(n * 3 + k)
    sp = (33, 45, 45)
    x = _cl(torch.randn((1, k) + sp, generator=gen))
    gy = _cl(torch.randn((1, n) + sp, generator=gen))
    scale = (torch.rand((k,), generator=gen) + 0.5).to(DEV)
    shift = (torch.randn((k,), generator=gen) * 0.2).to(DEV)
    x64, g64 = _rows(x), _rows(gy)
    a64 = F.leaky_relu(x64 * scale.double() + shift.double(), 0.01)
    import ctypes
    import json
    from nextou_amd import _lib
    L_ = _lib.lib()

    def run(mode):
        monkeypatch.setenv("NEXTOU_PW_SO", mode)
        L_.nextou_profile_enable(16)
        out = (hip.pw_wgrad(gy, x, 1), hip.pw_wgrad_fused(gy, x, 1, (scale, shift, 0.01)))
        torch.cuda.synchronize()
        buf = ctypes.create_string_buffer(1 << 16)
        m = L_.nextou_profile_report(buf, len(buf))
        L_.nextou_profile_enable(0)
        return out, [r["kernel"] for r in json.loads(buf.value[:m].decode())]

    (new, new_pro), names_new = run("1")
    (old, old_pro), names_old = run("0")
    assert any(s_.startswith("pw_wgrad_so_kernel") for s_ in names_new), names_new
    assert not any(s_.startswith("pw_wgrad_so_kernel") for s_ in names_old), names_old
    for got, other, ref, mag in ((new, old, g64.t() @ x64, g64.abs().t() @ x64.abs()), (new_pro, old_pro, g64.t() @ a64, g64.abs().t() @ a64.abs())):
        bound = 6e-7 * mag * (x64.shape[0] / 2048.0) ** 0.5
        assert bool(((got.double() - ref).abs() <= bound).all()), float(((got.double() - ref).abs() / bound).max())
        assert bool(((got.double() - other.double()).abs() <= 2 * bound).all())
    monkeypatch.setenv("NEXTOU_PW_SO", "1")
    assert torch.equal(new, hip.pw_wgrad(gy, x, 1))


@pytest.mark.parametrize("groups,k", [(6, 44), (4, 32), (8, 32), (2, 20)])
def test_small_group_rows_kernel(ops, monkeypatch, groups, k):
    """pw_rows_grp_kernel (one wave per group, full rows through LDS; MRConv's grouped 1x1 convolution at >= 32 768 points) against
    pw_rows_kernel (fp32 round-off apart) and the float64 product, plain and with the statistics epilogue, on a ragged point
    count."""
    hip = ops._HIP
    c = groups * k
    gen = torch.Generator().manual_seed(groups * 100 + k)
    sp = (21, 45, 45)                                              # 42 525 points: not a multiple of 64
    x = _cl(torch.randn((1, c) + sp, generator=gen))
    w = (torch.randn((c, k), generator=gen) * 0.2).to(DEV)
    import json
    import ctypes
    from nextou_amd import _lib
    L_ = _lib.lib()

    def run(mode):
        monkeypatch.setenv("NEXTOU_PW_GRP", mode)
        L_.nextou_profile_enable(64)
        out = {"plain": hip.pw_rows(x, w, None, groups), "stats": hip.pw_rows_fused(x, w, groups, want_stats=True)}
        torch.cuda.synchronize()
        buf = ctypes.create_string_buffer(1 << 16)
        n = L_.nextou_profile_report(buf, len(buf))
        L_.nextou_profile_enable(0)
        return out, [r["kernel"] for r in json.loads(buf.value[:n].decode())]

    new, names_new = run("1")
    old, names_old = run("0")
    assert all(n.startswith("pw_rows_grp_kernel") for n in names_new), names_new
    assert all(n.startswith("pw_rows_kernel") for n in names_old), names_old
    # (another assignment of k to the four slots of an MFMA step than pw_rows_kernel's: the same fp32 chain in another order)
    assert float((new["plain"] - old["plain"]).abs().max()) <= 2e-6 * float(old["plain"].abs().max())
    assert torch.equal(new["stats"][0], new["plain"])
    rows = x.permute(0, 2, 3, 4, 1).reshape(-1, groups, k)[::53].double()
    w64 = w.double().reshape(groups, k, k)
    want = torch.einsum("pgk,gnk->pgn", rows, w64).reshape(rows.shape[0], c)
    mag = torch.einsum("pgk,gnk->pgn", rows.abs(), w64.abs()).reshape(rows.shape[0], c)
    got = new["plain"].permute(0, 2, 3, 4, 1).reshape(-1, c)[::53].double()
    assert bool(((got - want).abs() <= 6e-7 * mag + 1e-30).all())
    y64 = new["plain"].permute(0, 2, 3, 4, 1).reshape(-1, c).double()
    s = new["stats"][1].sum(1)                                     # (C, T, 2) -> (C, 2)
    assert torch.allclose(s[:, 0], y64.sum(0), rtol=1e-6, atol=1e-6 * float(y64.abs().sum(0).max()))
    assert torch.allclose(s[:, 1], y64.square().sum(0), rtol=1e-6)
    so = old["stats"][1].sum(1)
    assert torch.allclose(s, so, rtol=1e-6, atol=1e-6 * float(so.abs().max()))


@pytest.mark.parametrize("shape,cin,cout", [((8, 14, 12), 324, 1296), ((4, 7, 6), 1296, 324), ((16, 16), 32, 48)])
def test_small_volume_pointwise_conv_as_rows_gemm(ops, monkeypatch, shape, cin, cout):
    """graph_ops.rows_gemm (the 1x1 convolutions of volumes of <= 8192 points as a BLAS GEMM over the channels-last rows) against
    ATen's convolution: forward, data and weight gradient to fp32 round-off, channels-last in and out; the module picks it by size."""
    import torch.nn.functional as F
    from nextou_amd.network_architecture.norm_act import ConvBiasFolded2d, ConvBiasFolded3d
    gen = torch.Generator().manual_seed(cin + cout)
    mf = torch.channels_last_3d if len(shape) == 3 else torch.channels_last
    conv_f = F.conv3d if len(shape) == 3 else F.conv2d
    x = torch.randn((2, cin) + shape, generator=gen).to(DEV).contiguous(memory_format=mf).requires_grad_(True)
    cls = ConvBiasFolded3d if len(shape) == 3 else ConvBiasFolded2d
    conv = cls(cin, cout, 1, bias=True).to(DEV)
    assert ops.rows_gemm_eligible(conv, x, conv.weight)
    y = conv(x)                                                    # bias is folded away: the module returns conv(x) without it
    assert ops._dense_channels_last(y) is mf and y.shape == (2, cout) + shape
    g = torch.randn(y.shape, generator=gen).to(DEV).contiguous(memory_format=mf)
    gx, gw = torch.autograd.grad(y, [x, conv.weight], g)
    x64, w64 = x.detach().double().requires_grad_(True), conv.weight.detach().double().requires_grad_(True)
    y64 = conv_f(x64, w64)
    gx64, gw64 = torch.autograd.grad(y64, [x64, w64], g.double())
    for a, b in ((y, y64), (gx, gx64), (gw, gw64)):
        assert float((a.double() - b).abs().max()) <= 2e-5 * float(b.abs().max())
    monkeypatch.setenv("NEXTOU_PW_MM_MAX_POINTS", "0")
    assert not ops.rows_gemm_eligible(conv, x, conv.weight)
    big = torch.randn((2, cin, 32, 32, 32) if len(shape) == 3 else (2, cin, 128, 128), device=DEV).contiguous(memory_format=mf)
    monkeypatch.delenv("NEXTOU_PW_MM_MAX_POINTS")
    assert not ops.rows_gemm_eligible(conv, big, conv.weight)      # 65 536 / 32 768 points: MIOpen's kernels


@pytest.mark.parametrize("shape,classes,layout", [((2, 14, 9, 21, 17), 14, "cl"), ((2, 14, 9, 21, 17), 14, "nc"), ((3, 5, 33, 20), 5, "cl"),
                                                   ((1, 20, 6, 10, 11), 20, "cl"), ((2, 3, 4000), 3, "nc")])
def test_fused_mean_cross_entropy(ops, monkeypatch, shape, classes, layout):
    """K5c (nextou_ce_mean_fwd / _bwd through graph_ops.cross_entropy_mean and the trainers' RobustCrossEntropyLoss) against
    torch.nn.functional.cross_entropy in float64: loss and logit gradient, channels-last and NCDHW logits, ignored voxels, an odd class
    count, more than 16 classes; the gradient keeps the logits' memory layout."""
    import torch.nn.functional as F
    from nextou_amd.loss.nnunet_losses import HAVE_NNUNET, RobustCrossEntropyLoss
    gen = torch.Generator().manual_seed(sum(shape))
    x = (torch.randn(shape, generator=gen) * 3).to(DEV)
    mf = {4: torch.channels_last, 5: torch.channels_last_3d}.get(len(shape))
    if layout == "cl" and mf is not None:
        x = x.contiguous(memory_format=mf)
    x.requires_grad_(True)
    t = torch.randint(0, classes, (shape[0],) + shape[2:], generator=gen).to(DEV)
    t[0].view(-1)[::7] = -100                                     # ignored voxels
    assert ops.cross_entropy_mean_eligible(x, t)
    loss = ops.cross_entropy_mean(x, t)
    gx, = torch.autograd.grad(loss * 1.7, x)
    x64 = x.detach().double().requires_grad_(True)
    want = F.cross_entropy(x64, t)
    gw, = torch.autograd.grad(want * 1.7, x64)
    assert abs(float(loss) - float(want)) <= 2e-6 * abs(float(want))
    assert float((gx.double() - gw).abs().max()) <= 2e-6 * float(gw.abs().max())
    assert gx.stride() == x.stride()
    if not HAVE_NNUNET:
        ce = RobustCrossEntropyLoss()
        got = ce(x, t.unsqueeze(1).float())                       # nnU-Net hands the target as (B, 1, ...) float
        assert float(got) == float(loss)
        monkeypatch.setenv("NEXTOU_FUSED_CE", "0")
        assert abs(float(ce(x, t.unsqueeze(1).float())) - float(want)) <= 1e-5 * abs(float(want))


def test_pool_mrconv_grouped_conv_as_batched_gemm(ops, monkeypatch):
    """The Pool MRConv's grouped 1x1 convolution on the channel-major (B, 2C, N, 1, 1) tensor as the strided-batched GEMM
    (graph_ops.grouped_cm_gemm, reference torch_nn.py:66-92): values and all gradients against the float64 convolution, routing by
    point count, and the reference's pooled block goldens (g5) through it."""
    import torch.nn.functional as F
    import model_cases as mc
    from nextou_amd.network_architecture.norm_act import ConvBiasFolded3d
    gen = torch.Generator().manual_seed(5)
    for B, C2, N, g in ((2, 264, 10752, 6), (1, 48, 5000, 4), (2, 24, 4096, 6)):
        conv = ConvBiasFolded3d(C2, C2, 1, groups=g, bias=True).to(DEV)
        x = torch.randn(B, C2, N, 1, 1, generator=gen).to(DEV).requires_grad_(True)
        assert ops.grouped_cm_gemm_eligible(conv, x, conv.weight)
        y = conv(x)                                              # bias folded into the norm behind it: not added here
        gy = torch.randn(y.shape, generator=gen).to(DEV)
        gx, gw = torch.autograd.grad(y, [x, conv.weight], gy)
        x64, w64 = x.detach().double().requires_grad_(True), conv.weight.detach().double().requires_grad_(True)
        y64 = F.conv3d(x64, w64, None, groups=g)
        gx64, gw64 = torch.autograd.grad(y64, [x64, w64], gy.double())
        for got, want in ((y, y64), (gx, gx64), (gw, gw64)):
            assert float((got.double() - want).abs().max()) <= 2e-5 * float(want.abs().max())
        small = torch.randn(B, C2, 1344, 1, 1, device=DEV)
        assert not ops.grouped_cm_gemm_eligible(conv, small, conv.weight)          # below the measured crossover: MIOpen
    monkeypatch.setenv("NEXTOU_GROUPED_GEMM_MIN_POINTS", "1")
    for name in [n for n in mc.BLOCKS if "pool" in n]:
        for cl in (False, True):
            out, dx, g_out, g_dx, tape, entries = mc.run_block(name, "train", DEV, teacher_forced=True, channels_last=cl)
            assert float((out - g_out).abs().max()) <= 2e-5 * max(1.0, float(g_out.abs().max()))
            assert float((dx - g_dx).abs().max()) <= 5e-5 * max(1.0, float(g_dx.abs().max()))


def test_fused_rows_partial_count_follows_the_launch(ops):
    """nextou_pw_rows_tiles sized the statistics buffer from the dense plan while a strided launch fell back to the
    LDS-tiled kernel and wrote its (many more) partials past the caller's buffer.  Now the size query takes the launch's strides and
    flags, the launch re-checks the count it is handed, and strided rows give the numbers of their dense copy."""
    from nextou_amd import _lib
    L = _lib.lib()
    P, K, N = 70000, 132, 264
    x_wide = torch.randn(P, K + 12, device=DEV)                   # rows with a 12-float tail: ldx = 144 > K
    x = x_wide[:, :K]
    w = (torch.randn(N, K, device=DEV) * 0.1).contiguous()
    dense_tiles = int(L.nextou_pw_rows_tiles(P, N, K, 1, K, N, 0, 0))
    strided_tiles = int(L.nextou_pw_rows_tiles(P, N, K, 1, K + 12, N, 0, 0))
    assert strided_tiles != dense_tiles                            # stationary-weights kernel (<= CU count) vs 128-point tiles

    def launch(xp, ldx, tiles):
        y = torch.empty(P, N, device=DEV)
        part = torch.full((N, tiles, 2), float("nan"), dtype=torch.float64, device=DEV)
        guard = torch.zeros(1 << 20, dtype=torch.float64, device=DEV)     # allocated right behind: an overrun would land here
        rc = L.nextou_pw_rows_fused(xp.data_ptr(), w.data_ptr(), y.data_ptr(), P, N, K, 1, ldx, N, None, None, 1.0, part.data_ptr(), tiles,
                                    None, 0, None, None, None, None, 1.0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        return rc, y, part, guard
    rc, _, _, _ = launch(x_wide, K + 12, dense_tiles)              # the count of ANOTHER plan: refused, nothing written
    assert rc != 0 and b"partials per channel" in L.nextou_last_error()
    rc, y_s, part_s, guard = launch(x_wide, K + 12, strided_tiles)
    assert rc == 0 and float(guard.abs().sum()) == 0.0 and not bool(torch.isnan(part_s).any())
    rc, y_d, part_d, _ = launch(x.contiguous(), K, dense_tiles)
    assert rc == 0
    ref = x.double() @ w.double().t()
    for y in (y_s, y_d):
        assert float((y.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    for part in (part_s, part_d):
        tot = part.sum(1)
        assert float((tot[:, 0] - ref.sum(0)).abs().max()) <= 1e-6 * float(ref.abs().sum(0).max())
        assert float((tot[:, 1] - (ref * ref).sum(0)).abs().max()) <= 1e-6 * float((ref * ref).sum(0).max())


@pytest.mark.parametrize("shape,layout,batch_dice,do_bg,masked", [
    ((2, 14, 9, 21, 17), "cl", False, False, False), ((2, 14, 9, 21, 17), "nc", True, False, True), ((3, 5, 33, 20), "cl", False, True, False),
    ((1, 20, 6, 10, 11), "cl", True, True, True), ((2, 3, 4000), "nc", False, False, False)])
def test_fused_soft_dice(ops, monkeypatch, shape, layout, batch_dice, do_bg, masked):
    """K5d (nextou_dice_stats_fwd / _bwd through graph_ops.dice_stats and the Dice classes of the trainers' losses) against the same
    Dice module running nnU-Net's op sequence (softmax -> one-hot scatter -> products -> sums) in float64: loss and logit gradient,
    channels-last and NCDHW logits, odd and > 16 class counts, loss mask, batch dice, background on / off; the three sums themselves
    against their definition; the gradient keeps the logits' memory layout."""
    from nextou_amd.loss.nnunet_losses import MemoryEfficientSoftDiceLoss, softmax_helper_dim1
    gen = torch.Generator().manual_seed(sum(shape))
    x = (torch.randn(shape, generator=gen) * 3).to(DEV)
    mf = {4: torch.channels_last, 5: torch.channels_last_3d}.get(len(shape))
    if layout == "cl" and mf is not None:
        x = x.contiguous(memory_format=mf)
    x.requires_grad_(True)
    L = shape[1]
    y = torch.randint(0, L, (shape[0], 1) + shape[2:], generator=gen).float().to(DEV)       # nnU-Net: (B, 1, ...) float label map
    mask = (torch.rand((shape[0], 1) + shape[2:], generator=gen) > 0.3).to(DEV) if masked else None
    assert ops.dice_stats_eligible(x, y)
    inter, pred, gt = ops.dice_stats(x, y, mask)
    p = torch.softmax(x.detach().double(), 1)
    oh = torch.zeros_like(p).scatter_(1, y.long(), 1.0)
    w = mask.double() if masked else torch.ones_like(y, dtype=torch.float64)
    axes = tuple(range(2, x.dim()))
    for got, want in ((inter, (p * oh * w).sum(axes)), (pred, (p * w).sum(axes)), (gt, (oh * w).sum(axes))):
        assert float((got.double() - want).abs().max()) <= 2e-6 * max(1.0, float(want.abs().max()))
    dice = MemoryEfficientSoftDiceLoss(apply_nonlin=softmax_helper_dim1, batch_dice=batch_dice, do_bg=do_bg, smooth=1e-5, ddp=False)
    loss = dice(x, y, loss_mask=mask)
    gx, = torch.autograd.grad(loss * 1.3, x)
    assert L % 2 or gx.stride() == x.stride()                    # (odd class counts go through an NCDHW copy: the rows kernels read pairs)
    monkeypatch.setenv("NEXTOU_FUSED_DICE", "0")                 # the base class's op sequence, in float64
    x64 = x.detach().double().contiguous().requires_grad_(True)
    want = dice(x64, y, loss_mask=mask)
    gw, = torch.autograd.grad(want * 1.3, x64)
    assert abs(float(loss) - float(want)) <= 2e-6 * abs(float(want))
    assert float((gx.double() - gw).abs().max()) <= 5e-6 * float(gw.abs().max())


@pytest.mark.parametrize("dims,cin,cout,cskip", [(3, 72, 40, 40), (3, 324, 324, 324), (2, 24, 16, 16)])
def test_up_convolution_bias_folded_into_the_concatenation(ops, monkeypatch, dims, cin, cout, cskip):
    """norm_act.up_conv_cat (the decoder's cat((up-convolution(x), skip), 1) with the convolution run bias-free and its bias added by the
    one-pass concatenation kernel) against the plain path (NEXTOU_CAT_BIAS=0): same values, same gradients for x, weight, bias, skip."""
    from nextou_amd.network_architecture.norm_act import ConvTransposeOwnBias2d, ConvTransposeOwnBias3d, up_conv_cat
    gen = torch.Generator().manual_seed(cin + cout)
    mf = torch.channels_last_3d if dims == 3 else torch.channels_last
    sp = (4, 6, 5) if dims == 3 else (9, 7)
    k = (1, 2, 2) if dims == 3 else (2, 2)
    up = (ConvTransposeOwnBias3d if dims == 3 else ConvTransposeOwnBias2d)(cin, cout, k, k, bias=True).to(DEV)
    x = torch.randn((2, cin) + sp, generator=gen).to(DEV).contiguous(memory_format=mf).requires_grad_(True)
    out_sp = tuple(s * f for s, f in zip(sp, k))
    skip = torch.randn((2, cskip) + out_sp, generator=gen).to(DEV).contiguous(memory_format=mf).requires_grad_(True)
    g = torch.randn((2, cout + cskip) + out_sp, generator=gen).to(DEV).contiguous(memory_format=mf)

    def run():
        y = up_conv_cat(up, x, skip)
        return y, torch.autograd.grad(y, [x, up.weight, up.bias, skip], g)
    y1, g1 = run()
    monkeypatch.setenv("NEXTOU_CAT_BIAS", "0")
    y0, g0 = run()
    assert ops._dense_channels_last(y1) is mf and y1.shape == y0.shape
    assert torch.equal(y1, y0) or float((y1 - y0).abs().max()) <= 1e-6 * float(y0.abs().max())
    for a, b in zip(g1, g0):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())


@pytest.mark.parametrize("C,k,mode", [(264, 14, "train"), (324, 28, "train"), (264, 14, "eval")])
def test_swin_block_below_the_chain_threshold_with_fused_aggregate(ops, monkeypatch, C, k, mode):
    """Stages 3-5 of cfg 2 (groups of 88 / 108 channels, blocks below NEXTOU_PW_FUSE_MIN_POINTS): SwinGrapher with aggregate + window
    reverse + grouped convolution in ONE launch (graph_ops.mr_grouped_conv) and the norm / fc2 modules behind it, against the three
    launches (NEXTOU_MR_GROUPED=0): same output, gradients and running statistics; the launch labels prove which path ran."""
    import ctypes
    import json
    from nextou_amd import _lib
    from nextou_amd.network_architecture import NexToU_Encoder_Decoder as encdec
    from nextou_amd.network_architecture.norm_act import fuse_norm_act
    kw = dict(conv_op=nn.Conv3d, norm_op=nn.BatchNorm3d, norm_op_kwargs={'eps': 1e-5, 'affine': True})
    monkeypatch.setenv("NEXTOU_MR_GROUPED_MIN_WORKGROUPS", "0")     # (8 windows here; the product takes the launch from 256 workgroups on)
    torch.manual_seed(C + k)
    blk = encdec.SwinGrapher(C, (4, 12, 14), k, 1, 'mr', 'leakyrelu', 'instance', True, True, 0.2, 1, n=168, relative_pos=True,
                             window_size=(2, 6, 14), shift_size=[1, 3, 7], dropout_op=None, **kw)
    fuse_norm_act(blk)
    blk = blk.to(DEV).train(mode == "train")
    shape = (2, C, 4, 12, 14)
    x0 = torch.randn(shape, device=DEV).contiguous(memory_format=CL)
    gy = torch.randn(shape, device=DEV).contiguous(memory_format=CL)
    L_ = _lib.lib()
    results = {}
    for setting in ("1", "0"):
        monkeypatch.setenv("NEXTOU_MR_GROUPED", setting)
        m = copy.deepcopy(blk)
        x = x0.clone().requires_grad_(True)
        L_.nextou_profile_enable(512)
        y = m(x)
        grads = torch.autograd.grad(y, [x] + [p for p in m.parameters() if p.requires_grad], gy, allow_unused=True)
        torch.cuda.synchronize()
        buf = ctypes.create_string_buffer(1 << 16)
        n = L_.nextou_profile_report(buf, len(buf))
        L_.nextou_profile_enable(0)
        names = [r["kernel"] for r in json.loads(buf.value[:n].decode())]
        results[setting] = (y.detach(), grads, {k_: v.clone() for k_, v in m.state_dict().items() if "running" in k_}, names)
    assert sum(n_.startswith("mr_grp_rows_kernel<train>") for n_ in results["1"][3]) == 1
    assert sum(n_.startswith("mr_grp_rows_bwd_kernel") for n_ in results["1"][3]) == 1
    assert not any(n_.startswith(("mr_fwd", "mr_bwd")) for n_ in results["1"][3])      # (window_scatter stays: window_gather's backward)
    assert not any(n_.startswith("mr_grp_rows") for n_ in results["0"][3])
    (y1, g1, r1, _), (y0, g0, r0, _) = results["1"], results["0"]
    assert float((y1 - y0).abs().max()) <= 2e-5 * float(y0.abs().max())
    gscale = max(float(b.abs().max()) for b in g0 if b is not None)
    for a, b in zip(g1, g0):
        assert (a is None) == (b is None)
        if a is not None:
            assert float((a - b).abs().max()) <= max(1e-4 * float(b.abs().max()), 2e-6 * gscale)
    for k_ in r0:
        assert torch.allclose(r1[k_], r0[k_], rtol=1e-5, atol=1e-6), k_

