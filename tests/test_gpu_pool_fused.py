"""K2 + K7 for the pooled graphs (the channel-major half): csrc/mr_aggregate.hip mr_grp_cm_kernel —
max-relative aggregation of a pooled (xy) or self graph -> MRConv's grouped 1x1 convolution -> InstanceNorm statistics in one launch
(reference NexToU_Encoder_Decoder.py:401-418 inside PoolDyGraphConv :516-551; torch_nn.py:66-92) — against the launches it replaces:
aggregate and arg tape bit-identical to nextou_mr_aggregate_fwd, the convolution against the float64 grouped convolution of that
aggregate, the statistics against their definition, K6's apply on them against K6's own two-pass forward; the autograd of the block
against the op-by-op block; random shapes."""
import pytest
import torch
import torch.nn.functional as F
from hypothesis import given, settings, strategies as st

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from nextou_amd import _lib, graph_ops
    _lib.lib()
    assert "libnextou_hip.so" in open("/proc/self/maps").read(), "HIP extension not loaded into this process"
    return graph_ops


@pytest.fixture(autouse=True)
def _every_shape(monkeypatch):
    """the product takes the launch only from a workgroup per CU on (NEXTOU_MR_GROUPED_CM_MIN_WORKGROUPS = 256, a measured crossover):
    the tests want it at every size"""
    monkeypatch.setenv("NEXTOU_MR_GROUPED_CM_MIN_WORKGROUPS", "0")
    monkeypatch.setenv("NEXTOU_MR_GROUPED_CM_CHUNKS", "1")          # ... and for sources that stream through LDS in chunks


def _ops():
    import os
    os.environ["NEXTOU_MR_GROUPED_CM_MIN_WORKGROUPS"] = "0"          # (hypothesis bodies run outside function-scoped fixtures)
    os.environ["NEXTOU_MR_GROUPED_CM_CHUNKS"] = "1"
    from nextou_amd import graph_ops
    return graph_ops


def _check_forward(ops, B, C, N, M, K, groups, seed, Ng=None):
    be = ops._HIP
    Kg = 2 * C // groups
    Ng = Kg if Ng is None else Ng
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((B, C, N), generator=g).to(DEV)
    y = None if M is None else torch.randn((B, C, M), generator=g).to(DEV)
    m = N if M is None else M
    idx = torch.randint(0, m, (B, N, K), generator=g, dtype=torch.int32).to(DEV)
    w = (torch.randn((groups * Ng, Kg), generator=g) * 0.3).to(DEV)
    tiles = be.mr_grouped_cm_tiles(B, C, groups, Ng, N, m, K)
    assert tiles == -(-N // 128)
    a, arg, h, part = be.mr_grouped_cm(x, y, idx, K, 1, w, groups, True, True, True)
    a0, arg0 = be.mr_fwd(x, y, idx, None, K, 1, want_arg=True)
    assert torch.equal(a, a0), "the aggregate must be mr_fwd_qb_kernel's, bit for bit"
    assert torch.equal(arg, arg0), "... and so must the arg-max tape"
    want = F.conv1d(a0.double(), w.double().unsqueeze(-1), groups=groups)
    mag = F.conv1d(a0.double().abs(), w.double().abs().unsqueeze(-1), groups=groups) + 1e-30
    assert float(((h.double() - want).abs() / mag).max()) <= 1e-6
    h64 = h.double().reshape(B * groups * Ng, N)
    sums = part.sum(1)
    assert part.shape == (B * groups * Ng, tiles, 2)
    assert torch.allclose(sums[:, 0], h64.sum(1), rtol=0, atol=2e-6 * float(h64.abs().sum(1).max()) + 1e-12)
    assert torch.allclose(sums[:, 1], (h64 * h64).sum(1), rtol=2e-6, atol=1e-12)
    # without the training outputs: same h, bit for bit
    _, _, h2, part2 = be.mr_grouped_cm(x, y, idx, K, 1, w, groups, False, False, True)
    assert torch.equal(h, h2) and torch.equal(part, part2)
    return x, y, idx, w, h, part


@pytest.mark.parametrize("B,C,N,M,K,groups", [
    (2, 132, 1000, 168, 14, 6),       # cfg 2 Pool s2 (fewer queries): 44-channel groups, one source chunk
    (2, 264, 700, 1344, 28, 6),       # Pool s3: 88-channel groups, the source streams through LDS in chunks of quads
    (1, 324, 1344, None, 32, 6),      # Pool s4: self graph, 108-channel groups
    (2, 324, 168, None, 32, 6),       # Pool s5
    (2, 24, 300, 75, 9, 6),           # tiny model: 8-channel groups
    (3, 12, 256, 32, 4, 6),           # the g5 block fixtures: 4-channel groups (one quad, half empty)
    (1, 48, 129, 40, 5, 4),           # 2-D model's groups of 4; one query past a tile boundary
    (1, 132, 127, 168, 1, 6),         # K = 1
])
def test_pooled_mrconv_in_one_launch(ops, B, C, N, M, K, groups):
    x, y, idx, w, h, part = _check_forward(ops, B, C, N, M, K, groups, seed=C + N)
    # K6's apply from the launch's partials == K6's own statistics + apply on h (InstanceNorm + LeakyReLU), bit for bit or 1 ulp
    Co = w.shape[0]
    nw = (torch.rand(Co, generator=torch.Generator().manual_seed(1)) + 0.5).to(DEV)
    nb = (torch.randn(Co, generator=torch.Generator().manual_seed(2)) * 0.2).to(DEV)
    got = ops.norm_act(h, nw, nb, None, None, True, 0.0, 1e-5, 0.01, instance=True, stats_partial=part)
    want = ops.norm_act(h, nw, nb, None, None, True, 0.0, 1e-5, 0.01, instance=True)
    assert float((got - want).abs().max()) <= 2e-6 * float(want.abs().max())
    ref = F.leaky_relu(F.instance_norm(h.double(), None, None, nw.double(), nb.double(), True, 0.0, 1e-5), 0.01)
    assert float((got.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())


@pytest.mark.parametrize("B,C,N,M,K", [(2, 132, 600, 168, 14), (1, 264, 300, 1344, 28), (2, 24, 200, None, 7)])
def test_pooled_mrconv_block_autograd_matches_op_by_op(ops, monkeypatch, B, C, N, M, K):
    """graph_ops.mr_grouped_cm_block (fused launch + K6 apply) against the same modules op by op (NEXTOU_MR_GROUPED_CM=0: mr_aggregate ->
    grouped conv -> InstanceNormAct): output and every gradient — x, y, the conv weight, the norm's weight and bias."""
    from torch import nn
    from nextou_amd.network_architecture.NexToU_Encoder_Decoder import MRConv
    from nextou_amd.network_architecture.norm_act import fuse_norm_act
    torch.manual_seed(C)
    mr = MRConv(C, 2 * C, act='leakyrelu', norm='instance', bias=True, conv_op=nn.Conv3d)
    fuse_norm_act(mr)
    mr = mr.to(DEV).train()
    with torch.no_grad():
        mr.nn[1].weight.uniform_(0.5, 1.5)
        mr.nn[1].bias.normal_(0, 0.2)
    g = torch.Generator().manual_seed(N)
    x0 = torch.randn((B, C, N), generator=g).to(DEV)
    y0 = None if M is None else torch.randn((B, C, M), generator=g).to(DEV)
    idx = torch.randint(0, N if M is None else M, (B, N, K), generator=g, dtype=torch.int32).to(DEV)
    gout = torch.randn((B, 2 * C, N, 1, 1), generator=g).to(DEV)
    results = []
    calls = []
    real = ops._HIP.mr_grouped_cm
    monkeypatch.setattr(ops._HIP, "mr_grouped_cm", staticmethod(lambda *a, **k: (calls.append(1), real(*a, **k))[1]))
    for mode in ("1", "0"):
        monkeypatch.setenv("NEXTOU_MR_GROUPED_CM", mode)
        x = x0.clone().requires_grad_(True)
        y = None if y0 is None else y0.clone().requires_grad_(True)
        mr.zero_grad(set_to_none=True)
        out = mr.aggregate(x, idx, y)
        out.backward(gout)
        results.append([out.detach(), x.grad, None if y is None else y.grad] + [p.grad.clone() for p in mr.parameters() if p.grad is not None])
    assert len(calls) == 1, "the fused launch must have been taken exactly once (mode 1)"
    assert len(results[0]) == len(results[1])
    for a, b in zip(*results):
        if a is None:
            assert b is None
            continue
        assert float((a - b).abs().max()) <= 2e-5 * max(float(b.abs().max()), 1e-6), (a.shape, float((a - b).abs().max()), float(b.abs().max()))


@settings(max_examples=20, deadline=None, derandomize=True)
@given(B=st.integers(1, 3), cg=st.one_of(st.integers(1, 12), st.sampled_from([22, 44, 54])), groups=st.sampled_from([1, 2, 4, 6]),
       N=st.integers(1, 400), M=st.one_of(st.none(), st.integers(2, 700)), K=st.integers(1, 32), seed=st.integers(0, 10 ** 6))
def test_pooled_mrconv_any_shape(B, cg, groups, N, M, K, seed):
    ops = _ops()
    C = cg * groups
    m = N if M is None else M
    if m < 2:
        return
    K = min(K, m)
    if ops._HIP.mr_grouped_cm_tiles(B, C, groups, 2 * cg, N, m, K) <= 0:
        return
    _check_forward(ops, B, C, N, M, K, groups, seed)


def test_pooled_mrconv_is_taken_from_a_workgroup_per_cu_on(ops, monkeypatch):
    """the default, measured crossovers: cfg 2's Pool s2 shape (1 008 workgroups, the group's source in LDS in one piece) takes the launch;
    s3 (the source streams in chunks), s4 (132 workgroups) and s5 (12) keep the three launches"""
    monkeypatch.delenv("NEXTOU_MR_GROUPED_CM_MIN_WORKGROUPS")
    monkeypatch.delenv("NEXTOU_MR_GROUPED_CM_CHUNKS")
    be = ops._HIP
    assert be.mr_grouped_cm_tiles(2, 132, 6, 44, 10752, 168, 14) == 84
    assert be.mr_grouped_cm_tiles(2, 264, 6, 88, 10752, 1344, 28) == 0
    assert be.mr_grouped_cm_tiles(2, 324, 6, 108, 1344, 1344, 32) == 0 and be.mr_grouped_cm_tiles(2, 324, 6, 108, 168, 168, 32) == 0
