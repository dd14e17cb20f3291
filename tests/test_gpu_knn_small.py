"""knn_small_kernel (csrc/knn_graph.hip, round 6): a handful of <= 192-point self graphs — cfg 2's stage-4 / 5 windows — in ONE launch:
slabs double-buffered through registers, one 16 x 16 MFMA tile per wave, selection by counting ranks.  Bar: neighbour ids bit-identical to
the oracle (reference torch_edge.py:58-110, 151-163 restated canonically) and to the three-launch path, exact ties included."""
import ctypes
import json

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from nextou_amd import _lib, graph_ops
    _lib.lib()
    assert "libnextou_hip.so" in open("/proc/self/maps").read(), "HIP extension not loaded into this process"
    return graph_ops


@pytest.fixture(scope="module")
def ora():
    import oracle
    oracle.lib()
    return oracle.CanonicalBackend


def _rand(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


def _labels(fn):
    from nextou_amd import _lib
    L = _lib.lib()
    L.nextou_profile_enable(64)
    out = fn()
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 16)
    n = L.nextou_profile_report(buf, len(buf))
    L.nextou_profile_enable(0)
    return out, [r["kernel"] for r in json.loads(buf.value.decode())] if n else []


@pytest.mark.parametrize("B,C,N,k,relpos", [
    (2, 324, 168, 32, True),     # cfg-2 stage 5, Pool graph
    (2, 324, 168, 28, True),     # cfg-2 stage 5, Swin window
    (16, 324, 168, 14, True),    # cfg-2 stage 4, Swin windows
    (1, 3, 4, 4, False),         # the smallest graph the kernel takes: every point is every point's neighbour
    (3, 65, 192, 32, False),     # the 192-point limit, one slab + one channel
    (5, 64, 100, 9, True),       # ragged tiles (100 = 6 * 16 + 4), exactly one slab
    (2, 130, 16, 16, True),      # one tile, K = N
    (7, 7, 180, 1, False),       # K = 1: every point finds itself
])
def test_small_graphs_bit_exact(ops, ora, B, C, N, k, relpos):
    x = _rand((B, C, N), 100 + N + k)
    rp = _rand((N, N), 101 + N + k, 0.05) if relpos else None
    want = ora.knn_graph(x, None, rp, k).numpy()
    xd, rd = x.to(DEV), None if rp is None else rp.to(DEV)
    got, labels = _labels(lambda: ops.knn_graph(xd, None, rd, k, algo="fused"))
    assert len(labels) == 1 and labels[0].startswith("knn_small_kernel"), labels       # ONE launch, this kernel
    got = got.cpu().numpy()
    bad = (got != want).any(-1)
    assert not bad.any(), "%d of %d rows differ (first: %s got %s want %s)" % (bad.sum(), bad.size, np.argwhere(bad)[0], got[bad][0], want[bad][0])
    three = ops.knn_graph(xd, xd.clone(), rd, k, algo="fused")            # an explicit y = x graph takes prep + fused (+ merge)
    assert torch.equal(torch.from_numpy(got).to(DEV), three)
    if k == 1:
        assert (got[..., 0] == np.arange(N)).all()


def test_small_graphs_exact_ties_break_by_index(ops, ora):
    """Duplicated points and an all-equal window: the rank sum detects the ties, the full (dist, index) comparison orders them."""
    half = _rand((2, 20, 48), 7)
    x = torch.cat([half, half], 2).contiguous()                     # every point twice: each query has pairs of equal distances
    for k in (4, 16, 32):
        want = ora.knn_graph(x, None, None, k).numpy()
        got, labels = _labels(lambda: ops.knn_graph(x.to(DEV), None, None, k, algo="fused"))
        assert labels[0].startswith("knn_small_kernel")
        np.testing.assert_array_equal(got.cpu().numpy(), want)
    const = torch.ones(1, 6, 40)
    got = ops.knn_graph(const.to(DEV), None, None, 8).cpu().numpy()
    np.testing.assert_array_equal(got, np.broadcast_to(np.arange(8), (1, 40, 8)))


def test_small_kernel_declines_what_it_does_not_take(ops):
    """Many windows (stage 2 / 3), a ragged point count, a pooled (x, y) graph: the other kernels, same entry point."""
    for shape, y_shape in (((600, 12, 168), None), ((2, 8, 166), None), ((2, 8, 168), (2, 8, 40))):
        x = _rand(shape, 3).to(DEV)
        y = None if y_shape is None else _rand(y_shape, 4).to(DEV)
        _, labels = _labels(lambda: ops.knn_graph(x, y, None, 7, algo="fused"))
        assert labels and not any(l.startswith("knn_small_kernel") for l in labels), labels


def test_small_kernel_on_guard_pages(ops):
    from tools.guard_alloc import GuardScope
    x, rp = _rand((2, 70, 168), 11).to(DEV), _rand((168, 168), 12, 0.05).to(DEV)
    want = ops.knn_graph(x, None, rp, 28, algo="fused")
    torch.cuda.synchronize()
    for flush in ("end", "start"):
        scope = GuardScope(flush=flush, align=16)
        try:
            gx, grp = scope.like(x.clone()), scope.like(rp.clone())
            with scope.patched_outputs():
                got = ops.knn_graph(gx, None, grp, 28, algo="fused")
            torch.cuda.synchronize()
            assert torch.equal(got, want)
        finally:
            torch.cuda.synchronize()
            scope.close()
