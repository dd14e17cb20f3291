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


def test_static_switches_turn_the_round_6_small_kernels_off():
    """NEXTOU_KNN_SMALL=0, NEXTOU_K6_ONE=0 and NEXTOU_KNN_GLDS=0 are read once per process (no getenv on the launch path), so the off position needs its own process:
    the same calls must take the multi-launch kernels and return the same bits."""
    import os
    import subprocess
    import sys
    code = r'''
import ctypes, json, torch
from nextou_amd import _lib, graph_ops
L = _lib.lib()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)
x = torch.randn((2, 70, 168), generator=g).to(dev)
y = torch.randn((2, 8, 4, 7, 6), generator=g).to(dev).contiguous(memory_format=torch.channels_last_3d)
w, b = torch.rand(8, generator=g).to(dev) + 0.5, torch.randn(8, generator=g).to(dev)
L.nextou_profile_enable(64)
ids = graph_ops.knn_graph(x, None, None, 9, algo="fused")
xq, yc = torch.randn((2, 40, 1024), generator=g).to(dev), torch.randn((2, 40, 640), generator=g).to(dev)
rp = (torch.randn((1024, 640), generator=g) * 0.05).to(dev)
long_ids = graph_ops.knn_graph(xq, yc, rp, 28, algo="fused")          # K > 16, M >= 512: the 128-wide kernel (direct-to-LDS slabs unless NEXTOU_KNN_GLDS=0)
out = graph_ops.norm_act(y, w, b, None, None, True, 0.1, 1e-5, 0.01)
torch.cuda.synchronize()
buf = ctypes.create_string_buffer(1 << 16)
n = L.nextou_profile_report(buf, len(buf))
labels = [r["kernel"] for r in json.loads(buf.value.decode())] if n else []
pos = torch.arange(long_ids.numel(), device=dev).view_as(long_ids)
print(json.dumps({"labels": labels, "long": [int(long_ids.sum()), int((long_ids.long() * pos % 1000003).sum())], "ids": int(ids.sum()), "ids_hash": int((ids.long() * torch.arange(ids.numel(), device=dev).view_as(ids) % 1000003).sum()),
                  "out": out.double().sum().item(), "out_abs": out.double().abs().sum().item()}))
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    runs = {}
    for name, extra in (("on", {}), ("off", {"NEXTOU_KNN_SMALL": "0", "NEXTOU_K6_ONE": "0", "NEXTOU_KNN_GLDS": "0"})):
        env = dict(os.environ, PYTHONPATH=root, **extra)
        for k in ("NEXTOU_KNN_SMALL", "NEXTOU_K6_ONE", "NEXTOU_KNN_GLDS"):
            if k not in extra:
                env.pop(k, None)
        p = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        runs[name] = json.loads(p.stdout.strip().splitlines()[-1])
    on, off = runs["on"], runs["off"]
    assert any(l.startswith("knn_small_kernel") for l in on["labels"]) and any(l.startswith("bn_one") for l in on["labels"]), on["labels"]
    assert not any(l.startswith("knn_small_kernel") or l.startswith("bn_one") for l in off["labels"]), off["labels"]
    assert (on["ids"], on["ids_hash"]) == (off["ids"], off["ids_hash"])                      # neighbour ids: the same bits
    assert any(l.startswith("knn_fused_kernel<28,4>") for l in on["labels"]) and on["long"] == off["long"]      # register-staged and direct-to-LDS slabs: the same ids
    assert abs(on["out"] - off["out"]) <= 1e-6 * on["out_abs"] + 1e-9                        # K6: the same sums in another order


@pytest.mark.parametrize("name,shape,y_shape,k", [
    ("small", (2, 70, 168), None, 9),              # knn_small_kernel (counting selection)
    ("window", (600, 12, 168), None, 7),           # knn_window_kernel (whole windows per workgroup)
    ("window split", (40, 12, 168), None, 7),      # knn_window_kernel over 64-wide candidate splits + merge
    ("fused self", (2, 40, 1344), None, 32),       # prep + fused + merge of partial lists
    ("fused xy", (2, 64, 2048), (2, 64, 512), 28),  # pooled graph
    ("naive", (1, 16, 300), None, 40),             # K > 32: materialised distances + selection
])
def test_non_finite_features_still_give_valid_neighbour_ids(ops, name, shape, y_shape, k):
    """A feature that overflowed (fp16 autocast: GradScaler expects to skip such a step, not to lose the process) makes distances NaN / inf.
    The reference's topk still returns ids inside the candidate set (torch_edge.py:58-110); so must every kernel here — an id outside
    [0, M) is an out-of-bounds gather in the aggregation that follows.  Non-finite distances sort last, by index."""
    x = _rand(shape, 31)
    y = None if y_shape is None else _rand(y_shape, 32)
    src = x if y is None else y
    M = src.shape[2]
    bad = [5, 77, M - 1]
    src[:, :, bad[0]] = float("nan")               # a whole candidate is NaN
    src[0, 3, bad[1]] = float("inf")               # one channel of one candidate overflowed: inf / inf = NaN after the normalisation
    src[-1, 0, bad[2]] = float("-inf")
    got = ops.knn_graph(x.to(DEV), None if y is None else y.to(DEV), None, k)
    torch.cuda.synchronize()
    got = got.cpu()
    assert got.shape == (shape[0], shape[2], k)
    assert int(got.min()) >= 0 and int(got.max()) < M, (name, int(got.min()), int(got.max()))
    srt = got.sort(-1).values
    assert bool((srt[..., 1:] != srt[..., :-1]).all()), "%s: an id twice in one row" % name
    finite_query = torch.isfinite(x).all(1)                                # (B, N)
    bad_cand = ~torch.isfinite(src).all(1)                                 # (B, M)
    hits = torch.gather(bad_cand.unsqueeze(1).expand(-1, shape[2], -1), 2, got.long())     # (B, N, k): neighbour is a non-finite candidate
    assert not bool((hits & finite_query.unsqueeze(-1)).any()), "%s: a non-finite candidate ahead of finite ones" % name
