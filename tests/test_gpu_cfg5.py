"""BASELINE.json configs[4] as a TEST: one train step of the 96 x 256 x 256 network (base 33 / max 324, batch 2)
with the conv stages under bf16 autocast and the graph kernels in fp32, on the MI355X; every one of the 14 kNN graph constructions
of the forward (reference NexToU_Encoder_Decoder.py:960-1006 hyper-parameters, torch_edge.py:151-163) is checked INSIDE the model run
against the oracle, bit for bit, on row / window subsets of the fp32 features the kernel really received (bf16-rounded values)."""
import os
import sys
import time

import pytest
import torch

from conftest import REPO

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.mark.timeout(3000)
def test_cfg5_bf16_train_step_with_knn_checked_inside_the_run(monkeypatch):
    monkeypatch.setenv("NEXTOU_FAST_RELPOS", "1")          # the literal 24 389^2 float64 position table is 4.8 GB
    for k in ("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD", "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD", "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW"):
        monkeypatch.setenv(k, "0")
    if REPO not in sys.path:
        sys.path.insert(0, REPO)
    import bench
    import oracle
    from nextou_amd import _lib, graph_ops
    from nextou_amd.harness import downsample_targets, synthetic_batch
    _lib.lib()
    oracle.lib()
    ora = oracle.CanonicalBackend
    # MIOpen immediate mode: the find step over the bf16 96 x 256 x 256 convolutions costs minutes and this test is about one correct step
    monkeypatch.setattr(torch.backends.cudnn, "benchmark", False)

    torch.cuda.empty_cache()       # (the step needs most of the GPU: blocks cached by earlier tests would be freed one failed allocation at a time)
    t0 = time.time()
    trainer, cfg, batch, classes = bench.build_trainer("cfg5", DEV, False)
    bench.move_to(trainer, DEV)
    data, target = synthetic_batch(cfg, 1, classes, batch, DEV, seed=1234)
    targets = downsample_targets(target, bench._head_shapes(cfg))
    step = bench.make_step(trainer, data, targets, None, bf16=True)
    torch.cuda.synchronize()
    print("cfg5 network + batch built: %.1f s" % (time.time() - t0))

    records = []
    original = graph_ops._HipBackend.knn_graph

    def recording(x, y, relpos, k_total, algo=0, normalize=True):
        out = original(x, y, relpos, k_total, algo, normalize)
        assert x.dtype == torch.float32 and (y is None or y.dtype == torch.float32)        # K1 computes in fp32 under autocast
        B, C, N = x.shape
        if y is None:                                   # self graph: whole windows / samples (queries == candidates)
            sub = sorted({0, B // 2, B - 1})[: (3 if N <= 384 else 1)]
            records.append(("self", x[sub].cpu(), None, None if relpos is None else relpos.cpu(), k_total, out[sub].cpu(), (B, C, N, N)))
        else:                                           # pooled graph: a strided subset of the query rows, all candidates
            rows = torch.arange(0, N, 193, device=x.device)
            records.append(("xy", x[:, :, rows].cpu(), y.cpu(), None if relpos is None else relpos[rows].cpu(), k_total,
                            out[:, rows].cpu(), (B, C, N, y.shape[2])))
        return out

    monkeypatch.setattr(graph_ops._HipBackend, "knn_graph", staticmethod(recording))
    t0 = time.time()
    loss = step()
    torch.cuda.synchronize()
    print("cfg5 step (incl. MIOpen kernel builds on a fresh box): %.1f s" % (time.time() - t0))
    assert torch.isfinite(loss), float(loss)
    assert all(torch.isfinite(p.grad).all() for p in trainer.network.parameters() if p.grad is not None)
    monkeypatch.setattr(graph_ops._HipBackend, "knn_graph", staticmethod(original))

    assert len(records) == 14, [r[6] for r in records]
    shapes = sorted({r[6] for r in records})
    # SURVEY.md §A.1, cfg 5: windows of 384 points, pooled graphs 24 576 x 384 / 24 576 x 3 072, 3 072 and 384-point self graphs
    assert (1024, 132, 384, 384) in shapes and (2, 264, 24576, 3072) in shapes and (2, 132, 24576, 384) in shapes, shapes
    t0 = time.time()
    for kind, x, y, rp, k, got, shape in records:
        want = ora.knn_graph(x.contiguous(), y, rp, k)
        assert torch.equal(got, want), (kind, shape, k)
    print("oracle checks of the 14 graphs: %.1f s" % (time.time() - t0))


def _bench_line(args, timeout):
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("NEXTOU_REDUCED_PRECISION_FILTERS",)}
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.timeout(1800)
def test_bench_tiny_under_bf16_autocast_in_find_mode():
    """`bench.py --autocast-bf16` with MIOpen's find mode ON (the mode the bench warms up in; VERDICT r5 missing #3): the reduced-precision
    bench path runs to its JSON line.  Round 5's closing tree died here with a GPU memory fault; round 6 convicted the library's find pass
    over the transposed convolution with a channels-last bf16 filter (profiles/r06_bf16/README.md), which channel_pad.filter_layout_for
    no longer hands it."""
    rec = _bench_line(["--workload", "tiny", "--autocast-bf16", "--steps", "3", "--warmup", "2", "--no-cpu-baseline"], 1500)
    assert rec["dtype"].startswith("bf16") and rec["value"] > 0 and rec["config"]["step_replayed_as_hipgraph"] is True


@pytest.mark.timeout(2400)
def test_bench_cfg2_under_bf16_autocast_in_find_mode():
    """The headline shape under bf16 autocast, find mode, three timed steps — the exact command that faulted at round 5's closing tree."""
    rec = _bench_line(["--workload", "cfg2", "--autocast-bf16", "--steps", "3", "--warmup", "2", "--no-cpu-baseline"], 2100)
    assert rec["dtype"].startswith("bf16") and rec["config"]["name"] == "cfg2" and rec["steps"] == 3
    assert 0 < rec["ms_per_step"] < 150          # ~77 ms on an MI355X; the fp32 step is ~167
