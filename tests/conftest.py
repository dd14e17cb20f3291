import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")
for p in (REPO, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


@pytest.fixture(autouse=True)
def _flush_native_stdio():
    """MIOpen's CK kernels print applicability diagnostics ("GridwiseOp: Problemsize descriptor dimension check failure") through C
    stdio.  With stdout a pipe that text sits in the process's stdio buffer until exit and lands AFTER pytest's summary, outside any
    test's capture (it buried the tail of GPUTEST_r03).  Flushing at every teardown puts it into the capture of the test that caused
    it, which pytest drops for passing tests."""
    yield
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle
    oracle.lib()
    return oracle


@pytest.fixture()
def cpu_checker(oracle_lib):
    """Routes CPU tensors of nextou_amd.graph_ops to the canonical oracle for the test's duration."""
    from nextou_amd import graph_ops
    graph_ops.install_cpu_checker(oracle_lib.CanonicalBackend)
    yield oracle_lib.CanonicalBackend
    graph_ops.install_cpu_checker(None)


@pytest.fixture()
def torch_ref_checker(oracle_lib):
    from nextou_amd import graph_ops
    from oracle.ref_ops import TorchRefBackend
    graph_ops.install_cpu_checker(TorchRefBackend)
    yield TorchRefBackend
    graph_ops.install_cpu_checker(None)


def knn_rows_equal_as_sets(a, b):
    """(B,N,k) integer arrays -> bool (B,N): same neighbour set in the row."""
    return (np.sort(np.asarray(a), axis=-1) == np.sort(np.asarray(b), axis=-1)).all(-1)
