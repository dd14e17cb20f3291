"""Repository contract: the C-ABI exports what include/nextou_hip.h declares, the product never
touches the oracle, and the driver entry points exist."""
import os
import re

import pytest

from conftest import REPO


def test_cabi_exports_every_declared_symbol():
    """Loads libnextou_hip.so (no GPU needed) and resolves every function the header declares."""
    from nextou_amd import _lib
    header = open(os.path.join(REPO, "include", "nextou_hip.h")).read()
    declared = set(re.findall(r"\b(nextou_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    lib = _lib.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), "libnextou_hip.so does not export %s" % name
    assert declared == set(_lib.EXPORTED_SYMBOLS)
    assert lib.nextou_abi_version() == _lib.ABI_VERSION


def test_cabi_argument_errors_do_not_need_a_gpu():
    from nextou_amd import _lib
    lib = _lib.lib()
    rc = lib.nextou_knn_graph(None, None, None, None, None, 0, 1, 1, 1, 1, 1, 0, 1, None)
    assert rc == -1 and b"null pointer" in lib.nextou_last_error()
    rc = lib.nextou_bti_critical_map(1, 1, 1, 3, 1, 1, 4, 4, 4, 5, 1, None)
    assert rc == -1 and b"connectivity" in lib.nextou_last_error()
    assert lib.nextou_knn_workspace_bytes(2, 8, 100, 50, 4, 1, 0) > 2 * 8 * 150 * 4


def test_product_never_imports_the_oracle():
    pat = re.compile(r"^\s*(from|import)\s+oracle\b", re.M)
    for root, _, files in os.walk(os.path.join(REPO, "nextou_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert not pat.search(src), "%s imports the oracle" % os.path.join(root, f)


def test_no_reference_source_in_repo():
    """Fixtures are data: no .py under tests/golden other than the generator and the formula."""
    allowed = {"make_golden.py", "formula.py"}
    for f in os.listdir(os.path.join(REPO, "tests", "golden")):
        assert f.endswith(".npz") or f in allowed or f == "__pycache__", f


def test_entry_points_present():
    import __graft_entry__ as ge
    assert callable(ge.build) and callable(ge.smoke)
    for f in ("bench.py", "DESIGN.md", "INTEGRATION.md", "include/nextou_hip.h", "oracle/nextou_oracle.c"):
        assert os.path.exists(os.path.join(REPO, f)), f


def test_trainer_plugins_are_discoverable_by_name():
    """nnU-Net finds trainers by class name inside the module of the same name."""
    import importlib
    for name in ("nnUNetTrainer_NexToU", "nnUNetTrainer_NexToU_NoMirroring", "nnUNetTrainer_NexToU_BTI_Synapse",
                 "nnUNetTrainer_NexToU_BTI_RAVIR", "nnUNetTrainer_NexToU_BTI_ICA_NoMirroring",
                 "nnUNetTrainer_NexToU_TI", "nnUNetTrainer_NexToU_TI_NoMirroring"):
        mod = importlib.import_module("nextou_amd.nnUNetTrainer." + name)
        cls = getattr(mod, name)
        assert hasattr(cls, "build_network_architecture")
