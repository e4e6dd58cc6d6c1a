import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A session that holds GPU tests brings torch's device context up first: torch ships its own HIP runtime, and when libkmahip.so's
    (the system's) has taken the device before it in the same process, torch finds "No HIP GPUs" -- which made the tests that hand
    torch tensors to the C-ABI depend on the order the files ran in. On a machine without a GPU nothing happens."""
    if "not gpu" in (config.getoption("markexpr", "") or "") or not any(it.get_closest_marker("gpu") for it in items):
        return
    try:
        import torch
        if torch.cuda.device_count() > 0:
            torch.zeros(1, device="cuda:0")
    except Exception:  # noqa: BLE001  (no torch, no device: the tests that need them say so themselves)
        pass


@pytest.fixture(scope="session")
def golden_se(tmp_path_factory):
    """Unpacked tests/golden/se fixture: dict with paths + parsed streams."""
    import golden_util
    return golden_util.load_se(tmp_path_factory.mktemp("golden_se"))


@pytest.fixture(scope="session")
def golden_long(tmp_path_factory):
    import golden_util
    return golden_util.load_se(tmp_path_factory.mktemp("golden_long"), "long")


@pytest.fixture(scope="session")
def golden_pe(tmp_path_factory):
    import golden_util
    return golden_util.load_pe(tmp_path_factory.mktemp("golden_pe"))
