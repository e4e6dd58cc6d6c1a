import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
