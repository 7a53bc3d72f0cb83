import os, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "merlin-zkevm-prover_amd"))
sys.path.insert(0, ROOT)


# logical-shard discipline for every multi-device test of the suite and for the children they start (csrc/capi.hip mi_own_check): on
# one GPU every shard is device 0, so a buffer, a program or an event of one shard used for another would otherwise go unnoticed
os.environ.setdefault("MI_MULTI_CHECK", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
