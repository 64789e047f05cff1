import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLD = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(GOLD, "data")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gmg():
    """the glimmer-mg_amd package (ctypes binding over libgmg.so); builds the library if needed"""
    import _gmg_pkg
    pkg = _gmg_pkg.load()
    pkg.build.build_lib()
    return pkg


@pytest.fixture(scope="session")
def oracle():
    """ctypes binding over the CPU oracle (test infrastructure)"""
    import oracle_py
    return oracle_py.load()


@pytest.fixture(scope="session")
def gpu(gmg):
    """initialised device 0; the -m gpu tests call the product through this"""
    gmg.init(0)
    return gmg


def built_binary(*parts):
    """path of a binary that only the build container can make (it needs /root/reference): integration/_build/* (the drop-in
    builds, product side) or oracle/_ref/* (the all-reference builds, checker side).  Missing: the test is skipped -- unless
    GMG_EXPECT_REF=1 says the binaries were pushed with the tree (the round's GPU runs), then it fails."""
    exe = os.path.join(ROOT, *parts)
    if not os.access(exe, os.X_OK):
        msg = "%s not built (needs /root/reference in the build container: make -C oracle ref && make -C integration)" % os.path.join(*parts)
        if os.environ.get("GMG_EXPECT_REF") == "1":
            pytest.fail(msg)
        pytest.skip(msg)
    return exe


@pytest.fixture
def request_finalizers():
    """callbacks run after the test (library switches set with gmg_set_option go back to their defaults)"""
    todo = []
    yield todo
    for f in reversed(todo):
        f()


@pytest.fixture(scope="session")
def seqs_fa(gmg):
    hdrs, seqs = gmg.read_fasta(os.path.join(DATA, "seqs.fa"))
    return hdrs, seqs
