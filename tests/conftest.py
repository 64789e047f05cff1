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
    the tree carries other such builds (then this one went missing: the test FAILS) or GMG_EXPECT_REF=1 says so; GMG_EXPECT_REF=0
    turns that off."""
    exe = os.path.join(ROOT, *parts)
    if not os.access(exe, os.X_OK):
        msg = "%s not built (needs /root/reference in the build container: make -C oracle ref && make -C integration)" % os.path.join(*parts)
        # a tree that carries SOME of those builds (the build container after build(), the GPU box that got its copy) must carry
        # all of them: a missing one fails there; only a tree without any (a fresh clone with no reference around) skips
        have_some = any(os.path.isdir(d) and any(os.access(os.path.join(d, f), os.X_OK) and os.path.isfile(os.path.join(d, f)) for f in os.listdir(d))
                        for d in (os.path.join(ROOT, "oracle", "_ref"), os.path.join(ROOT, "integration", "_build")))
        if os.environ.get("GMG_EXPECT_REF") == "1" or (have_some and os.environ.get("GMG_EXPECT_REF") != "0"):
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
