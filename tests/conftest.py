import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def ctx():
    """GPU context through the C ABI.  No fallback: a missing library or GPU is an error."""
    from visual_odometry_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


@pytest.fixture
def options(ctx):
    """options(name, value) sets a per-context switch (vo_ctx_set_option) for the rest of the test;
    whatever a test changed is put back afterwards."""
    before = {}

    def set_option(name, value):
        before.setdefault(name, ctx.get_option(name))
        ctx.set_option(name, value)

    yield set_option
    for name, value in before.items():
        ctx.set_option(name, value)
